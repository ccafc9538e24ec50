#!/bin/bash
# ablation builds of csrc/afno_mlp.hip (timing experiments; several of them compute garbage):
#   scripts/afno_variants.sh NAME -DFLAG...  ->  dpot_amd/lib/variants/libdpot_hip_NAME.so  (use with DPOT_HIP_LIB=...)
set -e
NAME=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/dpot_amd/lib/variants
O=$R/dpot_amd/lib/variants/afno_mlp_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -I$R/include "$@" -c $R/dpot_amd/csrc/afno_mlp.hip -o $O
OBJS=$(ls $R/dpot_amd/lib/*.o | grep -v "/afno_mlp.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so $OBJS $O
echo $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so
