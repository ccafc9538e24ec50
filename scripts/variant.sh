#!/bin/bash
# ablation / experiment build of one translation unit:
#   scripts/variant.sh FILE NAME -DFLAG...  ->  dpot_amd/lib/variants/libdpot_hip_NAME.so   (use with DPOT_HIP_LIB=...)
set -e
F=$1; NAME=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/dpot_amd/lib/variants
O=$R/dpot_amd/lib/variants/${F}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -I$R/include "$@" -c $R/dpot_amd/csrc/$F.hip -o $O
OBJS=$(ls $R/dpot_amd/lib/*.o | grep -v "/$F.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so $OBJS $O
echo $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so
