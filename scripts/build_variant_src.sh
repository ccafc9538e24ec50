#!/bin/bash
# build a variant of libdpot_hip.so with extra -D flags on ONE source file (kernel experiments):
#   scripts/build_variant_src.sh NAME gemm_bf16p -DFOO ...  ->  dpot_amd/lib/variants/libdpot_hip_NAME.so  (DPOT_HIP_LIB=...)
set -e
NAME=$1; SRC=$2; shift; shift
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/dpot_amd/lib/variants
O=$R/dpot_amd/lib/variants/${SRC}_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -I$R/include "$@" -c $R/dpot_amd/csrc/$SRC.hip -o $O
OBJS=$(ls $R/dpot_amd/lib/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so $OBJS $O
echo $R/dpot_amd/lib/variants/libdpot_hip_$NAME.so
