#!/bin/bash
mkdir -p gpurun_out
DPOT_HIP_LIB=dpot_amd/lib/variants/libdpot_hip_aftiming.so timeout 300 python scripts/afno_layer_phases.py > gpurun_out/r05_f4_phases.txt 2>&1
cat gpurun_out/r05_f4_phases.txt
