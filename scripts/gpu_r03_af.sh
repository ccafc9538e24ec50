#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "on_the_load or groupnorm or dft or pack" 2>&1 | tail -8 > gpurun_out/r03af_tests.log
timeout 2400 python -m pytest tests/test_gpu_sizes.py tests/test_gpu_model.py -m gpu -q -x 2>&1 | tail -5 >> gpurun_out/r03af_tests.log
for c in L M L20; do for d in 1 0; do st=6; [ $c = L20 ] && st=3; DPOT_GN_ONLOAD=$d timeout 900 python bench.py --config $c --steps $st --warmup 2 2>/dev/null | head -c 200; echo " onload=$d"; done; done > gpurun_out/r03af_bench.txt
cat gpurun_out/r03af_tests.log gpurun_out/r03af_bench.txt
