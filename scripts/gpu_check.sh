#!/bin/bash
# one GPU-box round trip: parity tests, smoke, a short bench (logs land in gpurun_out/)
mkdir -p gpurun_out
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
nproc >> gpurun_out/env.log; lscpu | grep -E "Model name|Socket|Core|Thread" >> gpurun_out/env.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
