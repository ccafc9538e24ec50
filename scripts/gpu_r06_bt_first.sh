mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bf16" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_train2.py -x -q -m gpu -k "adam_writes or segmented_graph_step_equals_eager_step_bf16 or graph_warmup" 2>&1 | tail -5
python scripts/gemm_yardstick.py M L16 > gpurun_out/r06_gemm_yardstick_bt.txt 2>&1; grep -v "^{" gpurun_out/r06_gemm_yardstick_bt.txt | cut -c1-170
rm -f gpurun_out/r06_bt_step_ab.txt
bash scripts/ab_config.sh gpurun_out/r06_bt_step_ab.txt M 20 "DPOT_BF16P_BT=0 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=1" "DPOT_BF16P_BT=0 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=1" > /dev/null
bash scripts/ab_config.sh gpurun_out/r06_bt_step_ab.txt L 8 "DPOT_BF16P_BT=0 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=1" "DPOT_BF16P_BT=0 DPOT_ADAM_PACKS=0" "DPOT_BF16P_BT=1 DPOT_ADAM_PACKS=1"
