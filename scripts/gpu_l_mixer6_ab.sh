mkdir -p gpurun_out
for t in "mixer6=0" "mixer6=1"; do DPOT_TUNE=$t timeout 900 python bench.py --config L --steps 10 --warmup 3 --no-other-configs --no-alt --no-pipeline 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$t', d['ms_per_step'], d['value'], d['config'].get('final_loss'))"; done
