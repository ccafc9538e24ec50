#!/bin/bash
# round 4, batch D: B-direct as default (auto CPW); A/B vs BD=0 on ONE box; skew experiment; M census; DP host time; opt-out
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
O=gpurun_out/r04_d
run() { # tag, env..., config steps warm
  local tag=$1; shift
  env "$@" timeout 300 python bench.py --config $CFG --brief --steps $ST --warmup $WU 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$CFG $tag', d['ms_per_step'], d['value'])" >> ${O}_step.txt
}
for rep in 1 2; do
  for c in "S 20 5" "M 20 5" "L 8 3"; do
    set -- $c; CFG=$1; ST=$2; WU=$3
    run "BD=0" DPOT_BF16P_BD=0
    run "BD=auto" DPOT_BF16P_BD=1
    run "BD cpw1" DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=1
    run "BD cpw2" DPOT_BF16P_BD=1 DPOT_BF16P_BD_CPW=2
  done
done
for sk in "0 8" "2 8" "4 8" "2 3" "4 3" "2 0" "4 0"; do
  set -- $sk
  echo "== skew $1 bit $2" >> ${O}_skew.txt
  DPOT_BF16P_BD_SKEW=$1 DPOT_BF16P_BD_SKEWBIT=$2 timeout 300 python scripts/bf16p_train_bench.py M L 2>&1 | grep "fc1 fwd, bf16\|fc2 dgrad, bf16\|^[ML] " >> ${O}_skew.txt
done
timeout 300 python scripts/r04/dp_host_time.py 2>&1 | grep -v amdgpu > ${O}_dp_host_time.txt
bash scripts/gpu_census_M.sh M bf16 > /dev/null 2>&1; cp gpurun_out/censusM.txt ${O}_census_M.txt
( time timeout 1500 python -m pytest tests/test_gpu_optout.py -m gpu -q -s -k "BF16P" ) 2>&1 | grep -v amdgpu.ids | tail -8 > ${O}_optout.log
cat ${O}_step.txt; cat ${O}_skew.txt; cat ${O}_dp_host_time.txt; head -12 ${O}_census_M.txt; cat ${O}_optout.log
