#!/bin/bash
# A/B of a train step (hipGraph replay, `bench.py --brief`) under environment switches, pairs on ONE box:
#   scripts/ab_config.sh OUTFILE CONFIG STEPS "ENV=a" "ENV=b" ["ENV=a" "ENV=b" ...]      (CONFIG: T | S | M | L | L20)
# one line per run: "config M DPOT_TUNE=packs=0: 12.68 ms/step 2523.3 samples/s" - the form of every profiles/r0*_step_ab.txt
O=$1; CFG=$2; STEPS=$3; shift; shift; shift
mkdir -p "$(dirname "$O")"
for v in "$@"; do
  echo -n "config $CFG $v: " >> $O
  env $v timeout 600 python bench.py --config $CFG --brief --skip-cpu-baseline --no-other-configs --no-alt --steps $STEPS --warmup 5 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], 'ms/step', d['value'], d['unit'])" >> $O 2>&1
done
cat $O
