#!/bin/bash
# SQ counters of the fused tail kernels (scripts/tail_bench.py): separate --pmc passes, --kernel-trace only.
# SLOW: ~3 GPU-minutes per pass (counter collection serialises every launch of the benchmark) - run with --timeout 1500.
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY")
i=0
for C in "${SETS[@]}"; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmct_$i -o p -- python $R/scripts/tail_bench.py > $R/gpurun_out/pmct_$i.log 2>&1
  i=$((i+1))
done
cd $R
python - <<'PY'
import csv, glob, json, collections
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmct_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "out_tail" in k:
            vals["fwd" if "fwd" in k else "bwd"][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in vals.items()}
json.dump(res, open("gpurun_out/pmc_tail_r02.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmct_*
