#!/bin/bash
# second counter set for the bf16x6 mixer kernel: instruction fetch, VMEM / LDS queue pressure, wave levels
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in ${FORMS:-l-fwd}; do
  for C in "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD" "SQ_WAIT_ANY SQ_LEVEL_WAVES" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_SALU" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_BUSY_CYCLES SQ_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_WAVES"; do
    T=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc6_${F}_$T -o p -- python $R/scripts/afno_mlp6_run.py $F > $R/gpurun_out/pmc6.log 2>&1 || echo "pass $F $C failed: $(tail -2 $R/gpurun_out/pmc6.log | cut -c1-200)"
  done
done
cd $R
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for form in os.environ.get("FORMS", "l-fwd").split():
    ent = {}
    for d in sorted(glob.glob(f"gpurun_out/pmc6_{form}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            vals = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "afno_mlp6_kernel" in k:
                    vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            for c, v in vals.items():
                v = v[-12:]
                ent[c] = round(sum(v) / len(v), 1)
    out[form] = ent
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/r06_pmc_mlp6b.json", "w"), indent=1)
PY
rm -rf gpurun_out/pmc6_* gpurun_out/pmc6.log
