#!/bin/bash
# counters of the bf16x6 mixer kernel (csrc/afno_mlp6.hip): separate --pmc passes with --kernel-trace only -> gpurun_out/r06_pmc_mlp6.json
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in ${FORMS:-l-fwd l-bwd l1-fwd}; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
    T=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc6_${F}_$T -o p -- python $R/scripts/afno_mlp6_run.py $F > $R/gpurun_out/pmc6.log 2>&1 || echo "pass $F $C failed: $(tail -2 $R/gpurun_out/pmc6.log | cut -c1-200)"
  done
done
cd $R
python - <<'PY'
import csv, glob, json, collections, os
out = {}
for form in os.environ.get("FORMS", "l-fwd l-bwd l1-fwd").split():
    ent = {}
    for d in sorted(glob.glob(f"gpurun_out/pmc6_{form}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            vals = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "afno_mlp6_kernel" in k:
                    vals[(k.split("(")[0][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
            for (k, c), v in vals.items():
                v = v[-12:]
                ent["kernel"] = k
                ent[c] = round(sum(v) / len(v), 1)
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            ds = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "afno_mlp6_kernel" in r.get("Kernel_Name", "")]
            if ds:
                ent["us"] = round(sum(ds[-12:]) / len(ds[-12:]) / 1e3, 1)
    g = ent.get
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        ent["bytes_guide_MB"] = round((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e6, 1)
    if g("SQ_BUSY_CU_CYCLES"):
        ent["mfma_util"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), 3)
    if g("SQ_LDS_IDX_ACTIVE"):
        ent["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("SQ_WAVE_CYCLES"):
        ent["wait_any_frac_of_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
    out[form] = ent
json.dump(out, open("gpurun_out/r06_pmc_mlp6.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf gpurun_out/pmc6_* gpurun_out/pmc6.log
