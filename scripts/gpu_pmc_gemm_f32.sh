#!/bin/bash
# VERDICT r4 #3a: counters of the two native fp32 GEMM kernels that are 54 % of the headline step (separate --pmc passes with
# --kernel-trace only) -> gpurun_out/r05_pmc_gemm_f32.json
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in panel panel_lin tn; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"; do
    T=$(echo $C | tr ' ' '_')
    timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcf_${F}-$T -o p -- python $R/scripts/gemm_f32_one.py $F > $R/gpurun_out/pmcf.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, json, collections
res = {}
for d in sorted(glob.glob("gpurun_out/pmcf_*")):
    form = d.split("pmcf_")[1].split("-")[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        vals = collections.defaultdict(list)
        durs = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gemm_panel_kernel" in k or "gemm_tn" in k:
                vals[(k[:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
                durs[k[:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for (k, c), v in vals.items():
            v = v[-10:]
            e = res.setdefault(form, {}).setdefault(k, {})
            e[c] = round(sum(v) / len(v), 1)
            e["us_under_counters"] = round(sum(durs[k][-10:]) / len(durs[k][-10:]), 2)
json.dump(res, open("gpurun_out/r05_pmc_gemm_f32.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcf_*
