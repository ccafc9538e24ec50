#!/bin/bash
# HBM traffic + matrix-pipe / LDS counters of the bf16 channel-MLP GEMM launches (DPOT-M shapes, training forms): separate
# --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section) -> gpurun_out/pmc_bf16p_r03.json
mkdir -p gpurun_out
R=$PWD
SHAPE=${1:-M}
cd /tmp && export TMPDIR=/tmp
for F in fc1_fwd fc2_fwd fc2_dgrad fc1_dgrad pair; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    T=$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcb_${F}_$T -o p -- python $R/scripts/bf16p_one.py $SHAPE $F > $R/gpurun_out/pmcb.log 2>&1
  done
done
cd $R
python - <<PY
import csv, glob, json, collections
res = {}
for d in sorted(glob.glob("gpurun_out/pmcb_*")):
    form = d.split("pmcb_")[1]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        vals = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gemm_bf16p" in k and "pack" not in k:
                vals[(k[:48], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in vals.items():
            v = v[-12:]                      # the 12 launches after the set-up calls
            res.setdefault(form.split("_FETCH")[0].split("_WRITE")[0].split("_SQ")[0], {})[c] = {"kernel": k, "launches": len(v), "mean": sum(v) / len(v)}
json.dump(res, open("gpurun_out/pmc_bf16p_r03_$SHAPE.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcb_*
