#!/bin/bash
# HBM traffic of the fused AFNO mixer kernel (DPOT-Tiny shape, training form) from the L2 memory-side counters:
# separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r05_pmc_mixer.json
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcm_$C -o p -- python $R/scripts/afno_mlp_bench.py tiny-train > $R/gpurun_out/pmcm_$C.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmcm_{c}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c and "afno_mlp" in row.get("Kernel_Name", ""):
                vals[row["Kernel_Name"][:60]].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        res[c] = {"kernel": k, "launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
res["note"] = ("rocprofv3 --pmc, one counter per pass; launches = the training form of the DPOT-Tiny mixer (M=4608, nb=4, bs=128) "
               "from scripts/afno_mlp_bench.py tiny-train; bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md")
json.dump(res, open("gpurun_out/r05_pmc_mixer.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcm_FETCH_SIZE gpurun_out/pmcm_WRITE_SIZE
