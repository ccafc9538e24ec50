#!/bin/bash
# HBM traffic of the AFNO mixer kernel from the L2 memory-side counters (separate --pmc passes, kernel-trace only)
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmc_$C -o p -- python $R/bench.py --steps 1 --warmup 1 --no-graph --skip-cpu-baseline > $R/gpurun_out/pmc_$C.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True)
    vals = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if row.get("Counter_Name") == c:
                vals[name].append(float(row["Counter_Value"]))
    for k, v in vals.items():
        if "gemm_f32_kernel<64, 64, false, false, true, 1>" in k:      # the AFNO mixer instantiation
            res[c] = {"kernel": k[:80], "launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    top = sorted(((sum(v) / len(v), k) for k, v in vals.items()), reverse=True)[:8]
    res[c + "_top"] = [(round(a, 1), k[:70]) for a, k in top]
json.dump(res, open("gpurun_out/pmc_mixer.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
