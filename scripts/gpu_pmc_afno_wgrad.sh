#!/bin/bash
# counters of the AFNO weight-gradient launches (separate --pmc passes with --kernel-trace only) -> gpurun_out/r05_pmc_afno_wgrad.json
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in T M L L4; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    T=$(echo $C | tr ' ' '_')
    if [ $F = L4 ]; then export DPOT_TUNE=wgrad_gauss=0; FF=L; else unset DPOT_TUNE; FF=$F; fi
    timeout 240 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcw_${F}-$T -o p -- python $R/scripts/afno_wgrad_one.py $FF > $R/gpurun_out/pmcw.log 2>&1
  done
done
cd $R
python - <<'PY'
import csv, glob, json, collections
shape = {"T": (4608, 4, 128), "M": (4608, 8, 128), "L": (8704, 16, 96), "L4": (8704, 16, 96)}
res = {}
for d in sorted(glob.glob("gpurun_out/pmcw_*")):
    form = d.split("pmcw_")[1].split("-")[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        vals, durs = collections.defaultdict(list), collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "gemm_tn" in k:
                vals[(k.split("(")[0][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
                durs[k.split("(")[0][:60]].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
        for (k, c), v in vals.items():
            v = v[-10:]
            e = res.setdefault(form, {})
            e["kernel"] = k
            e[c] = round(sum(v) / len(v), 1)
            e["us_under_counters"] = round(sum(durs[k][-10:]) / len(durs[k][-10:]), 2)
for form, e in res.items():
    Mm, nb, bs = shape[form]
    alg = 4.0 * Mm * 2 * nb * bs * 4 + 2.0 * 2 * nb * bs * bs * 4        # four operands read once + the gradients written
    e["algorithmic_MB"] = round(alg / 1e6, 1)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["bytes_guide"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["traffic_over_algorithmic"] = round(e["bytes_guide"] / alg, 2)
    if e.get("SQ_BUSY_CU_CYCLES"):
        e["mfma_util"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * e["SQ_BUSY_CU_CYCLES"]), 3)
out = {"note": "rocprofv3 --kernel-trace --pmc, one counter group per pass (scripts/gpu_pmc_afno_wgrad.sh, scripts/afno_wgrad_one.py); mean of the "
               "last 10 launches; T / M: gemm_tn_kernel three-product form (DPOT-Tiny / DPOT-S,-M), L: gemm_tn96g_kernel, L4: the 192 x 192 "
               "four-product kernel it replaced; bytes_guide = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md); traffic includes the "
               "split-K partials (written by the kernel, read by the reduce launch - not counted here)",
       "forms": res}
json.dump(out, open("gpurun_out/r05_pmc_afno_wgrad.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf gpurun_out/pmcw_*
