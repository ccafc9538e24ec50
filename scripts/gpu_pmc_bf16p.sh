#!/bin/bash
# HBM traffic + matrix-pipe / LDS counters of the bf16 channel-MLP GEMM launches (DPOT-M shapes, training forms, the kernels
# of the default selection): separate --pmc passes with --kernel-trace only (MI355X_MICROARCH.md, HBM section)
# -> gpurun_out/r06_pmc_bf16p_M.json  (bytes_guide = (2 FETCH_SIZE + WRITE_SIZE) KiB: the guide's gfx950 correction)
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
M, E, mh = {"S": (8192, 1024, 1024), "M": (8192, 1024, 4096), "L16": (16384, 1536, 6144), "L4": (4096, 1536, 6144)}["$SHAPE"]
A, H = 2.0 * M * E, 2.0 * M * mh                         # bf16 bytes of a [tokens, E] / [tokens, hidden] pack
Wb = 2.0 * E * mh
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
from dpot_amd import ops
npk = 2      # packs of the hidden layer / its gradient written
alg = {"fc1_fwd": A + Wb + (npk + 1) * H, "fc2_fwd": H + Wb + 2 * 4.0 * M * E, "fc2_dgrad": A + Wb + H + npk * H,
       "fc1_dgrad": H + Wb + 4.0 * M * E, "pair": 2 * (A + H) + 2 * 4.0 * E * mh}
forms = {}
for form in alg:
    ent = {}
    for d in sorted(glob.glob(f"gpurun_out/pmcb_{form}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            vals = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if "gemm_bf16p" in k and "pack" not in k and "reduce" not in k:
                    vals[(k.split("(")[0][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
            for (k, c), v in vals.items():
                v = v[-12:]                  # the 12 launches after the set-up calls
                ent["kernel"] = k
                ent[c] = sum(v) / len(v)
    if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
        ent["FETCH_SIZE_KiB"], ent["WRITE_SIZE_KiB"] = ent.pop("FETCH_SIZE"), ent.pop("WRITE_SIZE")
        ent["bytes_raw"] = (ent["FETCH_SIZE_KiB"] + ent["WRITE_SIZE_KiB"]) * 1024
        ent["bytes_guide"] = (2 * ent["FETCH_SIZE_KiB"] + ent["WRITE_SIZE_KiB"]) * 1024
        ent["algorithmic_MB"] = round(alg[form] / 1e6, 1)
        ent["traffic_over_algorithmic"] = round(ent["bytes_guide"] / alg[form], 2)
    if ent.get("SQ_BUSY_CU_CYCLES"):
        ent["mfma_util"] = round(ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * ent["SQ_BUSY_CU_CYCLES"]), 3)
    forms[form] = ent
out = {"shape": {"name": "$SHAPE", "tokens": M, "E": E, "hidden": mh},
       "note": "rocprofv3 --kernel-trace --pmc, one counter group per pass, scripts/gpu_pmc_bf16p.sh; mean of 12 launches per "
               "form (scripts/bf16p_one.py); bytes_guide = (2*FETCH_SIZE + WRITE_SIZE)*1024 (MI355X_MICROARCH.md: FETCH_SIZE "
               "reports half of a wide streaming read on gfx950); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)",
       "forms": forms}
json.dump(out, open("gpurun_out/r06_pmc_bf16p_$SHAPE.json", "w"), indent=1)
print(json.dumps(forms, indent=1))
PY
rm -rf gpurun_out/pmcb_*
