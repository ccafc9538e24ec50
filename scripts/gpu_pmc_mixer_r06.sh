#!/bin/bash
# Counters of the AFNO mixer kernels: HBM traffic AND matrix-pipe / LDS / wait counters, for the three-product
# forward (training form), its data-gradient form, and the one-launch layer kernel.  Separate --pmc passes with --kernel-trace
# only (MI355X_MICROARCH.md, HBM section) -> gpurun_out/r06_pmc_mixer.json
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for F in tiny-train tiny-bwd fused-fwd; do
  for C in FETCH_SIZE WRITE_SIZE "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS"; do
    T=$(echo $C | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/pmcm_${F}_$T -o p -- python $R/scripts/afno_mlp_bench.py $F > $R/gpurun_out/pmcm.log 2>&1 || echo "pass $F $C failed: $(tail -2 $R/gpurun_out/pmcm.log)"
  done
done
cd $R
python - <<'PY'
import csv, glob, json, collections
out = {}
for form, pat in (("tiny-train", "afno_mlp3"), ("tiny-bwd", "afno_mlp3"), ("fused-fwd", "afno_fused_fwd")):
    ent = {}
    for d in sorted(glob.glob(f"gpurun_out/pmcm_{form}_*")):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            vals = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                if pat in k:
                    vals[(k.split("(")[0][:70], row["Counter_Name"])].append(float(row["Counter_Value"]))
            for (k, c), v in vals.items():
                v = v[-20:]
                ent["kernel"] = k
                ent[c] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
    g = lambda c: ent.get(c, {}).get("mean")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        ent["bytes_guide"] = (2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024
    if g("SQ_BUSY_CU_CYCLES"):
        ent["mfma_util"] = round(g("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * g("SQ_BUSY_CU_CYCLES")), 3)
    if g("SQ_LDS_IDX_ACTIVE"):
        ent["lds_conflict_frac"] = round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"), 4)
    if g("SQ_WAVE_CYCLES"):
        ent["wait_any_frac_of_wave_cycles"] = round(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"), 3)
    out[form] = ent
# keep the keys bench.py reads (FETCH_SIZE / WRITE_SIZE of the training-form forward) at the top level
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    if c in out.get("tiny-train", {}):
        out[c] = dict(out["tiny-train"][c], kernel=out["tiny-train"].get("kernel"))
out["note"] = ("rocprofv3 --kernel-trace --pmc, one counter group per pass (scripts/gpu_pmc_mixer_r06.sh); forms: tiny-train = "
               "afno_mlp3 forward, training form, DPOT-Tiny batch 32 (M=4608, nb=4, bs=128); tiny-bwd = its data-gradient form; "
               "fused-fwd = afno_fused_fwd_kernel, DPOT-S/-M batch 32 (256 workgroups); bytes_guide = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
               "(MI355X_MICROARCH.md); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES)")
json.dump(out, open("gpurun_out/r06_pmc_mixer.json", "w"), indent=1)
print(json.dumps({k: ({kk: vv for kk, vv in v.items() if not isinstance(vv, dict)} if isinstance(v, dict) else v) for k, v in out.items()}, indent=1))
PY
rm -rf gpurun_out/pmcm_*
