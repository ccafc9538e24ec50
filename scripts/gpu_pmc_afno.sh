#!/bin/bash
# SQ counters of the fused AFNO MLP kernel (one pass, kernel-trace only): where do its cycles go?
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
for SET in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU"; do
  TAG=$(echo $SET | md5sum | cut -c1-6)
  rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $R/gpurun_out/pmc_afno_$TAG -o g -- python $R/scripts/afno_mlp_bench.py > $R/gpurun_out/pmc_afno_$TAG.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_afno_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "afno_mlp2" not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:50], r["Grid_Size"] if "Grid_Size" in r else "")
        rows[key].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        rows[key].setdefault("dur_us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in rows.items():
    m = {a: sum(b) / len(b) for a, b in v.items()}
    gui = m.get("GRBM_GUI_ACTIVE", 0)
    print(k, {a: round(b, 1) for a, b in m.items()}, "clk_GHz(sum over 8 XCDs /8)=%.2f" % (gui / 8 / (m["dur_us"] * 1e3)) if gui else "")
PY
rm -rf gpurun_out/pmc_afno_*/
