#!/bin/bash
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E -o "\b(GRBM_GUI_ACTIVE|SQ_VALU_MFMA_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|SQ_LDS_BANK_CONFLICT|SQ_BUSY_CYCLES|SQ_INSTS_MFMA|SQ_INST_CYCLES_VMEM|SQ_WAIT_INST_LDS|SQ_LDS_IDX_ACTIVE|SQ_ACTIVE_INST_LDS|SQ_INSTS_VALU_MFMA_MOPS_F32)\b" | sort -u > $R/gpurun_out/pmc_avail.txt
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_gemm -o g -- python $R/scripts/gemm_one.py > $R/gpurun_out/pmc_gemm.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/pmc_gemm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_f32" not in r["Kernel_Name"]: continue
        key = (r["Dispatch_Id"], r["Kernel_Name"][:60])
        rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
        rows[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k in sorted(rows, key=lambda x: int(x[0])):
    v = rows[k]
    gui = v.get("GRBM_GUI_ACTIVE", 0)
    print(k[1], {a: round(b, 1) for a, b in v.items()}, "clk_GHz=%.2f" % (gui / (v["dur_us"] * 1e3)) if gui else "")
PY
rm -rf gpurun_out/pmc_gemm
