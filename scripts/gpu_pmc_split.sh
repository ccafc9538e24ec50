#!/bin/bash
# PMC counters of the bf16x6 GEMM kernel; $1 = tag for the output, DPOT_HIP_LIB selects a variant library
TAG=${1:-split}
R=$PWD; mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$TAG -o g -- python $R/scripts/gemm_one_split.py > $R/gpurun_out/pmc_$TAG.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc2_$TAG -o g -- python $R/scripts/gemm_one_split.py >> $R/gpurun_out/pmc_$TAG.log 2>&1
cd $R
python - $TAG <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
for d in ("pmc_", "pmc2_"):
    rows = collections.defaultdict(dict)
    for f in glob.glob(f"gpurun_out/{d}{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_f32x" not in r["Kernel_Name"]: continue
            key = (r["Dispatch_Id"], r["Kernel_Name"][:44], r["Grid_Size"])
            rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
            rows[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen = set()
    for k in sorted(rows, key=lambda x: int(x[0])):
        if (k[1], k[2]) in seen: continue
        seen.add((k[1], k[2]))
        v = rows[k]
        gui = v.get("GRBM_GUI_ACTIVE", 0)
        print(k[1], "grid", k[2], {a: round(b / 1e6, 2) if a != "dur_us" else b for a, b in v.items()}, "(counters in 1e6)",
              "clk_GHz=%.2f" % (gui / 8 / (v["dur_us"] * 1e3)) if gui else "")
PY
rm -rf gpurun_out/pmc_$TAG gpurun_out/pmc2_$TAG
