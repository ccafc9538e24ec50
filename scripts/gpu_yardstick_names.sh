#!/bin/bash
# which kernels hipBLASLt (torch.mm on bf16) runs for the channel-MLP shapes: kernel names carry macro-tile / wave layout
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/yard_names -o y -- python $R/scripts/gemm_yardstick.py M L16 > $R/gpurun_out/yard_names.log 2>&1
cd $R
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/yard_names/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open("gpurun_out/r06_yardstick_kernel_names.txt", "w") as o:
        for r in rows[:40]:
            line = f'{int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:9.1f} us  {r["Name"][:400]}'
            print(line); o.write(line + "\n")
PY
rm -rf gpurun_out/yard_names
