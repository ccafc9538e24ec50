#!/usr/bin/env python
"""census of one train step from a rocpd database: per-kernel launches / time between two optimiser steps"""
import re
import sqlite3
import sys
import collections

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                   f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
# the LAST launch of an optimiser step: adam_kernel, or - Adam that writes the bf16 weight packs - adam_ranges_kernel
idx = [i for i, r in enumerate(rows) if "adam_kernel" in r[0] or "adam_ranges_kernel" in r[0]]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
c = collections.defaultdict(lambda: [0, 0.0])
small = []
for name, s, e, g0, g1, g2, w in step:
    wgs = (g0 // max(w, 1)) * g1 * g2
    if "gemm" in name and wgs < 256:
        small.append((wgs, (e - s) / 1e3, name[:60]))
    n = re.sub(r"\(.*", "", name)
    n = re.sub(r"^void ", "", n)
    c[n[:70]][0] += 1
    c[n[:70]][1] += (e - s) / 1e3
tot = sum(v[1] for v in c.values())
print(f"{len(step)} launches, {tot:.1f} us of kernel time, wall {(step[-1][2] - step[0][1]) / 1e3:.1f} us")
for k, v in sorted(c.items(), key=lambda x: -x[1][1]):
    print(f"{v[0]:4d} {v[1]:9.1f} us  {v[1] / v[0]:7.1f} avg  {k}")
print("GEMM launches with fewer than 256 workgroups:")
for wgs, us, name in small:
    print(f"   {wgs:4d} WGs {us:7.1f} us  {name}")
if len(sys.argv) > 2 and sys.argv[2] == "--seq":
    print("dispatch sequence of the step (start offset, duration, workgroups, kernel):")
    t0 = step[0][1]
    for name, s, e, g0, g1, g2, w in step:
        n = re.sub(r"\(.*$", "", name).replace("void ", "").replace("dpot::", "")
        print(f"{(s - t0) / 1e3:9.1f} us {(e - s) / 1e3:8.2f} us  wgs={(g0 // max(w, 1)) * g1 * g2:6d}  {n[:90]}")
