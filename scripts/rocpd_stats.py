#!/usr/bin/env python
"""summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / avg / share (like --stats)"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
rows = cur.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                   f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s}")
for name, n, t, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("dpot::", "").replace("void ", "")
    print(f"{name[:90]:90s} {n:6d} {t/1e6:9.3f} {t/n/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:8.2f} {100*t/tot:6.2f}")
print(f"total kernel time {tot/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
