#!/usr/bin/env python
"""per-kernel launch count / average duration of the LAST n dispatches in a rocpd database (rocprofv3 --kernel-trace)"""
import collections, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
c = collections.defaultdict(lambda: [0, 0.0])
for name, s, e in rows[-n:]:
    k = re.sub(r"\(.*", "", name).replace("void ", "")[:80]
    c[k][0] += 1; c[k][1] += (e - s) / 1e3
for k, v in sorted(c.items(), key=lambda x: -x[1][1]):
    print(f"{v[0]:5d} {v[1] / v[0]:9.2f} us avg  {k}")
