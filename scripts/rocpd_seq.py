#!/usr/bin/env python
"""dump the kernel dispatch sequence (name, grid, duration) of the last N dispatches of a rocpd database"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
gx = "grid_size_x" if "grid_size_x" in cols else cols[0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.{gx}, d.grid_size_y, d.grid_size_z, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 250
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = rows[len(rows) - n - skip: len(rows) - skip]
t0 = rows[0][1]
for name, s, e, g0, g1, g2, w in rows:
    name = re.sub(r"\(.*$", "", name).replace("void ", "").replace("dpot::", "")
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.2f} us  wgs={g0 // max(w,1):6d}x{g1}x{g2:<4d} {name[:70]}")
