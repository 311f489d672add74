#!/usr/bin/env python3
"""Print the dispatch sequence of ONE acting vector step of the device env loop from a rocpd database: everything after one k_env_observe2 up to and including the next."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, grid_x, workgroup_x, lds_size, vgpr_count from kernels order by start").fetchall()
idxs = [i for i, r in enumerate(rows) if "k_env_observe2" in r[0]]
a, b = idxs[-3] + 1, idxs[-2] + 1
t0 = rows[a][1]
print(f"{'t_us':>9s} {'dur_us':>8s} {'grid':>8s} {'wg':>4s} {'lds':>6s} {'vgpr':>4s}  kernel")
for r in rows[a:b]:
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.2f} {r[3]:8d} {r[4]:4d} {r[5]:6d} {r[6]:4d}  {r[0][:70]}")
print(f"step span {(rows[b][1]-t0)/1e3:.1f} us, kernel time {sum(r[2]-r[1] for r in rows[a:b])/1e3:.1f} us, {b-a} dispatches")
