#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (rocprofv3 --kernel-trace [--stats] -d DIR -o NAME) into the per-kernel
table committed under profiles/.  usage: rocprof_summary.py results.db [skip_first_n_dispatches_per_kernel]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        short = name.split("(")[0]
        st = stats.setdefault(short, [])
        st.append(e - s)
    tot = sum(sum(v) for v in stats.values())
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[:70]:70s} {len(v):7d} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} {max(v)/1e3:10.2f} {sum(v)/1e6:10.3f} {100*sum(v)/tot:6.2f}")
    print(f"total kernel time {tot/1e6:.3f} ms over {len(rows)} dispatches; wall span {(rows[-1][2]-rows[0][1])/1e6:.3f} ms")


if __name__ == "__main__":
    main()
