#!/usr/bin/env python3
"""Per-kernel mean PMC counter values from a rocprofv3 rocpd database (rocprofv3 --kernel-trace --pmc ...)."""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
name_k = "kernel_name" if "kernel_name" in ix else "name"
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    k = r[ix[name_k]].split("(")[0]
    acc[(k, r[ix["grid_size"]] if "grid_size" in ix else 0)][r[ix["counter_name"]]].append(r[ix["value"]])
ctrs = sorted({cn for v in acc.values() for cn in v})
print("columns:", cols)
print(f"{'kernel':42s} {'grid':>8s} " + " ".join(f"{cn[-22:]:>22s}" for cn in ctrs))
for (k, g), v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get(ctrs[0], [0]))):
    print(f"{k[:42]:42s} {g:8d} " + " ".join(f"{(sum(v[cn])/max(1,len(v[cn]))):22.0f}" if cn in v else f"{'-':>22s}" for cn in ctrs))
