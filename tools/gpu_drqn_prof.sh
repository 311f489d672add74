#!/bin/bash
# usage (GPU box, repo root): tools/gpu_drqn_prof.sh <tag>  -- rocprofv3 kernel trace of the config-4 DRQN bench: per-kernel table + the dispatch sequence of the last steps
tag=$1; R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/$tag -o r -- python $R/tools/drqn_bench.py --steps 300 > $R/gpurun_out/${tag}.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_summary.txt
python - <<PY
import sqlite3
c = sqlite3.connect("$R/gpurun_out/$tag/r_results.db")
rows = c.execute("select name, start, end from kernels order by start").fetchall()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
cp = []
for t in tabs:
    if "memory_cop" in t.lower():
        try: cp = c.execute(f"select name, start, end from {t} order by start").fetchall(); break
        except Exception as ex: pass
ev = sorted([(s, e, n[:60]) for n, s, e in rows] + [(s, e, "COPY " + str(n)[:40]) for n, s, e in cp])
ev = ev[-24:]
t0 = ev[0][0]
with open("$R/gpurun_out/${tag}_tail.txt", "w") as f:
    for s, e, n in ev: f.write(f"{(s-t0)/1e3:9.1f} {(e-s)/1e3:8.2f}  {n}\n")
PY
rm -rf $R/gpurun_out/$tag
tail -2 $R/gpurun_out/${tag}.log; head -8 $R/gpurun_out/${tag}_summary.txt; cat $R/gpurun_out/${tag}_tail.txt
