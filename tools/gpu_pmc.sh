#!/bin/bash
# usage: tools/gpu_pmc.sh <tag> "<counters>" [bench args]   (PMC pass only: --kernel-trace + --pmc, nothing else)
tag=$1; ctrs=$2; shift; shift
R=$(pwd); mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/$tag -o r -- python $R/bench.py --steps 6 --warmup 2 --profile-steps 1 --no-cpu-baseline --no-graph --sustained-steps 0 --per-call-steps 0 --no-secondary --replay 2000 "$@" > $R/gpurun_out/${tag}.log 2>&1
python $R/tools/rocprof_pmc.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_pmc.txt
rm -rf $R/gpurun_out/$tag
