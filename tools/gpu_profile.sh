#!/bin/bash
# usage (on the GPU box, from the repo root): tools/gpu_profile.sh <tag> [bench args...]
# runs rocprofv3 --kernel-trace --stats on bench.py (eager launches so every kernel is a dispatch) and writes
# gpurun_out/<tag>_summary.txt (per-kernel table) + gpurun_out/<tag>_step.txt (one step's dispatch sequence)
tag=$1; shift
R=$(pwd)
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o r -- python $R/bench.py --steps 50 --warmup 5 --profile-steps 1 --no-cpu-baseline --no-graph --sustained-steps 2000 --per-call-steps 0 --no-secondary "$@" > $R/gpurun_out/${tag}.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_summary.txt
python $R/tools/rocprof_step.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_step.txt
rm -rf $R/gpurun_out/$tag
tail -1 $R/gpurun_out/${tag}.log | cut -c1-300
