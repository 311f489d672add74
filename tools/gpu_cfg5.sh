#!/bin/bash
# usage (GPU box, repo root): tools/gpu_cfg5.sh <tag>   -- config 5 (B=512, u8 replay of 1e6 transitions): parity tests, rocprofv3 kernel
# stats + one step's dispatch sequence, the bench line (graph replay)
tag=${1:-r02_f}
# (the parity tests of this shape run in the full suite: tools/gpu_evidence.sh)
R=$(pwd); mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$tag -o r -- python $R/bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 50 --warmup 5 --profile-steps 1 --no-cpu-baseline --env-steps 0 --no-graph --sustained-steps 100 --per-call-steps 0 > $R/gpurun_out/${tag}.log 2>&1
python $R/tools/rocprof_summary.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_cfg5_kernels.txt
python $R/tools/rocprof_step.py $R/gpurun_out/$tag/r_results.db > $R/gpurun_out/${tag}_cfg5_one_step.txt
rm -rf $R/gpurun_out/$tag
cat $R/gpurun_out/${tag}_cfg5_one_step.txt
cd $R; python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 3 > gpurun_out/${tag}_cfg5_bench.json 2>/dev/null; python tools/bench_summary.py gpurun_out/${tag}_cfg5_bench.json | head -3
