#!/bin/bash
# same-box A/B: in-tree vs a build with the tanh / sigmoid bodies compiled out (instruction-footprint probe)
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree     |"
  DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/$1 $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|$1 |"
done 2>&1 | tee gpurun_out/r05_e_ab.txt
DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/$1 $B > gpurun_out/r05_e_bench.json 2>/dev/null; python tools/bench_summary.py gpurun_out/r05_e_bench.json
$B > gpurun_out/r05_e_bench0.json 2>/dev/null; python tools/bench_summary.py gpurun_out/r05_e_bench0.json
