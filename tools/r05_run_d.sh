#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_reduce_head or nature_dqn_b32 or mailbox or pipelined" > gpurun_out/r05_d_pytest1.log 2>&1; tail -3 gpurun_out/r05_d_pytest1.log
python tools/red_head_phases.py 2>&1 | tail -11
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  DQN_NO_RED_HEAD=1 $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|no_red_head |"
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree     |"
done 2>&1 | tee gpurun_out/r05_d_ab.txt
$B > gpurun_out/r05_d_bench.json 2>/dev/null; python tools/bench_summary.py gpurun_out/r05_d_bench.json
