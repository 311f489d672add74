#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_reduce_head or nature_dqn_b32 or pipelined" > gpurun_out/r05_m_pytest1.log 2>&1; tail -2 gpurun_out/r05_m_pytest1.log
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  DQN_NO_ST_WT=1 $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|no_st_wt |"
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree  |"
  for v in 1 3 7; do DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/adam_st$v.so $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|adam_st$v |"; done
done 2>&1 | tee gpurun_out/r05_m_store_ab.txt
