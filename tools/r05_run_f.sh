#!/bin/bash
python -m pytest tests -q -m gpu > gpurun_out/r05_f_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05_f_pytest.log
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree (noinline act)  |"
  DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/probe_notrans.so $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|probe_notrans           |"
done 2>&1 | tee gpurun_out/r05_f_ab.txt
python tools/drqn_bench.py 2>&1 | tail -1
python tools/cfg1_bench.py 2>&1 | head -1
