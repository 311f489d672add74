#!/bin/bash
# r05: parity of the fused reduce + head launch, then a same-box A/B: base build / in-tree with DQN_NO_RED_HEAD (= only the padding-wave change) / in-tree
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_reduce_head or nature_dqn_b32 or mailbox" > gpurun_out/r05_c_pytest1.log 2>&1; tail -5 gpurun_out/r05_c_pytest1.log
python -m pytest tests -q -m gpu > gpurun_out/r05_c_pytest.log 2>&1; tail -3 gpurun_out/r05_c_pytest.log
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/base_r05.so $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|base        |"
  DQN_NO_RED_HEAD=1 $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|no_red_head |"
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree     |"
done 2>&1 | tee gpurun_out/r05_c_ab.txt
$B > gpurun_out/r05_c_bench.json 2>/dev/null; python tools/bench_summary.py gpurun_out/r05_c_bench.json
DQN_NO_RED_HEAD=1 $B > gpurun_out/r05_c_bench_nrh.json 2>/dev/null; python tools/bench_summary.py gpurun_out/r05_c_bench_nrh.json
