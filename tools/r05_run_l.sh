#!/bin/bash
# same-box A/B of store flavours for the GEMM launches' outputs: in-tree (plain / nt) vs sc1 for activations+slabs+dX (1), for dW (2), for both (3)
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_reduce_head" > gpurun_out/r05_l_pytest1.log 2>&1; tail -2 gpurun_out/r05_l_pytest1.log
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 --no-secondary"
for i in 1 2 3; do
  $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|in-tree |"
  for v in 1 2 3; do DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/st_sc$v.so $B 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|st_sc$v  |"; done
done 2>&1 | tee gpurun_out/r05_l_store_ab.txt
for v in 1 3; do DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/st_sc$v.so python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --per-call-steps 0 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|cfg5 st_sc$v |"; done | tee -a gpurun_out/r05_l_store_ab.txt
python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --per-call-steps 0 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|cfg5 in-tree |" | tee -a gpurun_out/r05_l_store_ab.txt
