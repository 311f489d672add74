#!/bin/bash
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused_reduce_head or config5 or wide_sample or u8" > gpurun_out/r05_k_pytest1.log 2>&1; tail -4 gpurun_out/r05_k_pytest1.log
for i in 1 2; do
DQN_NO_RED_HEAD=1 python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --per-call-steps 0 2>/dev/null > /tmp/c0.json; python tools/bench_summary.py /tmp/c0.json | head -12 | tr '\n' ' ' | sed 's/^/no_red_head /'; echo
python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --per-call-steps 0 2>/dev/null > /tmp/c1.json; python tools/bench_summary.py /tmp/c1.json | head -12 | tr '\n' ' ' | sed 's/^/red_head    /'; echo
done | tee gpurun_out/r05_k_cfg5_ab.txt
