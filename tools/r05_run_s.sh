#!/bin/bash
# r05_s: wide dX body with 16-deep K tiles (build variant -DDQN_DX_KT=16: 24 KB instead of 46.6 KB of LDS -> the fused backward launches at 4 workgroups per CU): parity, then same-box A/B at config 5
mkdir -p gpurun_out
V=$PWD/deepqlearning.jl_amd/build/dx_kt16.so
DQN_MI355X_LIB=$V timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "u8 or config5 or wide_sample or four_columns or 32x32 or weights_resident" 2>&1 | grep -E "^E|passed|failed|Error" | tail -5
for i in 1 2 3; do
for so in "" "$V"; do
  DQN_MI355X_LIB=$so timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|cfg5 ${so:+dx_kt16}${so:-in-tree} |"
done; done 2>&1 | tee gpurun_out/r05_s_cfg5_dx_kt16_ab.txt
DQN_MI355X_LIB=$V timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -14
