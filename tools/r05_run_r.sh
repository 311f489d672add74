#!/bin/bash
# r05_r: k_head_cols4 (large-batch head level, four columns per workgroup): parity, then same-box A/B at config 5 against k_head_td (DQN_NO_HEAD_COLS4=1)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "four_columns or fused_reduce_head_launch or u8 or config5 or wide_sample or 32x32 or weights_resident" 2>&1 | grep -E "^E|passed|failed|Error" | tail -8
for i in 1 2; do
for k in "" 1; do
  DQN_NO_HEAD_COLS4=$k timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null | python tools/bench_summary.py /dev/stdin | head -1 | sed "s|^|cfg5 no_cols4=${k:-0} |"
done; done 2>&1 | tee gpurun_out/r05_r_cfg5_head_cols4_ab.txt
timeout 400 bash tools/gpu_cfg5.sh r05_r 2>&1 | tail -16
