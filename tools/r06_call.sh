#!/bin/bash
mkdir -p gpurun_out
DQN_DW_M32=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or wide_sample or four_columns or 512 or half_units or large_batch or nature_dqn_b32" 2>&1 | grep -E "^E  |passed|failed|rror" | tail -4
bash tools/run_ab.sh -n 3 -c both -e "DQN_DW_M32=1" "" 2>&1 | tee gpurun_out/r06_s_dw_m32_ab.txt
