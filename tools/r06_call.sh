#!/bin/bash
mkdir -p gpurun_out
B=$PWD/deepqlearning.jl_amd/build/base.so
for i in 1 2; do for m in 0 1; do
DQN_ADAM_MODE=$m DQN_MI355X_LIB=$B timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base adam_mode $m', round(d['value'],1), round(1e3*d['ms_per_step'],2), [(x['launch'], x['avg_us']) for x in d['roofline']['launches']])"
done; done 2>&1 | tee gpurun_out/r06_p_adam_mode_cfg5.txt
