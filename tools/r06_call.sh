#!/bin/bash
mkdir -p gpurun_out
DQN_FWD_NB=4 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config5 or wide_sample or four_columns or 512" 2>&1 | grep -E "^E  |passed|failed|rror" | tail -3
for i in 1 2; do for v in 0 2 4 8; do
DQN_FWD_NB=$v timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nb $v', round(d['value'],1), round(1e3*d['ms_per_step'],2), [(x['launch'].split('+')[0], x['avg_us']) for x in d['roofline']['launches'] if x['launch'][:3] in ('fwd',)])"
done; done 2>&1 | tee gpurun_out/r06_m_fwd_nb.txt
