#!/bin/bash
mkdir -p gpurun_out
for i in 1 2; do for w in 0 512 768 1100 1536; do
timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary --dw-wgs $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dw-wgs $w', round(d['value'],1), round(1e3*d['ms_per_step'],2), [(x['launch'].split('+')[0], x['avg_us']) for x in d['roofline']['launches'] if x['launch'][:2] in ('dw','ad')])"
done; done 2>&1 | tee gpurun_out/r06_k_dw_wgs_cfg5.txt
