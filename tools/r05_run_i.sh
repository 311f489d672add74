#!/bin/bash
python -m pytest tests/test_gpu_parity.py tests/test_envs_gpu.py tests/test_solve_gpu.py -q -m gpu -x -k "distinct or sampler or pipelined or envs or solve" > gpurun_out/r05_i_pytest1.log 2>&1; tail -5 gpurun_out/r05_i_pytest1.log
B="python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 100 --no-secondary"
for i in 1 2; do
  $B 2>/dev/null > /tmp/a.json; python -c "import json;d=json.load(open('/tmp/a.json'));print('default  sustained', round(d['sustained']['value'],1), 'per_call sync us', round(d['per_call']['sync']['us_per_call'],1))"
  $B --distinct 2>/dev/null > /tmp/b.json; python -c "import json;d=json.load(open('/tmp/b.json'));print('distinct sustained', round(d['sustained']['value'],1), 'per_call sync us', round(d['per_call']['sync']['us_per_call'],1))"
done 2>&1 | tee gpurun_out/r05_i_distinct_ab.txt
