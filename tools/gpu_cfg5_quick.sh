#!/bin/bash
# usage (GPU box, repo root): tools/gpu_cfg5_quick.sh <tag> [env...]   -- config-5 bench line + per-launch table (no rocprof)
tag=$1
timeout 200 python bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 > gpurun_out/${tag}_cfg5_bench.json 2>gpurun_out/${tag}_cfg5_bench.err
python tools/bench_summary.py gpurun_out/${tag}_cfg5_bench.json | head -22
