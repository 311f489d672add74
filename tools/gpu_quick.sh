#!/bin/bash
# usage (GPU box, repo root): tools/gpu_quick.sh <tag> [pytest -k expr]   -- the iteration loop of a kernel change: GPU parity suite (stop at the first
# failure), the config-2 bench line without the CPU baseline, the dispatch list of one steady-state step.  Everything lands in gpurun_out/<tag>_*.
tag=$1; kexpr=$2
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then python -m pytest tests -q -m gpu -x -k "$kexpr" > gpurun_out/${tag}_pytest.log 2>&1; else python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_pytest.log 2>&1; fi
grep -E "^E |passed|failed|Error" gpurun_out/${tag}_pytest.log | head -20
python bench.py --no-cpu-baseline --env-steps 0 --sustained-steps 0 > gpurun_out/${tag}_bench.json 2>gpurun_out/${tag}_bench.err; python tools/bench_summary.py gpurun_out/${tag}_bench.json | head -2
bash tools/gpu_profile.sh $tag --replay 10000 --env-steps 0 --sustained-steps 0 > /dev/null
cat gpurun_out/${tag}_step.txt
