#!/bin/bash
# usage (GPU box, repo root): tools/gpu_evidence.sh <tag>   -- the evidence set of a round: full -m gpu suite, smoke, rocprofv3 kernel stats and
# one-step dispatch sequence of the config-2 bench, the bench line (default flags and the driver's --steps 20 --warmup 5), PMC passes
tag=$1
python -m pytest tests -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/${tag}_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -2 gpurun_out/${tag}_smoke.log
bash tools/gpu_profile.sh $tag --replay 10000 --env-steps 0
python bench.py > gpurun_out/${tag}_bench.json 2>gpurun_out/${tag}_bench.err; python tools/bench_summary.py gpurun_out/${tag}_bench.json
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_k20.json 2>/dev/null; python tools/bench_summary.py gpurun_out/${tag}_bench_k20.json | head -1
bash tools/gpu_pmc.sh ${tag}_fetch "FETCH_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_write "WRITE_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --env-steps 0
ls gpurun_out | grep $tag
