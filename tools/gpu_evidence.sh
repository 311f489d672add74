#!/bin/bash
# usage (GPU box, repo root): tools/gpu_evidence.sh <tag> [quick]   -- a round's evidence set, written under gpurun_out/ with the names profiles/ uses (copy them there):
#   <tag>_pytest.log / _smoke.log                      full -m gpu suite, __graft_entry__.smoke()
#   <tag>_kernel_trace_summary.txt / _one_step_dispatches.txt   rocprofv3 --kernel-trace --stats of the config-2 bench (eager launches): per-kernel table, one step's dispatch list
#   <tag>_bench.json / _bench_driver_flags.json        bench.py with its defaults / with the driver's --gpus 1 --steps 20 --warmup 5
#   <tag>_pmc_fetch.txt / _pmc_write.txt / _pmc_sq.txt  separate --pmc passes of the config-2 bench (never combined with another trace domain)
#   <tag>_cfg5_kernels.txt / _cfg5_one_step.txt / _cfg5_bench.json / _cfg5_pmc_{fetch,write,sq}.txt   the same for config 5 (B = 512, u8 replay of 1e6 transitions)
# `quick` skips the test suite and the default-flag bench (A/B evidence between commits).
tag=$1; quick=$2
mkdir -p gpurun_out
if [ -z "$quick" ]; then
  python -m pytest tests -q -m gpu > gpurun_out/${tag}_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/${tag}_pytest.log
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -1 gpurun_out/${tag}_smoke.log
fi
bash tools/gpu_profile.sh $tag --replay 10000 --env-steps 0 > /dev/null
mv gpurun_out/${tag}_summary.txt gpurun_out/${tag}_kernel_trace_summary.txt; mv gpurun_out/${tag}_step.txt gpurun_out/${tag}_one_step_dispatches.txt; rm -f gpurun_out/${tag}.log
cat gpurun_out/${tag}_one_step_dispatches.txt
if [ -z "$quick" ]; then python bench.py > gpurun_out/${tag}_bench.json 2>gpurun_out/${tag}_bench.err; python tools/bench_summary.py gpurun_out/${tag}_bench.json | head -3; fi
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench_driver_flags.json 2>/dev/null; python tools/bench_summary.py gpurun_out/${tag}_bench_driver_flags.json | head -1
for c in fetch:FETCH_SIZE write:WRITE_SIZE "sq:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  n=${c%%:*}; ctr=${c#*:}
  bash tools/gpu_pmc.sh ${tag}_x_$n "$ctr" --env-steps 0; grep -v "^columns" gpurun_out/${tag}_x_${n}_pmc.txt > gpurun_out/${tag}_pmc_$n.txt; rm -f gpurun_out/${tag}_x_${n}_pmc.txt gpurun_out/${tag}_x_$n.log
  bash tools/gpu_pmc_cfg5.sh ${tag}_y_$n "$ctr"; mv gpurun_out/${tag}_y_${n}_pmc.txt gpurun_out/${tag}_cfg5_pmc_$n.txt; rm -f gpurun_out/${tag}_y_$n.log
done
bash tools/gpu_cfg5.sh $tag | tail -14
rm -f gpurun_out/${tag}.log
ls gpurun_out | grep "^$tag"
