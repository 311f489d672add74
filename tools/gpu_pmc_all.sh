#!/bin/bash
# usage: tools/gpu_pmc_all.sh <tag>   -- the PMC passes quoted by bench.py / profiles/README.md, one rocprofv3 --kernel-trace --pmc run per counter
# group (never combined with other trace domains): HBM bytes (FETCH_SIZE, WRITE_SIZE) and MFMA / LDS activity, for the config-2 bench and
# (cfg5 suffix) the B=512 / 1e6-transition u8 run.
tag=$1
bash tools/gpu_pmc.sh ${tag}_fetch "FETCH_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_write "WRITE_SIZE" --env-steps 0
bash tools/gpu_pmc.sh ${tag}_sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --env-steps 0
R=$(pwd); cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${tag}_c5 -o r -- python $R/bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 6 --warmup 2 --profile-steps 1 --no-cpu-baseline --env-steps 0 --no-graph > $R/gpurun_out/${tag}_c5.log 2>&1
  python $R/tools/rocprof_pmc.py $R/gpurun_out/${tag}_c5/r_results.db > $R/gpurun_out/${tag}_cfg5_${c}_pmc.txt
  rm -rf $R/gpurun_out/${tag}_c5
done
