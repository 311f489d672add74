#!/bin/bash
# usage: tools/gpu_pmc_cfg5.sh <tag> "<counters>"   -- one rocprofv3 --kernel-trace --pmc pass of the config-5 bench (B=512, u8, 1e6 transitions)
tag=$1; ctrs=$2
R=$(pwd); mkdir -p $R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs -d $R/gpurun_out/$tag -o r -- python $R/bench.py --batch 512 --u8 --replay 1000000 --device-fill --steps 6 --warmup 3 --profile-steps 1 --no-cpu-baseline --env-steps 0 --no-graph > $R/gpurun_out/${tag}.log 2>&1
python $R/tools/rocprof_pmc.py $R/gpurun_out/$tag/r_results.db | grep -v "^columns" > $R/gpurun_out/${tag}_pmc.txt
rm -rf $R/gpurun_out/$tag
