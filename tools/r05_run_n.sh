#!/bin/bash
R=$(pwd); mkdir -p gpurun_out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_n_act -o r -- python $R/tools/rollout_profile.py --steps 200 > $R/gpurun_out/r05_n_act.log 2>&1
python $R/tools/rocprof_act_step.py $R/gpurun_out/r05_n_act/r_results.db | tee $R/gpurun_out/r05_n_act_step.txt
rm -rf $R/gpurun_out/r05_n_act; cd $R
tail -1 gpurun_out/r05_n_act.log
python tools/rollout_profile.py --steps 400 --graph | tail -1
