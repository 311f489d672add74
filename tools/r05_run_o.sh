#!/bin/bash
timeout 300 python -m pytest tests/test_envs_gpu.py tests/test_solve_gpu.py -q -m gpu -x > gpurun_out/r05_o_pytest1.log 2>&1; tail -4 gpurun_out/r05_o_pytest1.log
R=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_o_act -o r -- python $R/tools/rollout_profile.py --steps 200 > $R/gpurun_out/r05_o_act.log 2>&1
python $R/tools/rocprof_act_step.py $R/gpurun_out/r05_o_act/r_results.db | tee $R/gpurun_out/r05_o_act_step.txt
rm -rf $R/gpurun_out/r05_o_act; cd $R
for i in 1 2; do DQN_NO_RED_HEAD=1 timeout 60 python tools/rollout_profile.py --steps 400 --graph | tail -1 | sed 's/^/no_red_head /'; timeout 60 python tools/rollout_profile.py --steps 400 --graph | tail -1 | sed 's/^/in-tree     /'; done
