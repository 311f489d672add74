#!/bin/bash
# usage (GPU box, repo root): tools/gpu_act_tail.sh <tag>   -- the acting step's tail (act_head.hip) against the four-launch tail (DQN_NO_ACT_HEAD=1):
# env-loop parity tests, one acting vector step's dispatch list under rocprofv3 for both schedules, the env-loop numbers of the bench line for both.
tag=$1; R=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_envs_gpu.py tests/test_solve_gpu.py -q -x > gpurun_out/${tag}_pytest.log 2>&1; tail -5 gpurun_out/${tag}_pytest.log
cd /tmp; export TMPDIR=/tmp
for v in new old; do
  if [ $v = old ]; then export DQN_NO_ACT_HEAD=1; else unset DQN_NO_ACT_HEAD; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_act_$v -o r -- python $R/tools/rollout_profile.py --steps 200 > $R/gpurun_out/${tag}_act_$v.log 2>&1
  python $R/tools/rocprof_act_step.py $R/gpurun_out/${tag}_act_$v/r_results.db > $R/gpurun_out/${tag}_act_step_$v.txt 2>&1; cat $R/gpurun_out/${tag}_act_step_$v.txt
  rm -rf $R/gpurun_out/${tag}_act_$v
done
cd $R
for rep in 1 2; do for v in new old; do
  if [ $v = old ]; then export DQN_NO_ACT_HEAD=1; else unset DQN_NO_ACT_HEAD; fi
  timeout 600 python bench.py --no-cpu-baseline --no-secondary --sustained-steps 0 --per-call-steps 0 > gpurun_out/${tag}_bench_$v$rep.json 2>gpurun_out/${tag}_bench_$v$rep.err
  python - <<P
import json
d=json.load(open("gpurun_out/${tag}_bench_$v$rep.json")); e=d["env_loop"]
print("$v$rep", "value", round(d["value"],1), "act us", round(e["act_only_ms_per_vector_step"]*1e3,1), "loop train/s", round(e["loop_train_steps_per_s"],1), "us/vstep", round(e["loop_ms_per_vector_step"]*1e3,1), "refcadence train/s", round(e["refcadence_train_steps_per_s"],1), "us/vstep", round(e["refcadence_ms_per_vector_step"]*1e3,1))
P
done; done
unset DQN_NO_ACT_HEAD
