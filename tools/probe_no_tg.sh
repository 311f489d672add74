#!/bin/bash
# TIMING PROBE (GPU box): upper bound of hiding the target-network forward -- the steady-state step with the target problems dropped from the forward
# launches (wrong numbers, right schedule) against the normal step.  usage: tools/probe_no_tg.sh <tag>
tag=$1
for v in 0 1; do
  if [ $v = 1 ]; then export DQN_PROBE_NO_TG=1; else unset DQN_PROBE_NO_TG; fi
  python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 2 --per-call-steps 0 > gpurun_out/${tag}_notg${v}.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/${tag}_notg${v}.json")); print("NO_TG=$v value", round(d["value"],1), "sustained", round(d["sustained"]["value"],1), "us/step", round(1e6/d["sustained"]["value"],2))
for r in d["roofline"]["launches"]: print("   ", r["launch"], r["avg_us"])
PY
done
