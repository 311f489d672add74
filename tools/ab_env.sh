#!/bin/bash
# usage (GPU box, repo root): tools/ab_env.sh <VAR> [bench args...]   -- same-box A/B of one engine switch on the config-2 bench: sustained steps/s and the per-launch table,
# twice each, alternating (boxes differ by a few per cent; only same-box comparisons decide)
var=$1; shift
for rep in 1 2; do for v in 0 1; do
  if [ $v = 1 ]; then export $var=1; else unset $var; fi
  python bench.py --no-cpu-baseline --env-steps 0 --sustained-seconds 3 --per-call-steps 0 "$@" 2>/dev/null > /tmp/ab.json
  python - <<PY
import json; d=json.load(open("/tmp/ab.json")); L={r["launch"]: r["avg_us"] for r in d["roofline"]["launches"]}
print("$var=$v  sustained %.1f steps/s (%.2f us/step)  head_td %.2f  reduce %.2f  eager %.1f" % (d["sustained"]["value"], 1e6/d["sustained"]["value"], L.get("head_td",0), L.get("fwd_reduce_dense3",0), d["roofline"]["eager_step_us"]))
PY
done; done
