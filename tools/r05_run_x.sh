#!/bin/bash
# r05_x: what the driver's short timed region (--steps 20 --warmup 5) loses against the sustained rate, by warm-up length and region length
for a in "20 5" "20 50" "100 5" "20 5" "20 50" "100 5" "20 5"; do set -- $a
  python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --sustained-seconds 1 --per-call-steps 0 $EXTRA 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2: value %.1f  (%.2f us/step)  sustained %.1f' % (d['value'], d['ms_per_step']*1e3, d['sustained']['value']))"
done
