for a in "20 5" "20 5" "20 5" "300 30" "100 5"; do set -- $a
  python bench.py --gpus 1 --steps $1 --warmup $2 --no-cpu-baseline --no-secondary --sustained-seconds 2 --per-call-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timed_region_us']; print('steps $1 warmup $2: value %.1f  (%.2f us/step)  call %.1f   sustained %.1f' % (d['value'], d['ms_per_step']*1e3, t['dqn_train_steps_call'], d['sustained']['value']))"
done
