for env in "DQN_MID_GROUP=4" "DQN_MID_GROUP=6" "DQN_MID_GROUP=9" "DQN_MID_GROUP=18" "DQN_MID_GROUP=3" "DQN_MID_GROUP=4" "DQN_MID_GROUP=6" "DQN_MID_GROUP=9" "DQN_MID_GROUP=18" "DQN_MID_GROUP=3"; do
  env $env python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustained-seconds 1 --per-call-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timed_region_us']; print('$env: value %.1f  (%.2f us/step)  call %.1f   sustained %.1f' % (d['value'], d['ms_per_step']*1e3, t['dqn_train_steps_call'], d['sustained']['value']))"
done
