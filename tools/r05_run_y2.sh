for env in "" "DQN_MID_BIG=0" "DQN_MID_GROUP=1" "" "DQN_MID_BIG=0" "DQN_MID_GROUP=1"; do
  env $env python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --sustained-seconds 1 --per-call-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['timed_region_us']; print('${env:-default}: value %.1f  (%.2f us/step)  call %.1f sync %.1f torch %.1f   sustained %.1f' % (d['value'], d['ms_per_step']*1e3, t['dqn_train_steps_call'], t['engine_stream_sync'], t['torch_cuda_synchronize'], d['sustained']['value']))"
done
