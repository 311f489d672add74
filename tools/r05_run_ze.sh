for k in "" 1 "" 1; do
DQN_NO_DW_IN_ADAM=$k timeout 300 python bench.py --no-graph --no-cpu-baseline --sustained-seconds 1 --per-call-steps 0 --no-secondary --env-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager no_dw_in_adam=${k:-0}: %.1f steps/s, sustained %.1f' % (d['value'], d['sustained']['value']))"
done
