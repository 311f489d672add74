#!/bin/bash
# r05_z: the first layer's dW inside the Adam launch (k_adam_pg_dw): parity, then same-box A/B (DQN_NO_DW_IN_ADAM=1 = two launches)
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_layer_dw_inside or pipelined_gather or soak or distinct" 2>&1 | grep -E "^E|passed|failed|Error" | tail -8
for i in 1 2 3; do
for k in "" 1; do
  DQN_NO_DW_IN_ADAM=$k timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 2 --per-call-steps 0 --no-secondary --env-steps 0 2>/dev/null > gpurun_out/z.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/z.json").read().strip().splitlines()[-1])
L={x["launch"]:x["avg_us"] for x in d["roofline"]["launches"]}
print("no_dw_in_adam=%-2s" % "${k:-0}", "%.1f steps/s  sustained %.1f" % (d["value"], d["sustained"]["value"]), "  ".join("%s %.1f" % (k2, L[k2]) for k2 in L if k2.startswith("dw_conv0") or k2.startswith("adam")))
PY
done; done 2>&1 | tee gpurun_out/r05_z_dw_in_adam_ab.txt
