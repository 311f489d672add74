#!/bin/bash
# r05_v: dedicated db workgroups in the dW launches at large batches (db_chunk_body): parity, then same-box A/B at config 5 (DQN_NO_DB_BLOCKS=1 = bias sums inside the dW workgroups)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "u8 or config5 or wide_sample or four_columns or 32x32 or weights_resident or fuzz" 2>&1 | grep -E "^E|passed|failed|Error" | tail -5
for i in 1 2 3; do
for k in "" 1; do
  DQN_NO_DB_BLOCKS=$k timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null > gpurun_out/v_$k.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/v_$k.json").read().strip().splitlines()[-1])
L={x["launch"]:x["avg_us"] for x in d["roofline"]["launches"]}
print("no_db_blocks=%-2s" % "${k:-0}", "%.1f steps/s" % d["value"], "  ".join("%s %.1f" % (k.split("+")[0], L[k]) for k in L if k.startswith("dw")))
PY
done; done 2>&1 | tee gpurun_out/r05_v_cfg5_db_blocks_ab.txt
