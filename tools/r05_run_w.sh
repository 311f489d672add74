#!/bin/bash
# r05_w: three register stages in the dW body for long chunks: parity, then same-box A/B vs the previous build (build/base2.so) at config 5 and config 2
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "u8 or config5 or wide_sample or four_columns or 32x32 or weights_resident or fuzz or fixtures" 2>&1 | grep -E "^E|passed|failed|Error" | tail -5
V=$PWD/deepqlearning.jl_amd/build/base2.so
for i in 1 2 3; do
for so in "" "$V"; do
  DQN_MI355X_LIB=$so timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null > gpurun_out/w.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/w.json").read().strip().splitlines()[-1])
L={x["launch"]:x["avg_us"] for x in d["roofline"]["launches"]}
print("${so:+base    }${so:-in-tree }", "%.1f steps/s" % d["value"], "  ".join("%s %.1f" % (k.split("+")[0], L[k]) for k in L if k.startswith("dw")))
PY
done; done 2>&1 | tee gpurun_out/r05_w_cfg5_dw_stages_ab.txt
bash tools/ab_lib.sh deepqlearning.jl_amd/build/base2.so 2>&1 | head -6
