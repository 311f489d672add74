#!/bin/bash
mkdir -p gpurun_out
DQN_DW_BIAS_FIRST=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "u8 or config5 or wide_sample or four_columns or fuzz or fixtures or 512 or 32x32 or weights_resident" 2>&1 | grep -E "^E  |passed|failed|rror" | tail -4
run() { # label, env...
  lbl=$1; shift
  env "$@" timeout 300 python bench.py --batch 512 --u8 --replay 200000 --device-fill --steps 100 --warmup 10 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null > gpurun_out/_ab.json
  python - "$lbl" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/_ab.json").read().strip().splitlines()[-1])
L = d["roofline"]["launches"]
print("%-26s %8.1f steps/s  %7.2f us  " % (sys.argv[1], d["value"], 1e3 * d["ms_per_step"]) + "  ".join("%s %.1f" % (x["launch"].split("+")[0].replace("fwd_",""), x["avg_us"]) for x in L))
PY
}
B=$PWD/deepqlearning.jl_amd/build/base.so
for i in 1 2; do
  run base DQN_MI355X_LIB=$B
  run "in-tree (dw128, equal wres)" A=1
  run "wres prop" DQN_WRES_PROP=1
  run "bias first" DQN_DW_BIAS_FIRST=1
  run "dw64" DQN_DW_SPLIT=64
  run "dw96" DQN_DW_SPLIT=96
  run "dw192" DQN_DW_SPLIT=192
done 2>&1 | tee gpurun_out/r06_h_ab.txt
