#!/bin/bash
# r05_u: TIMING PROBES (wrong numbers, right schedule) of the dW body at config 5: build variants -DDQN_DWP=bits (1 no MFMA, 2 no byte conversion, 4 every tile re-reads tile 0,
# 8 no bias sums, 16 no LDS stores, 32 no barriers); per-launch HIP-event durations of the launches that run dw_lds_body
mkdir -p gpurun_out
for v in ${DWPS:-0 1 2 4 8 16 48 6 0}; do
  DQN_MI355X_LIB=$PWD/deepqlearning.jl_amd/build/dwp$v.so timeout 200 python bench.py --batch 512 --u8 --replay 100000 --device-fill --steps 30 --warmup 5 --no-cpu-baseline --env-steps 0 --sustained-seconds 0 --no-secondary 2>/dev/null > gpurun_out/dwp$v.json
  python - <<PY
import json
d=json.loads(open("gpurun_out/dwp$v.json").read().strip().splitlines()[-1])
L={x["launch"]:x["avg_us"] for x in d["roofline"]["launches"]}
print("DWP=%-3s" % "$v", "  ".join("%s %.1f" % (k.split("+")[0], L[k]) for k in L if k.startswith("dw")), " step %.1f us" % (d["ms_per_step"]*1e3))
PY
done 2>&1 | tee gpurun_out/r05_u_dw_probes.txt
