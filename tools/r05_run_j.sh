#!/bin/bash
python -m pytest tests -q -m gpu > gpurun_out/r05_j_pytest.log 2>&1; grep -E "passed|failed" gpurun_out/r05_j_pytest.log; grep -E "^FAILED|^E  " gpurun_out/r05_j_pytest.log | head -10
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_j_bench_k20.json 2>gpurun_out/r05_j_bench.err; tail -2 gpurun_out/r05_j_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r05_j_bench_k20.json"))
print("value", d["value"], "sustained", d["sustained"]["value"], "value_distinct", d["value_distinct"])
print("env_loop", d["env_loop"])
print("cpu", {k:v for k,v in d["cpu_baseline"].items() if k in ("value","cores","port_value","port_cores","p90_over_p10","medians_ms")}, d["cpu_baseline"]["torch_cpu"].get("runs_steps_per_s"))
for k,v in d["secondary"].items():
    if isinstance(v, dict): print(k, {a:b for a,b in v.items() if a in ("steps_per_s","ms_per_step","per_call_sync_us","error")}, v.get("roofline",{}).get("frac"))
print("dominant", d["roofline"]["dominant_kernel"])
PY
