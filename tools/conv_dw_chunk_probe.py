"""PROBE: positions per split-K chunk of the conv layers' dW at config 2 (plan.dw_kc = positions x B), one layer at a time: train_steps rate and the dW / Adam launch
durations.  usage (GPU box): python tools/conv_dw_chunk_probe.py"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4))
layers, _ = nn.lower(net)
B = 32
hp = pkg.default_hparams(batch_size=B, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=2000, seed=1)
rng = np.random.default_rng(0)
S_ = rng.random((2048, 4, 84, 84), dtype=np.float32)
A_ = rng.integers(0, 4, 2048).astype(np.int32); R_ = rng.standard_normal(2048).astype(np.float32); D_ = np.zeros(2048, np.uint8)
p = nn.glorot_params(net, seed=1)
base = pkg.default_plan(layers, hp)
print("default plan (fwd_kc, dx_kc, dw_kc):", base[:3])
cases = [("default", {})] + [(f"conv1 {k} pos", {0: k}) for k in (4, 5, 6, 8)] + [(f"conv3 {k} pos", {2: k}) for k in (2,)] + [("default", {})]
if os.environ.get("PROBE_CASES"):      # e.g. PROBE_CASES="1:2,1:4,2:4,1:3+2:4" (layer index : positions per chunk; + joins layers of one case)
    cases = [("default", {})] + [(c, {int(x.split(":")[0]): int(x.split(":")[1]) for x in c.split("+")}) for c in os.environ["PROBE_CASES"].split(",")] + [("default", {})]      # "1:3+2:4" = both at once
for rep in range(2):
    for name, chg in cases:
        plan = [(q[0], q[1], chg[i] * B) if i in chg else q for i, q in enumerate(base)]
        eng = pkg.Engine(layers, hp, plan=plan)
        eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p, pkg.NET_TARGET)
        eng.replay_add(S_, A_, R_, S_, D_)
        eng.train_steps(300); eng.sync()
        acc = {}
        for _ in range(30):
            for n, ms in eng.profile_step(steady=True):
                acc.setdefault(n, []).append(ms * 1e3)
        eng.train_steps(100); eng.sync()
        t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
        sel = [n for n in acc if n.startswith("dw") or n.startswith("adam")]
        print(f"{name:14s} {3000 / dt:8.1f} steps/s   " + "  ".join(f"{n.split('+')[0]} {np.median(acc[n]):.2f}" for n in sel))
        eng.close()
