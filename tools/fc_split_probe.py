"""PROBE: the split-K depth of the hidden dense layers' forward (plan.fwd_kc) at config 2, now that the slabs are reduced by the chip-filling k_red_head launch:
train_steps rate and the FC forward / red_head launch durations for S = 4 / 7 (default) / 10 / 14.  usage (GPU box): python tools/fc_split_probe.py"""
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
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=2000, seed=1)
rng = np.random.default_rng(0)
S_ = rng.random((2048, 4, 84, 84), dtype=np.float32)
A_ = rng.integers(0, 4, 2048).astype(np.int32); R_ = rng.standard_normal(2048).astype(np.float32); D_ = np.zeros(2048, np.uint8)
p = nn.glorot_params(net, seed=1)
for rep in range(2):
    for kc in (448, 224, 320, 896, 448):
        plan = pkg.default_plan(layers, hp)
        plan = [(kc, q[1], q[2]) if q[0] == 448 else q for q in plan]      # (fwd_kc, dx_kc, dw_kc) per layer
        eng = pkg.Engine(layers, hp, plan=plan)
        eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p, pkg.NET_TARGET)
        eng.replay_add(S_, A_, R_, S_, D_)
        eng.train_steps(300); eng.sync()
        acc = {}
        for _ in range(30):
            for name, ms in eng.profile_step(steady=True):
                acc.setdefault(name, []).append(ms * 1e3)
        eng.train_steps(100); eng.sync()
        t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
        names = list(acc)
        fc = [n for n in names if n.startswith("fwd_dense")]; hd = [n for n in names if "head" in n or "reduce" in n]
        print(f"fwd_kc {kc:4d} (S = {-(-3136 // kc):2d}): {3000 / dt:8.1f} steps/s   " + "  ".join(f"{n} {np.median(acc[n]):.2f}" for n in fc + hd) + f"   sum {sum(np.median(v) for v in acc.values()):.1f} us")
        eng.close()
