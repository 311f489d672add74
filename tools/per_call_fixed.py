"""PROBE: the fixed cost of one dqn_train_steps(n) call at config 2 (what the driver's --steps 20 run pays once per 20 steps): T(n) = a + b n by least squares over
n in {1 .. 200}, and the cost of the two synchronisations the bench contract brackets the timed region with.  usage (GPU box): python tools/per_call_fixed.py"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4))
layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=2000, seed=1)
eng = pkg.Engine(layers, hp)
p = nn.glorot_params(net, seed=1); eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p, pkg.NET_TARGET)
rng = np.random.default_rng(0)
for _ in range(8):
    s = rng.random((256, 4, 84, 84), dtype=np.float32)
    eng.replay_add(s, rng.integers(0, 4, 256).astype(np.int32), rng.standard_normal(256).astype(np.float32), s, np.zeros(256, np.uint8))
torch.zeros(1, device="cuda")
eng.train_steps(500); eng.sync()
ns = [1, 2, 3, 5, 10, 20, 40, 80, 200]
T = {}
for n in ns:
    ts = []
    for _ in range(30):
        eng.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter(); eng.train_steps(n); t1 = time.perf_counter(); eng.sync(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2))
    a = np.median(np.array(ts), axis=0) * 1e6
    T[n] = a
    print(f"n = {n:4d}: train_steps {a[0]:9.1f} us ({a[0] / n:7.2f} per step)   + eng.sync {a[1]:5.1f} us   + torch.cuda.synchronize {a[2]:5.1f} us   -> bench clock {(a.sum()) / n:7.2f} us/step")
x = np.array(ns[3:], float); y = np.array([T[n][0] for n in ns[3:]])
b, a = np.polyfit(x, y, 1)
print(f"fit over n >= 5: T(n) = {a:.1f} us + {b:.2f} us * n")
