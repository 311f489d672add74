"""TIMING PROBE: phases of the fused reduce + head launch (red_head.hip) at BASELINE config 2, from s_memrealtime stamps (100 MHz).
usage (GPU box): python tools/red_head_phases.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

os.environ["DQN_DRQN_STAMPS"] = "1"
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
eng.train_steps(50); eng.sync()
lib = pkg.lib(); f = lib.dqn_debug_drqn_stamps; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]; f.restype = C.c_int
A = []
for _ in range(30):
    eng.train_steps(3); eng.sync()
    buf = (C.c_uint64 * 32)(); assert f(eng._h, buf, 32) == 0
    A.append(np.array(buf[:18], np.float64) * 0.01)
    # re-arm the min/max words
A = np.array(A)
pn = ["A: loads + slab sums + LDS (wg 0)", "B: head chunk chains", "C: drain of the write-through stores", "C: ticket"]
for i, n in enumerate(pn):
    print(f"  {n:44s} {np.median(A[:, i + 1] - A[:, i]):6.2f} us")
ln = ["entry -> ticket (group 0's last arriver)", "D: partials + y loads + chunk sums", "(unused)", "D: Q columns + TD (one lane per column)", "D: head dX + stores"]
for i, n in enumerate(ln):
    print(f"  {n:44s} {np.median(A[:, i + 9] - A[:, i + 8]):6.2f} us")
print(f"  last arriver of group 0, entry -> end        {np.median(A[:, 13] - A[:, 8]):6.2f} us")
