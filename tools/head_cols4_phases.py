"""TIMING PROBE: phases of k_head_cols4 (red_head.hip) at BASELINE config 5's batch (B = 512), from s_memrealtime stamps (100 MHz).
usage (GPU box): python tools/head_cols4_phases.py [batch]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

os.environ["DQN_DRQN_STAMPS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4))
layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=B, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=4096, seed=1, obs_dtype=pkg.OBS_U8)
eng = pkg.Engine(layers, hp)
p = nn.glorot_params(net, seed=1); eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p, pkg.NET_TARGET)
rng = np.random.default_rng(0)
for _ in range(8):
    s = rng.integers(0, 256, (256, 4, 84, 84), dtype=np.uint8)
    eng.replay_add(s, rng.integers(0, 4, 256).astype(np.int32), rng.standard_normal(256).astype(np.float32), s, np.zeros(256, np.uint8))
eng.train_steps(20); eng.sync()
print([n for n, _ in eng.profile_step()])
lib = pkg.lib(); f = lib.dqn_debug_drqn_stamps; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]; f.restype = C.c_int
A = []
for _ in range(20):
    eng.train_steps(3); eng.sync()
    buf = (C.c_uint64 * 32)(); assert f(eng._h, buf, 32) == 0
    A.append(np.array(buf[:18], np.float64) * 0.01)
A = np.array(A)
pn = ["1: loads -> LDS (wg 0)", "2: chunk chains", "3: chunk sums + bias + act", "4: Q columns + TD (one lane per column)", "5: head dX + stores"]
for i, n in enumerate(pn):
    print(f"  {n:44s} {np.median(A[:, i + 1] - A[:, i]):6.2f} us")
print(f"  workgroup 0 entry -> end                     {np.median(A[:, 5] - A[:, 0]):6.2f} us")
