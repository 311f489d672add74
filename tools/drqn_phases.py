"""TIMING PROBE: phase boundaries of the fused recurrent step (drqn_cols.hip) at BASELINE config 4, from s_memrealtime stamps of workgroup 0.
usage (GPU box): DQN_DRQN_STAMPS=1 python tools/drqn_phases.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

os.environ["DQN_DRQN_STAMPS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn"); envs = importlib.import_module(pkg.__name__ + ".envs"); S = importlib.import_module(pkg.__name__ + ".solver")
model = nn.Chain(nn.flattenbatch, nn.LSTM(25, 32), nn.Dense(32, 4))
layers, _ = nn.lower(model)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=1, obs_h=5, obs_w=5, gamma=0.99, double_q=1, dueling=0, prioritized_replay=0, buffer_size=1000, recurrence=1, trace_length=8, learning_rate=1e-3)
eng = pkg.Engine(layers, hp)
eng.set_params(nn.glorot_params(model, seed=1), pkg.NET_ONLINE); eng.sync_target()
replay = S.HIPEpisodeReplayBuffer(eng)
S.populate_episode_replay(replay, envs.TestMDP((5, 5), 1, 6, n=1, seed=7), max_pop=400, rng=np.random.default_rng(0))
eng.train_steps(50); eng.sync()
names = ["params + episode rows", "input projections (registers)", "recurrence (T steps, all sets)", "heads", "TD / Huber", "head dX", "BPTT", "dW slab"]
lib = pkg.lib(); f = lib.dqn_debug_drqn_stamps; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]; f.restype = C.c_int
acc = np.zeros(len(names))
for _ in range(20):
    eng.train_steps(1); eng.sync()
    buf = (C.c_uint64 * 32)(); assert f(eng._h, buf, 32) == 0
    st = np.array(buf[:len(names) + 1], np.float64) * 0.01          # 100 MHz -> us
    acc += np.diff(st)
for nme, v in zip(names, acc / 20):
    print(f"  {nme:34s} {v:7.2f} us")
print(f"  {'workgroup 0 total':34s} {acc.sum() / 20:7.2f} us")
