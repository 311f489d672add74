"""Config 4: TestMDP((5,5),1,6) observations (25), Chain(flattenbatch, LSTM(25,32), Dense(32,4)) (benchmark/flux_dqn.jl:35-36),
trace_length 8, B = 32, double-Q, no dueling.  Prints DRQN train steps/s (hipGraph replay, sampler on the device) and the per-launch table."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=500)
ap.add_argument("--hidden", type=int, default=32)
ap.add_argument("--trace", type=int, default=8)
ap.add_argument("--profile", action="store_true")
ap.add_argument("--no-mfma", action="store_true")
args = ap.parse_args()
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
envs = importlib.import_module(pkg.__name__ + ".envs")
S = importlib.import_module(pkg.__name__ + ".solver")
model = nn.Chain(nn.flattenbatch, nn.LSTM(25, args.hidden), nn.Dense(args.hidden, 4))
layers, _ = nn.lower(model)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=1, obs_h=5, obs_w=5, gamma=0.99, double_q=1, dueling=0, prioritized_replay=0,
                         buffer_size=1000, recurrence=1, trace_length=args.trace, learning_rate=1e-3, use_mfma=0 if args.no_mfma else 1)
eng = pkg.Engine(layers, hp)
eng.set_params(nn.glorot_params(model, seed=1), pkg.NET_ONLINE)
eng.sync_target()
env = envs.TestMDP((5, 5), 1, 6, n=1, seed=7)
replay = S.HIPEpisodeReplayBuffer(eng)
S.populate_episode_replay(replay, env, max_pop=400, rng=np.random.default_rng(0))
eng.train_steps(20)
eng.sync()
t0 = time.perf_counter()
loss, gn = eng.train_steps(args.steps)
eng.sync()
dt = time.perf_counter() - t0
print(f"config 4 DRQN: {args.steps / dt:.0f} train steps/s ({dt / args.steps * 1e6:.1f} us/step), loss {loss:.4g}")
if args.profile:
    acc = {}
    for _ in range(5):
        for name, ms in eng.profile_step():
            a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:28s} x{v[1] // 5:<3d} {v[0] / 5 * 1e3:8.1f} us/step")
