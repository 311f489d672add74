"""Profile target: the device-resident env loop alone (config-3 shape: 32 envs/rank of the 84x84x4 image MDP, Nature-DQN dueling).
usage: rocprofv3 --kernel-trace --stats -- python tools/rollout_profile.py [--graph] [--steps N] [--train-freq F]"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--envs", type=int, default=32)
ap.add_argument("--train-freq", type=int, default=0)
ap.add_argument("--graph", action="store_true")
ap.add_argument("--u8", action="store_true")
args = ap.parse_args()
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
envs = importlib.import_module(pkg.__name__ + ".envs")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4))
layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, obs_dtype=pkg.OBS_U8 if args.u8 else pkg.OBS_F32, gamma=0.99,
                         buffer_size=10000, use_graph=1 if args.graph else 0)
eng = pkg.Engine(layers, hp)
eng.set_params(nn.glorot_params(net, seed=1), pkg.NET_ONLINE)
eng.sync_target()
env = envs.TestMDP((84, 84), 4, 6, n=args.envs, seed=7, u8=args.u8)
eng.envs_create(env, seed=3)
eng.rollout(20, t0=1, train_freq=args.train_freq, stats=False)
eng.sync()
t0 = time.perf_counter()
eng.rollout(args.steps, t0=21, train_freq=args.train_freq, stats=False)
eng.sync()
dt = time.perf_counter() - t0
print(f"{dt / args.steps * 1e6:.1f} us per vector step, {args.envs * args.steps / dt:.0f} env steps/s")
