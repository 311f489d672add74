"""End-to-end on the headline configuration: TestMDP((84,84),4,6) image MDP, Nature-DQN dueling, double-Q, prioritized replay, B = 32,
32 device-resident environment copies; the reference's dqn_train! loop through the solve() mirror with device_envs=True.
Prints the evaluation return (optimal 2.1; the reference's test threshold for the small-image variant is 1.5)."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=3000)
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--envs", type=int, default=32)
ap.add_argument("--train-freq", type=int, default=4)
ap.add_argument("--u8", action="store_true")
args = ap.parse_args()
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
envs = importlib.import_module(pkg.__name__ + ".envs")
S = importlib.import_module(pkg.__name__ + ".solver")
env = envs.TestMDP((84, 84), 4, 6, n=args.envs, seed=7, u8=args.u8)
expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=args.steps / 2))
solver = S.DeepQLearningSolver(qnetwork=nn.nature_dqn(n_actions=4, in_channels=4), max_steps=args.steps, learning_rate=args.lr, exploration_policy=expl,
                               train_freq=args.train_freq, target_update_freq=500, eval_freq=1000, num_ep_eval=64, log_freq=500, double_q=True, dueling=True,
                               prioritized_replay=True, buffer_size=20000, train_start=640, verbose=True, logdir=None, device_envs=True,
                               obs_dtype=pkg.OBS_U8 if args.u8 else pkg.OBS_F32)
t0 = time.perf_counter()
policy = S.solve(solver, env)
dt = time.perf_counter() - t0
r, st = policy.engine.evaluate(64, 100, seed=99)
print(f"trained {args.steps} vector steps x {args.envs} envs ({args.steps // args.train_freq} train steps) in {dt:.1f} s; greedy return {r:.3f} over 64 episodes ({st:.1f} steps)")
