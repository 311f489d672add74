"""Config 1 (README.md:26-46): SimpleGridWorld, Chain(Dense(2,32), Dense(32,4)) dueling + double-Q + prioritized replay, B = 32.
Prints train steps/s (hipGraph replay) and the device-resident env loop rate with 256 copies."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
envs = importlib.import_module(pkg.__name__ + ".envs")
net = nn.create_dueling_network(nn.Chain(nn.Dense(2, 32), nn.Dense(32, 4)))
layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=2, obs_h=1, obs_w=1, gamma=0.95, buffer_size=10000)
eng = pkg.Engine(layers, hp)
eng.set_params(nn.glorot_params(net, seed=1), pkg.NET_ONLINE)
eng.sync_target()
eng.envs_create(envs.SimpleGridWorld(n=256), max_episode_length=100, seed=1)
eng.rollout(40, t0=1, train_freq=0, eps=(1.0, 1.0, 1.0), stats=False)      # 10 240 transitions
eng.train_steps(50); eng.sync()
t0 = time.perf_counter(); eng.train_steps(2000); eng.sync(); dt = time.perf_counter() - t0
print(f"config 1: {2000 / dt:.0f} train steps/s ({dt / 2000 * 1e6:.1f} us/step)")
t0 = time.perf_counter(); eng.rollout(1000, t0=41, train_freq=0, stats=False); eng.sync(); dt = time.perf_counter() - t0
print(f"device env loop, 256 GridWorld copies, acting only: {dt / 1000 * 1e6:.1f} us per vector step = {256 * 1000 / dt:.0f} env steps/s")
t0 = time.perf_counter(); st = eng.rollout(1000, t0=1041, train_freq=4); dt = time.perf_counter() - t0
print(f"with train_freq=4: {dt / 1000 * 1e6:.1f} us per vector step = {256 * 1000 / dt:.0f} env steps/s + {st['train_steps'] / dt:.0f} train steps/s")
if "--profile" in sys.argv:
    acc = {}
    for _ in range(20):
        for name, ms in eng.profile_step(steady=True):
            a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
    print("per-launch (HIP events, eager, steady-state step):")
    for k, (t, c) in acc.items():
        print(f"    {k:36s} {t / c * 1e3:7.2f} us")
