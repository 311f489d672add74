"""PROBE: per-launch HIP-event durations of the config-2 train step with the Adam job in its own launch (DQN_ADAM_MODE=0, default) vs carried as tail workgroups of the
backward launches as soon as a layer's gradient is final (DQN_ADAM_MODE=1).  usage (GPU box): python tools/adam_mode_probe.py"""
import importlib
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import __graft_entry__ as ge
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
    eng.train_steps(200); eng.sync()
    acc = {}
    order = []
    for _ in range(40):
        for name, ms in eng.profile_step(steady=True):
            if name not in acc:
                acc[name] = []; order.append(name)
            acc[name].append(ms * 1e3)
    tot = 0.0
    for n in order:
        m = float(np.median(acc[n])); tot += m
        print(f"    {n:44s} {m:7.2f} us")
    print(f"    {'sum':44s} {tot:7.2f} us")
    import time
    eng.train_steps(300); eng.sync()
    t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
    print(f"    train_steps(3000): {3000 / dt:8.1f} steps/s ({dt / 3000 * 1e6:.2f} us/step)")
else:
    for mode in (sys.argv[1:] or ["0", "2", "0", "2"]):
        print(f"DQN_ADAM_MODE={mode}")
        env = dict(os.environ, DQN_ADAM_MODE=mode)
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, timeout=200)
