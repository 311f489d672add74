"""PROBE: is the Adam launch bound by its Float64 arithmetic or by its bytes?  Same launch with hp.adam_f64_scalars = 1 (Flux semantics, default) and 0 (pure fp32)."""
import importlib, os, sys, argparse
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
for f64 in (1, 0, 1, 0):
    nn = pkg.nn
    net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4)); layers, _ = nn.lower(net)
    hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=10000, seed=1, adam_f64_scalars=f64)
    eng = pkg.Engine(layers, hp); p = nn.glorot_params(net, seed=1); eng.set_params(p, 0); eng.set_params(p, 1)
    env = pkg.envs.TestMDP((84, 84), 4, 6, n=1024, seed=7); eng.envs_create(env, n_envs=1024, max_episode_length=100, seed=1)
    eng.rollout(10, t0=1, train_freq=0, target_update_freq=0, eps=(1.0, 1.0, 1.0), stats=False); eng.sync()
    eng.train_steps(50); eng.sync()
    acc = {}
    for _ in range(20):
        for name, ms in eng.profile_step(steady=True):
            a = acc.setdefault(name, [0.0, 0]); a[0] += ms; a[1] += 1
    import time
    t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
    print(f"adam_f64_scalars={f64}: adam+gather {acc['adam+gather'][0] / acc['adam+gather'][1] * 1e3:.2f} us (HIP events), {3000 / dt:.0f} steps/s")
    eng.close()
