"""Soak of the acting step's fused tail (act_head.hip) against the four-launch tail (DQN_NO_ACT_HEAD=1) on the headline configuration: two engines, same seeds, SOAK_STEPS
vector steps each (default 60 000: 1.9 M env steps, a train step every 4 vector steps, some chunks at the reference's env-step cadence, an evaluation per chunk) -- the
two runs must agree BIT FOR BIT on parameters, replay priorities, env state and evaluation results at every checkpoint (a missed hand-off of the ticketed last arriver
would show as a diverging trajectory).  usage (GPU box): python tools/act_tail_soak.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn"); envs = importlib.import_module(pkg.__name__ + ".envs")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4)); layers, _ = nn.lower(net)
hp = pkg.default_hparams(batch_size=32, n_actions=4, obs_c=4, obs_h=84, obs_w=84, gamma=0.99, buffer_size=20000, learning_rate=1e-4)
steps = int(os.environ.get("SOAK_STEPS", "60000")); chunk = 5000


def run(fused):
    if fused: os.environ.pop("DQN_NO_ACT_HEAD", None)
    else: os.environ["DQN_NO_ACT_HEAD"] = "1"
    eng = pkg.Engine(layers, hp); eng.set_params(nn.glorot_params(net, seed=1), 0); eng.sync_target()
    eng.envs_create(envs.TestMDP((84, 84), 4, 6, n=32, seed=7), seed=1)
    assert eng.envs_info() == (32, fused)
    out = []; t = 1; t0 = time.perf_counter()
    for c in range(steps // chunk):
        cad = c % 3 == 2
        n = chunk // 8 if cad else chunk      # (the env-step cadence runs 8 train steps per vector step)
        st = eng.rollout(n, t0=t, train_freq=4, target_update_freq=500 if not cad else 16000, eps=(1.0, 0.01, 20000.0), env_step_cadence=cad); t += n
        ev = eng.evaluate(64, 100, seed=c)
        assert np.isfinite(st["loss"]) and np.isfinite(st["grad_norm"]), st
        out.append((st, ev, eng.get_params(0).copy(), eng.replay_priorities().copy(), [x.copy() for x in eng.envs_peek()]))
    dt = time.perf_counter() - t0
    eng.close()
    return out, dt, t - 1


a, ta, na = run(True)
b, tb, nb = run(False)
assert na == nb
for c, (x, y) in enumerate(zip(a, b)):
    assert x[0] == y[0] and x[1] == y[1], (c, x[0], y[0], x[1], y[1])
    np.testing.assert_array_equal(x[2], y[2]); np.testing.assert_array_equal(x[3], y[3])
    for u, v in zip(x[4], y[4]): np.testing.assert_array_equal(u, v)
print(f"{na} vector steps per schedule ({len(a)} checkpoints; train steps in the last chunk {a[-1][0]['train_steps']}): fused {ta:.1f} s, four-launch tail {tb:.1f} s; "
      f"parameters, priorities, env state, rollout statistics and evaluations identical at every checkpoint; last eval return {a[-1][1][0]:.3f}")
