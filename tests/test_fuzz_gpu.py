"""GPU (-m gpu): seeded random network shapes / batch sizes / options through the engine and the canonical-order CPU twin -- every kernel family
(LDS-tiled MFMA, direct MFMA, VALU fallbacks; with and without split-K; sibling fusion on and off) gets shapes at and off its tile multiples.
Bit-exact: TD errors, loss, grad norm, gradients and parameters after two steps, and the sampled indices."""
import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import ref

pytestmark = pytest.mark.gpu
ACTS = [O.ACT_RELU, O.ACT_TANH, O.ACT_IDENTITY, O.ACT_SIGMOID]


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.lib()
    return p


def random_net(rng):
    conv = rng.random() < 0.6
    layers = []
    if conv:
        c, h, w = int(rng.choice([1, 2, 3, 4])), int(rng.integers(8, 15)), int(rng.integers(8, 15))
        obs = (c, h, w)
        nconv = int(rng.integers(1, 3))
        for _ in range(nconv):
            k = int(rng.choice([2, 3, 4])); st = int(rng.choice([1, 2])); co = int(rng.choice([4, 8, 16, 32]))
            if (h - k) // st + 1 < 2 or (w - k) // st + 1 < 2:
                break
            layers.append(O.Conv(k, c, co, int(rng.choice(ACTS[:2])), st))
            c, h, w = co, (h - k) // st + 1, (w - k) // st + 1
        feat = c * h * w
    else:
        feat = int(rng.choice([2, 6, 25, 33, 64]))
        obs = (feat,)
    for _ in range(int(rng.integers(1, 3))):
        n = int(rng.choice([8, 16, 24, 32, 48, 64, 96]))
        layers.append(O.Dense(feat, n, int(rng.choice(ACTS))))
        feat = n
    nA = int(rng.choice([2, 3, 4, 5, 7]))
    layers.append(O.Dense(feat, nA, O.ACT_IDENTITY))
    dueling = rng.random() < 0.6
    if dueling:
        b, v, a = O.create_dueling_network(layers)
        return O.Network(obs, b, v, a), dueling
    return O.Network(obs, layers), dueling


@pytest.mark.parametrize("seed", list(range(64)))
def test_random_configuration_bit_exact(pkg, seed):
    rng = np.random.default_rng(1000 + seed)
    net, dueling = random_net(rng)
    B = int(rng.choice([1, 3, 8, 16, 17, 32, 48, 64, 96]))
    cap = int(rng.choice([B + 5, 2 * B + 1, 128, 300]))
    kw = dict(batch_size=B, buffer_size=max(cap, B), learning_rate=float(rng.choice([1e-3, 1e-4])), gamma=float(rng.choice([0.9, 0.99])),
              double_q=int(rng.random() < 0.7), prioritized_replay=int(rng.random() < 0.7), obs_dtype=int(rng.random() < 0.3),
              use_mfma=int(rng.random() < 0.8), use_graph=int(rng.random() < 0.7), adam_f64_scalars=int(rng.random() < 0.8), seed=int(rng.integers(0, 1 << 30)))
    hp = ref.hparams_for(net, **kw)
    layers = ref.layers_from_network(net)
    plan = pkg.default_plan(layers, hp)
    g, t = pkg.Engine(layers, hp, plan=plan), ref.Twin(layers, hp, plan=plan, threads=4)
    n_fill = int(hp.buffer_size + rng.integers(0, 20))            # wraps the ring when > capacity
    if hp.obs_dtype:
        s = rng.integers(0, 256, (n_fill,) + net.obs_shape).astype(np.uint8); sp = rng.integers(0, 256, (n_fill,) + net.obs_shape).astype(np.uint8)
    else:
        s = rng.random((n_fill,) + net.obs_shape, dtype=np.float32); sp = rng.random((n_fill,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, n_fill).astype(np.int32); r = (2 * rng.standard_normal(n_fill)).astype(np.float32); d = (rng.random(n_fill) < 0.2).astype(np.uint8)
    p_on = O.Network.flatten(O.init_params(net, seed=seed)); p_on = (p_on + 0.02 * rng.standard_normal(p_on.shape)).astype(np.float32)
    p_tg = (p_on + 0.05 * rng.standard_normal(p_on.shape)).astype(np.float32)
    for h in (g, t):
        h.replay_add(s, a, r, sp, d); h.set_params(p_on, 0); h.set_params(p_tg, 1)
    for step in range(2):
        lg, gg, tdg = g.train_step(); lt, gt, tdt = t.train_step()
        np.testing.assert_array_equal(g.last_indices(), t.last_indices())
        np.testing.assert_array_equal(tdg, tdt)
        assert lg == lt and gg == gt, (lg, lt, gg, gt)
        np.testing.assert_array_equal(g.get_grads(), t.get_grads())
    np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())
    obs = s[:5].astype(np.float32) / np.float32(255) if hp.obs_dtype else s[:5]
    np.testing.assert_array_equal(g.forward(obs), t.forward(obs))
    np.testing.assert_array_equal(g.greedy_action(obs), t.greedy_action(obs))
    g.close(); t.close()


# ------------------------------------------------------------------ DRQN and the device env loop under random shapes
from drqn_common import draws, feed, make_episodes, make_handle      # noqa: E402


@pytest.mark.parametrize("seed", list(range(16)))
def test_random_drqn_configuration_bit_exact(pkg, seed):
    """random LSTM widths / batch / trace lengths: hidden sizes on both sides of the whole-sequence kernels' LDS limit, with and without a dense
    layer in front, dueling or not (src/solver.jl:239-287; src/episode_replay.jl:71-95)."""
    rng = np.random.default_rng(500 + seed)
    n_in = int(rng.choice([4, 6, 16, 25])); H = int(rng.choice([4, 8, 16, 32, 48, 80])); nA = int(rng.choice([2, 3, 4, 5]))
    B = int(rng.choice([2, 4, 8, 16, 32])); T = int(rng.choice([2, 3, 5, 8, 10]))
    base = []
    feat = n_in
    if rng.random() < 0.4:
        feat = int(rng.choice([8, 12, 16])); base.append(O.Dense(n_in, feat, O.ACT_RELU))
    base.append(O.LSTM(feat, H))
    if rng.random() < 0.5:
        net = O.RecurrentNetwork((n_in,), base, [O.Dense(H, 1, O.ACT_IDENTITY)], [O.Dense(H, nA, O.ACT_IDENTITY)])
    else:
        net = O.RecurrentNetwork((n_in,), base + [O.Dense(H, nA, O.ACT_IDENTITY)])
    kw = dict(gamma=float(rng.choice([0.9, 0.99])), double_q=int(rng.random() < 0.7), use_mfma=int(rng.random() < 0.7), use_graph=int(rng.random() < 0.7))
    cap = B + int(rng.integers(1, 6))
    g, hp, layers = make_handle(pkg.Engine, net, B, T, kw, cap=cap)
    t = ref.Twin(layers, hp, plan=g.plan(), threads=4)
    eps = make_episodes(net, cap + 2, T, rng)
    feed(g, eps); feed(t, eps)
    ring = [None] * cap
    for i, ep in enumerate(eps):
        ring[i % cap] = ep
    p_on = (O.Network.flatten(O.init_params_recurrent(net, seed)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    p_tg = (p_on + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    for h in (g, t):
        h.set_params(p_on, 0); h.set_params(p_tg, 1)
    for step in range(3):
        idx, start = draws(ring, B, rng)
        assert g.train_step_drqn(idx, start) == t.train_step_drqn(idx, start)
    np.testing.assert_array_equal(g.get_grads(), t.get_grads())
    np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    g.close(); t.close()


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_device_env_loop_bit_exact(pkg, seed):
    """random env counts / batch / cadence / storage type for the device-resident loop (dqn_rollout, dqn_evaluate) on both built-in MDPs."""
    import importlib
    envs = importlib.import_module(pkg.__name__ + ".envs")
    rng = np.random.default_rng(900 + seed)
    grid = rng.random() < 0.4
    if grid:
        hid = int(rng.choice([8, 16, 32]))
        net = O.Network((2,), *O.create_dueling_network([O.Dense(2, hid, O.ACT_RELU), O.Dense(hid, 4, O.ACT_IDENTITY)]))
        n = int(rng.choice([1, 5, 16, 64])); spec = envs.SimpleGridWorld(n=n, tprob=float(rng.choice([0.7, 1.0]))); u8 = 0
    else:
        hw = (int(rng.integers(5, 10)), int(rng.integers(5, 10))); stack = int(rng.choice([1, 2, 4]))
        feat = stack * hw[0] * hw[1]
        hid = int(rng.choice([8, 16, 32]))
        net = O.Network((stack, hw[1], hw[0]), [O.Dense(feat, hid, O.ACT_TANH), O.Dense(hid, 4, O.ACT_IDENTITY)])
        n = int(rng.choice([1, 3, 8, 32])); spec = envs.TestMDP(hw, stack, int(rng.choice([4, 6])), n=n, seed=int(rng.integers(0, 100))); u8 = int(rng.random() < 0.5)
    B = int(rng.choice([2, 8, 32])); cap = int(max(B, n) * rng.integers(2, 6))
    hp = ref.hparams_for(net, batch_size=B, buffer_size=cap, obs_dtype=u8, learning_rate=1e-3, prioritized_replay=int(rng.random() < 0.7), double_q=int(rng.random() < 0.7))
    layers = ref.layers_from_network(net); plan = pkg.default_plan(layers, hp)
    g, t = pkg.Engine(layers, hp, plan=plan), ref.Twin(layers, hp, plan=plan, threads=4)
    p = O.Network.flatten(O.init_params(net, seed=seed)); p = (p + 0.05 * rng.standard_normal(p.shape)).astype(np.float32)
    mel = int(rng.choice([3, 7, 100])); es = int(rng.integers(0, 1000))
    for h in (g, t):
        h.set_params(p, 0); h.sync_target(); h.envs_create(spec, max_episode_length=mel, seed=es)
    t0 = 1
    for chunk in (int(rng.integers(1, 6)), int(rng.integers(5, 25))):
        cfg = dict(t0=t0, train_freq=int(rng.choice([1, 2, 4])), target_update_freq=int(rng.choice([0, 3, 8])), eps=(1.0, 0.1, float(rng.choice([5, 20]))))
        assert g.rollout(chunk, **cfg) == t.rollout(chunk, **cfg)
        t0 += chunk
        for x, y in zip(g.envs_peek(), t.envs_peek()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())
        np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    ne = int(rng.choice([1, 4, 9])); assert g.evaluate(ne, mel, seed=3) == t.evaluate(ne, mel, seed=3)
    g.close(); t.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_random_simulated_ranks_equal_concatenated_batch(pkg, monkeypatch, seed):
    """random wide-dense networks (first layer or after a conv trunk, dueling or not) under DQN_SIM_WORLD=k: the gather exchange must equal the twin's
    single-device step on the concatenated k*B batch (wide layers bit for bit, everything else to round-off)."""
    rng = np.random.default_rng(4000 + seed)
    k = int(rng.choice([2, 3, 4])); B = 32
    wide = int(rng.choice([256, 512]))
    if rng.random() < 0.5:
        obs = (3, 12, 14); trunk = [O.Conv(3, 3, 8, O.ACT_RELU, 1), O.Conv(4, 8, 16, O.ACT_RELU, 2)]; feat = 16 * 4 * 5
    else:
        obs = (320,); trunk = []; feat = 320
    layers_ = trunk + [O.Dense(feat, wide, O.ACT_RELU), O.Dense(wide, 4, O.ACT_IDENTITY)]
    net = O.Network(obs, *O.create_dueling_network(layers_)) if rng.random() < 0.6 else O.Network(obs, layers_)
    layers = ref.layers_from_network(net)
    hp_g = ref.hparams_for(net, batch_size=B, buffer_size=200, learning_rate=1e-3, gamma=0.99, double_q=int(rng.random() < 0.7))
    hp_t = ref.hparams_for(net, batch_size=B * k, buffer_size=200, learning_rate=1e-3, gamma=0.99, double_q=hp_g.double_q)
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g = pkg.Engine(layers, hp_g, plan=pkg.default_plan(layers, hp_g))
    monkeypatch.delenv("DQN_SIM_WORLD")
    t = ref.Twin(layers, hp_t, plan=pkg.default_plan(layers, hp_t), threads=8)
    n = 150
    s = rng.random((n,) + obs, dtype=np.float32); sp = rng.random((n,) + obs, dtype=np.float32)
    a = rng.integers(0, 4, n).astype(np.int32); r = (2 * rng.standard_normal(n)).astype(np.float32); d = (rng.random(n) < 0.2).astype(np.uint8)
    p = O.Network.flatten(O.init_params(net, seed=seed)); p = (p + 0.01 * rng.standard_normal(p.shape)).astype(np.float32)
    for h in (g, t):
        h.replay_add(s, a, r, sp, d); h.set_params(p, 0); h.set_params(p, 1)
    idx = rng.choice(n, B, replace=False).astype(np.int64)
    lg, gg, tdg = g.train_step(idx); lt, gt, tdt = t.train_step(np.tile(idx, k))
    np.testing.assert_array_equal(tdg, tdt[:B])
    Gg, Gt = g.get_grads() / np.float32(k), t.get_grads()
    np.testing.assert_allclose(Gg, Gt, rtol=2e-4, atol=2e-7)
    if k in (2, 4):     # 1/k exact: the wide blocks are then bit-identical
        for x, y in zip(net.unflatten(Gg), net.unflatten(Gt)):
            if x.ndim == 2 and wide in x.shape and feat in x.shape:
                np.testing.assert_array_equal(x, y)
    g.close(); t.close()
