"""Shared DRQN test driver (TEST INFRASTRUCTURE): random episodes -> engine/twin -> fp64 oracle on the same sampled batch."""
import numpy as np

import dqn_oracle as O
import ref

I = O.ACT_IDENTITY


def drqn_nets():
    return {
        "cfg4_lstm_plain": (O.RecurrentNetwork((1, 5, 5), [O.LSTM(25, 32), O.Dense(32, 4, I)]), 32, 8, dict(gamma=0.99, double_q=1)),   # benchmark/flux_dqn.jl:35-36
        "dense_lstm_dueling": (O.RecurrentNetwork((6,), [O.Dense(6, 12, O.ACT_RELU), O.LSTM(12, 16)], [O.Dense(16, 1, I)], [O.Dense(16, 5, I)]), 6, 5, dict(gamma=0.95, double_q=1)),
        "lstm_single_q": (O.RecurrentNetwork((6,), [O.LSTM(6, 8), O.Dense(8, 3, I)]), 4, 3, dict(gamma=0.9, double_q=0)),
        "lstm16_dueling_b16": (O.RecurrentNetwork((16,), [O.LSTM(16, 32)], [O.Dense(32, 1, I)], [O.Dense(32, 4, I)]), 16, 10, dict(gamma=0.95, double_q=1)),   # test/runtests.jl:131-147 shape
    }


def make_episodes(net, n_ep, T, rng):
    eps = []
    for _ in range(n_ep):
        L = int(rng.integers(1, T + 4))
        ep = []
        for t in range(L):
            s = rng.random(net.obs_shape, dtype=np.float32)
            sp = rng.random(net.obs_shape, dtype=np.float32)
            ep.append((s, int(rng.integers(0, net.n_actions)), np.float32(2 * rng.standard_normal()), sp, t == L - 1))
        eps.append(ep)
    return eps


def feed(h, eps):
    for ep in eps:
        s = np.stack([x[0] for x in ep]); sp = np.stack([x[3] for x in ep])
        a = np.array([x[1] for x in ep], np.int32); r = np.array([x[2] for x in ep], np.float32); d = np.array([x[4] for x in ep], np.uint8)
        h.episode_add(s, a, r, sp, d)      # the last transition has done = 1 -> the episode is stored (episode_replay.jl:46-52)


def make_handle(Engine, net, B, T, kw, cap=12, **ekw):
    hp = ref.hparams_for(net, batch_size=B, buffer_size=cap, recurrence=1, trace_length=T, learning_rate=1e-3, prioritized_replay=0, **kw)
    layers = ref.layers_from_network(net)
    return Engine(layers, hp, **ekw), hp, layers


def draws(eps_in_ring, B, rng):
    idx = rng.permutation(len(eps_in_ring))[:B].astype(np.int64)
    start = np.array([rng.integers(0, len(eps_in_ring[i])) for i in idx], np.int32)
    return idx, start


def check_against_oracle(h, net, eps_ring, B, T, kw, rng, params):
    p_on, p_tg = params
    idx, start = draws(eps_ring, B, rng)
    batch = h.episode_get_batch(idx, start)
    exp = O.episode_sample(eps_ring, idx, start, T, net.obs_shape)
    for got, want in zip(batch, (np.stack(exp[0]), np.stack(exp[1]), np.stack(exp[2]), np.stack(exp[3]), np.stack(exp[4]), np.stack(exp[5]))):
        np.testing.assert_array_equal(got.reshape(want.shape), want)
    s, a, r, sp, d, m = batch
    ob = ([x for x in s], [x for x in a], [x for x in r], [x for x in sp], [x for x in d], [x for x in m])
    adam = O.AdamState([np.asarray(p, np.float64) for p in net.unflatten(p_on)], 1e-3)
    o = O.drqn_train_step(net, net.unflatten(p_on), net.unflatten(p_tg), ob, gamma=float(np.float32(kw["gamma"])), double_q=bool(kw["double_q"]), adam=adam)
    loss, gn = h.train_step_drqn(idx, start)
    np.testing.assert_allclose(loss, o["loss"], rtol=2e-5, atol=1e-7)
    g = h.get_grads(); go = O.Network.flatten(o["grads"]); sc = np.abs(go).max() + 1e-30
    np.testing.assert_allclose(g, go, atol=3e-5 * sc, rtol=1e-4)
    np.testing.assert_allclose(gn, o["grad_norm"], rtol=1e-4)
    newp = h.get_params(0)
    diff = np.abs(newp - O.Network.flatten(o["new_params"]))
    assert diff.max() <= 2.1e-3 and (diff > 5e-6).mean() < 1e-3     # Adam at |g| ~ eps, see test_twin_vs_oracle.py
    return idx, start, loss, gn
