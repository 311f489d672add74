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
        # (nset + 1) * T * cg * n_out NOT a multiple of 4 (single-Q, odd T, 3 actions; B = 6 -> column groups of 2, B = 5 -> of 1): the fused step's LDS arrays behind the
        # head outputs (the padded Wh copy BPTT reads 16 bytes at a time) must stay 16-byte aligned
        "lstm_q3_b6_t5": (O.RecurrentNetwork((6,), [O.LSTM(6, 8), O.Dense(8, 3, I)]), 6, 5, dict(gamma=0.9, double_q=0)),
        "lstm_q3_b5_t3": (O.RecurrentNetwork((7,), [O.LSTM(7, 16), O.Dense(16, 3, I)]), 5, 3, dict(gamma=0.9, double_q=0)),
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


def oracle_recur_state(net, p_on, xs):
    """(h, c) of every LSTM layer after running the policy network over xs from the reset state: the fp64 oracle's Recur state
    (what hiddenstates(m), src/helpers.jl:61-63, returns after len(xs) calls of the policy, src/policy.jl:38-46), each [out, streams]."""
    qs, caches = O._seq_forward(net, [p.astype(np.float64) for p in net.unflatten(p_on)], [x.astype(np.float64) for x in xs])
    out = []
    for c in caches[-1]:
        if c[0] == "lstm":
            _, _, _, hp, cp, i, f, gc, o, tc = c
            out.append(((o * tc).T, (f * cp + i * gc).T))
    return qs, out


def check_hidden_state_protocol(h, net, p_on, rng, twin=None):
    """hiddenstates / sethiddenstates! / resetstate! (src/helpers.jl:61-79, src/policy.jl:32-34) and their use around batch_train!
    (src/solver.jl:137-139: hs = hiddenstates(active_q); batch_train!; sethiddenstates!(active_q, hs)).
    h: engine or twin handle with episodes in its replay; twin: optional second handle in the same state, compared bit for bit."""
    hs_all = [h] + ([twin] if twin is not None else [])
    xs = [rng.random((1,) + net.obs_shape).astype(np.float32) for _ in range(5)]
    lstm = [l for l in net.base if l.kind == "lstm"]
    for g in hs_all:
        g.reset_state()
    # fresh state == state0 of the online network (Flux.reset!)
    sl = net.param_slices(); ps = net.unflatten(p_on)
    st0 = [(ps[sl[li][0] + 3], ps[sl[li][0] + 4]) for li, l in enumerate(net.base) if l.kind == "lstm"]
    for (hh, cc), (h0, c0) in zip(h.get_hidden(), st0):
        np.testing.assert_array_equal(hh[:, 0], h0); np.testing.assert_array_equal(cc[:, 0], c0)
    # after k policy forwards the state equals the oracle's to 1e-5 (and the twin's bit for bit)
    for k in range(3):
        for g in hs_all:
            g.forward(xs[k])
    _, want = oracle_recur_state(net, p_on, xs[:3])
    saved = h.get_hidden()
    assert len(saved) == len(lstm) == len(want)
    for (hh, cc), (ho, co) in zip(saved, want):
        np.testing.assert_allclose(hh, ho, atol=1e-5, rtol=1e-5); np.testing.assert_allclose(cc, co, atol=1e-5, rtol=1e-5)
    if twin is not None:
        for (a, b), (c, d) in zip(saved, twin.get_hidden()):
            np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)
    # the policy's Recur state survives a train step (the reference saves and restores it around batch_train!, src/solver.jl:137-139;
    # the engine keeps it apart from the train step's sequences)
    for g in hs_all:
        g.train_step_drqn()
    for (a, b), (c, d) in zip(saved, h.get_hidden()):
        np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)
    # ... so the next forward continues the sequence -- with the UPDATED parameters, exactly what the reference computes after sethiddenstates!
    p_new = h.get_params(0)
    q3 = h.forward(xs[3])
    st3 = h.get_hidden()
    # set -> forward reproduces: restore the saved state, the same observation gives the same Q and the same next state, bit for bit
    h.forward(xs[4])
    h.set_hidden(saved)
    for (a, b), (c, d) in zip(saved, h.get_hidden()):
        np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)
    np.testing.assert_array_equal(h.forward(xs[3]), q3)
    for (a, b), (c, d) in zip(st3, h.get_hidden()):
        np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)
    # and that forward is the oracle's step from the saved state with the new parameters (1e-5)
    psn = [p.astype(np.float64) for p in net.unflatten(p_new)]
    x = xs[3].astype(np.float64)
    k = 0
    for li, l in enumerate(net.base):
        a, b_ = sl[li]
        if l.kind == "lstm":
            Wi, Wh, b, _, _ = psn[a:b_]
            hp, cp = saved[k][0].T.astype(np.float64), saved[k][1].T.astype(np.float64)
            g = x.reshape(1, -1) @ Wi + hp @ Wh + b
            H = l.n_out
            i, f, gc, o = O._sigm(g[:, :H]), O._sigm(g[:, H:2 * H]), np.tanh(g[:, 2 * H:3 * H]), O._sigm(g[:, 3 * H:])
            c = f * cp + i * gc
            x = o * np.tanh(c)
            np.testing.assert_allclose(st3[k][0], x.T, atol=1e-5, rtol=1e-5); np.testing.assert_allclose(st3[k][1], c.T, atol=1e-5, rtol=1e-5)
            k += 1
        else:
            x, _ = O.layer_forward(l, x, *psn[a:b_])
    if twin is not None:
        np.testing.assert_array_equal(twin.forward(xs[3]), q3)
        for (a, b), (c, d) in zip(st3, twin.get_hidden()):
            np.testing.assert_array_equal(a, c); np.testing.assert_array_equal(b, d)
    # a wrong-sized buffer is refused
    import pytest
    with pytest.raises(Exception, match="buffer too small"):
        buf = np.zeros(max(1, h.hidden_size() - 1), np.float32)
        h._check(h.f["get_hidden"](h._h, buf.ctypes.data_as(ref.abi._f32p), buf.size))
