"""Reader / consumers for fixtures a maintainer generates with Julia + the reference: `julia oracle/make_golden.jl` -> tests/golden/julia_*.dqnvec
(TEST INFRASTRUCTURE).  No such file can be produced in the build image (no Julia); the consumers below are exercised by a self-test that writes
files of the same format from the NumPy oracle (tests/test_julia_golden_cpu.py::test_consumer_selftest_*)."""
import os

import numpy as np

import dqn_oracle as O
import ref

R, I, T_ = O.ACT_RELU, O.ACT_IDENTITY, O.ACT_TANH
_DT = {"Float32": np.float32, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64, "UInt8": np.uint8, "Bool": np.uint8}
_JL = {np.dtype(np.float32): "Float32", np.dtype(np.float64): "Float64", np.dtype(np.int32): "Int32", np.dtype(np.int64): "Int64", np.dtype(np.uint8): "UInt8"}


def read_dqnvec(path):
    """{name: array}; a Julia array of size (d1..dN) (column-major) comes back as a C-order NumPy array of shape (dN..d1) -- the same bytes."""
    out = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            if not line.strip():
                continue
            name, ty, nd, *dims = line.decode().split()
            dims = [int(x) for x in dims]
            assert int(nd) == len(dims)
            dt = np.dtype(_DT[ty])
            n = int(np.prod(dims)) if dims else 1
            buf = f.read(n * dt.itemsize)
            assert len(buf) == n * dt.itemsize, f"{path}: truncated array {name}"
            out[name] = np.frombuffer(buf, dt).reshape(dims[::-1]).copy()
    return out


def write_dqnvec(path, arrays):
    """the inverse (self-test only): C-order arrays of shape (dN..d1) are written as Julia arrays of size (d1..dN)"""
    with open(path, "wb") as f:
        for name, a in arrays.items():
            a = np.ascontiguousarray(a)
            f.write(f"{name} {_JL[a.dtype]} {a.ndim} {' '.join(str(d) for d in a.shape[::-1])}\n".encode())
            f.write(a.tobytes()); f.write(b"\n")


def _mlp():
    return [O.Dense(100, 8, T_), O.Dense(8, 4, I)]


def _conv():
    return [O.Conv(3, 4, 8, R, 2), O.Conv(2, 8, 16, R, 1), O.Dense(144, 32, R), O.Dense(32, 4, I)]


def _duel(obs, layers):
    b, v, a = O.create_dueling_network(layers)
    return O.Network(obs, b, v, a)


JULIA_FF_CASES = {      # the networks oracle/make_golden.jl builds, by case name
    "mlp_tanh_plain": lambda: O.Network((4, 5, 5), _mlp()),
    "mlp_dueling_ddqn_per": lambda: _duel((4, 5, 5), _mlp()),
    "conv_dueling_ddqn_per": lambda: _duel((4, 10, 10), _conv()),
}
JULIA_DRQN_CASES = {"drqn_lstm": lambda: O.RecurrentNetwork((1, 5, 5), [O.LSTM(25, 8), O.Dense(8, 4, I)])}


def find(golden_dir, case):
    p = os.path.join(golden_dir, f"julia_{case}.dqnvec")
    return p if os.path.exists(p) else None


def consume_ff(Engine, case, f, tol_q=1e-5, **ekw):
    """One batch_train! of the REFERENCE (src/solver.jl:191-236) against an engine fed through the replay protocol with the same transitions."""
    net = JULIA_FF_CASES[case]()
    B, gamma, lr, double_q, dueling, n, alpha, beta, eps, prioritized = [float(x) for x in f["meta"].ravel()[:10]]
    B, n = int(B), int(n)
    assert bool(dueling) == bool(net.dueling)
    hp = ref.hparams_for(net, batch_size=B, gamma=gamma, double_q=int(double_q), learning_rate=lr, buffer_size=n, prioritized_replay=int(prioritized),
                         prio_alpha=alpha, prio_beta=beta, prio_eps=eps)
    h = Engine(ref.layers_from_network(net), hp, **ekw)
    h.set_params(f["p_on"], 0); h.set_params(f["p_tg"], 1)
    np.testing.assert_array_equal(h.get_params(0), f["p_on"])
    a0 = f["ra"].astype(np.int32) - 1                                  # stored 1-based (src/solver.jl:84-88), 0-based at the ABI
    h.replay_add(f["rs"], a0, f["rr"], f["rsp"], f["rdone"].astype(np.uint8), td_err=np.abs(f["rr"]))      # populate_replay_buffer!: add_exp!(replay, exp, abs(rew))
    np.testing.assert_allclose(h.replay_priorities(), f["rprio"], rtol=2e-7)      # Float32^Float32: <= 1 ulp
    idx = f["idx"].astype(np.int64) - 1
    s, a, r, sp, done, w = h.get_batch(idx)
    np.testing.assert_array_equal(s, f["bs"].reshape(s.shape)); np.testing.assert_array_equal(sp, f["bsp"].reshape(sp.shape))
    np.testing.assert_array_equal(a, f["ba"].astype(np.int32) - 1); np.testing.assert_array_equal(r, f["br"]); np.testing.assert_array_equal(done, f["bdone"])
    np.testing.assert_allclose(w, f["bw"], rtol=2e-6)                  # fp32 `sum` order differs
    pr_before = h.replay_priorities()
    loss, gn, td = h.train_step(idx)
    q = h.last_q()
    for k, name in (("q_on_s", "q_on_s"), ("q_on_sp", "q_on_sp"), ("q_tg_sp", "q_tg_sp")):
        np.testing.assert_allclose(q[k], f[name], atol=tol_q, rtol=1e-5, err_msg=f"{case}: {k} vs Flux")       # north_star: Q within 1e-5; Julia (nA, B) == C [B][nA]
    if double_q:
        top2 = np.sort(f["q_on_sp"].astype(np.float64), axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 2e-5
        np.testing.assert_array_equal(q["best_a"][clear], np.argmax(f["q_on_sp"], axis=1)[clear])               # greedy indices (first-max)
    np.testing.assert_allclose(td, f["td"].ravel(), atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(loss, float(f["loss"].ravel()[0]), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(gn, float(f["grad_norm"].ravel()[0]), rtol=1e-4)
    g = h.get_grads(); sc = np.abs(f["grads"]).max() + 1e-30
    np.testing.assert_allclose(g, f["grads"].ravel(), atol=3e-5 * sc, rtol=1e-4)
    diff = np.abs(h.get_params(0) - f["p_new"].ravel())
    assert diff.max() <= 2.1 * lr and (diff > 5e-6).mean() < 1e-3       # Adam's first step at |g| ~ eps: see tests/test_twin_vs_oracle.py
    pr = h.replay_priorities()
    if prioritized:
        np.testing.assert_allclose(pr, f["rprio_new"], rtol=1e-4, atol=1e-6)
    else:
        np.testing.assert_array_equal(pr, pr_before); np.testing.assert_allclose(pr, f["rprio_new"], rtol=2e-7)
    h.close()
    return dict(loss=loss, gn=gn, td=td)


def consume_drqn(Engine, case, f, **ekw):
    """One recurrent batch_train! of the REFERENCE (src/solver.jl:239-287) against an engine holding the same episodes and given the same draws."""
    net = JULIA_DRQN_CASES[case]()
    B, gamma, lr, double_q, _, n, T = [float(x) for x in f["meta"].ravel()[:7]]
    B, n, T = int(B), int(n), int(T)
    hp = ref.hparams_for(net, batch_size=B, buffer_size=n, recurrence=1, trace_length=T, learning_rate=lr, prioritized_replay=0, gamma=gamma, double_q=int(double_q))
    h = Engine(ref.layers_from_network(net), hp, **ekw)
    h.set_params(f["p_on"], 0); h.set_params(f["p_tg"], 1)
    E = int(np.prod(net.obs_shape))
    # Julia (E, T, n) == C [n][T][E]; (T, n) == C [n][T]
    h.episode_import(f["es"].reshape(n, T, E), f["esp"].reshape(n, T, E), f["ea"].reshape(n, T).astype(np.int32) - 1, f["er"].reshape(n, T), f["edone"].reshape(n, T), f["ep_len"])
    idx, start = f["ep_idx"].astype(np.int64) - 1, f["ep_start"].astype(np.int32) - 1
    s, a, r, sp, d, m = h.episode_get_batch(idx, start)
    np.testing.assert_array_equal(m, f["bmask"].reshape(T, B))                       # the prefix-copy quirk (episode_replay.jl:82-92)
    np.testing.assert_array_equal(s.reshape(T, B, E), f["bs"].reshape(T, B, E)); np.testing.assert_array_equal(sp.reshape(T, B, E), f["bsp"].reshape(T, B, E))
    np.testing.assert_array_equal(r, f["br"].reshape(T, B)); np.testing.assert_array_equal(d, f["bdone"].reshape(T, B))
    mb = m.astype(bool)
    np.testing.assert_array_equal(a[mb], (f["ba"].reshape(T, B).astype(np.int32) - 1)[mb])      # masked rows: CartesianIndex(1,1) in the reference, 0 here
    loss, gn = h.train_step_drqn(idx, start)
    np.testing.assert_allclose(loss, float(f["loss"].ravel()[0]), rtol=3e-5, atol=1e-7)
    np.testing.assert_allclose(gn, float(f["grad_norm"].ravel()[0]), rtol=1e-4)
    diff = np.abs(h.get_params(0) - f["p_new"].ravel())
    assert diff.max() <= 2.1 * lr and (diff > 5e-6).mean() < 1e-3
    h.close()
    return dict(loss=loss, gn=gn)


# ------------------------------------------------------------------ self-test material: the same files, written from the NumPy oracle
def selftest_ff_file(path, case, seed=0, n=48, B=16, gamma=0.99, lr=1e-3, double_q=1, prioritized=1):
    net = JULIA_FF_CASES[case]()
    rng = np.random.default_rng(seed)
    p_on = O.init_params(net, seed=seed + 1); p_tg = [np.asarray(p * 0.9, np.float32) for p in p_on]
    rs, rsp = rng.random((n,) + net.obs_shape, dtype=np.float32), rng.random((n,) + net.obs_shape, dtype=np.float32)
    ra = rng.integers(0, net.n_actions, n).astype(np.int32); rr = rng.choice(np.array([-0.1, 0.0, 0.1, 1.0, -1.0], np.float32), n); rdone = (rng.random(n) < 0.2).astype(np.uint8)
    prio = O.priority_from_td(np.abs(rr), np.float32(1e-3), np.float32(0.6))
    idx = rng.choice(n, B, replace=False).astype(np.int64)
    w = O.is_weights(prio[idx], prio, 0.4)
    batch = (rs[idx], ra[idx], rr[idx], rsp[idx], rdone[idx].astype(np.float32), w)
    adam = O.AdamState([np.asarray(p, np.float64) for p in p_on], lr)
    o = O.batch_train_step(net, p_on, p_tg, batch, gamma=float(np.float32(gamma)), double_q=bool(double_q), adam=adam)
    prio_new = prio.copy()
    if prioritized:
        prio_new[idx] = O.priority_from_td(np.abs(o["td"]).astype(np.float32), np.float32(1e-3), np.float32(0.6))
    f32 = lambda x: np.asarray(x, np.float32)
    write_dqnvec(path, dict(
        meta=np.array([B, np.float32(gamma), np.float32(lr), double_q, int(net.dueling), n, np.float32(0.6), np.float32(0.4), np.float32(1e-3), prioritized], np.float64),
        p_on=O.Network.flatten(p_on), p_tg=O.Network.flatten(p_tg), rs=rs, rsp=rsp, ra=ra + 1, rr=rr, rdone=rdone, rprio=prio,
        idx=idx + 1, bs=batch[0], ba=batch[1] + 1, br=batch[2], bsp=batch[3], bdone=batch[4], bw=w,
        q_on_s=f32(o["q"]), q_on_sp=f32(o["q_on_sp"]), q_tg_sp=f32(o["q_tg_sp"]), td=f32(o["td"]), grads=f32(O.Network.flatten(o["grads"])),
        grad_norm_closure=f32([o["grad_norm"]]), loss=f32([o["loss"]]), grad_norm=f32([o["grad_norm"]]), p_new=f32(O.Network.flatten(o["new_params"])), rprio_new=prio_new))
