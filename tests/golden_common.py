"""Shared checks against the committed golden fixtures (tests/golden/*.npz, made by oracle/make_golden.py from an INDEPENDENT torch float64
autograd model of src/solver.jl:191-287).  TEST INFRASTRUCTURE.

Since round 4 the feed-forward fixtures' IS weights are the ones get_batch computes (src/prioritized_experience_replay.jl:93-102) for a replay holding
exactly the fixture's B transitions added with td_err = |r| (src/solver.jl:91-94), so an engine fed the same way reproduces the fixture's `w` and every
stored output -- loss, td, Q, gradients, the Adam step -- can be compared with the torch values directly."""
import os

import numpy as np

import dqn_oracle as O
import ref
from nets import GOLDEN_CASES, golden_params

DRQN_GOLDEN = {
    "drqn_cfg4_lstm_plain": lambda: O.RecurrentNetwork((1, 5, 5), [O.LSTM(25, 32), O.Dense(32, 4, O.ACT_IDENTITY)]),
    "drqn_dense_lstm_dueling": lambda: O.RecurrentNetwork((6,), [O.Dense(6, 12, O.ACT_RELU), O.LSTM(12, 16)], [O.Dense(16, 1, O.ACT_IDENTITY)], [O.Dense(16, 5, O.ACT_IDENTITY)]),
    "drqn_lstm_single_q": lambda: O.RecurrentNetwork((6,), [O.LSTM(6, 8), O.Dense(8, 3, O.ACT_IDENTITY)]),
}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def rel(x, y):
    x, y = np.asarray(x, np.float64), np.asarray(y, np.float64)
    return float(np.max(np.abs(x - y)) / (1e-300 + np.max(np.abs(y))))


# ------------------------------------------------------------------ the fp64 oracle reproduces EVERY stored field
def oracle_reproduces_ff(name, g):
    net = GOLDEN_CASES[name]()
    p_on, p_tg = golden_params(name, net, g)
    lr = float(g["lr"])
    adam = O.AdamState([np.asarray(p, np.float64) for p in net.unflatten(p_on)], lr)
    batch = (g["s"], g["a"], g["r"], g["sp"], g["done"], g["w"])
    o = O.batch_train_step(net, net.unflatten(p_on), net.unflatten(p_tg), batch, gamma=float(g["gamma"]), double_q=bool(g["double_q"]), adam=adam, dtype=np.float64)
    errs = dict(loss=rel(o["loss"], g["loss"]), td=rel(o["td"], g["td"]), q=rel(o["q"], g["q"]), grad_norm=rel(o["grad_norm"], g["grad_norm"]),
                grad_sums=rel([x.sum() for x in o["grads"]], g["grad_sums"]), grad_abs_sums=rel([np.abs(x).sum() for x in o["grads"]], g["grad_abs_sums"]),
                newp_sums=rel([x.sum() for x in o["new_params"]], g["newp_sums"]))
    if "grads" in g:
        errs["grads"] = rel(O.Network.flatten(o["grads"]), g["grads"])
        errs["new_params"] = rel(O.Network.flatten(o["new_params"]), g["new_params"])
    return errs


def drqn_batch(g):
    T = int(g["T"])
    return tuple([g[k][t] for t in range(T)] for k in ("s", "a", "r", "sp", "done", "mask"))


def oracle_reproduces_drqn(name, g):
    net = DRQN_GOLDEN[name]()
    p_on, p_tg = g["p_on"].astype(np.float32), g["p_tg"].astype(np.float32)
    adam = O.AdamState([np.asarray(p, np.float64) for p in net.unflatten(p_on)], float(g["lr"]))
    o = O.drqn_train_step(net, net.unflatten(p_on), net.unflatten(p_tg), drqn_batch(g), gamma=float(g["gamma"]), double_q=bool(g["double_q"]), adam=adam)
    return dict(loss=rel(o["loss"], g["loss"]), q=rel(o["q"], g["q"]), grads=rel(O.Network.flatten(o["grads"]), g["grads"]),
                new_params=rel(O.Network.flatten(o["new_params"]), g["new_params"]))


# ------------------------------------------------------------------ an engine (twin or GPU) against the torch values of a feed-forward fixture
def engine_vs_ff_fixture(h, name, g, out, tol_q=1e-5):
    """h: a handle that already ran ONE train step on the fixture's batch (replay = the B transitions added with td_err = |r|, indices 0..B-1);
    out = dict(w, loss, gn, td, q, grads, newp).  Compares with the stored torch float64 values at fp32 round-off."""
    assert int(g["w_from_priorities"]) == 1
    np.testing.assert_allclose(out["w"], g["w"], rtol=2e-6)                     # get_batch's IS weights ARE the fixture's
    np.testing.assert_allclose(out["q"], g["q"], atol=tol_q, rtol=1e-5)           # north_star: Q within 1e-5
    np.testing.assert_allclose(out["td"], g["td"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(out["loss"], float(g["loss"]), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["gn"], float(g["grad_norm"]), rtol=1e-4)
    net = GOLDEN_CASES[name]()
    parts = [np.asarray(x, np.float64) for x in net.unflatten(out["grads"])]
    newp = [np.asarray(x, np.float64) for x in net.unflatten(out["newp"])]
    gscale = float(g["grad_norm"])
    for k, (x, want, wabs) in enumerate(zip(parts, g["grad_sums"], g["grad_abs_sums"])):
        assert abs(x.sum() - want) <= 2e-5 * gscale * np.sqrt(x.size) + 1e-5 * abs(want) + 3e-5 * wabs / np.sqrt(x.size), (name, "grad sum", k)
        np.testing.assert_allclose(np.abs(x).sum(), wabs, rtol=2e-4, atol=2e-5 * gscale * np.sqrt(x.size), err_msg=f"{name}: grad |sum| {k}")
    for k, (x, want) in enumerate(zip(newp, g["newp_sums"])):
        # Adam's first step moves every parameter by <= lr (and by exactly +-lr wherever |g| >> eps): the sum is pinned to a few lr
        assert abs(x.sum() - want) <= 2.1 * float(g["lr"]) * max(1.0, 1e-3 * x.size) + 1e-6 * np.abs(x).sum(), (name, "new param sum", k)
    if "grads" in g:
        np.testing.assert_allclose(out["grads"], g["grads"], atol=2e-5 * np.abs(g["grads"]).max(), rtol=1e-4)
        diff = np.abs(out["newp"] - g["new_params"])
        assert diff.max() <= 2.1 * float(g["lr"]) and (diff > 2e-6).mean() < 1e-5       # Adam at |g| ~ eps: see test_twin_vs_oracle.py


# ------------------------------------------------------------------ an engine against a DRQN fixture
def run_drqn_fixture(Engine, name, g, **ekw):
    """Feeds the fixture's sequences as episodes (through the checkpoint-import seam, the only one that accepts arbitrary `done` flags inside an
    episode), draws them with explicit (episode, start) pairs so that episode_get_batch returns the fixture's batch wherever mask == 1 (and zeros
    elsewhere -- the reference's behaviour, src/episode_replay.jl:82-92; masked rows reach neither the loss nor, being a suffix, any valid row's state),
    runs ONE recurrent train step and compares loss / Q on valid rows / gradients / the Adam step with the torch float64 values."""
    net = DRQN_GOLDEN[name]()
    B, T = int(g["B"]), int(g["T"])
    hp = ref.hparams_for(net, batch_size=B, buffer_size=B, recurrence=1, trace_length=T, learning_rate=float(g["lr"]), prioritized_replay=0,
                         gamma=float(g["gamma"]), double_q=int(g["double_q"]))
    h = Engine(ref.layers_from_network(net), hp, **ekw)
    p_on, p_tg = g["p_on"].astype(np.float32), g["p_tg"].astype(np.float32)
    h.set_params(p_on, 0); h.set_params(p_tg, 1)
    mask = g["mask"]                                  # [T, B], a prefix per column
    lens = mask.sum(0).astype(np.int32)
    assert all(np.array_equal(mask[:, b], (np.arange(T) < lens[b]).astype(mask.dtype)) for b in range(B))
    tb = lambda x: np.ascontiguousarray(np.swapaxes(x, 0, 1))     # [T, B, ...] -> [B (episode), T, ...]
    s, sp, a, r, d = tb(g["s"]), tb(g["sp"]), tb(g["a"]), tb(g["r"]), tb(g["done"]).astype(np.uint8)
    # a column with no valid row: an episode longer than T drawn at start = T contributes min(len, T) - start = 0 rows (the prefix quirk)
    ep_len = np.where(lens > 0, lens, T + 1).astype(np.int32)
    start = np.where(lens > 0, 0, T).astype(np.int32)
    h.episode_import(s, sp, a, r, d, ep_len)
    idx = np.arange(B, dtype=np.int64)
    bs, ba, br, bsp, bd, bm = h.episode_get_batch(idx, start)
    np.testing.assert_array_equal(bm, mask)
    m = mask.astype(bool)
    bs, bsp = bs.reshape(g["s"].shape), bsp.reshape(g["sp"].shape)
    np.testing.assert_array_equal(bs[m], g["s"][m]); np.testing.assert_array_equal(bsp[m], g["sp"][m])
    np.testing.assert_array_equal(ba[m], g["a"][m]); np.testing.assert_array_equal(br[m], g["r"][m]); np.testing.assert_array_equal(bd[m], g["done"][m])
    assert not bs[~m].any() and not bsp[~m].any()
    loss, gn = h.train_step_drqn(idx, start)
    np.testing.assert_allclose(loss, float(g["loss"]), rtol=2e-5, atol=1e-7)
    grads = h.get_grads(); sc = np.abs(g["grads"]).max()
    np.testing.assert_allclose(grads, g["grads"], atol=3e-5 * sc, rtol=1e-4)
    np.testing.assert_allclose(gn, np.abs(g["grads"]).max(), rtol=1e-4)
    diff = np.abs(h.get_params(0) - g["new_params"])
    assert diff.max() <= 2.1 * float(g["lr"]) and (diff > 5e-6).mean() < 1e-3
    out = dict(loss=loss, gn=gn, grads=grads, newp=h.get_params(0))
    h.close()
    return out
