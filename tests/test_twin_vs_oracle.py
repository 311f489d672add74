"""CPU: the canonical-order C twin (oracle/dqn_ref.c) against the NumPy fp64 oracle and the
torch-autograd-pinned golden fixtures (tests/golden, oracle/make_golden.py)."""
import os

import numpy as np
import pytest

import dqn_oracle as O
import ref
from nets import GOLDEN_CASES, fill_replay_from_batch, golden_params


def check_priorities_after_step(h, hp, idx, pr_before, td_engine, td_oracle, w):
    """update_priorities!(replay, idx, td) (src/solver.jl:231-233, ...replay.jl:76-80): p[idx] = (|td| + eps)^alpha with the UNWEIGHTED td,
    only when solver.prioritized_replay; every other priority is untouched.  idx must hold no duplicates."""
    pr = h.replay_priorities()
    if not hp.prioritized_replay:
        np.testing.assert_array_equal(pr, pr_before)
        return
    eps, alpha = np.float32(hp.prio_eps), np.float32(hp.prio_alpha)
    np.testing.assert_allclose(pr[idx], O.priority_from_td(np.abs(td_engine), eps, alpha), rtol=2e-7)   # the formula on the engine's own td: 1 ulp of pow
    want = O.priority_from_td(np.abs(td_oracle), eps, alpha, np.float64)                                  # and on the fp64 oracle's td
    slack = 0.6 * (np.abs(td_oracle) + float(eps)) ** (float(alpha) - 1.0) * 3e-5 + 1e-6 * want           # d p / d td  x  the td tolerance
    assert np.all(np.abs(pr[idx] - want) <= slack), np.abs(pr[idx] - want).max()
    rest = np.setdiff1d(np.arange(pr.size), idx)
    np.testing.assert_array_equal(pr[rest], pr_before[rest])
    # the misreading (priorities from the IS-WEIGHTED td) must be distinguishable on this batch, else the check above proves nothing
    wrong = O.priority_from_td(np.abs(np.asarray(w, np.float64) * td_oracle), eps, alpha, np.float64)
    assert np.any(np.abs(wrong - want) > 10 * slack), "batch cannot tell weighted from unweighted td"


def run_case(name, golden_dir, Engine, tol_q=1e-5, prioritized=1, **engine_kw):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    net = GOLDEN_CASES[name]()
    B = int(g["B"])
    p_on, p_tg = golden_params(name, net, g)
    hp = ref.hparams_for(net, batch_size=B, gamma=float(g["gamma"]), double_q=int(g["double_q"]),
                         learning_rate=float(g["lr"]), buffer_size=max(64, B), prioritized_replay=prioritized)
    layers = ref.layers_from_network(net)
    h = Engine(layers, hp, **engine_kw)
    h.set_params(p_on, 0)
    h.set_params(p_tg, 1)
    np.testing.assert_array_equal(h.get_params(0), p_on)  # layout round trip (conv flip + transpose)
    idx = fill_replay_from_batch(h, g)
    s, a, r, sp, done, w = h.get_batch(idx)
    np.testing.assert_array_equal(s, g["s"].reshape(s.shape))
    np.testing.assert_array_equal(a, g["a"])
    np.testing.assert_array_equal(done, g["done"])
    # fp64 oracle on the engine's own batch (IS weights included)
    adam = O.AdamState([np.asarray(p, np.float64) for p in net.unflatten(p_on)], float(g["lr"]))
    o = O.batch_train_step(net, net.unflatten(p_on), net.unflatten(p_tg), (s, a, r, sp, done, w),
                           gamma=float(np.float32(g["gamma"])), double_q=bool(g["double_q"]), adam=adam)
    w64 = O.is_weights(h.replay_priorities()[idx], h.replay_priorities(), hp.prio_beta, np.float64)
    np.testing.assert_allclose(w, w64, rtol=2e-6)
    pr_before = h.replay_priorities()
    loss, gn, td = h.train_step(idx)
    q = h.last_q()
    check_priorities_after_step(h, hp, idx, pr_before, td, o["td"], w)
    np.testing.assert_allclose(q["q_on_s"], o["q"], atol=tol_q, rtol=1e-5)
    np.testing.assert_allclose(q["q_tg_sp"], o["q_tg_sp"], atol=tol_q, rtol=1e-5)
    np.testing.assert_array_equal(q["best_a"], o["best_a"])
    np.testing.assert_allclose(td, o["td"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(loss, o["loss"], rtol=1e-5, atol=1e-7)
    grads = h.get_grads()
    go = O.Network.flatten(o["grads"])
    scale = np.abs(go).max()
    np.testing.assert_allclose(grads, go, atol=2e-5 * scale, rtol=1e-4)
    np.testing.assert_allclose(gn, o["grad_norm"], rtol=1e-4)
    newp = h.get_params(0)
    # Adam's first step is lr*g/(|g|+eps): where |g| ~ eps (1e-8) fp32 round-off of g legitimately moves
    # the update by up to lr; everywhere else the parameters agree to 2e-6.
    diff = np.abs(newp - O.Network.flatten(o["new_params"]))
    assert diff.max() <= 2.1 * float(g["lr"])
    assert (diff > 2e-6).mean() < 1e-5
    # golden (torch float64 autograd): with prioritized replay on, get_batch's IS weights ARE the fixture's (oracle/make_golden.py derives them from
    # the priorities this replay holds), so every stored torch output is compared directly
    np.testing.assert_allclose(q["q_on_s"], g["q"], atol=tol_q, rtol=1e-5)
    if prioritized:
        from golden_common import engine_vs_ff_fixture
        engine_vs_ff_fixture(h, name, g, dict(w=w, loss=loss, gn=gn, td=td, q=q["q_on_s"], grads=grads, newp=newp), tol_q=tol_q)
    h.close()
    return dict(loss=loss, td=td, q=q["q_on_s"], q_all=q, grads=grads, newp=newp, w=w, gn=gn)


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_twin_matches_fp64_oracle(name, golden_dir):
    run_case(name, golden_dir, ref.Twin, threads=8)


@pytest.mark.parametrize("name", ["cfg1_gridworld_mlp_dueling", "small_conv_dueling"])
def test_twin_prioritized_replay_off_leaves_priorities(name, golden_dir):
    run_case(name, golden_dir, ref.Twin, prioritized=0, threads=4)     # src/solver.jl:231: no update_priorities! call


def test_twin_thread_count_does_not_change_bits(golden_dir):
    a = run_case("small_conv_dueling", golden_dir, ref.Twin, threads=1)
    b = run_case("small_conv_dueling", golden_dir, ref.Twin, threads=8)
    assert a["loss"] == b["loss"]
    np.testing.assert_array_equal(a["grads"], b["grads"])
    np.testing.assert_array_equal(a["newp"], b["newp"])
