"""CPU: the hand-derived end-to-end known answer and the sampler-distribution check (tests/parity_common.py) against the CPU twin, and the
same hand-derived numbers against the NumPy fp64 oracle -- so the oracle's composition order (huber(w*td), unweighted td into the
priorities, first-max argmax, Adam's first step) is pinned by numbers that were NOT produced by any code in this repository."""
import numpy as np

import dqn_oracle as O
import ref
from parity_common import hand_derived_known_answer, sampler_distinct, sampler_distribution


def test_hand_derived_known_answer_twin():
    hand_derived_known_answer(ref.Twin, threads=1)


def test_sampler_distribution_twin():
    sampler_distribution(ref.Twin, threads=1)


def test_sampler_distinct_twin():
    """hp.sample_distinct = 1: the twin's restatement of the reference's replace=false draw (no duplicates, uniform redraws on the residual mass)"""
    sampler_distinct(ref.Twin, threads=1)


def test_hand_derived_known_answer_oracle():
    net = O.Network((2,), [O.Dense(2, 2, O.ACT_IDENTITY)])
    # the oracle holds a Dense weight as (in, out) == the bytes of Julia's column-major (out, in)
    p_on = [np.array([[1, 2], [3, -1]], np.float64).T.copy(), np.array([0, 1], np.float64)]
    p_tg = [np.array([[2, 0], [0, 1]], np.float64).T.copy(), np.array([1, 0], np.float64)]
    rp = O.PrioritizedReplay((2,), 4, 2, alpha=1.0, beta=1.0, eps=0.5)
    s = np.array([[1, 0], [0, 2], [5, 5], [7, 7]], np.float32); sp = np.array([[0, 1], [1, 1], [6, 6], [8, 8]], np.float32)
    for i, (a, r, d, te) in enumerate(zip([0, 1, 0, 1], [1, -2, 0, 0], [0, 1, 0, 0], [0.5, 3.5, 1.5, 0.5])):
        rp.add_exp(s[i], a, r, sp[i], d, td_err=te)
    np.testing.assert_array_equal(rp.prio, [1, 4, 2, 1])
    idx = np.array([0, 1])
    batch = rp.get_batch(idx, np.float64)
    np.testing.assert_array_equal(batch[5], [2.0, 0.5])
    adam = O.AdamState(p_on, 0.25)
    o = O.batch_train_step(net, p_on, p_tg, batch, gamma=0.5, double_q=True, adam=adam)
    np.testing.assert_array_equal(o["q"], [[1, 4], [4, -1]])
    np.testing.assert_array_equal(o["best_a"], [0, 0])
    np.testing.assert_array_equal(o["y"], [1.5, -2.0])
    np.testing.assert_array_equal(o["td"], [-0.5, 1.0])
    assert o["loss"] == 0.3125 and o["grad_norm"] == 1.0
    np.testing.assert_array_equal(o["grads"][0].T, [[-1, 0], [0, 0.25]])
    np.testing.assert_array_equal(o["grads"][1], [-1, 0.125])
    np.testing.assert_allclose(o["new_params"][0].T, [[1.25, 2], [3, -1.25]], rtol=1e-7)
    np.testing.assert_allclose(o["new_params"][1], [0.25, 0.75], rtol=1e-7)
    rp.update_priorities(idx, o["td"])
    np.testing.assert_array_equal(rp.prio, [1.0, 1.5, 2, 1])
