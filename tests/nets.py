"""Shared network zoo for the tests (oracle-side descriptions; TEST INFRASTRUCTURE)."""
import numpy as np

import dqn_oracle as O

R, I, T = O.ACT_RELU, O.ACT_IDENTITY, O.ACT_TANH


def cfg1_mlp_dueling():
    b, v, a = O.create_dueling_network([O.Dense(2, 32, I), O.Dense(32, 4, I)])  # README.md:38
    return O.Network((2,), b, v, a)


def testmdp_mlp_tanh():
    return O.Network((4, 5, 5), [O.Dense(100, 8, T), O.Dense(8, 4, I)])  # test/runtests.jl:49


def _small_conv_layers():
    return [O.Conv(4, 3, 8, R, 2), O.Conv(3, 8, 16, R, 1), O.Dense(16 * 3 * 4, 32, R), O.Dense(32, 5, I)]


def small_conv_dueling():
    b, v, a = O.create_dueling_network(_small_conv_layers())
    return O.Network((3, 12, 14), b, v, a)


def small_conv_plain():
    return O.Network((3, 12, 14), _small_conv_layers())


def mid_conv_plain():
    """32-channel convolutions and a 64-wide dense layer: large enough for every LDS-tiled kernel (cin % 32 == 0, N % 32 == 0), small enough for the twin."""
    return O.Network((4, 20, 20), [O.Conv(4, 4, 32, R, 2), O.Conv(3, 32, 32, R, 1), O.Dense(32 * 7 * 7, 64, R), O.Dense(64, 5, I)])


def mid_conv_dueling():
    b, v, a = O.create_dueling_network([O.Conv(4, 4, 32, R, 2), O.Conv(3, 32, 32, R, 1), O.Dense(32 * 7 * 7, 64, R), O.Dense(64, 5, I)])
    return O.Network((4, 20, 20), b, v, a)


def nature_dueling():
    nat = [O.Conv(8, 4, 32, R, 4), O.Conv(4, 32, 64, R, 2), O.Conv(3, 64, 64, R, 1),
           O.Dense(3136, 512, R), O.Dense(512, 4, I)]
    b, v, a = O.create_dueling_network(nat)
    return O.Network((4, 84, 84), b, v, a)


GOLDEN_CASES = {
    "cfg1_gridworld_mlp_dueling": cfg1_mlp_dueling,
    "testmdp_mlp_tanh_plain": testmdp_mlp_tanh,
    "small_conv_dueling": small_conv_dueling,
    "small_conv_plain_single_q": small_conv_plain,
    "cfg2_nature_dueling_b4": nature_dueling,
}


def golden_params(name, net, g):
    """(p_on, p_tg) flat fp32 for a golden case (regenerated from the seed when not stored)."""
    if "p_on" in g:
        return g["p_on"].astype(np.float32), g["p_tg"].astype(np.float32)
    seed = int(g["seed"])
    rng = np.random.default_rng(seed)
    p_on, p_tg = O.init_params(net, seed=seed + 1), O.init_params(net, seed=seed + 2)
    for i in range(1, len(p_on), 2):  # same draws as oracle/make_golden.py
        p_on[i] = (0.1 * rng.standard_normal(p_on[i].shape)).astype(np.float32)
        p_tg[i] = (0.1 * rng.standard_normal(p_tg[i].shape)).astype(np.float32)
    return O.Network.flatten(p_on), O.Network.flatten(p_tg)


def fill_replay_from_batch(h, g):
    """Put the golden batch into a replay so that get_batch(0..B-1) returns it with the golden IS
    weights: we add with td_err chosen so that the priorities reproduce g['w'] is NOT possible in
    general, so golden cases drive the step through explicit indices and compare with the oracle
    using the engine's own IS weights (returned by get_batch)."""
    B = int(g["B"])
    h.replay_add(g["s"], g["a"], g["r"], g["sp"], g["done"].astype(np.uint8), td_err=np.abs(g["r"]))
    return np.arange(B, dtype=np.int64)
