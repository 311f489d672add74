"""Shared pieces of the device-environment tests: networks sized for TestMDP / SimpleGridWorld."""
import numpy as np

import dqn_oracle as O

R, I = O.ACT_RELU, O.ACT_IDENTITY


def testmdp_conv_dueling(h=12, w=14, stack=4):
    """image-obs TestMDP((w,h), stack, 6): 4 actions, a small conv trunk + dueling heads."""
    conv = [O.Conv(4, stack, 8, R, 2), O.Conv(3, 8, 16, R, 1)]
    oh, ow = (h - 4) // 2 + 1 - 2, (w - 4) // 2 + 1 - 2
    b, v, a = O.create_dueling_network(conv + [O.Dense(16 * oh * ow, 32, R), O.Dense(32, 4, I)])
    return O.Network((stack, h, w), b, v, a)


def testmdp_wide_fc_dueling(h=20, w=20, stack=4):
    """a trunk whose flattened output (32 x 7 x 7 = 1568 > 1024) makes the dueling streams' hidden layers split-K: the shape class of the Nature network, where the
    fused reduce + head launch (red_head.hip) applies -- in the train step and, in its acting form, in the env loop's policy forward"""
    b, v, a = O.create_dueling_network([O.Conv(4, stack, 32, R, 2), O.Conv(3, 32, 32, R, 1), O.Dense(32 * 7 * 7, 128, R), O.Dense(128, 4, I)])
    return O.Network((stack, h, w), b, v, a)


def testmdp_wide_fc_plain(h=20, w=20, stack=4):
    """the same trunk with a plain (non-dueling) Q head: one stream through the fused acting tail"""
    return O.Network((stack, h, w), [O.Conv(4, stack, 32, R, 2), O.Conv(3, 32, 32, R, 1), O.Dense(32 * 7 * 7, 64, R), O.Dense(64, 4, I)])


def gridworld_mlp_dueling():
    b, v, a = O.create_dueling_network([O.Dense(2, 32, R), O.Dense(32, 4, I)])   # README.md:38
    return O.Network((2,), b, v, a)


def same_params(handles, net, seed=3):
    p = O.Network.flatten(O.init_params(net, seed=seed))
    rng = np.random.default_rng(seed)
    p = (p + 0.01 * rng.standard_normal(p.shape)).astype(np.float32)
    for h in handles:
        h.set_params(p, 0)
        h.sync_target()
    return p
