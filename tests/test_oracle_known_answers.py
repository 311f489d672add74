"""CPU: closed-form known answers that follow directly from the cited reference lines -- the pins of
the oracle (NumPy) and of the C twin that do not depend on any other implementation."""
import importlib

import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import ref
from nets import cfg1_mlp_dueling, small_conv_dueling


def test_huber_known_values():  # src/helpers.jl:14-19
    assert O.huber_loss(np.float64(0.5)) == 0.125
    assert O.huber_loss(np.float64(2.0)) == 1.5
    assert O.huber_loss(np.float64(-2.0)) == 1.5
    assert O.huber_loss(np.float64(1.0)) == 0.5


def test_uniform_priorities_give_unit_is_weights():  # ...replay.jl:101-102
    pr = np.full(37, 0.3, np.float32)
    np.testing.assert_allclose(O.is_weights(pr[:8], pr, 0.4), 1.0, rtol=1e-6)


def test_per_off_priority_is_eps_pow_alpha():  # src/solver.jl:94, ...replay.jl:67
    assert O.priority_from_td(0.0, 1e-3, 0.6) == np.float32(np.float64(np.float32(1e-3)) ** np.float64(np.float32(0.6)))


def test_dueling_mean_identity_and_split_rule():  # src/dueling.jl:10, :36-58
    net = small_conv_dueling()
    assert [l.kind for l in net.base] == ["conv", "conv"] and len(net.val) == 2 and len(net.adv) == 2
    assert net.val[-1].n_out == 1 and net.val[-1].n_in == net.adv[-1].n_in
    ps = [p.astype(np.float64) for p in O.init_params(net, seed=2)]
    x = np.random.default_rng(0).random((5,) + net.obs_shape)
    q = O.network_forward(net, ps, x)
    nb = 2 * len(net.base)
    xb, _ = O._chain_forward(net.base, ps[:nb], x)
    v, _ = O._chain_forward(net.val, ps[nb:nb + 2 * len(net.val)], xb)
    np.testing.assert_allclose((q - v).mean(axis=1), 0.0, atol=1e-12)     # mean_a(Q - V) = 0 per column
    mlp = cfg1_mlp_dueling()                                              # all-Dense chain: duel_layer = 0, nothing shared
    assert mlp.base == [] and [l.n_out for l in mlp.val] == [32, 1] and [l.n_out for l in mlp.adv] == [32, 4]
    with pytest.raises(ValueError, match="incompatible with dueling"):
        O.create_dueling_network([])


def test_done_means_target_is_reward_and_first_max_tie_rule():  # src/solver.jl:212-217
    q_on = np.array([[1.0, 3.0, 3.0, 0.0], [2.0, 2.0, 2.0, 2.0]])
    q_tg = np.array([[10.0, 20.0, 30.0, 40.0], [1.0, 2.0, 3.0, 4.0]])
    y, best = O.bellman_targets(q_on, q_tg, np.array([0.5, -1.0]), np.array([1.0, 0.0]), 0.9, True)
    assert list(best) == [1, 0]                     # first maximum wins
    assert y[0] == 0.5                              # done = 1 => y = r
    assert y[1] == -1.0 + 0.9 * 1.0                 # target net indexed by the ONLINE argmax
    y2, _ = O.bellman_targets(q_on, q_tg, np.array([0.5, -1.0]), np.array([0.0, 0.0]), 0.9, False)
    assert y2[0] == 0.5 + 0.9 * 40.0                # non-double: max of the target net


def test_replay_ring_wrap_twin_and_oracle():  # mod1 ring, ...replay.jl:70
    net = cfg1_mlp_dueling()
    hp = ref.hparams_for(net, batch_size=4, buffer_size=5)
    tw = ref.Twin(ref.layers_from_network(net), hp)
    orc = O.PrioritizedReplay(net.obs_shape, 5, 4)
    for i in range(8):
        s = np.full((1, 2), i, np.float32)
        tw.replay_add(s, [i % 4], [float(i)], s + 0.5, [i % 2])
        orc.add_exp(s[0], i % 4, float(i), s[0] + 0.5, i % 2)
    assert tw.replay_size() == (5, 5)
    b = tw.get_batch(np.array([0, 1, 2, 3], np.int64))
    np.testing.assert_array_equal(b[0][:, 0, 0, 0], [5, 6, 7, 3])       # slots 0..2 were overwritten by transitions 5..7
    np.testing.assert_array_equal(b[0].reshape(4, 2), orc.get_batch([0, 1, 2, 3])[0])
    np.testing.assert_array_equal(tw.replay_priorities(), orc.prio)
    tw.close()


def test_testmdp_known_answer():  # test/test_env.jl:7-8: optimal return 2.1 with policy [2,1,2,1,3]
    pkg = ge.load_package()
    envs = importlib.import_module(pkg.__name__ + ".envs")
    env = envs.TestMDP((5, 5), 4, 6, n=1)
    tot = 0.0
    for a in [2, 1, 2, 1, 3]:
        tot += float(env.act(np.array([a - 1]))[0])
    assert env.terminated()[0]
    assert abs(tot - 2.1) < 1e-6
    o = envs.TestMDP((84, 84), 4, 6, n=3).observe()
    assert o.shape == (3, 4, 84, 84) and o.dtype == np.float32 and 0 < o.min() and o.max() <= 200 / 255 + 1e-6
