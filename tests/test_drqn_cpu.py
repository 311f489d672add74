"""CPU: the C twin's DRQN path (EpisodeReplayBuffer with the prefix-copy quirk, canonical-order LSTM + BPTT) against the
fp64 oracle (itself pinned by torch autograd, oracle/make_golden.py) on the same sampled batches."""
import numpy as np
import pytest

import dqn_oracle as O
import ref
from drqn_common import check_against_oracle, drqn_nets, feed, make_episodes, make_handle


@pytest.mark.parametrize("name", list(drqn_nets()))
def test_twin_drqn_matches_fp64_oracle(name):
    net, B, T, kw = drqn_nets()[name]
    rng = np.random.default_rng(5)
    h, hp, layers = make_handle(ref.Twin, net, B, T, kw, cap=max(12, B + 4), threads=4)
    eps = make_episodes(net, max(12, B + 4) + 3, T, rng)     # more than the ring holds: exercises the episode ring wrap
    feed(h, eps)
    cap = h.episode_count()[1]
    assert h.episode_count() == (cap, cap)
    ring = [None] * cap
    for i, ep in enumerate(eps):
        ring[i % cap] = ep
    p_on = (O.Network.flatten(O.init_params_recurrent(net, 3)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    p_tg = (O.Network.flatten(O.init_params_recurrent(net, 4)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    h.set_params(p_on, 0); h.set_params(p_tg, 1)
    np.testing.assert_array_equal(h.get_params(0), p_on)
    check_against_oracle(h, net, ring, B, T, kw, rng, (p_on, p_tg))
    h.close()


def test_recurrent_model_without_recurrence_is_rejected():
    net, B, T, kw = drqn_nets()["lstm_single_q"]
    hp = ref.hparams_for(net, batch_size=B, buffer_size=8, recurrence=0)
    with pytest.raises(ref.abi.DQNError, match="recurrent model but recurrence is set to false"):   # src/solver.jl:45-47
        ref.Twin(ref.layers_from_network(net), hp)


@pytest.mark.parametrize("name", ["cfg4_lstm_plain", "dense_lstm_dueling"])
def test_twin_hidden_state_save_restore(name):
    """hiddenstates / sethiddenstates! (src/helpers.jl:61-79) around batch_train! (src/solver.jl:137-139)"""
    from drqn_common import check_hidden_state_protocol
    net, B, T, kw = drqn_nets()[name]
    rng = np.random.default_rng(17)
    h, hp, layers = make_handle(ref.Twin, net, B, T, kw, cap=B + 4, threads=2)
    feed(h, make_episodes(net, B + 4, T, rng))
    p_on = (O.Network.flatten(O.init_params_recurrent(net, 3)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    h.set_params(p_on, 0); h.set_params(p_on * 0.9, 1)
    check_hidden_state_protocol(h, net, p_on, rng)
    h.close()


def test_recurrence_with_u8_replay_is_rejected():
    """the episode replay stores Float32 rows; a u8 replay would make dqn_episode_add read 4x past the caller's byte buffer"""
    net, B, T, kw = drqn_nets()["lstm_single_q"]
    hp = ref.hparams_for(net, batch_size=B, buffer_size=8, recurrence=1, trace_length=T, obs_dtype=ref.abi.OBS_U8)
    with pytest.raises(ref.abi.DQNError, match="u8 is not supported with recurrence"):
        ref.Twin(ref.layers_from_network(net), hp)
