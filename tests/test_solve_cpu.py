"""CPU: the host-side mirror of the reference interface (deepqlearning.jl_amd/solver.py: solve, dqn_train, batch_train dispatch, replay
protocol, NNPolicy, exploration schedule, evaluation, model save/restore) run end to end with the CPU twin standing in for the engine --
the twin exposes the same Handle API as the product, so the host logic is exercised without a GPU.  Same scenarios as the reference's own
end-to-end tests (test/runtests.jl:45-147, 165-234), with smaller step counts where the assertion is only "runs and has the right shape"."""
import importlib

import numpy as np
import pytest

import __graft_entry__ as ge
import ref

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
envs = importlib.import_module(pkg.__name__ + ".envs")
S = importlib.import_module(pkg.__name__ + ".solver")


def twin_engine(layers, hp, device=0):
    return ref.Twin(layers, hp, plan=None, threads=4)


def evaluate(env, policy, n_ep=30, max_steps=100):      # test/runtests.jl:28-42
    tot = 0.0
    for _ in range(n_ep):
        env.reset(); policy.resetstate()
        r, step = 0.0, 0
        while not env.terminated()[0] and step < max_steps:
            r += float(env.act(np.array([policy.action(env.observe()[0])]))[0]); step += 1
        tot += r
    return tot / n_ep


@pytest.mark.parametrize("double_q,dueling,per", [(False, False, False), (True, True, True)], ids=["vanilla", "prioritized_ddqn"])
def test_testmdp_learning_threshold_host_loop(double_q, dueling, per):
    """test/runtests.jl:45-61 and :96-111: TestMDP((5,5),4,6), Chain(flattenbatch, Dense(100,8,tanh), Dense(8,4)), 10 000 steps, lr 0.005: return >= 1.5."""
    env = envs.TestMDP((5, 5), 4, 6, n=1, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.Dense(100, 8, nn.tanh), nn.Dense(8, env.n_actions))
    max_steps = 10000
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=max_steps / 2), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=max_steps, learning_rate=0.005, exploration_policy=expl, eval_freq=2000, num_ep_eval=20,
                                   log_freq=500, double_q=double_q, dueling=dueling, prioritized_replay=per, verbose=False, logdir=None)
    policy = S.solve(solver, env, engine_cls=twin_engine)
    assert evaluate(env, policy) >= 1.5
    env.reset()
    assert policy.actionvalues(env.observe()[0]).shape == (env.n_actions,)                      # test/runtests.jl:60
    with pytest.raises(pkg.DQNError, match="NNPolicyError: was expecting an array with 3 dimensions"):   # src/policy.jl:44
        policy.action(np.zeros((5,), np.float32))
    assert policy.value(env.observe()[0]) == pytest.approx(float(policy.actionvalues(env.observe()[0]).max()))


def test_device_env_loop_through_the_mirror():
    """solver.device_envs = True routes dqn_train! through dqn_envs_create / dqn_rollout / dqn_evaluate (here: the twin's restatement of them)."""
    env = envs.TestMDP((5, 5), 4, 6, n=8, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.Dense(100, 8, nn.tanh), nn.Dense(8, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=600))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=1500, learning_rate=0.005, exploration_policy=expl, eval_freq=500, num_ep_eval=8, train_freq=1,
                                   log_freq=500, double_q=False, dueling=False, prioritized_replay=True, verbose=False, logdir=None, device_envs=True,
                                   buffer_size=4096, train_start=64)
    policy = S.solve(solver, env, engine_cls=twin_engine)
    assert policy.engine.evaluate(8, 100, seed=3)[0] >= 1.5
    with pytest.raises(pkg.DQNError, match="device_envs drives the feed-forward path"):
        S.dqn_train(S.DeepQLearningSolver(qnetwork=model, exploration_policy=expl, recurrence=True, device_envs=True, verbose=False, logdir=None), env, policy, None)


def test_drqn_loop_and_recurrence_check():
    """test/runtests.jl:115-147 shape: recurrent model, EpisodeReplayBuffer dispatch of batch_train!, resetstate!; and the solver's own
    consistency check (src/solver.jl:45-47)."""
    env = envs.TestMDP((5, 5), 1, 6, n=1, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.LSTM(25, 8), nn.Dense(8, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=400), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=800, learning_rate=0.005, exploration_policy=expl, eval_freq=400, num_ep_eval=5, log_freq=500,
                                   double_q=True, dueling=False, recurrence=True, trace_length=6, verbose=False, logdir=None)
    policy = S.solve(solver, env, engine_cls=twin_engine)
    assert policy.actionvalues(env.observe()[0]).shape == (env.n_actions,)
    assert np.isfinite(evaluate(env, policy, n_ep=5))
    with pytest.raises(pkg.DQNError, match="recurrent model but recurrence is set to false"):
        S.solve(S.DeepQLearningSolver(qnetwork=model, exploration_policy=expl, recurrence=False, verbose=False, logdir=None), env, engine_cls=twin_engine)


def test_gridworld_config1_and_model_save_restore(tmp_path):
    """README.md:26-46 (config 1) through solve(); save_model / restore_best_model (src/solver.jl:290-318) round trip through the log directory."""
    env = envs.SimpleGridWorld(n=1, seed=3)
    model = nn.Chain(nn.Dense(2, 32), nn.Dense(32, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=500), rng=np.random.default_rng(2))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=1200, exploration_policy=expl, learning_rate=0.005, log_freq=500, eval_freq=400, save_freq=400,
                                   num_ep_eval=5, double_q=True, dueling=True, prioritized_replay=True, verbose=True, logdir=str(tmp_path))
    policy = S.solve(solver, env, engine_cls=twin_engine)
    assert policy.actionvalues(np.array([1.0, 1.0], np.float32)).shape == (4,)
    assert policy.action(np.array([9.0, 2.0], np.float32)) in range(4)
    bson = importlib.import_module(pkg.__name__ + ".bson")
    saved, sizes = bson.load_qnetwork(tmp_path / "qnetwork.bson")  # joinpath(logdir, "qnetwork.bson"), src/solver.jl:292
    assert sizes == [(32, 2), (32,), (1, 32), (1,), (32, 2), (32,), (4, 32), (4,)]    # Flux.params order base | val | adv, Julia sizes (out, in)
    assert saved.shape == policy.getnetwork().shape
    np.testing.assert_array_equal(saved, policy.getnetwork())      # verbose=True: the best model was restored at the end (src/solver.jl:170-176)


def test_replay_protocol_and_schedule():
    """HIPReplayBuffer mirrors add_exp! / sample / get_batch / update_priorities! / is_full / max_size (…replay.jl:61-104); LinearDecaySchedule and
    EpsGreedyPolicy (POMDPTools) behave as the reference's exploration policy."""
    env = envs.SimpleGridWorld(n=4, seed=1)
    net = nn.create_dueling_network(nn.Chain(nn.Dense(2, 8), nn.Dense(8, 4)))
    solver = S.DeepQLearningSolver(qnetwork=net, batch_size=4, buffer_size=12)
    eng = S.make_engine(twin_engine, solver, env, net, 0.95)
    eng.set_params(nn.glorot_params(net, seed=1), 0)
    rb = S.HIPReplayBuffer(eng)
    assert rb.max_size() == 12 and not rb.is_full() and rb.batch_size == 4
    S.populate_replay_buffer(rb, env, max_pop=12, rng=np.random.default_rng(0))
    assert rb.is_full()
    s, a, r, sp, d, idx, w = rb.sample()                      # (s, a, r, sp, done, indices, weights), ...replay.jl:86,103
    assert s.shape[0] == 4 and idx.shape == (4,) and (w > 0).all()
    rb.update_priorities(idx, np.array([0.5, -1.0, 2.0, 0.0], np.float32))
    got = rb.get_batch(idx)
    assert len(got) == 7 and np.array_equal(got[-1], idx)
    sch = S.LinearDecaySchedule(start=1.0, stop=0.1, steps=10)
    assert sch(0) == 1.0 and sch(5) == pytest.approx(0.55) and sch(10) == pytest.approx(0.1) and sch(1000) == pytest.approx(0.1)
    pol = S.NNPolicy(env, eng, list(range(4)), 1)
    expl = S.EpsGreedyPolicy(env, 0.0, rng=np.random.default_rng(0))
    obs = env.observe()
    np.testing.assert_array_equal(expl.action(pol, 1, obs), pol.action(obs))          # eps = 0: greedy
    assert set(np.unique(S.EpsGreedyPolicy(env, 1.0, rng=np.random.default_rng(0)).action(pol, 1, np.repeat(obs, 50, 0)))) <= {0, 1, 2, 3}
