"""GPU: the reference's own end-to-end tests (test/runtests.jl:45-111) run through the solve() mirror and the HIP engine:
TestMDP((5,5),4,6), Chain(flattenbatch, Dense(100,8,tanh), Dense(8,4)), 10 000 steps, lr 0.005, four DQN variants,
average return >= 1.5 of the optimum 2.1 (test/test_env.jl:7-8)."""
import importlib

import numpy as np
import pytest

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    pkg = ge.load_package()
    pkg.lib()
    return pkg, importlib.import_module(pkg.__name__ + ".nn"), importlib.import_module(pkg.__name__ + ".envs"), importlib.import_module(pkg.__name__ + ".solver")


def evaluate(env, policy, n_ep=100, max_steps=100):   # test/runtests.jl:28-42
    tot = 0.0
    for _ in range(n_ep):
        env.reset()
        r, step = 0.0, 0
        while not env.terminated()[0] and step < max_steps:
            a = policy.action(env.observe()[0])
            r += float(env.act(np.array([a]))[0])
            step += 1
        tot += r
    return tot / n_ep


@pytest.mark.parametrize("double_q,dueling,per", [(False, False, False), (True, False, False), (False, True, False), (True, True, True)],
                         ids=["vanilla", "double_q", "dueling", "prioritized_ddqn"])
def test_testmdp_learning_threshold(mods, double_q, dueling, per):
    pkg, nn, envs, S = mods
    env = envs.TestMDP((5, 5), 4, 6, n=1, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.Dense(100, 8, nn.tanh), nn.Dense(8, env.n_actions))
    max_steps = 10000
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=max_steps / 2), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=max_steps, learning_rate=0.005, exploration_policy=expl, eval_freq=2000,
                                   num_ep_eval=100, log_freq=500, double_q=double_q, dueling=dueling, prioritized_replay=per,
                                   verbose=False, logdir=None)
    policy = S.solve(solver, env)
    r = evaluate(env, policy)
    assert r >= 1.5, r
    env.reset()
    assert policy.actionvalues(env.observe()[0]).shape == (env.n_actions,)      # test/runtests.jl:60
    with pytest.raises(pkg.DQNError, match="NNPolicyError: was expecting an array with 3 dimensions"):   # src/policy.jl:44
        policy.action(np.zeros((5,), np.float32))
    policy.engine.close()


def test_gridworld_config1_runs(mods):
    """BASELINE config 1: SimpleGridWorld, Chain(Dense(2,32), Dense(32,4)), double_q + dueling + prioritized (README.md:26-46)."""
    pkg, nn, envs, S = mods
    env = envs.SimpleGridWorld(n=1, seed=3)
    model = nn.Chain(nn.Dense(2, 32), nn.Dense(32, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=1000), rng=np.random.default_rng(2))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=2000, exploration_policy=expl, learning_rate=0.005, log_freq=500,
                                   double_q=True, dueling=True, prioritized_replay=True, verbose=False, logdir=None)
    policy = S.solve(solver, env)
    assert policy.actionvalues(np.array([1.0, 1.0], np.float32)).shape == (4,)
    assert policy.action(np.array([9.0, 2.0], np.float32)) in range(4)
    policy.engine.close()


def test_testmdp_drqn(mods):
    """test/runtests.jl:115-129: TestMDP((5,5),1,6), Chain(flattenbatch, LSTM(25,8), Dense(8,4)), recurrence=true, double_q; return >= 0."""
    pkg, nn, envs, S = mods
    env = envs.TestMDP((5, 5), 1, 6, n=1, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.LSTM(25, 8), nn.Dense(8, env.n_actions))
    max_steps = 4000
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=max_steps / 2), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=max_steps, learning_rate=0.005, exploration_policy=expl, eval_freq=2000, num_ep_eval=20,
                                   log_freq=500, double_q=True, dueling=False, recurrence=True, verbose=False, logdir=None)
    policy = S.solve(solver, env)
    tot = 0.0
    for _ in range(50):
        env.reset(); policy.resetstate()
        r, step = 0.0, 0
        while not env.terminated()[0] and step < 100:
            r += float(env.act(np.array([policy.action(env.observe()[0])]))[0]); step += 1
        tot += r
    assert tot / 50 >= 0.0
    policy.engine.close()


def test_gridworld_ddrqn_dueling(mods):
    """test/runtests.jl:131-147: SimpleGridWorld, LSTM(2,32) -> Dense(32,4), trace_length 10, dueling + double-Q DRQN runs end to end."""
    pkg, nn, envs, S = mods
    env = envs.SimpleGridWorld(n=1, seed=3)
    model = nn.Chain(nn.flattenbatch, nn.LSTM(2, 32), nn.Dense(32, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=1000), rng=np.random.default_rng(2))
    solver = S.DeepQLearningSolver(qnetwork=model, prioritized_replay=False, max_steps=2000, exploration_policy=expl, learning_rate=0.001, log_freq=500,
                                   recurrence=True, trace_length=10, double_q=True, dueling=True, verbose=False, logdir=None)
    policy = S.solve(solver, env)
    assert policy.actionvalues(np.array([3.0, 4.0], np.float32)).shape == (4,)
    with pytest.raises(pkg.DQNError, match="recurrent model but recurrence is set to false"):
        S.solve(S.DeepQLearningSolver(qnetwork=model, exploration_policy=expl, recurrence=False, verbose=False, logdir=None), env)
    policy.engine.close()


def test_testmdp_device_envs(mods):
    """The reference's TestMDP end-to-end test (test/runtests.jl:45-57: return >= 1.5 of the optimal 2.1) with the env loop on the device:
    16 lock-stepped copies, eps-greedy and add_exp! in HBM (dqn_rollout), same solver settings."""
    pkg, nn, envs, S = mods
    env = envs.TestMDP((5, 5), 4, 6, n=16, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.Dense(100, 8, nn.tanh), nn.Dense(8, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=1000 / 2))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=1000, learning_rate=0.005, exploration_policy=expl, eval_freq=500, num_ep_eval=16, train_freq=1,
                                   log_freq=500, double_q=False, dueling=False, prioritized_replay=True, verbose=False, logdir=None, device_envs=True,
                                   buffer_size=4096, train_start=64)
    policy = S.solve(solver, env)
    ev = envs.TestMDP((5, 5), 4, 6, n=1, seed=7)
    tot = 0.0
    for _ in range(20):
        ev.reset()
        r = 0.0
        while not ev.terminated()[0]:
            r += float(ev.act(np.array([policy.action(ev.observe()[0])]))[0])
        tot += r
    assert tot / 20 >= 1.5
    policy.engine.close()
