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


class _UserEnv:
    """Minimal user-defined environment in the envs.py protocol (n lock-stepped copies): what CommonRLInterface / POMDPs.jl
    adapters give the reference's solve()."""
    n_actions, obs_shape, discount = 2, (1,), 0.95

    def __init__(self, n=1):
        self.n = n
        self.reset()

    def reset(self, mask=None):
        if mask is None:
            self.s = np.ones(self.n, np.int64)
        else:
            self.s[mask] = 1

    def observe(self):
        return self.s.astype(np.float32)[:, None]

    def terminated(self):
        return self.s >= 3


class SimpleEnv(_UserEnv):
    """test/runtests.jl:199-216: actions [-1, +1]; act! returns the state BEFORE the move; s = max(1, s + a); terminal at s >= 3."""

    def act(self, a):
        r = self.s.astype(np.float32)
        self.s = np.maximum(1, self.s + np.where(np.asarray(a) == 0, -1, 1))
        return r


class StaticArrayMDP(_UserEnv):
    """test/runtests.jl:165-181: state SVector(1); actions [0, 1]; sp = s + a; r = m.state[1]^2 == 1; terminal at s >= 3."""

    def act(self, a):
        self.s = self.s + np.asarray(a)
        return np.ones(self.n, np.float32)


@pytest.mark.parametrize("cls", [SimpleEnv, StaticArrayMDP], ids=["common_rl_env", "static_array_env"])
def test_user_defined_env_ten_steps(mods, cls):
    """test/runtests.jl:165-234: solve() on a user-defined env with a 1-element observation, Chain(Dense(1,32), Dense(32,2)), max_steps = 10,
    double_q + dueling + prioritized; the rolled-out return is > 1.0."""
    pkg, nn, envs, S = mods
    env = cls()
    model = nn.Chain(nn.Dense(1, 32), nn.Dense(32, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=5), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=10, exploration_policy=expl, learning_rate=0.005, log_freq=500,
                                   recurrence=False, double_q=True, dueling=True, prioritized_replay=True, verbose=False, logdir=None)
    policy = S.solve(solver, env)
    assert evaluate(env, policy, n_ep=1, max_steps=50) > 1.0
    assert policy.actionvalues(np.array([1.0], np.float32)).shape == (2,)
    policy.engine.close()


class TigerPOMDP:
    """POMDPModels.TigerPOMDP(r_listen, r_findtiger, r_escapetiger, p_listen_correctly, discount) (third-party; recalled) as the env the
    reference's POMDP path produces: observation = convert_o(Vector, o) = [tiger heard on the left?]; actions listen / open-left / open-right;
    opening a door ends the episode."""
    n_actions, obs_shape = 3, (1,)

    def __init__(self, r_listen=0.01, r_find=-1.0, r_escape=0.1, p_correct=0.8, discount=0.95, n=1, seed=0):
        self.r_listen, self.r_find, self.r_escape, self.p_correct, self.discount, self.n = r_listen, r_find, r_escape, p_correct, discount, n
        self.rng = np.random.default_rng(seed)
        self.reset()

    def reset(self, mask=None):
        new = self.rng.integers(0, 2, self.n)
        if mask is None:
            self.tiger, self.done, self.o = new, np.zeros(self.n, bool), np.zeros(self.n, np.float32)
        else:
            self.tiger[mask], self.done[mask], self.o[mask] = new[mask], False, 0.0

    def observe(self):
        return self.o[:, None].copy()

    def terminated(self):
        return self.done

    def act(self, a):
        a = np.asarray(a)
        correct = self.rng.random(self.n) < self.p_correct
        heard = np.where(correct, self.tiger, 1 - self.tiger)
        self.o = np.where(a == 0, heard, self.rng.integers(0, 2, self.n)).astype(np.float32)
        opened_tiger = (a - 1) == self.tiger
        r = np.where(a == 0, self.r_listen, np.where(opened_tiger, self.r_find, self.r_escape)).astype(np.float32)
        self.done = a != 0
        return r


def test_tiger_pomdp_ddrqn_shape(mods):
    """test/runtests.jl:149-163: POMDP path, Chain(flattenbatch, LSTM(input_dims, 4), Dense(4, 3)), recurrence, trace_length 10, dueling + double-Q,
    target_update_freq 1000; asserts only size(actionvalues(policy, o)) == (n_actions,).  (2000 of the reference's 10 000 steps.)"""
    pkg, nn, envs, S = mods
    env = TigerPOMDP(0.01, -1.0, 0.1, 0.8, 0.95)
    model = nn.Chain(nn.flattenbatch, nn.LSTM(1, 4), nn.Dense(4, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=1000), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, prioritized_replay=False, max_steps=2000, learning_rate=0.0001, exploration_policy=expl, log_freq=500,
                                   target_update_freq=1000, recurrence=True, trace_length=10, double_q=True, dueling=True, max_episode_length=100,
                                   verbose=False, logdir=None)
    policy = S.solve(solver, env)
    assert policy.actionvalues(np.array([1.0], np.float32)).shape == (env.n_actions,)
    policy.engine.close()


def test_headline_config_end_to_end_on_device(mods):
    """BASELINE configs 2/3 end to end: TestMDP((84,84),4,6) (84x84x4 observations), Nature-DQN 3-conv + 2-dense dueling, double-Q, prioritized replay,
    B = 32, 32 device-resident env copies, the dqn_train! loop on the device (dqn_rollout + dqn_evaluate).  Known answer of the MDP: optimal return 2.1
    (test/test_env.jl:7-8); the reference's own threshold on its small-image variant is 1.5 (test/runtests.jl:110)."""
    pkg, nn, envs, S = mods
    env = envs.TestMDP((84, 84), 4, 6, n=32, seed=7)
    steps = 3000
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.01, steps=steps / 2))
    solver = S.DeepQLearningSolver(qnetwork=nn.nature_dqn(n_actions=4, in_channels=4), max_steps=steps, learning_rate=1e-4, exploration_policy=expl,
                                   train_freq=4, target_update_freq=500, eval_freq=1000, num_ep_eval=32, log_freq=1000, double_q=True, dueling=True,
                                   prioritized_replay=True, buffer_size=20000, train_start=640, verbose=False, logdir=None, device_envs=True)
    policy = S.solve(solver, env)
    r, st = policy.engine.evaluate(64, 100, seed=99)
    assert r >= 1.5 and st == 5.0, (r, st)
    policy.engine.close()


@pytest.mark.parametrize("device_envs", [False, True], ids=["host_loop", "device_envs"])
def test_save_model_and_restore_best_model_through_the_hip_engine(mods, tmp_path, monkeypatch, capsys, device_envs):
    """SURVEY 8(f)-3 / src/solver.jl:290-318 on the HIP engine: solve(...; logdir) writes qnetwork.bson at the evaluations that follow a
    save_freq mark whenever the score did not get worse; the arrays in the file are the engine's online parameters AT SAVE TIME bit for bit
    (Flux.params order, Julia sizes), and restore_best_model (verbose route, :170-172) puts exactly them back into the engine."""
    pkg, nn, envs, S = mods
    bson = importlib.import_module(pkg.__name__ + ".bson")
    env = envs.TestMDP((5, 5), 4, 6, n=8 if device_envs else 1, seed=7)
    model = nn.Chain(nn.flattenbatch, nn.Dense(100, 8, nn.tanh), nn.Dense(8, env.n_actions))
    expl = S.EpsGreedyPolicy(env, S.LinearDecaySchedule(start=1.0, stop=0.05, steps=600), rng=np.random.default_rng(1))
    solver = S.DeepQLearningSolver(qnetwork=model, max_steps=1200, learning_rate=0.005, exploration_policy=expl, eval_freq=200, save_freq=200,
                                   num_ep_eval=20, log_freq=400, double_q=True, dueling=True, prioritized_replay=True, train_start=64,
                                   verbose=True, logdir=str(tmp_path / "log"), **({"device_envs": True} if device_envs else {}))
    saved = []                                        # (flat online parameters at save time, shapes) of every save
    real_save = bson.save_qnetwork

    def spy(path, flat, shapes):
        saved.append((np.array(flat, np.float32, copy=True), list(shapes)))
        return real_save(path, flat, shapes)

    monkeypatch.setattr(bson, "save_qnetwork", spy)
    policy = S.solve(solver, env)
    out = capsys.readouterr().out
    path = tmp_path / "log" / "qnetwork.bson"
    assert path.exists() and saved, "no model was saved"
    assert "Saving new model with eval reward" in out
    w, sizes = bson.load_qnetwork(path)
    last_flat, last_shapes = saved[-1]
    np.testing.assert_array_equal(w, last_flat)                                   # the file holds the parameters of the LAST save, bit for bit
    assert sizes == [s for s, _ in last_shapes] == [s for s, _ in bson.julia_param_shapes(policy.qnetwork)]
    assert sum(int(np.prod(s)) for s in sizes) == policy.engine.P
    # verbose route: restore_best_model ran at the end of dqn_train! -> the engine's online network IS the saved one again
    np.testing.assert_array_equal(policy.engine.get_params(pkg.NET_ONLINE), w)
    # and the restore works on its own: perturb the engine, restore, compare
    policy.engine.set_params(w * np.float32(0.5), pkg.NET_ONLINE)
    S.restore_best_model(solver, policy)
    np.testing.assert_array_equal(policy.engine.get_params(pkg.NET_ONLINE), w)
    policy.engine.close()
