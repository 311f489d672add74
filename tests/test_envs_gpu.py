"""GPU (-m gpu): the device-resident env loop (dqn_envs_create / dqn_rollout) against the CPU twin's
restatement -- BIT-EXACT trajectories (observations, eps-greedy actions, rewards, terminals), replay contents and
priorities, episode statistics, and the parameters after interleaved training (src/solver.jl:82-145)."""
import importlib

import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import envs_common as EC
import ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.lib()
    return p


@pytest.fixture(scope="module")
def envs(pkg):
    return importlib.import_module(pkg.__name__ + ".envs")


def make_pair(pkg, net, B, cap, plan_edit=None, **kw):
    hp = ref.hparams_for(net, batch_size=B, buffer_size=cap, **kw)
    layers = ref.layers_from_network(net)
    plan = pkg.default_plan(layers, hp)
    if plan_edit:
        plan = plan_edit(plan)
    return pkg.Engine(layers, hp, plan=plan), ref.Twin(layers, hp, plan=plan, threads=8), hp


def compare_state(g, t):
    for x, y in zip(g.envs_peek(), t.envs_peek()):
        np.testing.assert_array_equal(x, y)
    assert g.replay_size() == t.replay_size()
    np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())


@pytest.mark.parametrize("u8,mfma", [(False, 1), (True, 1), (False, 0)])
def test_testmdp_rollout_bit_exact(pkg, envs, u8, mfma):
    net = EC.testmdp_conv_dueling()
    g, t, hp = make_pair(pkg, net, B=8, cap=96, obs_dtype=1 if u8 else 0, use_mfma=mfma)
    EC.same_params([g, t], net)
    spec = envs.TestMDP((14, 12), 4, 6, n=6, seed=3)
    for h in (g, t):
        h.envs_create(spec, max_episode_length=100, seed=17)
    compare_state(g, t)
    t0 = 1
    for chunk in (1, 3, 7, 12, 9):          # 32 vector steps: the 96-slot ring wraps, episodes end every 5 steps
        sg = g.rollout(chunk, t0=t0, train_freq=2, target_update_freq=5, eps=(1.0, 0.1, 20.0))
        st = t.rollout(chunk, t0=t0, train_freq=2, target_update_freq=5, eps=(1.0, 0.1, 20.0))
        t0 += chunk
        assert sg == st, (sg, st)
        compare_state(g, t)
        np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
        np.testing.assert_array_equal(g.get_params(1), t.get_params(1))
    assert sg["episodes"] > 0 or st["train_steps"] > 0
    idx = np.arange(8, dtype=np.int64) * 11
    for x, y in zip(g.get_batch(idx), t.get_batch(idx)):
        np.testing.assert_array_equal(x, y)


def test_gridworld_rollout_bit_exact(pkg, envs):
    net = EC.gridworld_mlp_dueling()
    g, t, hp = make_pair(pkg, net, B=32, cap=1024)
    EC.same_params([g, t], net)
    spec = envs.SimpleGridWorld(n=64)
    for h in (g, t):
        h.envs_create(spec, max_episode_length=20, seed=5)
    t0 = 1
    tot_eps = 0
    for chunk in (5, 40, 55):
        sg = g.rollout(chunk, t0=t0, train_freq=4, target_update_freq=25, eps=(1.0, 0.05, 60.0))
        st = t.rollout(chunk, t0=t0, train_freq=4, target_update_freq=25, eps=(1.0, 0.05, 60.0))
        t0 += chunk
        assert sg == st, (sg, st)
        compare_state(g, t)
        np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
        tot_eps = sg["episodes"]
    assert tot_eps >= 64 * 4          # max_episode_length 20 over 100 steps


@pytest.mark.parametrize("n_envs", [8, 12])
def test_rollout_with_fused_acting_head_bit_exact(pkg, envs, monkeypatch, n_envs):
    """r05: the device env loop on a network whose dueling streams have split-K hidden layers (the Nature shape class): its train steps take the fused reduce + head launch
    (red_head.hip).  Trajectories, replay, priorities, parameters and evaluation equal the twin's bit for bit, and the schedule without the fused launch (DQN_NO_RED_HEAD)
    walks the same trajectory.  (An ACTING form of that launch -- reduce + heads of the policy forward in one -- was built and measured no faster: 9.6 us vs 4.7 + 4.8,
    profiles/history/r05_o_acting_step_with_fused_head_dropped.txt; dropped.)"""
    net = EC.testmdp_wide_fc_dueling()
    outs = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("DQN_NO_RED_HEAD", "1")
        g, t, hp = make_pair(pkg, net, B=8, cap=128)
        monkeypatch.delenv("DQN_NO_RED_HEAD", raising=False)
        EC.same_params([g, t], net)
        spec = envs.TestMDP((20, 20), 4, 6, n=n_envs, seed=3)
        for h in (g, t):
            h.envs_create(spec, max_episode_length=100, seed=17)
        t0 = 1
        for chunk in (1, 4, 9, 6):
            kw = dict(t0=t0, train_freq=2, target_update_freq=7, eps=(0.6, 0.05, 15.0))
            sg = g.rollout(chunk, **kw); st = t.rollout(chunk, **kw)
            t0 += chunk
            assert sg == st, (sg, st)
            compare_state(g, t)
            np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
        assert g.evaluate(8, 50, seed=5) == t.evaluate(8, 50, seed=5)
        outs.append((g.get_params(0), g.replay_priorities()))
        g.close(); t.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0]); np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("case", ["wide_fc_dueling", "wide_fc_dueling_u8", "wide_fc_dueling_s13", "wide_fc_plain", "wide_fc_dueling_uniform", "gridworld", "gridworld_1024", "small_fc_dueling"])
def test_rollout_fused_acting_tail_both_schedules(pkg, envs, monkeypatch, case):
    """r06: the acting step's tail as ONE launch -- k_act_head (act_head.hip): split-K reduce of the heads' producers + heads + Q / argmax + eps-greedy +
    act! + add_exp!'s per-experience part, one wave per (four copies, stream, plan chunk) with a ticketed last arriver -- and the sum-tree ancestors as workgroup 0 of the observe
    launch.  Trajectories, replay (rows, metadata, priorities incl. a wrapping ring), episode statistics, parameters and evaluation equal the twin's bit for bit, and the
    four-launch tail (DQN_NO_ACT_HEAD=1) walks the same trajectory.  Cases: split-K dueling streams (the Nature shape class; f32 and u8 rows; 13 slabs; uniform replay without double-Q), a plain Q head (one stream),
    the GridWorld MLP (unsplit producers, one chunk), a dueling net whose producers are unsplit."""
    grid = case.startswith("gridworld")      # (gridworld_1024: the largest copy count, 256 groups = every RolloutDev record in use)
    net = {"wide_fc_dueling": EC.testmdp_wide_fc_dueling, "wide_fc_dueling_u8": EC.testmdp_wide_fc_dueling, "wide_fc_dueling_s13": EC.testmdp_wide_fc_dueling, "wide_fc_dueling_uniform": EC.testmdp_wide_fc_dueling, "wide_fc_plain": EC.testmdp_wide_fc_plain,
           "gridworld": EC.gridworld_mlp_dueling, "gridworld_1024": EC.gridworld_mlp_dueling, "small_fc_dueling": EC.testmdp_conv_dueling}[case]()
    n_envs = {"wide_fc_dueling": 8, "wide_fc_dueling_u8": 12, "wide_fc_dueling_s13": 8, "wide_fc_dueling_uniform": 8, "wide_fc_plain": 4, "gridworld": 64, "gridworld_1024": 1024, "small_fc_dueling": 8}[case]
    # plan edits (the twin takes the same plan): 13 slabs per hidden layer (the 16-slab instantiation of the launch); 32-row chunks for the plain net's 64-input head (the default
    # plan leaves it unsplit, which the launch does not cover)
    plan_edit = {"wide_fc_dueling_s13": lambda pl: [(128, dx, dw) if fk == 392 else (fk, dx, dw) for fk, dx, dw in pl],
                 "wide_fc_plain": lambda pl: pl[:-1] + [(32, pl[-1][1], pl[-1][2])]}.get(case)
    outs = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("DQN_NO_ACT_HEAD", "1")
        extra = dict(prioritized_replay=0, double_q=0) if case.endswith("uniform") else {}      # uniform replay (add_exp! with priority 0f0, no pre-drawn batches), plain Q targets
        g, t, hp = make_pair(pkg, net, B=8, cap=(3000 if n_envs == 1024 else 1024) if grid else 80, plan_edit=plan_edit, obs_dtype=1 if case.endswith("u8") else 0, **extra)
        monkeypatch.delenv("DQN_NO_ACT_HEAD", raising=False)
        EC.same_params([g, t], net)
        spec = envs.SimpleGridWorld(n=n_envs) if grid else envs.TestMDP((14, 12) if case == "small_fc_dueling" else (20, 20), 4, 6, n=n_envs, seed=3)
        for h in (g, t):
            h.envs_create(spec, max_episode_length=20 if grid else 100, seed=17)
        compare_state(g, t)
        assert g.envs_info() == (n_envs, fused), "the schedule under test is not the one this case names"
        t0 = 1
        for chunk, cad in ((1, False), (4, False), (3, True), (9, False), (5, True), (6, False)):      # the 80-slot ring wraps; episodes end every 5 steps (TestMDP) / within 20 (GridWorld)
            kw = dict(t0=t0, train_freq=2 if cad else 3, target_update_freq=7, eps=(0.6, 0.05, 15.0), env_step_cadence=cad)      # cad: n / 2 train steps per vector step, the whole vector step as one graph
            sg = g.rollout(chunk, **kw); st = t.rollout(chunk, **kw)
            t0 += chunk
            assert sg == st, (sg, st)
            compare_state(g, t)
            np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
            np.testing.assert_array_equal(g.get_params(1), t.get_params(1))
        idx = (np.arange(8, dtype=np.int64) * 7) % int(g.replay_size()[0])
        for x, y in zip(g.get_batch(idx), t.get_batch(idx)):
            np.testing.assert_array_equal(x, y)
        assert g.evaluate(8, 50, seed=5) == t.evaluate(8, 50, seed=5)
        assert g.evaluate(4, 30, seed=6) == t.evaluate(4, 30, seed=6)
        compare_state(g, t)
        outs.append((g.get_params(0), g.replay_priorities(), g.envs_peek()))
        g.close(); t.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0]); np.testing.assert_array_equal(outs[0][1], outs[1][1])
    for x, y in zip(outs[0][2], outs[1][2]):
        np.testing.assert_array_equal(x, y)


def test_rollout_env_step_cadence_bit_exact(pkg, envs):
    """r05 (VERDICT r04 missing #4): dqn_rollout with cadence_env_steps = 1 trains every train_freq ENV steps like the reference's loop (src/solver.jl:136-140) -- n / train_freq
    train steps per vector step, run back to back through the pipelined dqn_train_steps path -- and syncs the target net on env-step multiples: trajectories, replay,
    priorities and both parameter vectors equal the twin's restatement bit for bit, chunk by chunk, mixed with the vector-step cadence"""
    net = EC.testmdp_conv_dueling()
    g, t, hp = make_pair(pkg, net, B=8, cap=96)
    EC.same_params([g, t], net)
    spec = envs.TestMDP((14, 12), 4, 6, n=6, seed=3)
    for h in (g, t):
        h.envs_create(spec, max_episode_length=100, seed=17)
    t0 = 1; total = 0
    for chunk, cad in ((2, True), (5, True), (3, False), (7, True), (4, True)):
        kw = dict(t0=t0, train_freq=4, target_update_freq=25, eps=(1.0, 0.1, 20.0), env_step_cadence=cad)
        sg = g.rollout(chunk, **kw); st = t.rollout(chunk, **kw)
        t0 += chunk; total += sg["train_steps"]
        assert sg == st, (sg, st)
        compare_state(g, t)
        np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
        np.testing.assert_array_equal(g.get_params(1), t.get_params(1))
    assert total >= 6 * 18 // 4 - 2          # ~1.5 train steps per vector step in the env-step chunks


@pytest.mark.parametrize("kind", ["testmdp", "testmdp_u8", "gridworld"])
def test_evaluate_bit_exact_and_isolated(pkg, envs, kind):
    """dqn_evaluate (device basic_evaluation) == the twin's: average return (Float64) and steps identical; the training envs, the
    replay and the rollout that follows are not disturbed by an evaluation in between."""
    if kind == "gridworld":
        net, spec, n, mel = EC.gridworld_mlp_dueling(), envs.SimpleGridWorld(n=8), 8, 30
        g, t, hp = make_pair(pkg, net, B=8, cap=256)
    else:
        net, spec, n, mel = EC.testmdp_conv_dueling(), envs.TestMDP((14, 12), 4, 6, n=8, seed=3), 8, 100
        g, t, hp = make_pair(pkg, net, B=8, cap=256, obs_dtype=1 if kind.endswith("u8") else 0)
    EC.same_params([g, t], net)
    for h in (g, t):
        h.envs_create(spec, max_episode_length=mel, seed=2)
        h.rollout(6, t0=1, train_freq=2, target_update_freq=0, eps=(0.5, 0.5, 1.0))
    for n_eval, mx, seed in ((16, mel, 7), (16, mel, 8), (5, 3, 7), (40, mel, 1)):
        assert g.evaluate(n_eval, mx, seed) == t.evaluate(n_eval, mx, seed)
    compare_state(g, t)
    sg = g.rollout(9, t0=7, train_freq=2, target_update_freq=4, eps=(0.5, 0.5, 1.0))
    st = t.rollout(9, t0=7, train_freq=2, target_update_freq=4, eps=(0.5, 0.5, 1.0))
    assert sg == st
    compare_state(g, t)
    np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    assert g.evaluate(16, mel, 7) == t.evaluate(16, mel, 7)


def test_rollout_errors(pkg, envs):
    net = EC.gridworld_mlp_dueling()
    g, t, hp = make_pair(pkg, net, B=8, cap=64)
    with pytest.raises(pkg.DQNError, match="dqn_envs_create"):
        g.rollout(1)
    with pytest.raises(pkg.DQNError, match="TestMDP"):
        g.envs_create(envs.TestMDP((14, 12), 4, 6, n=2))
    with pytest.raises(pkg.DQNError, match="n_envs"):
        g.envs_create(envs.SimpleGridWorld(n=65))           # more envs than replay slots
    g.envs_create(envs.SimpleGridWorld(n=4))
    with pytest.raises(pkg.DQNError, match="t0"):
        g.rollout(1, t0=0)


def test_odd_shapes_recreate_and_zero_steps(pkg, envs):
    """Shapes off the vectorised paths (E = 25 and n = 3 are not multiples of 4 -> scalar observe kernel; u8 rows), re-creating the env set with
    another size on the same engine, a zero-step rollout and single-copy evaluation -- all still identical to the twin."""
    net = O.Network((1, 5, 5), [O.Dense(25, 16, O.ACT_RELU), O.Dense(16, 4, O.ACT_IDENTITY)])
    for u8 in (0, 1):
        g, t, hp = make_pair(pkg, net, B=4, cap=40, obs_dtype=u8, dueling=0)
        EC.same_params([g, t], net)
        for n in (3, 7):
            spec = envs.TestMDP((5, 5), 1, 6, n=n, seed=4)
            for h in (g, t):
                h.envs_create(spec, max_episode_length=4, seed=11 + n)
                assert h.rollout(0, t0=1)["train_steps"] == 0
            compare_state(g, t)
            sg = g.rollout(13, t0=1, train_freq=3, target_update_freq=6, eps=(0.9, 0.2, 10.0))
            st = t.rollout(13, t0=1, train_freq=3, target_update_freq=6, eps=(0.9, 0.2, 10.0))
            assert sg == st
            compare_state(g, t)
            np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
            assert g.evaluate(1, 3, seed=2) == t.evaluate(1, 3, seed=2)
