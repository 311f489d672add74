"""GPU: DRQN (EpisodeReplayBuffer + recurrent batch_train!, src/episode_replay.jl, src/solver.jl:239-287) through the C ABI:
bit-exact vs the CPU twin over several steps, fp32 round-off vs the fp64 oracle, recurrent policy forward vs the oracle."""
import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import ref
from drqn_common import check_against_oracle, draws, drqn_nets, feed, make_episodes, make_handle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package(); p.lib(); return p


def setup(pkg, name, mfma=1, graph=1):
    net, B, T, kw = drqn_nets()[name]
    rng = np.random.default_rng(5)
    cap = max(12, B + 4)
    gpu, hp, layers = make_handle(pkg.Engine, net, B, T, dict(kw, use_mfma=mfma, use_graph=graph), cap=cap)
    cpu = ref.Twin(layers, hp, plan=gpu.plan(), threads=4)
    eps = make_episodes(net, cap + 3, T, rng)
    feed(gpu, eps); feed(cpu, eps)
    ring = [None] * cap
    for i, ep in enumerate(eps):
        ring[i % cap] = ep
    p_on = (O.Network.flatten(O.init_params_recurrent(net, 3)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    p_tg = (O.Network.flatten(O.init_params_recurrent(net, 4)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    for h in (gpu, cpu):
        h.set_params(p_on, 0); h.set_params(p_tg, 1)
    np.testing.assert_array_equal(gpu.get_params(0), p_on)      # LSTM block layout round trip (Wi, Wh, b, h0, c0)
    return net, B, T, kw, rng, gpu, cpu, ring, (p_on, p_tg)


@pytest.mark.parametrize("mfma", [0, 1])
@pytest.mark.parametrize("name", list(drqn_nets()))
def test_drqn_bit_exact_vs_twin_and_oracle(pkg, name, mfma):
    net, B, T, kw, rng, gpu, cpu, ring, params = setup(pkg, name, mfma=mfma)
    assert gpu.episode_count() == cpu.episode_count()
    idx, start = draws(ring, B, np.random.default_rng(9))
    for a, b in zip(gpu.episode_get_batch(idx, start), cpu.episode_get_batch(idx, start)):
        np.testing.assert_array_equal(a, b)
    check_against_oracle(gpu, net, ring, B, T, kw, np.random.default_rng(11), params)     # also advances the engine by one step
    cpu.train_step_drqn(*draws(ring, B, np.random.default_rng(11)))                       # same draws -> same state
    for step in range(4):
        idx, start = draws(ring, B, rng)
        lg, gg = gpu.train_step_drqn(idx, start); lc, gc = cpu.train_step_drqn(idx, start)
        assert lg == lc and gg == gc, (step, lg, lc, gg, gc)
        if step == 1:
            gpu.sync_target(); cpu.sync_target()
    np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    mg, vg, bg = gpu.get_adam_state(); mc, vc, bc = cpu.get_adam_state()
    np.testing.assert_array_equal(mg, mc); np.testing.assert_array_equal(vg, vc); np.testing.assert_array_equal(bg, bc)
    gpu.close(); cpu.close()


def test_recurrent_policy_forward_carries_state(pkg):
    net, B, T, kw, rng, gpu, cpu, ring, (p_on, p_tg) = setup(pkg, "dense_lstm_dueling")
    xs = [rng.random((1,) + net.obs_shape).astype(np.float32) for _ in range(6)]
    qs64, _ = O._seq_forward(net, [p.astype(np.float64) for p in net.unflatten(p_on)], [x.astype(np.float64) for x in xs])
    gpu.reset_state()
    for t, x in enumerate(xs):                     # Flux Recur: the hidden state persists between calls (src/policy.jl:38-46)
        np.testing.assert_allclose(gpu.forward(x), qs64[t], atol=1e-5, rtol=1e-5)
    gpu.reset_state()                              # resetstate!(policy): back to state0
    np.testing.assert_allclose(gpu.forward(xs[0]), qs64[0], atol=1e-5, rtol=1e-5)
    assert gpu.greedy_action(xs[1])[0] == int(np.argmax(qs64[1][0]))
    # the sampled-draws path (engine draws episodes itself) runs and keeps training finite
    l, g = gpu.train_step_drqn()
    assert np.isfinite(l) and g >= 0
    with pytest.raises(pkg.DQNError, match="use dqn_train_step_drqn"):
        gpu.train_step()
    gpu.close(); cpu.close()


def test_lstm_without_recurrence_flag_is_rejected(pkg):
    net, B, T, kw = drqn_nets()["lstm_single_q"]
    hp = ref.hparams_for(net, batch_size=B, buffer_size=8, recurrence=0)
    with pytest.raises(pkg.DQNError, match="recurrent model but recurrence is set to false"):   # src/solver.jl:45-47
        pkg.Engine(ref.layers_from_network(net), hp)


@pytest.mark.parametrize("name", list(drqn_nets()))
def test_recurrent_policy_and_sampled_draws_bit_exact_vs_twin(pkg, name):
    """The Recur state carried by the policy path (action / actionvalues between resetstate! calls, src/policy.jl:32-46) and the engine's own
    episode draws (sample without replacement + random start, src/episode_replay.jl:75,81 -- SplitMix64 stream) against the twin's restatement."""
    net, B, T, kw, rng, gpu, cpu, ring, (p_on, p_tg) = setup(pkg, name)
    for n in (1, 3):
        xs = [rng.random((n,) + net.obs_shape).astype(np.float32) for _ in range(5)]
        gpu.reset_state(); cpu.reset_state()
        for x in xs:
            np.testing.assert_array_equal(gpu.forward(x), cpu.forward(x))
        np.testing.assert_array_equal(gpu.greedy_action(xs[0]), cpu.greedy_action(xs[0]))       # state advanced identically on both
        gpu.reset_state(); cpu.reset_state()
        np.testing.assert_array_equal(gpu.forward(xs[0], which=1), cpu.forward(xs[0], which=1))    # target weights, shared carried state
    for step in range(4):                                   # no draws given: both sides sample the same episodes and start offsets
        assert gpu.train_step_drqn() == cpu.train_step_drqn()
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    gpu.close(); cpu.close()


def test_drqn_checkpoint_resume_is_bit_exact(pkg, tmp_path):
    """SURVEY 8(f)-3 for BASELINE config 4: dqn_episode_export/import + counters (episodes, ring cursor, the host sampler's draw counter) +
    parameters + Adam state: a DRQN run continued in a NEW engine reproduces the uninterrupted run bit for bit, with the engine's own
    episode draws (train_step_drqn()) -- and the next episode lands in the same ring slot."""
    name = "cfg4_lstm_plain" if "cfg4_lstm_plain" in drqn_nets() else list(drqn_nets())[0]
    net, B, T, kw, rng, a, cpu, ring, params = setup(pkg, name)
    cpu.close()
    for _ in range(3):
        a.train_step_drqn()
    ck = a.checkpoint()
    np.savez(tmp_path / "ck.npz", **ck)
    want = [a.train_step_drqn() for _ in range(4)]
    cap = max(12, B + 4)
    b, hp, layers = make_handle(pkg.Engine, net, B, T, dict(kw, use_mfma=1, use_graph=1), cap=cap)
    b.restore(dict(np.load(tmp_path / "ck.npz")))
    assert b.get_counters() == {k: int(v) for k, v in zip(("size", "widx", "sample_ctr", "train_steps"), ck["counters"])}
    assert b.episode_count() == a.episode_count()
    got = [b.train_step_drqn() for _ in range(4)]
    assert got == want, (got, want)
    np.testing.assert_array_equal(a.get_params(0), b.get_params(0)); np.testing.assert_array_equal(a.get_params(1), b.get_params(1))
    ma, va, ba = a.get_adam_state(); mb, vb, bb = b.get_adam_state()
    np.testing.assert_array_equal(ma, mb); np.testing.assert_array_equal(va, vb); np.testing.assert_array_equal(ba, bb)
    # the ring cursor survives: one more episode lands in the same slot of both buffers
    ep = make_episodes(net, 1, T, np.random.default_rng(77))
    feed(a, ep); feed(b, ep)
    for x, y in zip(a.episode_export(), b.episode_export()):
        np.testing.assert_array_equal(x, y)
    with pytest.raises(pkg.DQNError, match="capacity"):
        z = np.zeros((cap + 1, T) + net.obs_shape, np.float32)
        b.episode_import(z, z, np.zeros((cap + 1, T), np.int32), np.zeros((cap + 1, T), np.float32), np.zeros((cap + 1, T), np.uint8), np.ones(cap + 1, np.int32))
    a.close(); b.close()


@pytest.mark.parametrize("name", ["cfg4_lstm_plain", "dense_lstm_dueling", "lstm16_dueling_b16"])
def test_hidden_state_save_restore_vs_oracle_and_twin(pkg, name):
    """dqn_get_hidden / dqn_set_hidden == hiddenstates / sethiddenstates! (src/helpers.jl:61-79): the policy's Recur state after k forwards equals the fp64
    oracle's to 1e-5 and the twin's bit for bit, survives a recurrent train step (src/solver.jl:137-139), and set -> forward reproduces."""
    from drqn_common import check_hidden_state_protocol
    net, B, T, kw, rng, gpu, cpu, ring, (p_on, p_tg) = setup(pkg, name)
    check_hidden_state_protocol(gpu, net, p_on, np.random.default_rng(23), twin=cpu)
    gpu.close(); cpu.close()


def test_recurrence_with_u8_replay_is_rejected(pkg):
    net, B, T, kw = drqn_nets()["lstm_single_q"]
    hp = ref.hparams_for(net, batch_size=B, buffer_size=8, recurrence=1, trace_length=T, obs_dtype=pkg.OBS_U8)
    with pytest.raises(pkg.DQNError, match="u8 is not supported with recurrence"):
        pkg.Engine(ref.layers_from_network(net), hp, device=0)


@pytest.mark.parametrize("mfma", [0, 1])
@pytest.mark.parametrize("name", ["drqn_cfg4_lstm_plain", "drqn_dense_lstm_dueling", "drqn_lstm_single_q"])
def test_drqn_golden_fixture_torch_values(pkg, name, mfma, golden_dir):
    """the three committed DRQN fixtures (torch float64 autograd of src/solver.jl:239-287): loss, gradients, Adam step DIRECTLY, and the twin bit for bit"""
    from golden_common import load, run_drqn_fixture
    g = load(golden_dir, name)
    outs = []
    class Eng(pkg.Engine):
        def __init__(self, layers, hp, **kw):
            hp.use_mfma = mfma
            super().__init__(layers, hp, device=0, **kw)
            outs.append(self.plan())
    a = run_drqn_fixture(Eng, name, g)
    b = run_drqn_fixture(ref.Twin, name, g, plan=outs[0], threads=4)
    assert a["loss"] == b["loss"] and a["gn"] == b["gn"]
    np.testing.assert_array_equal(a["grads"], b["grads"]); np.testing.assert_array_equal(a["newp"], b["newp"])


@pytest.mark.parametrize("name", ["cfg4_lstm_plain", "lstm16_dueling_b16", "lstm_single_q"])
def test_drqn_multilaunch_program_with_contiguous_dw_plan(pkg, name):
    """the networks the fused column-parallel step covers still run the multi-launch recurrent program when the plan asks for contiguous dW chunks
    (dw_kc >= 0): both schedules stay under test, each bit-exact against the twin handed the same plan -- and they differ from each other only in rounding"""
    net, B, T, kw = drqn_nets()[name]
    rng = np.random.default_rng(5)
    cap = max(12, B + 4)
    hp = ref.hparams_for(net, batch_size=B, buffer_size=cap, recurrence=1, trace_length=T, learning_rate=1e-3, prioritized_replay=0, **kw)
    layers = ref.layers_from_network(net)
    dplan = pkg.default_plan(layers, hp)
    assert all(p[2] < 0 for p in dplan)                                  # covered: column-group chunks by default
    plan0 = [(p[0], p[1], 0) for p in dplan]
    eps = make_episodes(net, cap + 3, T, rng)
    p_on = (O.Network.flatten(O.init_params_recurrent(net, 3)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    outs = []
    for plan in (plan0, dplan):
        gpu = pkg.Engine(layers, hp, plan=plan, device=0); cpu = ref.Twin(layers, hp, plan=plan, threads=4)
        assert gpu.plan() == plan
        for h in (gpu, cpu):
            feed(h, eps); h.set_params(p_on, 0); h.set_params(p_on * np.float32(0.9), 1)
        r = np.random.default_rng(3)
        ring = [None] * cap
        for i, ep in enumerate(eps):
            ring[i % cap] = ep
        for step in range(3):
            idx, start = draws(ring, B, r)
            assert gpu.train_step_drqn(idx, start) == cpu.train_step_drqn(idx, start)
        np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
        np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
        outs.append(gpu.get_params(0)); gpu.close(); cpu.close()
    np.testing.assert_allclose(outs[0], outs[1], atol=3e-3, rtol=0)     # same math, different summation order (Adam at |g| ~ eps moves a parameter by up to lr per step)


def test_drqn_column_group_plan_on_uncovered_network_is_refused(pkg):
    net, B, T, kw = drqn_nets()["dense_lstm_dueling"]
    hp = ref.hparams_for(net, batch_size=B, buffer_size=12, recurrence=1, trace_length=T, prioritized_replay=0, **kw)
    layers = ref.layers_from_network(net)
    plan = [(p[0], p[1], -2) for p in pkg.default_plan(layers, hp)]
    gpu = pkg.Engine(layers, hp, plan=plan, device=0)
    feed(gpu, make_episodes(net, 12, T, np.random.default_rng(0)))
    with pytest.raises(pkg.DQNError, match="column-group dW chunks"):
        gpu.train_step_drqn()
    gpu.close()


def test_drqn_fused_step_long_run_wraps_the_draw_ring(pkg):
    """the fused recurrent step takes its episode draws from DQN_DRAW_SLOTS = 32 mapped host slots; a slot is a LAUNCH PARAMETER fixed per graph node (two alternating 8-step
    graphs own slots 0-7 / 8-15, two alternating single-step graphs 16 / 17) and the host waits for the event behind a graph's previous launch before rewriting its slots:
    300 steps -- single calls with the engine's own sampler, explicit draws, and dqn_train_steps runs of 8-step graphs, in every alignment -- reuse every slot many times and
    must leave the twin's parameters bit for bit"""
    net, B, T, kw, rng, gpu, cpu, ring, params = setup(pkg, "cfg4_lstm_plain")
    assert all(p[2] < 0 for p in gpu.plan())                        # the fused path
    done = 0
    for n in (1, 7, 8, 9, 30, 64, 3, 65, 16, 40):
        lg = gpu.train_steps(n)
        for _ in range(n):
            lc = cpu.train_step_drqn()
        assert lg == lc, (done, n, lg, lc)
        done += n
        idx, start = draws(ring, B, rng)                              # an explicit-draw step in between shifts the alignment of the next run
        assert gpu.train_step_drqn(idx, start) == cpu.train_step_drqn(idx, start)
        done += 1
    assert done > 3 * 64
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    mg, vg, bg = gpu.get_adam_state(); mc, vc, bc = cpu.get_adam_state()
    np.testing.assert_array_equal(mg, mc); np.testing.assert_array_equal(vg, vc); np.testing.assert_array_equal(bg, bc)
    gpu.close(); cpu.close()


@pytest.mark.parametrize("name", ["cfg4_lstm_plain", "lstm16_dueling_b16"])
def test_drqn_default_plan_with_communicator(pkg, monkeypatch, name):
    """a recurrent engine created with the DEFAULT plan (column-group dW chunks: the fused single-device step) and then given a communicator must keep
    training: dqn_comm_init re-derives the default plan without the column-group rule, the step runs the multi-launch program with the all-reduce between backward and
    Adam (a real RCCL communicator at world 1, forced on), bit-exact against the twin handed the engine's plan; a caller-written column-group plan is refused AT
    dqn_comm_init with the reason"""
    net, B, T, kw = drqn_nets()[name]
    rng = np.random.default_rng(5); cap = max(12, B + 4)
    monkeypatch.setenv("DQN_FORCE_ALLREDUCE", "1")
    gpu, hp, layers = make_handle(pkg.Engine, net, B, T, kw, cap=cap)
    assert all(p[2] < 0 for p in gpu.plan())
    gpu.comm_init(pkg.comm_unique_id(), 0, 1)
    plan = gpu.plan()
    assert all(p[2] >= 0 for p in plan), plan
    cpu = ref.Twin(layers, hp, plan=plan, threads=4)
    eps = make_episodes(net, cap + 3, T, rng)
    ring = [None] * cap
    for i, ep in enumerate(eps):
        ring[i % cap] = ep
    p_on = (O.Network.flatten(O.init_params_recurrent(net, 3)) + 0.05 * rng.standard_normal(net.n_params())).astype(np.float32)
    for h in (gpu, cpu):
        feed(h, eps); h.set_params(p_on, 0); h.set_params(p_on * np.float32(0.9), 1)
    for step in range(3):
        idx, start = draws(ring, B, rng)
        assert gpu.train_step_drqn(idx, start) == cpu.train_step_drqn(idx, start)
    assert gpu.train_steps(3) == [cpu.train_step_drqn() for _ in range(3)][-1]
    info = gpu.comm_info()
    assert info["rccl_nranks"] == 1 and info["exchange"] == 2      # all-reduce of the flat gradient
    np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    gpu.close(); cpu.close()
    g2 = pkg.Engine(layers, hp, plan=pkg.default_plan(layers, hp), device=0)      # the same plan, but WRITTEN by the caller
    with pytest.raises(pkg.DQNError, match="column-group dW plan"):
        g2.comm_init(pkg.comm_unique_id(), 0, 1)
    g2.close()
