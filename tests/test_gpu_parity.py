"""GPU (-m gpu): the HIP engine, called through the C ABI, against
  * the canonical-order CPU twin  -> BIT-EXACT (TD loss, td, greedy/best actions, indices, IS weights, parameters)
  * the NumPy fp64 oracle         -> Q-values within 1e-5, gradients/loss to fp32 round-off
  * the torch-autograd-pinned golden fixtures (tests/golden).
Tolerances are the ones BASELINE.json's north_star states: TD loss and greedy action indices bit-exact,
Q-values within 1e-5 fp32."""
import os

import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import ref
from nets import GOLDEN_CASES, cfg1_mlp_dueling, mid_conv_dueling, mid_conv_plain, nature_dueling, small_conv_dueling, small_conv_plain
from nets import testmdp_mlp_tanh as mlp_tanh_net
from parity_common import hand_derived_known_answer, sampler_distinct, sampler_distribution
from test_twin_vs_oracle import check_priorities_after_step, run_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.lib()  # fail loudly if the HIP library is missing
    return p


def make_pair(pkg, net, B, cap=128, mfma=1, graph=1, tiny=True, **kw):
    """tiny=False: networks that fit in LDS take the multi-launch program instead of the single-launch step (tiny_step.hip; DQN_NO_TINY is read at
    dqn_engine_create) -- both schedules of the same arithmetic stay under test"""
    tiny = bool(kw.pop("_tiny", tiny))
    hp = ref.hparams_for(net, batch_size=B, buffer_size=cap, use_mfma=mfma, use_graph=graph, **kw)
    layers = ref.layers_from_network(net)
    plan = pkg.default_plan(layers, hp)
    if not tiny:
        os.environ["DQN_NO_TINY"] = "1"
    try:
        eng = pkg.Engine(layers, hp, plan=plan)
    finally:
        os.environ.pop("DQN_NO_TINY", None)
    return eng, ref.Twin(layers, hp, plan=plan, threads=8), hp


def fill(handles, net, n, seed=0, u8=False):
    rng = np.random.default_rng(seed)
    if u8:
        s = rng.integers(0, 256, (n,) + net.obs_shape).astype(np.uint8)
        sp = rng.integers(0, 256, (n,) + net.obs_shape).astype(np.uint8)
    else:
        s = rng.random((n,) + net.obs_shape, dtype=np.float32)
        sp = rng.random((n,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, n).astype(np.int32)
    r = (rng.standard_normal(n) * 2).astype(np.float32)
    d = (rng.random(n) < 0.2).astype(np.uint8)
    for h in handles:
        h.replay_add(s, a, r, sp, d)
    return s, a, r, sp, d


def set_same_params(handles, net, seed=1):
    p_on = O.Network.flatten(O.init_params(net, seed=seed))
    p_tg = O.Network.flatten(O.init_params(net, seed=seed + 100))
    rng = np.random.default_rng(seed)
    p_on = (p_on + 0.01 * rng.standard_normal(p_on.shape)).astype(np.float32)  # non-zero biases
    for h in handles:
        h.set_params(p_on, 0)
        h.set_params(p_tg, 1)
    return p_on, p_tg


def assert_step_bit_exact(gpu, cpu, idx=None):
    lg, gg, tg = gpu.train_step(idx)
    lc, gc, tc = cpu.train_step(idx)
    np.testing.assert_array_equal(gpu.last_indices(), cpu.last_indices())
    qg, qc = gpu.last_q(), cpu.last_q()
    for k in ("q_on_s", "q_on_sp", "q_tg_sp", "best_a", "y"):
        np.testing.assert_array_equal(qg[k], qc[k], err_msg=k)
    np.testing.assert_array_equal(tg, tc)
    assert lg == lc, (lg, lc)
    assert gg == gc, (gg, gc)
    return lg, gg


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_engine_matches_fp64_oracle_and_golden(pkg, name, golden_dir):
    run_case(name, golden_dir, pkg.Engine)


@pytest.mark.parametrize("name", ["cfg1_gridworld_mlp_dueling", "small_conv_dueling"])
def test_prioritized_replay_off_leaves_priorities(pkg, name, golden_dir):
    run_case(name, golden_dir, pkg.Engine, prioritized=0)      # src/solver.jl:231: no update_priorities! call


def test_hand_derived_known_answer(pkg):
    """every number of one batch_train! worked out by hand from the cited reference lines (tests/parity_common.py)"""
    hand_derived_known_answer(pkg.Engine)


def test_sampler_inclusion_frequencies(pkg):
    """sum-tree sampler vs p / sum(p) (...replay.jl:85); the twin runs the same check on the CPU and the two samplers are bit-identical"""
    sampler_distribution(pkg.Engine)


@pytest.mark.parametrize("mfma", [0, 1])
@pytest.mark.parametrize("netf,B,kw", [
    (cfg1_mlp_dueling, 32, dict(gamma=0.95)),
    (mlp_tanh_net, 32, dict(gamma=0.99, double_q=0)),
    (cfg1_mlp_dueling, 6, dict(gamma=0.9)),                               # B % 4 != 0: the scalar item loops of the single-launch step
    (small_conv_dueling, 16, dict(gamma=0.99)),
    (small_conv_plain, 8, dict(gamma=0.9, double_q=0, prioritized_replay=0)),
    (small_conv_dueling, 5, dict(gamma=0.99, adam_f64_scalars=0)),
])
def test_multi_step_bit_exact_vs_twin(pkg, netf, B, kw, mfma):
    net = netf()
    gpu, cpu, hp = make_pair(pkg, net, B, cap=100, mfma=mfma, tiny=bool(mfma), learning_rate=1e-3, **kw)      # the MLPs: single-launch step (mfma=1) and multi-launch VALU program (mfma=0)
    fill((gpu, cpu), net, 137)  # > cap: exercises the ring wrap (mod1, ...replay.jl:70)
    set_same_params((gpu, cpu), net)
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    for step in range(6):
        assert_step_bit_exact(gpu, cpu)
        if step == 2:
            gpu.sync_target(); cpu.sync_target()
    np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.get_params(1), cpu.get_params(1))
    mg, vg, bg = gpu.get_adam_state(); mc, vc, bc = cpu.get_adam_state()
    np.testing.assert_array_equal(mg, mc); np.testing.assert_array_equal(vg, vc); np.testing.assert_array_equal(bg, bc)
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())



@pytest.mark.parametrize("graph", [1, 0])
@pytest.mark.parametrize("netf,B,kw", [
    (nature_dueling, 32, dict()),
    (nature_dueling, 32, dict(obs_dtype=1)),                              # u8 replay on the byte arena
    (small_conv_dueling, 16, dict()),
    (cfg1_mlp_dueling, 32, dict(gamma=0.95)),                             # fits in LDS: the single-launch step (tiny_step.hip)
    (cfg1_mlp_dueling, 32, dict(gamma=0.95, _tiny=0)),                    # ... and its multi-launch program
    (small_conv_plain, 8, dict(double_q=0, prioritized_replay=0)),       # not eligible for the pre-gather: must simply still be right
    (small_conv_dueling, 5, dict(adam_f64_scalars=0)),                    # ragged: 2B = 10 columns, E = 504 features (partial gather tiles)
    (mlp_tanh_net, 24, dict(double_q=0, _tiny=0)),                        # plain network, VALU-only launches
    (mlp_tanh_net, 24, dict(double_q=0)),                                 # ... and as ONE launch
    (small_conv_dueling, 16, dict(obs_dtype=1)),                          # u8 rows into the FLOAT arena: no pre-gather, still right
])
def test_train_steps_pipelined_gather_bit_exact(pkg, netf, B, kw, graph):
    """dqn_train_steps(n): step i's Adam launch gathers step i+1's batch (common.h PreGather) and step i+1 runs without a gather launch.  Another
    schedule of the same arithmetic: after train_steps(n) every piece of state equals the twin stepped n times one call at a time -- and single
    steps, replay writes and explicit-index steps in between must find nothing stale."""
    net = netf()
    kw = dict(kw)
    gpu, cpu, hp = make_pair(pkg, net, B, cap=128, graph=graph, learning_rate=1e-3, **kw)
    u8 = kw.get("obs_dtype", 0) == 1
    fill((gpu, cpu), net, 100, seed=7, u8=u8)
    set_same_params((gpu, cpu), net, seed=5)

    def same_state():
        np.testing.assert_array_equal(gpu.last_indices(), cpu.last_indices())
        np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
        np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
        mg, vg, bg = gpu.get_adam_state(); mc, vc, bc = cpu.get_adam_state()
        np.testing.assert_array_equal(mg, mc); np.testing.assert_array_equal(vg, vc); np.testing.assert_array_equal(bg, bc)

    lg = gpu.train_steps(5)
    for _ in range(5):
        lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    same_state()
    assert_step_bit_exact(gpu, cpu)                       # a plain step after the pipelined call
    fill((gpu, cpu), net, 9, seed=8, u8=u8)               # the replay changes: whatever was drawn ahead is stale
    lg = gpu.train_steps(3)
    for _ in range(3):
        lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    same_state()
    idx = np.random.default_rng(1).integers(0, 100, B)
    assert_step_bit_exact(gpu, cpu, idx)                  # an explicit-index step overwrites the arena
    lg = gpu.train_steps(2)
    for _ in range(2):
        lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    same_state()
    lg = gpu.train_steps(1); lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    same_state()
    # the steady-state profile runs two real steps
    names = [n for n, _ in gpu.profile_step(steady=True)]
    cpu.train_step(); cpu.train_step()
    same_state()
    if names == ["tiny_step"]:
        assert kw.get("_tiny", 1) and netf in (cfg1_mlp_dueling, mlp_tanh_net), names      # the whole step is one launch that samples and gathers itself
    elif kw.get("prioritized_replay", 1) and (kw.get("obs_dtype", 0) == 0 or gpu.batch_arena_elem_bytes() == 1):
        assert "adam+gather" in names and not any(n in ("gather", "sample_gather") for n in names), names      # eligible: no gather launch of its own
    else:
        assert "sample_gather" in names and "adam+gather" not in names, names
    assert_step_bit_exact(gpu, cpu)


@pytest.mark.parametrize("netf,B", [(small_conv_dueling, 16), (cfg1_mlp_dueling, 32)])
def test_train_steps_long_run_bit_exact(pkg, netf, B):
    """300 sampled steps issued as dqn_train_steps calls of assorted lengths (1, 2, 3, 7, 50, ...) with replay writes between some of them: the
    pipelined gather must never hand a stale or half-written batch to a step -- parameters, priorities and Adam state equal the twin's after every call."""
    net = netf()
    gpu, cpu, _ = make_pair(pkg, net, B, cap=256, learning_rate=1e-3, gamma=0.95)
    fill((gpu, cpu), net, 200, seed=51)
    set_same_params((gpu, cpu), net, seed=52)
    rng = np.random.default_rng(5)
    done = 0
    for call, n in enumerate([1, 2, 3, 7, 50, 1, 13, 64, 2, 31, 100, 5, 21]):
        lg = gpu.train_steps(n)
        for _ in range(n):
            lc = cpu.train_step()
        done += n
        assert lg[0] == lc[0] and lg[1] == lc[1], (call, n)
        np.testing.assert_array_equal(gpu.last_indices(), cpu.last_indices())
        np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
        np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
        if call % 3 == 2:
            fill((gpu, cpu), net, int(rng.integers(1, 9)), seed=100 + call)
        if call % 4 == 1:
            gpu.sync_target(); cpu.sync_target()
    assert done == 300
    mg, vg, bg = gpu.get_adam_state(); mc, vc, bc = cpu.get_adam_state()
    np.testing.assert_array_equal(mg, mc); np.testing.assert_array_equal(vg, vc); np.testing.assert_array_equal(bg, bc)


def test_graph_and_eager_agree(pkg):
    net = small_conv_dueling()
    a, cpu, _ = make_pair(pkg, net, 16, graph=1)
    b, _, _ = make_pair(pkg, net, 16, graph=0)
    fill((a, b), net, 64); set_same_params((a, b), net)
    for _ in range(4):
        ra, rb = a.train_step(), b.train_step()
        assert ra[0] == rb[0] and ra[1] == rb[1]
        np.testing.assert_array_equal(ra[2], rb[2])
    np.testing.assert_array_equal(a.get_params(0), b.get_params(0))


def test_replay_seams(pkg):
    net = small_conv_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 16, cap=50)
    s, a, r, sp, d = fill((gpu, cpu), net, 70, seed=5)
    assert gpu.replay_size() == (50, 50) == cpu.replay_size()
    # ring: slot j holds transition 50+j for j<20, else j
    idx = np.array([0, 19, 20, 49, 7, 7, 33, 1, 2, 3, 4, 5, 6, 8, 9, 10], np.int64)
    bg, bc = gpu.get_batch(idx), cpu.get_batch(idx)
    for x, y in zip(bg, bc):
        np.testing.assert_array_equal(x, y)
    src = np.where(idx < 20, idx + 50, idx)
    np.testing.assert_array_equal(bg[0], s[src]); np.testing.assert_array_equal(bg[3], sp[src])
    np.testing.assert_array_equal(bg[1], a[src]); np.testing.assert_array_equal(bg[2], r[src])
    # priorities (|r|+eps)^alpha and IS weights (n*p/sum)^-beta vs the fp64 formula
    pr = gpu.replay_priorities()
    np.testing.assert_allclose(pr, (np.abs(r[np.r_[50:70, 20:50]].astype(np.float64)) + 1e-3) ** 0.6, rtol=2e-7)
    np.testing.assert_allclose(bg[5], O.is_weights(pr[idx], pr, 0.4, np.float64), rtol=2e-6)
    # sampler: same Philox stream + same tree => same indices as the twin, call after call
    for _ in range(5):
        np.testing.assert_array_equal(gpu.replay_sample(), cpu.replay_sample())
    # update_priorities! with a duplicate: last write wins (...replay.jl:79)
    td = np.linspace(-2, 2, 16).astype(np.float32)
    gpu.update_priorities(idx, td); cpu.update_priorities(idx, td)
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    assert gpu.replay_priorities()[7] == np.float32((np.float64(abs(td[5]) + np.float32(1e-3))) ** np.float64(np.float32(0.6)))
    with pytest.raises(pkg.DQNError):
        gpu.get_batch(np.full(16, 50, np.int64))  # BoundsError
    with pytest.raises(pkg.DQNError):
        gpu.replay_add(s[:1], [99], r[:1], sp[:1], d[:1])  # bad action index


def test_u8_replay_bit_exact(pkg):
    net = small_conv_dueling()
    gpu, cpu, _ = make_pair(pkg, net, 16, cap=64, obs_dtype=1)
    s, *_ = fill((gpu, cpu), net, 64, u8=True)
    set_same_params((gpu, cpu), net)
    idx = np.arange(16, dtype=np.int64)
    np.testing.assert_array_equal(gpu.get_batch(idx)[0], s[:16].astype(np.float32) / np.float32(255))
    for _ in range(3):
        assert_step_bit_exact(gpu, cpu)


@pytest.mark.parametrize("u8", [False, True])
def test_large_batch_paths_bit_exact(pkg, u8):
    """B > 64 switches the step program to its large-batch shape (config 5 runs B = 512): the sampler is a launch of its own instead of
    being repeated inside every gather workgroup, the head split-K slabs go through the multi-workgroup reduce instead of the TD kernel,
    u8 rows use the 256-byte-segment gather, and the priority block of the Adam launch owns several tree nodes per thread."""
    net = small_conv_dueling()
    gpu, cpu, _ = make_pair(pkg, net, 320, cap=2048, obs_dtype=1 if u8 else 0, learning_rate=1e-3, gamma=0.99)
    fill((gpu, cpu), net, 1500, u8=u8)
    set_same_params((gpu, cpu), net)
    for _ in range(4):
        assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())


@pytest.mark.parametrize("netf,B", [(mid_conv_plain, 128), (mid_conv_plain, 256), (mid_conv_dueling, 128), (mid_conv_plain, 96)])
def test_wide_sample_dx_tiles_bit_exact(pkg, netf, B):
    """B % 128 == 0 switches the LDS-tiled dX kernels to 32-feature x 128-sample workgroup tiles (dx_lds_body_wide: conv and dense, one or two
    sources); B = 96 keeps the 32 x 32 tiles on the same network.  Another tiling of the same per-element chains: bit-identical to the twin."""
    net = netf()
    gpu, cpu, _ = make_pair(pkg, net, B, cap=1024, learning_rate=1e-3, gamma=0.99)
    fill((gpu, cpu), net, 700, seed=11)
    set_same_params((gpu, cpu), net, seed=12)
    for _ in range(3):
        assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())


@pytest.mark.parametrize("netf,B,cap,kw", [
    (cfg1_mlp_dueling, 8, 40, dict()),                    # sum-tree of 64 leaves: the priority block's general (two-phase) form inside the 1024-thread workgroup
    (cfg1_mlp_dueling, 4, 20, dict(double_q=0)),          # 32 leaves, no online Q(sp) columns
    (mlp_tanh_net, 12, 50, dict(obs_dtype=1)),            # u8 rows converted in the kernel's own gather
    (cfg1_mlp_dueling, 64, 5000, dict()),                 # the largest batch the single-launch step takes
])
def test_single_launch_step_edge_cases(pkg, netf, B, cap, kw):
    """tiny_step.hip at the corners of its eligibility: tiny replay capacities, batches that are not multiples of 4 x 8, byte observations, B = 64; single
    steps and dqn_train_steps(n) against the twin, priorities included."""
    net = netf()
    gpu, cpu, hp = make_pair(pkg, net, B, cap=cap, learning_rate=1e-3, **kw)
    fill((gpu, cpu), net, cap + 7, seed=3, u8=kw.get("obs_dtype", 0) == 1)
    set_same_params((gpu, cpu), net, seed=4)
    for _ in range(4):
        assert_step_bit_exact(gpu, cpu)
    lg = gpu.train_steps(6)
    for _ in range(6):
        lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    assert [n for n, _ in gpu.profile_step()] == ["tiny_step"]


@pytest.mark.parametrize("knob,val,B", [("DQN_FWD_M32", "1", 128), ("DQN_FWD_M32", "0", 384)])
def test_forward_32x32_mfma_blocks_bit_exact(pkg, monkeypatch, knob, val, B):
    """DQN_FWD_M32 (read at dqn_engine_create): the forward launches with 64-channel tiles use 2 x 2 blocks of v_mfma_f32_32x32x2_f32 per workgroup instead of
    four 16x16x4 accumulators per wave -- by default in the large launches (>= 1024 workgroups, r04: B = 384 here and in the config-5 tests), with =1 in every
    launch (B = 128), with =0 in none (the 16x16x4 form of the large launches stays under test).  The 32x32x2 instruction accumulates its two k steps in order like
    the 16x16x4 one accumulates its four: the same k-ascending chain, bit-identical to the twin (DESIGN.md sections 6.6, 6.10)."""
    monkeypatch.setenv(knob, val)
    net = nature_dueling()
    gpu, cpu, _ = make_pair(pkg, net, B, cap=512, learning_rate=1e-3, gamma=0.99)
    monkeypatch.delenv(knob)
    fill((gpu, cpu), net, 400, seed=21)
    set_same_params((gpu, cpu), net, seed=22)
    for _ in range(2):
        assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    names = [n for n, _ in gpu.profile_step()]
    assert any(n.startswith("fwd_conv") for n in names), names


def test_policy_forward_and_greedy(pkg):
    net = small_conv_dueling()
    gpu, cpu, _ = make_pair(pkg, net, 16)
    p_on, _ = set_same_params((gpu, cpu), net)
    rng = np.random.default_rng(3)
    for n in (1, 7, 64):
        obs = rng.random((n,) + net.obs_shape, dtype=np.float32)
        qg, qc = gpu.forward(obs), cpu.forward(obs)
        np.testing.assert_array_equal(qg, qc)
        q64 = O.network_forward(net, [p.astype(np.float64) for p in net.unflatten(p_on)], obs.astype(np.float64))
        np.testing.assert_allclose(qg, q64, atol=1e-5, rtol=1e-5)
        np.testing.assert_array_equal(gpu.greedy_action(obs), np.argmax(qc, axis=1))
    # dueling identity: mean_a(Q - V) == 0 per column (src/dueling.jl:10)
    assert qg.shape == (64, net.n_actions)


def test_nature_dqn_b32_full_size_bit_exact(pkg):
    """BASELINE config 2 at full size: 84x84x4, Nature-DQN dueling, B=32, double-Q, prioritized: bit-exact vs the canonical-order twin,
    then one more step on explicit indices against the NumPy fp64 oracle (north_star: Q-values within 1e-5, greedy indices equal)."""
    net = nature_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 32, cap=256, gamma=0.99)
    fill((gpu, cpu), net, 256, seed=9)
    p_on, p_tg = set_same_params((gpu, cpu), net, seed=1)
    for step in range(3):
        loss, gn = assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    # ---- fp64 oracle at FULL size on the engine's own batch and current parameters
    p_on, p_tg = gpu.get_params(0), gpu.get_params(1)
    idx = np.random.default_rng(123).choice(256, 32, replace=False).astype(np.int64)
    batch = gpu.get_batch(idx)
    o = O.batch_train_step(net, net.unflatten(p_on), net.unflatten(p_tg), batch, gamma=float(np.float32(0.99)), double_q=True, adam=None)
    pr_before = gpu.replay_priorities()
    np.testing.assert_allclose(batch[5], O.is_weights(pr_before[idx], pr_before, hp.prio_beta, np.float64), rtol=2e-6)
    loss, gn, td = gpu.train_step(idx)
    q = gpu.last_q()
    np.testing.assert_allclose(q["q_on_s"], o["q"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(q["q_on_sp"], o["q_on_sp"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(q["q_tg_sp"], o["q_tg_sp"], atol=1e-5, rtol=1e-5)
    top2 = np.sort(o["q_on_sp"], axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 1e-5                 # a greedy index can only differ where fp64 itself sees a near-tie
    assert clear.sum() >= 24
    np.testing.assert_array_equal(q["best_a"][clear], o["best_a"][clear])
    np.testing.assert_allclose(q["y"], o["y"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(td, o["td"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(loss, o["loss"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(gn, o["grad_norm"], rtol=1e-4)
    go = O.Network.flatten(o["grads"])
    np.testing.assert_allclose(gpu.get_grads(), go, atol=2e-5 * np.abs(go).max(), rtol=1e-4)
    check_priorities_after_step(gpu, hp, idx, pr_before, td, o["td"], batch[5])


def _wide_fc_plain():
    """a plain (non-dueling) Q network whose last hidden layer is wide enough for a split-K forward (K > 1024): one stream for the fused reduce + head launch"""
    return O.Network((4, 20, 20), [O.Conv(4, 4, 32, O.ACT_RELU, 2), O.Conv(3, 32, 32, O.ACT_RELU, 1), O.Dense(32 * 7 * 7, 128, O.ACT_RELU), O.Dense(128, 3, O.ACT_IDENTITY)])


def _wide_fc_dueling_tanh():
    b, v, a = O.create_dueling_network([O.Conv(4, 4, 32, O.ACT_RELU, 2), O.Conv(3, 32, 32, O.ACT_RELU, 1), O.Dense(32 * 7 * 7, 160, O.ACT_TANH), O.Dense(160, 5, O.ACT_IDENTITY)])
    return O.Network((4, 20, 20), b, v, a)


@pytest.mark.parametrize("netf,B,kw", [(nature_dueling, 32, dict(gamma=0.99)), (_wide_fc_plain, 16, dict(gamma=0.9)), (_wide_fc_plain, 8, dict(gamma=0.9, double_q=0)),
                                       (_wide_fc_dueling_tanh, 24, dict(gamma=0.95)), (_wide_fc_dueling_tanh, 4, dict(gamma=0.95, double_q=0, prioritized_replay=0)), (_wide_fc_plain, 96, dict(gamma=0.9))])
def test_fused_reduce_head_launch_both_schedules(pkg, monkeypatch, netf, B, kw):
    """r05: where the head layers sit on split-K dense hidden layers, k_reduce_multi + k_head_td are ONE launch (red_head.hip: workgroup = 4 batch columns x stream x chunk
    of 32 hidden rows, write-through hand-off to the column group's last arriver).  Both schedules (DQN_NO_RED_HEAD, read at dqn_engine_create) run the same chains: each
    bit-exact against the twin -- single steps, explicit indices, the pipelined dqn_train_steps -- and against each other."""
    net = netf()
    outs = []
    # r06: the fused launch's slabs are piece-major ([S][column quad][N][4], written so by the forward launch) and its last arriver reads a piece-major copy of the hidden
    # activations; DQN_NO_RH_PM keeps the [S][N][columns] form -- a third schedule of the same chains
    for fused, env in ((True, None), (True, "DQN_NO_RH_PM"), (False, "DQN_NO_RED_HEAD")):
        if env:
            monkeypatch.setenv(env, "1")
        gpu, cpu, hp = make_pair(pkg, net, B, cap=max(128, B + 40), **kw)
        if env:
            monkeypatch.delenv(env, raising=False)
        fill((gpu, cpu), net, max(100, B + 20), seed=5)
        set_same_params((gpu, cpu), net, seed=3)
        for _ in range(2):
            assert_step_bit_exact(gpu, cpu)
        assert_step_bit_exact(gpu, cpu, np.random.default_rng(1).integers(0, 100, B))      # duplicates allowed
        lg = gpu.train_steps(5)
        for _ in range(5):
            lc = cpu.train_step(want_td=False)
        assert lg == lc
        np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
        np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
        np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
        names = [n for n, _ in gpu.profile_step()]
        assert ("red_head" in names) == fused and ("head_td" in names) == (not fused), names
        if fused:
            assert not any("fwd_reduce" in n for n in names), names
        outs.append((gpu.get_params(0), gpu.get_adam_state()[0]))
        gpu.close(); cpu.close()
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][0], o[0]); np.testing.assert_array_equal(outs[0][1], o[1])


@pytest.mark.parametrize("netf,B,kw", [(nature_dueling, 128, dict(gamma=0.99)), (nature_dueling, 132, dict(gamma=0.99, double_q=0)), (_wide_fc_dueling_tanh, 160, dict(gamma=0.95)),
                                       (_wide_fc_plain, 256, dict(gamma=0.9, prioritized_replay=0))])
def test_head_level_four_columns_per_workgroup_both_schedules(pkg, monkeypatch, netf, B, kw):
    """r05: at batches whose hidden-layer forwards are NOT split-K the head level is k_head_cols4 (red_head.hip): one workgroup per group of four batch columns pulls its 12
    columns of the hidden layers as 16-byte pieces, runs the chunk chains, the TD arithmetic and the heads' dX (16-byte stores) -- instead of k_head_td's one workgroup per
    column reading a transposed copy.  Same chains in the same order: both schedules (DQN_NO_HEAD_COLS4=1 keeps k_head_td) bit-exact against the twin and each other."""
    net = netf()
    outs = []
    for fused in (True, 2, False):      # (2: k_head_cols4 reading the [K][columns] activations themselves -- its loader when no transposed copy exists)
        if fused is not True:
            monkeypatch.setenv("DQN_NO_HEAD_COLS4", "2" if fused else "1")
        gpu, cpu, hp = make_pair(pkg, net, B, cap=B + 72, **kw)
        monkeypatch.delenv("DQN_NO_HEAD_COLS4", raising=False)
        fill((gpu, cpu), net, B + 40, seed=5)
        set_same_params((gpu, cpu), net, seed=3)
        for _ in range(2):
            assert_step_bit_exact(gpu, cpu)
        assert_step_bit_exact(gpu, cpu, np.random.default_rng(1).integers(0, B + 40, B))
        lg = gpu.train_steps(3)
        for _ in range(3):
            lc = cpu.train_step(want_td=False)
        assert lg == lc
        np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
        np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
        np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
        names = [n for n, _ in gpu.profile_step()]
        if fused:
            assert "head_cols4" in names or "red_head" in names, names      # (a plan that splits the hidden layer's forward at this batch takes k_red_head)
            assert netf is not nature_dueling or "head_cols4" in names, names
        else:
            assert "head_cols4" not in names, names
        outs.append((gpu.get_params(0), gpu.get_adam_state()[0]))
        gpu.close(); cpu.close()
    for o in outs[1:]:
        np.testing.assert_array_equal(outs[0][0], o[0]); np.testing.assert_array_equal(outs[0][1], o[1])


def test_fused_reduce_head_hand_off_soak(pkg, monkeypatch):
    """the fused reduce + head launch hands data between workgroups INSIDE a launch (write-through stores, a drained ticket, L1-bypassing loads on the last arriver: MI355X guide,
    Guideline 16).  A stale or torn hand-off would not crash, it would change a number: 4000 train steps at the full Nature-DQN shape (B = 32, 256 workgroups per launch, 8 column
    groups) on the fused schedule and on the two-launch schedule must leave IDENTICAL parameters, Adam state and priorities -- 32 000 hand-offs checked through every bit they feed."""
    net = nature_dueling()
    hp = ref.hparams_for(net, batch_size=32, buffer_size=512, gamma=0.99, learning_rate=1e-4)
    layers = ref.layers_from_network(net)
    plan = pkg.default_plan(layers, hp)
    rng = np.random.default_rng(11)
    s = rng.random((512,) + net.obs_shape, dtype=np.float32); sp = rng.random((512,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, 512).astype(np.int32); r = rng.standard_normal(512).astype(np.float32); d = (rng.random(512) < 0.1).astype(np.uint8)
    p = O.Network.flatten(O.init_params(net, seed=3))
    outs = []
    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("DQN_NO_RED_HEAD", "1")
        g = pkg.Engine(layers, hp, plan=plan)
        monkeypatch.delenv("DQN_NO_RED_HEAD", raising=False)
        g.set_params(p, 0); g.set_params(p * np.float32(0.95), 1); g.replay_add(s, a, r, sp, d)
        for chunk in (1, 37, 962, 3000):
            loss, gn = g.train_steps(chunk)
            assert np.isfinite(loss) and np.isfinite(gn)
        outs.append((g.get_params(0), g.get_adam_state(), g.replay_priorities(), (loss, gn)))
        g.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    assert outs[0][3] == outs[1][3]


def test_nature_u8_b32_byte_arena_bit_exact(pkg):
    """u8 replay with the Nature-DQN first layer: the observation arena stays in BYTES (gather writes 1 byte per element, conv1's forward and dW
    tile loads convert byte / 255f0 exactly) -- every one of the 256 byte values occurs in the random rows; bit-exact vs the twin's plain
    (float)b / 255f0, with the fused sample+gather launch of small batches."""
    net = nature_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 32, cap=256, obs_dtype=1, gamma=0.99)
    fill((gpu, cpu), net, 256, seed=12, u8=True)
    set_same_params((gpu, cpu), net, seed=4)
    for _ in range(3):
        assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_grads(), cpu.get_grads())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))


def test_config5_nature_b512_u8_bit_exact(pkg):
    """BASELINE config 5's shape under pytest: Nature-DQN dueling, B = 512, u8 replay (4096 transitions here; the 1e6-transition property
    run is test_config5_million_transition_properties): the large-batch program (own sampler launch, multi-workgroup head reduce, u8 gather,
    side-stream priority update) bit-exact vs the twin for two sampled steps."""
    net = nature_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 512, cap=4096, obs_dtype=1, gamma=0.99)
    cpu.set_threads(64)
    fill((gpu, cpu), net, 4096, seed=11, u8=True)
    set_same_params((gpu, cpu), net, seed=2)
    for _ in range(2):
        assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())


@pytest.mark.parametrize("split", [0, 16, 4096])
def test_dw_section_half_units_bit_exact(pkg, monkeypatch, split):
    """dw_section (csrc/nn_gemm.hip, r06): the last DQN_DW_SPLIT units of a large-batch dW section (64-channel tiles, chunks of >= 4 K tiles) run as two
    halves along N so that a backward launch does not end in a drain of whole units.  Another assignment of the same chains to workgroups: bit-identical to
    the twin with none (0), a few (16) and as many as allowed (4096 -> half of the section's units) of the units cut; the default (128) is what every other
    large-batch test runs."""
    net = nature_dueling()
    monkeypatch.setenv("DQN_DW_SPLIT", str(split))
    gpu, cpu, hp = make_pair(pkg, net, 256, cap=1024, obs_dtype=1, gamma=0.99)
    monkeypatch.delenv("DQN_DW_SPLIT", raising=False)
    cpu.set_threads(64)
    fill((gpu, cpu), net, 700, seed=split + 5, u8=True)
    set_same_params((gpu, cpu), net, seed=4)
    assert_step_bit_exact(gpu, cpu)
    assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))


@pytest.mark.parametrize("B,u8,off", [(128, True, False), (160, True, False), (144, True, False), (128, False, False), (144, False, False), (128, True, True)])
def test_weights_resident_first_conv_forward_bit_exact(pkg, monkeypatch, B, u8, off):
    """k_fwd_wres (csrc/nn_gemm.hip): from 2048 M-groups up the forward of a layer of <= 32 channels keeps its weights in LDS and walks the output with
    long-lived workgroups, a wave owning 4 / 2 / 1 adjacent M-tiles (B = 128 / 160 / 144: the column tiles of [s;sp] and sp divide by 4 / 2 / 1), byte or
    float operands.  Another schedule of the same chains: bit-identical to the twin; off = DQN_NO_FWD_WRES keeps the per-tile kernel's large-launch
    form (16-deep K tiles) under test at the same shape."""
    net = nature_dueling()
    if off:
        monkeypatch.setenv("DQN_NO_FWD_WRES", "1")
    gpu, cpu, hp = make_pair(pkg, net, B, cap=1024, obs_dtype=1 if u8 else 0, gamma=0.99)
    monkeypatch.delenv("DQN_NO_FWD_WRES", raising=False)
    cpu.set_threads(64)
    fill((gpu, cpu), net, 700, seed=B, u8=u8)
    set_same_params((gpu, cpu), net, seed=3)
    assert_step_bit_exact(gpu, cpu)
    assert_step_bit_exact(gpu, cpu)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))


def test_large_batch_train_steps_pipelined(pkg):
    """B > 64: the priority update runs on the side stream and also draws the next step's indices; inside dqn_train_steps the Adam launch gathers
    the next batch (byte arena) and the next step starts without its sample and gather launches (k_td publishes the indices).  Mid-size network at
    B = 128 / u8: equal to the twin stepped one call at a time."""
    net = mid_conv_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 128, cap=1024, obs_dtype=1, learning_rate=1e-3, gamma=0.99)
    fill((gpu, cpu), net, 900, seed=41, u8=True)
    set_same_params((gpu, cpu), net, seed=42)
    lg = gpu.train_steps(4)
    for _ in range(4):
        lc = cpu.train_step()
    assert lg[0] == lc[0] and lg[1] == lc[1]
    np.testing.assert_array_equal(gpu.last_indices(), cpu.last_indices())
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    assert_step_bit_exact(gpu, cpu)
    names = [n for n, _ in gpu.profile_step(steady=True)]
    cpu.train_step(); cpu.train_step()
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    if gpu.batch_arena_elem_bytes() == 1:
        assert "adam+gather" in names and "gather" not in names and "sample" not in names, names
    assert_step_bit_exact(gpu, cpu)


def test_config5_million_transition_properties(pkg):
    """BASELINE config 5 at FULL size: replay of 1 000 000 u8 transitions (56 GB of rows in HBM), filled by the device env loop; size-independent
    properties: every sampled index in range and spread over the whole ring, finite loss, the sum-tree root equals the sum of the leaves, the
    step touches exactly the sampled leaves, gathered rows equal the stored rows."""
    import importlib
    envs = importlib.import_module(pkg.__name__ + ".envs")
    net = nature_dueling()
    N, B = 1_000_000, 512
    hp = ref.hparams_for(net, batch_size=B, buffer_size=N, obs_dtype=1, gamma=0.99, seed=3)
    layers = ref.layers_from_network(net)
    gpu = pkg.Engine(layers, hp)
    p = O.Network.flatten(O.init_params(net, seed=1))
    gpu.set_params(p, 0); gpu.set_params(p, 1)
    env = envs.TestMDP((84, 84), 4, 6, n=32, seed=7, u8=True)
    gpu.envs_create(env, n_envs=1000, max_episode_length=100, seed=5)
    gpu.rollout(N // 1000, t0=1, train_freq=0, target_update_freq=0, eps=(1.0, 1.0, 1.0), stats=False)
    assert gpu.replay_size() == (N, N)
    pr = gpu.replay_priorities()
    assert pr.shape == (N,) and np.all(pr > 0)
    seen = np.zeros(N, bool)
    for step in range(6):
        before = pr
        loss, gn, td = gpu.train_step()
        idx = gpu.last_indices()
        assert idx.min() >= 0 and idx.max() < N and np.isfinite(loss) and np.all(np.isfinite(td)) and gn > 0
        seen[idx] = True
        pr = gpu.replay_priorities()
        changed = np.flatnonzero(pr != before)
        assert np.isin(changed, idx).all()                                                   # only sampled leaves move
        last = {int(i): k for k, i in enumerate(idx)}                                        # duplicates: last write wins
        keep = np.array(sorted(last.values()))
        np.testing.assert_allclose(pr[idx[keep]], O.priority_from_td(np.abs(td[keep]), hp.prio_eps, hp.prio_alpha), rtol=2e-7)
    assert np.flatnonzero(seen).max() > N // 2 and np.flatnonzero(seen).min() < N // 2       # draws cover the ring, not a prefix
    # IS weights carry the tree root: w = (n p / root)^-beta  =>  root recovered from one weight must equal the fp64 sum of the leaves
    idx = np.arange(B, dtype=np.int64) * (N // B)
    s_rows, a, r, sp_rows, done, w = gpu.get_batch(idx)
    root = N * pr[idx].astype(np.float64) / w.astype(np.float64) ** (-1.0 / np.float64(np.float32(hp.prio_beta)))
    np.testing.assert_allclose(root, pr.astype(np.float64).sum(), rtol=2e-5)
    ex = gpu.replay_export(first=int(idx[5]), n=1)
    np.testing.assert_array_equal(s_rows[5], ex[0][0].astype(np.float32) / np.float32(255))
    np.testing.assert_array_equal(sp_rows[5], ex[1][0].astype(np.float32) / np.float32(255))
    gpu.close()


def test_errors_are_loud(pkg):
    net = cfg1_mlp_dueling()
    gpu, cpu, _ = make_pair(pkg, net, 32, cap=64)
    with pytest.raises(pkg.DQNError, match="_curr_size >= r.batch_size"):
        gpu.train_step()
    with pytest.raises(pkg.DQNError):
        gpu.set_params(np.zeros(3, np.float32))
    # update_priorities! asserts before it assigns (...replay.jl:77-79): a NaN TD error is refused and the priorities stay as they were
    fill((gpu,), net, 40)
    before = gpu.replay_priorities()
    idx = np.arange(32, dtype=np.int64)
    with pytest.raises(pkg.DQNError, match="new_priorities"):
        gpu.update_priorities(idx, np.full(32, np.nan, np.float32))
    np.testing.assert_array_equal(gpu.replay_priorities(), before)
    gpu.train_step()                                        # and the engine keeps working
    # zero-length and degenerate calls
    gpu.replay_add(np.zeros((0, 2), np.float32), np.zeros(0, np.int32), np.zeros(0, np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint8))
    with pytest.raises(pkg.DQNError, match="n must be >= 1"):
        gpu.forward(np.zeros((0, 2), np.float32))
    gpu.close()
    with pytest.raises(pkg.DQNError, match="closed"):      # not a crash
        gpu.train_step()
    gpu.close()                                            # idempotent


def test_rccl_path_world1_matches_plain(pkg, monkeypatch):
    """The data-parallel code path (dlopen'ed RCCL communicator, step graph cut in two around ncclAllReduce on the engine
    stream) forced on at world_size 1: an all-reduce over one rank is the identity, so results must equal the plain path."""
    monkeypatch.setenv("DQN_FORCE_ALLREDUCE", "1")
    net = small_conv_dueling()
    a, cpu, _ = make_pair(pkg, net, 16)
    b, _, _ = make_pair(pkg, net, 16)
    fill((a, b, cpu), net, 64); set_same_params((a, b, cpu), net)
    a.comm_init(pkg.comm_unique_id(), 0, 1)
    for _ in range(4):
        ra, rb, rc = a.train_step(), b.train_step(), cpu.train_step()
        assert ra[0] == rb[0] == rc[0] and ra[1] == rb[1]
        np.testing.assert_array_equal(ra[2], rb[2])
    np.testing.assert_array_equal(a.get_params(0), b.get_params(0))
    np.testing.assert_array_equal(a.get_params(0), cpu.get_params(0))


@pytest.mark.parametrize("u8", [False, True])
def test_checkpoint_resume_is_bit_exact(pkg, u8, tmp_path):
    """dqn_replay_export/import + counters + parameters + Adam state (the reference only saves the best network, src/solver.jl:290-318):
    a run continued from a checkpoint in a NEW engine reproduces the uninterrupted run bit for bit (sampled indices, TD errors, loss, parameters)."""
    net = small_conv_dueling()
    a, cpu, hp = make_pair(pkg, net, 16, cap=100, obs_dtype=1 if u8 else 0, learning_rate=1e-3)
    fill((a, cpu), net, 137, u8=u8); set_same_params((a, cpu), net)
    for _ in range(5):
        a.train_step()
    ck = a.checkpoint()
    np.savez(tmp_path / "ck.npz", **ck)
    ref_run = [a.train_step() for _ in range(4)]
    b, _, _ = make_pair(pkg, net, 16, cap=100, obs_dtype=1 if u8 else 0, learning_rate=1e-3)
    b.restore(dict(np.load(tmp_path / "ck.npz")))
    assert b.get_counters() == {k: int(v) for k, v in zip(("size", "widx", "sample_ctr", "train_steps"), ck["counters"])}
    for want in ref_run:
        got = b.train_step()
        assert got[0] == want[0] and got[1] == want[1]
        np.testing.assert_array_equal(got[2], want[2])
    np.testing.assert_array_equal(a.get_params(0), b.get_params(0)); np.testing.assert_array_equal(a.get_params(1), b.get_params(1))
    np.testing.assert_array_equal(a.replay_priorities(), b.replay_priorities())
    ma, va, ba = a.get_adam_state(); mb, vb, bb = b.get_adam_state()
    np.testing.assert_array_equal(ma, mb); np.testing.assert_array_equal(va, vb); np.testing.assert_array_equal(ba, bb)
    # the ring cursor survives: the next add lands in the same slot
    s1, a1, r1, sp1, d1 = fill((a, b), net, 3, seed=9, u8=u8)
    np.testing.assert_array_equal(a.replay_priorities(), b.replay_priorities())
    z = np.zeros((101,) + net.obs_shape, b.obs_np)
    with pytest.raises(pkg.DQNError, match="capacity"):
        b.replay_import(z, z, np.zeros(101, np.int32), np.zeros(101, np.float32), np.zeros(101, np.uint8), np.ones(101, np.float32))
    # the checkpoint records the summation plan and the version of its semantics (include/dqn_mi355x.h DQN_PLAN_VERSION): a resume that could not be bit-exact is refused
    assert int(ck["plan_version"]) == pkg.fns()["plan_version"]() == 3 and [tuple(r) for r in ck["plan"]] == a.plan()
    with pytest.raises(pkg.DQNError, match="plan version"):
        b.restore(dict(ck, plan_version=np.int32(2)))
    other = [(p[0], p[1], 64 if p[2] else p[2]) for p in a.plan()]
    if other != a.plan():
        c = pkg.Engine(ref.layers_from_network(net), hp, plan=other)
        with pytest.raises(pkg.DQNError, match="different summation plan"):
            c.restore(ck)
        c.close()


def test_sampler_distinct_gpu_equals_twin(pkg):
    """hp.sample_distinct = 1 (...replay.jl:85 replace=false): the engine's draws are distinct, pass the distribution checks of
    parity_common.sampler_distinct, and are the twin's draws index for index."""
    g = sampler_distinct(pkg.Engine)
    c = sampler_distinct(ref.Twin, threads=1)
    np.testing.assert_array_equal(g, c)


@pytest.mark.parametrize("netf,B,cap,kw", [(small_conv_dueling, 32, 64, {}), (small_conv_dueling, 128, 200, {}), (cfg1_mlp_dueling, 32, 64, {}), (cfg1_mlp_dueling, 32, 64, dict(_tiny=False)),
                                           (mid_conv_dueling, 16, 64, dict(obs_dtype=1)), (nature_dueling, 512, 1024, dict(obs_dtype=1))],
                         ids=["B32_fused_heads", "B128_large_batch_path", "cfg1_single_launch_step", "cfg1_multi_launch", "u8_byte_arena", "B512_config5_shape"])
def test_train_steps_with_distinct_sampling_bit_exact(pkg, netf, B, cap, kw):
    """train steps in distinct mode (...replay.jl:85, replace=false): sampled indices never repeat inside a batch even with a dominant priority, and indices / TD errors /
    loss / parameters equal the twin's bit for bit, single steps and train_steps(n).  r05: at B <= 64 the mode keeps the fast path -- the priority block dedupes the list
    it pre-draws, the fused sample + gather workgroups (and the single-launch step) dedupe a list they draw themselves, dqn_train_steps pre-gathers as in the default
    mode.  r06: larger batches too -- the priority workgroup that rides a backward launch dedupes its pre-drawn list (one wave runs the sequential redraw, common.h
    sample_distinct_fix) and the Adam launch pre-gathers it; B512_config5_shape is BASELINE configs[4]'s network and batch with ONE dominant priority, i.e. ~500 redraws per batch."""
    net = netf()
    u8 = kw.get("obs_dtype", 0) == 1
    gpu, cpu, hp = make_pair(pkg, net, B, cap=cap, sample_distinct=1, learning_rate=1e-3, **kw)
    if B >= 512:
        cpu.set_threads(64)
    fill((gpu, cpu), net, cap, seed=4, u8=u8)
    set_same_params((gpu, cpu), net)
    big = np.full(cap, 0.01, np.float32); big[5] = 50.0                 # one transition dominates: the stratified default would repeat it
    for h in (gpu, cpu):
        h.update_priorities(np.arange(cap, dtype=np.int64), big)
    for step in range(4):
        assert_step_bit_exact(gpu, cpu)
        idx = gpu.last_indices()
        assert len(set(idx.tolist())) == B, idx
    lg = gpu.train_steps(3)
    for _ in range(3):
        lc = cpu.train_step()
    assert lg[0] == lc[0], (lg, lc)
    np.testing.assert_array_equal(gpu.last_indices(), cpu.last_indices())
    assert len(set(gpu.last_indices().tolist())) == B
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    # a long pipelined run with replay writes in between (they invalidate the pre-drawn list: the next gather draws and dedupes itself)
    rng = np.random.default_rng(2)
    for n in (5, 1, 7):
        lg = gpu.train_steps(n)
        for _ in range(n):
            lc = cpu.train_step(want_td=False)
        assert lg == lc, (n, lg, lc)
        assert len(set(gpu.last_indices().tolist())) == B
        if u8:
            s = rng.integers(0, 256, (3,) + net.obs_shape).astype(np.uint8)
        else:
            s = rng.random((3,) + net.obs_shape, dtype=np.float32)
        for h in (gpu, cpu):
            h.replay_add(s, np.zeros(3, np.int32), np.full(3, 30.0, np.float32), s, np.zeros(3, np.uint8))      # heavy new transitions: duplicates likely among the stratified draws
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    names = [n for n, _ in gpu.profile_step()]
    if B <= 64 and (not u8 or gpu.batch_arena_elem_bytes() == 1):
        assert "sample" not in names, names          # no separate sample launch on the small-batch path (u8 rows into a FLOAT arena keep it: that gather kernel has its own index code)
    if B >= 512:                                      # inside dqn_train_steps the large-batch step has neither a sample nor a gather launch: the previous step's Adam launch gathered the deduped list
        steady = [n for n, _ in gpu.profile_step(steady=True)]
        assert "sample" not in steady and "gather" not in steady and any(n.startswith("adam+gather") for n in steady), steady
    gpu.close(); cpu.close()


@pytest.mark.parametrize("graph", [0, 1])
@pytest.mark.parametrize("netf,B", [(small_conv_dueling, 16), (cfg1_mlp_dueling, 32), (mid_conv_dueling, 32)])
def test_scalar_mailbox_and_async_step_bit_exact(pkg, netf, B, graph):
    """dqn_train_step with scalar outputs publishes (loss, grad_norm) from the step's last launch into a mapped host ring (no fold launch / D2H copy / stream
    synchronize), dqn_train_step_async returns before the step has run and dqn_step_scalars fetches a ticket's record later: every record equals what the
    twin's batch_train! returns (src/solver.jl:235), bit for bit, in any interleaving with the other entry points."""
    net = netf()
    gpu, cpu, hp = make_pair(pkg, net, B, cap=128, graph=graph, learning_rate=1e-3)
    fill((gpu, cpu), net, 100, seed=3)
    set_same_params((gpu, cpu), net, seed=2)
    for _ in range(3):                                    # synchronous, scalars only: the mailbox path
        assert gpu.train_step(want_td=False) == cpu.train_step(want_td=False)
    assert_step_bit_exact(gpu, cpu)                       # with td: the copy path, still the same numbers
    rng = np.random.default_rng(4)
    tickets, want = [], []
    for k in range(70):                                   # more than the ring holds
        idx = None if k % 3 else rng.integers(0, 100, B)
        tickets.append(gpu.train_step_async(idx)); want.append(cpu.train_step(idx, want_td=False))
    assert tickets == list(range(tickets[0], tickets[0] + 70))
    for t, w in list(zip(tickets, want))[-60:]:           # any of the 64 newest, in any order, more than once
        assert gpu.step_scalars(t) == w
    assert gpu.step_scalars(tickets[-1], wait=False) == want[-1]
    with pytest.raises(pkg.DQNError, match="older than"):
        gpu.step_scalars(tickets[0])
    with pytest.raises(pkg.DQNError, match="never issued"):
        gpu.step_scalars(tickets[-1] + 1)
    np.testing.assert_array_equal(gpu.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(gpu.replay_priorities(), cpu.replay_priorities())
    lg = gpu.train_steps(4)                               # the batched entry point afterwards
    for _ in range(4):
        lc = cpu.train_step(want_td=False)
    assert lg == lc
    gpu.close(); cpu.close()


def test_device_assertion_travels_in_the_mailbox_once(pkg):
    """assert all(new_priorities .> 0) (...replay.jl:78) fails on the DEVICE when a TD error is NaN.  The failure is consumed into exactly one mailbox record
    (k_publish_scalars), and the host reports it once, naming the step, at the first call that sweeps that record -- not once per outstanding ticket"""
    net = cfg1_mlp_dueling()
    gpu, cpu, hp = make_pair(pkg, net, 32, cap=128, learning_rate=1e-3)
    fill((gpu, cpu), net, 100, seed=3)
    set_same_params((gpu, cpu), net, seed=2)
    assert gpu.train_step(want_td=False) == cpu.train_step(want_td=False)
    rng = np.random.default_rng(0)
    s = rng.random((1,) + net.obs_shape, dtype=np.float32)
    gpu.replay_add(s, np.zeros(1, np.int32), np.full(1, np.nan, np.float32), s, np.zeros(1, np.uint8), td_err=np.ones(1, np.float32))      # a transition with a NaN reward at slot 100
    idx = np.arange(69, 101).astype(np.int64)
    t1 = gpu.train_step_async(idx)                        # its TD error is NaN -> a NaN priority -> the device-side assertion
    with pytest.raises(pkg.DQNError, match="new_priorities"):
        gpu.step_scalars(t1)
    l, g = gpu.step_scalars(t1)                           # reported once: the record itself stays readable
    assert np.isnan(l)
    gpu.close(); cpu.close()


def test_engine_switches_are_read_at_creation_and_stay_per_engine(pkg, monkeypatch):
    """experiment / test switches are read ONCE, in dqn_engine_create, into the engine (EngineOpts, csrc/engine.h): a second engine in the same process does
    not inherit the first one's, and changing the environment afterwards changes nothing for an existing engine"""
    net = cfg1_mlp_dueling()
    hp = ref.hparams_for(net, batch_size=32, buffer_size=64)
    layers = ref.layers_from_network(net)
    monkeypatch.setenv("DQN_NO_TINY", "1")
    a = pkg.Engine(layers, hp)                       # created under DQN_NO_TINY: the multi-launch program
    monkeypatch.delenv("DQN_NO_TINY")
    b = pkg.Engine(layers, hp)                       # created without: the single-launch step
    monkeypatch.setenv("DQN_NO_TINY", "1")           # ... and setting it again must not reach b
    fill((a, b), net, 40, seed=1); set_same_params((a, b), net, seed=1)
    na = [n for n, _ in a.profile_step()]; nb = [n for n, _ in b.profile_step()]
    assert "tiny_step" not in na and len(na) > 3, na
    assert nb == ["tiny_step"], nb
    for _ in range(3):                               # two schedules of the same arithmetic
        ra, rb = a.train_step(), b.train_step()
        assert ra[0] == rb[0] and ra[1] == rb[1]; np.testing.assert_array_equal(ra[2], rb[2])
    a.close(); b.close()
