"""GPU (-m gpu): the data-parallel exchange of DESIGN.md section 8 -- ONE all-gather of the wide dense layers' operands (X, dpre) and of the
small gradients instead of an all-reduce of the whole gradient -- checked on one GPU:
  * DQN_SIM_WORLD=k: one process plays k identical ranks (the collective is k local copies); the step must equal the CPU twin's single-device
    step on the CONCATENATED batch of k*B samples (SURVEY.md 8e's definition of multi-GPU parity) -- bit for bit on the wide dense layers
    (rank-major contraction order == the concatenated batch's column order), to round-off on the layers summed over ranks;
  * the real RCCL ncclAllGather (dlopen'ed communicator on the engine stream between the two step graphs) forced on at world_size 1 must
    reproduce the plain single-GPU path exactly."""
import numpy as np
import pytest

import __graft_entry__ as ge
import dqn_oracle as O
import ref

pytestmark = pytest.mark.gpu
R, I = O.ACT_RELU, O.ACT_IDENTITY


@pytest.fixture(scope="module")
def pkg():
    p = ge.load_package()
    p.lib()
    return p


def wide_dense_dueling():
    """conv trunk + Dense(192, 512) dueling streams: (K+1)*N = 98 816 floats of gradient per stream against 4*(K+N)*B = 90 112 of operands."""
    b, v, a = O.create_dueling_network([O.Conv(4, 3, 8, R, 2), O.Conv(3, 8, 16, R, 1), O.Dense(16 * 3 * 4, 512, R), O.Dense(512, 5, I)])
    return O.Network((3, 12, 14), b, v, a)


def concat_plan(pkg, layers, hp_rank, hp_concat):
    """summation-order plan of the single-device step on the concatenated batch: the per-sample contractions (forward, dX) are rounded as on the
    ranks (their chunking is per column and may depend on the per-rank batch size), the sample-axis contraction (dW) as ONE device would round
    the k*B-sample batch"""
    return [(pr[0], pr[1], pc[2]) for pr, pc in zip(pkg.default_plan(layers, hp_rank), pkg.default_plan(layers, hp_concat))]


def setup(pkg, net, B, Bt, cap=256, n_fill=200, seed=0):
    layers = ref.layers_from_network(net)
    hp_g = ref.hparams_for(net, batch_size=B, buffer_size=cap, learning_rate=1e-3, gamma=0.99)
    hp_t = ref.hparams_for(net, batch_size=Bt, buffer_size=cap, learning_rate=1e-3, gamma=0.99)
    g = pkg.Engine(layers, hp_g, plan=pkg.default_plan(layers, hp_g))
    t = ref.Twin(layers, hp_t, plan=concat_plan(pkg, layers, hp_g, hp_t), threads=8)
    rng = np.random.default_rng(seed)
    s = rng.random((n_fill,) + net.obs_shape, dtype=np.float32); sp = rng.random((n_fill,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, n_fill).astype(np.int32); r = (rng.standard_normal(n_fill) * 2).astype(np.float32); d = (rng.random(n_fill) < 0.2).astype(np.uint8)
    p = O.Network.flatten(O.init_params(net, seed=5))
    p = (p + 0.01 * rng.standard_normal(p.shape)).astype(np.float32)
    for h in (g, t):
        h.replay_add(s, a, r, sp, d); h.set_params(p, 0); h.set_params(p, 1)
    return g, t, rng


def wide_blocks(net, flat):
    blocks = [w for w in net.unflatten(flat) if w.ndim == 2 and 512 in w.shape and 192 in w.shape]      # the two Dense(192, 512) weight matrices
    assert len(blocks) == 2
    return blocks


@pytest.mark.parametrize("k", [2, 4, 8])
def test_simulated_ranks_equal_concatenated_batch(pkg, monkeypatch, k):
    net = wide_dense_dueling()
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g, t, rng = setup(pkg, net, 32, 32 * k)
    monkeypatch.delenv("DQN_SIM_WORLD")
    for step in range(3):
        idx = rng.choice(200, 32, replace=False).astype(np.int64)
        lg, gg, tdg = g.train_step(idx)
        lt, gt, tdt = t.train_step(np.tile(idx, k))
        np.testing.assert_array_equal(tdg, tdt[:32])                       # same per-sample TD errors
        np.testing.assert_allclose(lg, lt, rtol=1e-6)                      # mean over 32 vs over k copies of them
        Gg, Gt = g.get_grads() / np.float32(k), t.get_grads()              # sum over ranks, scaled by 1/world (exact power of two)
        for x, y in zip(wide_blocks(net, Gg), wide_blocks(net, Gt)):
            np.testing.assert_array_equal(x, y)                             # gathered contraction == the concatenated batch's, bit for bit
        np.testing.assert_allclose(Gg, Gt, rtol=1e-4, atol=1e-7)            # conv / head gradients: summed over ranks vs one long chain
        assert abs(gg - gt) <= 1e-6 * max(1.0, abs(gt))
        Pg, Pt = g.get_params(0), t.get_params(0)
        assert np.abs(Pg - Pt).max() <= 2.1e-3                              # Adam at |g| ~ eps moves up to lr
        assert (np.abs(Pg - Pt) > 2e-6).mean() < 1e-3
        np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())
        t.set_params(Pg, 0)                                                 # keep the two trajectories on the same parameters


def _distinct_rank_step(g, t, net, k, B, idx, wide_shape, lr):
    """one data-parallel step with k DISTINCT per-rank batches against the twin's single-device step on the true concatenated batch"""
    pr_before = g.replay_priorities()
    lossg, gg, tdg = g.sim_ranks_step(idx)
    lt, gt, tdt = t.train_step(idx.reshape(-1))
    np.testing.assert_array_equal(tdg.reshape(-1), tdt)                     # rank r's TD errors == columns r*B..(r+1)*B-1 of the big batch
    assert np.unique(tdg, axis=0).shape[0] == k                              # the ranks really did different work
    np.testing.assert_allclose(lossg.astype(np.float64).mean(), lt, rtol=2e-6)   # mean of per-rank means == mean over k*B
    Gg, Gt = g.get_grads() / np.float32(k), t.get_grads()
    wide = [(x, y) for x, y in zip(net.unflatten(Gg), net.unflatten(Gt)) if x.ndim == 2 and x.shape == wide_shape]
    assert len(wide) == 2
    for x, y in wide:
        np.testing.assert_array_equal(x, y)                                  # gathered contraction, rank-major == the concatenated batch's, bit for bit
    np.testing.assert_allclose(Gg, Gt, rtol=2e-4, atol=1e-6 * np.abs(Gt).max())   # conv / head gradients: per-rank chains summed over ranks vs one long chain
    assert abs(gg - gt) <= 1e-5 * max(1.0, abs(gt))
    Pg, Pt = g.get_params(0), t.get_params(0)
    assert np.abs(Pg - Pt).max() <= 2.1 * lr                                 # Adam at |g| ~ eps moves up to lr
    assert (np.abs(Pg - Pt) > 2e-6).mean() < 1e-3
    np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())   # every rank's update_priorities! landed
    assert (g.replay_priorities() != pr_before).sum() >= k * B - 2
    t.set_params(Pg, 0)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_distinct_ranks_equal_concatenated_batch(pkg, monkeypatch, k):
    """k simulated ranks with k DIFFERENT index lists (dqn_sim_ranks_step): a wrong rank offset anywhere in the exchange -- the pack layout,
    the rank stride of the gathered dW operands (DwStride), the sum over ranks -- reads another rank's data and turns this red (with identical
    ranks it could not)."""
    net = wide_dense_dueling()
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g, t, rng = setup(pkg, net, 32, 32 * k, cap=512, n_fill=400)
    monkeypatch.delenv("DQN_SIM_WORLD")
    for step in range(2):
        idx = rng.choice(400, (k, 32), replace=False).astype(np.int64)
        _distinct_rank_step(g, t, net, k, 32, idx, (192, 512), 1e-3)


def test_distinct_ranks_detects_a_wrong_rank_offset(pkg, monkeypatch):
    """the test above CAN fail: feeding the twin the batch with two rank blocks swapped (what a wrong rank offset would compute) breaks it"""
    net = wide_dense_dueling()
    k = 4
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g, t, rng = setup(pkg, net, 32, 32 * k, cap=512, n_fill=400)
    monkeypatch.delenv("DQN_SIM_WORLD")
    idx = rng.choice(400, (k, 32), replace=False).astype(np.int64)
    g.sim_ranks_step(idx)
    t.train_step(idx[[1, 0, 2, 3]].reshape(-1))
    Gg, Gt = g.get_grads() / np.float32(k), t.get_grads()
    wide = [(x, y) for x, y in zip(net.unflatten(Gg), net.unflatten(Gt)) if x.ndim == 2 and x.shape == (192, 512)]
    assert any(not np.array_equal(x, y) for x, y in wide)                    # same SET of samples, other ORDER: the bit-exact check notices


def test_config3_nature_dqn_8_ranks_distinct_batches(pkg, monkeypatch):
    """BASELINE config 3's exchange at FULL size on one GPU: Nature-DQN dueling (84x84x4), 8 ranks x B = 32 with distinct batches; the two
    3136x512 dense layers go through the operand all-gather (0.86 MB per rank), everything else through the sum over ranks."""
    from nets import nature_dueling
    net = nature_dueling()
    k, B = 8, 32
    layers = ref.layers_from_network(net)
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    hp_g = ref.hparams_for(net, batch_size=B, buffer_size=512, gamma=0.99)
    g = pkg.Engine(layers, hp_g, plan=pkg.default_plan(layers, hp_g))
    monkeypatch.delenv("DQN_SIM_WORLD")
    hp_t = ref.hparams_for(net, batch_size=B * k, buffer_size=512, gamma=0.99)
    t = ref.Twin(layers, hp_t, plan=concat_plan(pkg, layers, hp_g, hp_t), threads=64)
    rng = np.random.default_rng(3)
    n = 384
    s = rng.random((n,) + net.obs_shape, dtype=np.float32); sp = rng.random((n,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, n).astype(np.int32); r = (rng.standard_normal(n) * 2).astype(np.float32); d = (rng.random(n) < 0.2).astype(np.uint8)
    p = O.Network.flatten(O.init_params(net, seed=5))
    p = (p + 0.01 * rng.standard_normal(p.shape)).astype(np.float32)
    for h in (g, t):
        h.replay_add(s, a, r, sp, d); h.set_params(p, 0); h.set_params(p * np.float32(0.9), 1)
    idx = rng.choice(n, (k, B), replace=False).astype(np.int64)
    _distinct_rank_step(g, t, net, k, B, idx, (3136, 512), 1e-4)


def test_rccl_allgather_world1_matches_plain(pkg, monkeypatch):
    net = wide_dense_dueling()
    a, cpu, rng = setup(pkg, net, 32, 32, seed=1)
    b, _, _ = setup(pkg, net, 32, 32, seed=1)
    monkeypatch.setenv("DQN_FORCE_ALLREDUCE", "1")                         # run the communicator path although world_size == 1
    assert b.comm_info()["rccl_nranks"] == 0 and b.comm_info()["exchange"] == 0           # no communicator: nothing to report
    a.comm_init(pkg.comm_unique_id(), 0, 1)
    ci = a.comm_info()                                                                   # what the communicator ITSELF says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)
    assert (ci["rccl_nranks"], ci["rccl_rank"], ci["rccl_device"], ci["engine_world"], ci["engine_rank"]) == (1, 0, 0, 1, 0) and ci["exchange"] == -1
    for _ in range(4):
        ra, rb, rc = a.train_step(), b.train_step(), cpu.train_step()
        assert ra[0] == rb[0] == rc[0] and ra[1] == rb[1]
        np.testing.assert_array_equal(ra[2], rb[2])
    # r05: the scalar mailbox on replicas -- the publish launch is enqueued behind the exchange + Adam, so batch_train!'s (loss, grad_norm) cost no fold launch / D2H copy /
    # stream synchronize there either; synchronous and asynchronous forms, same records as the plain engine and the twin
    for _ in range(2):
        assert a.train_step(want_td=False) == b.train_step(want_td=False) == cpu.train_step(want_td=False)
    tk = [a.train_step_async() for _ in range(3)]
    want = [cpu.train_step(want_td=False) for _ in range(3)]
    assert [b.train_step(want_td=False) for _ in range(3)] == want
    assert [a.step_scalars(t) for t in tk] == want
    np.testing.assert_array_equal(a.get_params(0), b.get_params(0))
    np.testing.assert_array_equal(a.get_params(0), cpu.get_params(0))
    np.testing.assert_array_equal(a.get_grads(), b.get_grads())
    assert a.comm_info()["exchange"] == 1                                                # the all-gather of the wide dense layers' operands + small gradients


def test_rccl_world1_train_steps_pipelined_gather(pkg, monkeypatch):
    """dqn_train_steps(n) on the replica path (real RCCL communicator at world size 1, step cut in two around ncclAllGather): the second half's
    Adam launch gathers the next batch, the next first half runs without its gather launch -- rank-local work, so it must equal the twin stepped
    one call at a time, with replay writes and single steps in between."""
    monkeypatch.setenv("DQN_FORCE_ALLREDUCE", "1")
    net = wide_dense_dueling()
    g, t, _ = setup(pkg, net, 32, 32, seed=2)
    g.comm_init(pkg.comm_unique_id(), 0, 1)

    def same():
        np.testing.assert_array_equal(g.last_indices(), t.last_indices())
        np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
        np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())

    lg = g.train_steps(5)
    for _ in range(5):
        lt = t.train_step()
    assert lg[0] == lt[0] and lg[1] == lt[1]
    same()
    rg, rt = g.train_step(), t.train_step()
    assert rg[0] == rt[0]; np.testing.assert_array_equal(rg[2], rt[2])
    rng = np.random.default_rng(9)
    s = rng.random((7,) + net.obs_shape, dtype=np.float32); sp = rng.random((7,) + net.obs_shape, dtype=np.float32)
    for h in (g, t):
        h.replay_add(s, np.arange(7, dtype=np.int32) % net.n_actions, np.ones(7, np.float32), sp, np.zeros(7, np.uint8))
    lg = g.train_steps(3)
    for _ in range(3):
        lt = t.train_step()
    assert lg[0] == lt[0] and lg[1] == lt[1]
    same()
    names = [n for n, _ in g.profile_step(steady=True)]
    t.train_step(); t.train_step()
    same()
    assert "adam+gather" in names and "sample_gather" not in names and "dp_pack" in names, names


def test_simulated_ranks_with_two_tiles_per_rank(pkg, monkeypatch):
    """B = 64 per rank: each rank block of the gathered sample axis spans two 32-sample K tiles (the dW kernel's tiles-per-rank stride path);
    the wide layer here is the network's FIRST layer, so its X operand is packed out of the observation arena (leading dimension 2B)."""
    net = O.Network((768,), *O.create_dueling_network([O.Dense(768, 1024, R), O.Dense(1024, 5, I)]))
    k = 2
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g, t, rng = setup(pkg, net, 64, 64 * k)
    monkeypatch.delenv("DQN_SIM_WORLD")
    idx = rng.choice(200, 64, replace=False).astype(np.int64)
    lg, gg, tdg = g.train_step(idx)
    lt, gt, tdt = t.train_step(np.tile(idx, k))
    np.testing.assert_array_equal(tdg, tdt[:64])
    Gg, Gt = g.get_grads() / np.float32(k), t.get_grads()
    wide = [(x, y) for x, y in zip(net.unflatten(Gg), net.unflatten(Gt)) if x.ndim == 2 and 768 in x.shape]
    assert len(wide) == 2
    for x, y in wide:
        np.testing.assert_array_equal(x, y)
    np.testing.assert_allclose(Gg, Gt, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("env", [{"DQN_DP_OVERLAP": "1"}, {"DQN_DP_OVERLAP": "1", "DQN_DP_NO_ONE_GRAPH": "1"}, {"DQN_DP_NO_ONE_GRAPH": "1"}],
                         ids=["overlap_one_graph", "overlap_eager_collectives", "no_overlap_eager_collectives"])
def test_rccl_world1_exchange_variants_bit_exact(pkg, monkeypatch, env):
    """The three other schedules of the replica step (default: ONE graph holding [first half | ncclAllGather | second half]):
    DQN_DP_OVERLAP=1 -- the wide layers' operands X | dpre are packed right after k_head_td and all-gathered on the exchange stream WHILE the conv
    backward runs, the small gradients in a second all-gather after it; DQN_DP_NO_ONE_GRAPH=1 -- collectives enqueued
    eagerly between graph segments.  Real RCCL communicator at world size 1; every variant must reproduce the twin bit for bit over single steps,
    train_steps(n) with the pipelined gather, and replay writes in between."""
    monkeypatch.setenv("DQN_FORCE_ALLREDUCE", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    net = wide_dense_dueling()
    g, t, _ = setup(pkg, net, 32, 32, seed=4)
    g.comm_init(pkg.comm_unique_id(), 0, 1)
    for _ in range(3):
        rg, rt = g.train_step(), t.train_step()
        assert rg[0] == rt[0] and rg[1] == rt[1]
        np.testing.assert_array_equal(rg[2], rt[2])
    lg = g.train_steps(5)
    for _ in range(5):
        lt = t.train_step()
    assert lg[0] == lt[0] and lg[1] == lt[1]
    np.testing.assert_array_equal(g.last_indices(), t.last_indices())
    np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    np.testing.assert_array_equal(g.get_grads(), t.get_grads())
    np.testing.assert_array_equal(g.replay_priorities(), t.replay_priorities())
    names = [n for n, _ in g.profile_step(steady=True)]
    t.train_step(); t.train_step()
    np.testing.assert_array_equal(g.get_params(0), t.get_params(0))
    assert ("dp_pack_wide" in names) == ("DQN_DP_OVERLAP" in env), names
    g.close(); t.close()


def test_simulated_ranks_with_overlapped_exchange(pkg, monkeypatch):
    """DQN_DP_OVERLAP=1 under DQN_SIM_WORLD: k DISTINCT rank batches, the two segments of every rank's block landing in its slots of the two gathered
    buffers -- the step must still equal the twin's single-device step on the concatenated batch (and notice swapped rank blocks)."""
    monkeypatch.setenv("DQN_DP_OVERLAP", "1")
    net = wide_dense_dueling()
    k, B = 4, 32
    monkeypatch.setenv("DQN_SIM_WORLD", str(k))
    g, t, rng = setup(pkg, net, B, k * B, cap=512, n_fill=400, seed=6)
    monkeypatch.delenv("DQN_SIM_WORLD")
    idx = rng.choice(400, (k, B), replace=False).astype(np.int64)
    _distinct_rank_step(g, t, net, k, B, idx, (192, 512), 1e-3)
    names = [n for n, _ in g.profile_step()]
    assert "dp_pack_wide" in names and "dp_pack" in names, names
