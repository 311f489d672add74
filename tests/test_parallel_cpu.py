"""CPU, world_size 2 over gloo: the host plumbing of the data-parallel path (deepqlearning.jl_amd/parallel.py) and the
equivalence the design rests on -- mean over ranks of per-shard gradients == gradient of one step on the concatenated
batch with the loss averaged -- checked with the CPU twin as the checker (the engine itself needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import __graft_entry__ as ge
    import dqn_oracle as O
    import ref
    from nets import small_conv_dueling
    pkg = ge.load_package()
    import importlib
    par = importlib.import_module(pkg.__name__ + ".parallel")
    g = par.Group(backend="gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world=world)
    # 1. id broadcast
    uid = bytes(range(128)) if rank == 0 else None
    assert g.bcast_bytes(uid, 128, 0) == bytes(range(128))
    # 2. sharding covers everything exactly once
    lo, hi = par.shard(37, rank, world)
    cover = g.allreduce_mean(np.bincount(np.arange(lo, hi), minlength=37).astype(np.float64)) * world
    assert np.all(cover == 1)
    assert g.max_over_ranks(float(rank)) == world - 1
    # 3. DP equivalence with the twin: per-rank batch B, IS weights forced to 1 (uniform priorities)
    net = small_conv_dueling(); B = 8
    layers = ref.layers_from_network(net)
    rng = np.random.default_rng(0)
    n = B * world
    s = rng.random((n,) + net.obs_shape, dtype=np.float32); sp = rng.random((n,) + net.obs_shape, dtype=np.float32)
    a = rng.integers(0, net.n_actions, n).astype(np.int32); r = rng.standard_normal(n).astype(np.float32); d = (rng.random(n) < 0.2).astype(np.uint8)
    p = O.Network.flatten(O.init_params(net, seed=4))
    td0 = np.ones(n, np.float32)                        # equal priorities => IS weights == 1 on every rank and on the big batch
    def grads(sl, bsz):
        hp = ref.hparams_for(net, batch_size=bsz, buffer_size=64, gamma=0.99)
        t = ref.Twin(layers, hp, threads=2)
        t.set_params(p, 0); t.set_params(p * 0.5, 1)
        t.replay_add(s[sl], a[sl], r[sl], sp[sl], d[sl], td0[sl])
        t.train_step(np.arange(bsz, dtype=np.int64))
        gr = t.get_grads(); t.close(); return gr
    lo, hi = par.shard(n, rank, world)
    g_mean = g.allreduce_mean(grads(slice(lo, hi), B).astype(np.float64))
    g_big = grads(slice(0, n), n).astype(np.float64)
    err = np.abs(g_mean - g_big).max() / np.abs(g_big).max()
    q.put((rank, float(err)))
    g.barrier(); g.close()


def test_world2_gloo_plumbing_and_dp_equivalence():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err in res:
        assert err < 1e-5, (rank, err)
