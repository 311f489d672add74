"""
Data-parallel replicas (BASELINE config 3): one process per GPU, per-rank vectorised envs + per-rank replay and
sum-tree (no cross-rank sampling), identical parameter replicas, ONE exchange per train step over RCCL inside the engine
(dqn_comm_init; on the engine's stream): an all-gather of the wide dense layers' operands and of the small gradients
(csrc/dp.hip), or an all-reduce of the flat gradient as the fallback; Adam scales by 1/world, so every
rank applies the same update and replicas stay bit-identical.  Equivalent to a single-GPU step on the concatenated batch of
B*world with the loss averaged (tests/test_dp_gpu.py: simulated ranks against the CPU twin; tests/test_parallel_cpu.py: the
same equivalence and this module's plumbing over gloo).  The reference has no distributed code at all.

This module is the host-side plumbing only (torch.distributed -- gloo by default, so that the engine's communicator is
the only RCCL instance in the process -- is used for rendezvous, the 128-byte RCCL id broadcast, barriers and the
max-over-ranks timer); the data path collective never touches torch.
"""
from __future__ import annotations

import os

import numpy as np


def env_info():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard(n_total: int, rank: int, world: int):
    """contiguous shard [lo, hi) of n_total items (envs) for `rank`; sizes differ by at most one."""
    q, r = divmod(n_total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class Group:
    """Thin wrapper over torch.distributed (backend "nccl" == RCCL on ROCm; "gloo" for the CPU tests)."""

    def __init__(self, backend="nccl", device=None, init_method=None, rank=None, world=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        r, w, lr = env_info()
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.device = device if device is not None else (torch.device("cuda", lr) if backend == "nccl" else torch.device("cpu"))
        if self.world > 1 and not dist.is_initialized():
            kw = {}
            if init_method is not None:
                kw = dict(init_method=init_method, rank=self.rank, world_size=self.world)
            if backend == "nccl":
                kw["device_id"] = self.device
            dist.init_process_group(backend, **kw)

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def bcast_bytes(self, data: bytes | None, n: int, src=0) -> bytes:
        """broadcast n bytes from `src` (the RCCL unique id made by dqn_comm_unique_id on rank 0)."""
        if self.world == 1:
            return data
        t = self.torch.zeros(n, dtype=self.torch.uint8, device=self.device)
        if self.rank == src:
            t.copy_(self.torch.frombuffer(bytearray(data), dtype=self.torch.uint8))
        self.dist.broadcast(t, src)
        return bytes(t.cpu().numpy().tobytes())

    def max_over_ranks(self, x: float) -> float:
        if self.world == 1:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def allreduce_mean(self, a: np.ndarray) -> np.ndarray:
        """host-array mean over ranks (tests and parameter-consistency checks; NOT the training data path)."""
        if self.world == 1:
            return a
        t = self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return (t / self.world).cpu().numpy()

    def attach_engine(self, pkg, engine, init_comm=True):
        """create the engine's RCCL communicator: rank 0 makes the id, everyone joins.  init_comm=False (bench.py's one-GPU self-test of the
        N > 1 script path, engines created under DQN_SIM_WORLD): the id is still made and broadcast, only ncclCommInitRank is skipped."""
        if self.world == 1:
            return
        uid = pkg.comm_unique_id() if self.rank == 0 else None
        uid = self.bcast_bytes(uid, 128, 0)
        if init_comm:
            engine.comm_init(uid, self.rank, self.world)

    def close(self):
        if self.world > 1 and self.dist.is_initialized():
            self.dist.destroy_process_group()
