#!/usr/bin/env python3
"""Per-workgroup timeline of the backward GEMM launches (dqn_debug_ktrace; k_dwdx_lds / k_dw_lds records carry a role in word 3: 0 priority
block, 1 dX, 2 tail task, 3 dW): per launch (grid size) and role the entry spread, the lifetime distribution and the launch span, in us
(s_memtime ticks / TICKS_PER_US; 100 ticks per us on gfx950's constant 100 MHz counter)."""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
import importlib
import argparse
TPU = float(os.environ.get("TICKS_PER_US", "100"))
args = argparse.Namespace(batch=int(os.environ.get("KT_BATCH", "32")), u8=False, replay=2000, no_graph=True, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
pkg = ge.load_package()
pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
eng, *_ = bench.build_workload(pkg, args, 0, 0)
for _ in range(5):
    eng.train_step()
f = pkg.fns()["debug_ktrace"]
assert f(eng._h, None, 0) == 0
eng.train_step()
n = 1 + 8 * 65536
buf = np.zeros(n, np.uint64)
assert f(eng._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n) == 0
cnt = int(buf[0]); rec = buf[1:1 + 8 * cnt].reshape(cnt, 8).astype(np.int64)
roles = {0: "prio", 1: "dX", 2: "tail", 3: "dW"}
# forward records use words 3..7 as timestamps (large); backward records hold a role < 4 in word 3
bw = rec[rec[:, 3] < 4]
print("backward records", len(bw))
for grid in sorted(set(bw[:, 0])):
    r = bw[bw[:, 0] == grid]
    t0 = r[:, 2].min()
    print(f"grid {grid}: {len(r)} workgroups, span {(r[:, 7].max() - t0) / TPU:.2f} us, entry spread {(r[:, 2].max() - t0) / TPU:.2f} us")
    for role in sorted(set(r[:, 3])):
        q = r[r[:, 3] == role]
        life = (q[:, 7] - q[:, 2]) / TPU
        print(f"   {roles[int(role)]:5s} n={len(q):5d}  first entry {(q[:, 2].min() - t0) / TPU:6.2f}  last entry {(q[:, 2].max() - t0) / TPU:6.2f}  last exit {(q[:, 7].max() - t0) / TPU:6.2f}"
              f"  lifetime median {np.median(life):6.2f}  p10 {np.percentile(life, 10):6.2f}  p90 {np.percentile(life, 90):6.2f}  max {life.max():6.2f} us")
