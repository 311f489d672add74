#!/usr/bin/env python3
"""Per-workgroup timeline of the forward GEMM kernels (dqn_debug_ktrace): one eager train step of the bench workload, then per launch (grouped
by grid size) the distribution of phase durations in s_memtime ticks (100 MHz => 10 ns per tick) and the launch's first-entry-to-last-exit span."""
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
cnt = int(buf[0]); raw = buf[1:1 + 8 * cnt].reshape(cnt, 8)
rec = raw.astype(np.int64); rec[:, 0] &= 0xffffffff; rec[:, 1] &= 0xffffffff      # words 0 / 1 carry s_memrealtime in their upper halves (r03); low halves: grid size, block id
rec = rec[rec[:, 3] > 16]                                                            # forward records only: the backward kernels put a ROLE (0..3) in word 3 (tools/ktrace_bwd.py reads those)
print("records", cnt, "forward records", len(rec))
names = ["entry->tables", "tables->tile0", "tile0->loop end", "loop end->combined", "combined->stores"]
for grid in sorted(set(rec[:, 0])):
    r = rec[rec[:, 0] == grid]
    t = r[:, 2:8]
    t0 = t[:, 0].min()
    print(f"grid {grid}: {len(r)} workgroups, span {(t[:, 5].max() - t0) / 100:.2f} us; entry spread {(t[:, 0].max() - t0) / 100:.2f} us; last exit of the first-dispatched 256: {(t[np.argsort(t[:, 0])[:256], 5].max() - t0) / 100:.2f} us")
    for i, nm in enumerate(names):
        d = (t[:, i + 1] - t[:, i]) / 100.0
        print(f"   {nm:22s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    d = (t[:, 5] - t[:, 0]) / 100.0
    print(f"   {'workgroup lifetime':22s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
