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
args = argparse.Namespace(batch=int(os.environ.get("KT_BATCH", "32")), u8=bool(os.environ.get("KT_U8")), replay=2000, no_graph=True, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
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
rt0 = (raw[:, 0] >> np.uint64(32)).astype(np.int64); rt1 = (raw[:, 1] >> np.uint64(32)).astype(np.int64)      # s_memrealtime (100 MHz, common to all XCDs) at entry / exit
rec = raw.astype(np.int64); rec[:, 0] &= 0xffffffff; rec[:, 1] &= 0xffffffff      # words 0 / 1 carry s_memrealtime in their upper halves (r03); low halves: grid size, block id
rt0 = rt0[rec[:, 3] > 16]; rt1 = rt1[rec[:, 3] > 16]
rec = rec[rec[:, 3] > 16]                                                            # forward records only: the backward kernels put a ROLE (0..3) in word 3 (tools/ktrace_bwd.py reads those)
print("records", cnt, "forward records", len(rec))
names = ["entry->tables", "tables->tile0", "tile0->loop end", "loop end->combined", "combined->stores"]
for grid in sorted(set(rec[:, 0])):
    r = rec[rec[:, 0] == grid]
    if os.environ.get("KT_TIMELINE"):      # live workgroups every KT_TIMELINE us of the launch, on the common wall clock
        a0 = rt0[rec[:, 0] == grid]; a1 = rt1[rec[:, 0] == grid]; T0 = a0.min(); dt = float(os.environ["KT_TIMELINE"])
        life = (a1 - a0) / 100.0
        print(f"grid {grid}: wall clock: last entry {(a0.max() - T0) / 100:.2f} us, last exit {(a1.max() - T0) / 100:.2f} us; lifetime median {np.median(life):.2f} p10 {np.percentile(life, 10):.2f} p90 {np.percentile(life, 90):.2f} max {life.max():.2f}")
        print("   live: " + "  ".join(f"{t_:.0f}:{int(((a0 - T0) / 100 <= t_).sum() - ((a1 - T0) / 100 <= t_).sum())}" for t_ in np.arange(0.0, (a1.max() - T0) / 100, dt)))
        o = np.argsort(a0); k = max(1, len(o) // 8)
        print("   by entry order (eighths; median entry -> median lifetime): " + "  ".join(f"{np.median((a0[o[j:j + k]] - T0) / 100):5.1f}->{np.median(life[o[j:j + k]]):5.1f}" for j in range(0, len(o), k)))
    t = r[:, 2:8]
    t0 = t[:, 0].min()
    print(f"grid {grid}: {len(r)} workgroups, span {(t[:, 5].max() - t0) / 100:.2f} us; entry spread {(t[:, 0].max() - t0) / 100:.2f} us; last exit of the first-dispatched 256: {(t[np.argsort(t[:, 0])[:256], 5].max() - t0) / 100:.2f} us")
    for i, nm in enumerate(names):
        d = (t[:, i + 1] - t[:, i]) / 100.0
        print(f"   {nm:22s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    d = (t[:, 5] - t[:, 0]) / 100.0
    print(f"   {'workgroup lifetime':22s} median {np.median(d):6.2f}  p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
