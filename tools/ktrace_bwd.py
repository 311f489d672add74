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
args = argparse.Namespace(batch=int(os.environ.get("KT_BATCH", "32")), u8=bool(os.environ.get("KT_U8")), replay=int(os.environ.get("KT_REPLAY", "2000")), no_graph=True, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
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
rec = raw.astype(np.int64); rec[:, 0] &= 0xffffffff; rec[:, 1] &= 0xffffffff
roles = {0: "prio", 1: "dX", 2: "tail", 3: "dW"}
# forward records use words 3..7 as timestamps (large); backward records hold a role < 4 in word 3
sel = rec[:, 3] < 4
bw = rec[sel]; rt0 = rt0[sel]; rt1 = rt1[sel]
print("backward records", len(bw))
for grid in sorted(set(bw[:, 0])):
    g = bw[:, 0] == grid
    r = bw[g]; a0 = rt0[g]; a1 = rt1[g]
    t0 = r[:, 2].min(); T0 = a0.min()
    print(f"grid {grid}: {len(r)} workgroups; wall clock (s_memrealtime, 0.01 us ticks): first entry 0, last entry {(a0.max() - T0) / 100:.2f} us, last exit {(a1.max() - T0) / 100:.2f} us")
    for role in sorted(set(r[:, 3])):
        q = r[:, 3] == role
        e0 = (a0[q] - T0) / 100; e1 = (a1[q] - T0) / 100
        print(f"   {roles[int(role)]:5s} wall: entries {e0.min():6.2f} .. {e0.max():6.2f} (median {np.median(e0):6.2f}), exits {e1.min():6.2f} .. {e1.max():6.2f} (median {np.median(e1):6.2f}) us")
    if os.environ.get("KT_TIMELINE"):      # live workgroups per role every KT_TIMELINE us of the launch (common 100 MHz clock)
        dt = float(os.environ["KT_TIMELINE"]); span = (a1.max() - T0) / 100
        for t in np.arange(0.0, span, dt):
            live = [(int(role), int(((a0[r[:, 3] == role] - T0) / 100 <= t).sum() - ((a1[r[:, 3] == role] - T0) / 100 <= t).sum())) for role in sorted(set(r[:, 3]))]
            print(f"      t={t:6.1f} us  live: " + "  ".join(f"{roles[k]} {v}" for k, v in live))
    if os.environ.get("KT_GROUPS"):      # what decides a dW workgroup's lifetime: its XCD (blockIdx & 7), its position in the dispatch order, its row tile?
        q = r[:, 3] == 3
        if q.any():
            b = r[q, 1]; e0 = (a0[q] - T0) / 100; e1 = (a1[q] - T0) / 100; life = e1 - e0
            print("      dW wall lifetime by XCD (blockIdx & 7): " + "  ".join(f"{x}: {np.median(life[(b & 7) == x]):5.1f}/{life[(b & 7) == x].max():5.1f}" for x in range(8)))
            for m in (4, 8, 9):
                bb = b - b.min(); n = len(bb); qn, rn = n >> 3, n & 7; x = bb & 7; i = bb >> 3
                w = np.where(x < rn, x * (qn + 1), rn * (qn + 1) + (x - rn) * qn) + i
                print(f"      dW wall lifetime by (unit index mod {m}) (median/max): " + "  ".join(f"{k}: {np.median(life[w % m == k]):5.1f}/{life[w % m == k].max():5.1f}" for k in range(m)))
            o = np.argsort(e0); k = max(1, len(o) // 8)
            print("      dW wall lifetime by entry order (eighths; median entry -> median lifetime): " + "  ".join(f"{np.median(e0[o[j:j + k]]):5.1f}->{np.median(life[o[j:j + k]]):5.1f}" for j in range(0, len(o), k)))
    for role in sorted(set(r[:, 3])):
        q = r[r[:, 3] == role]
        life = (q[:, 7] - q[:, 2]) / TPU
        if int(role) == 0 and (q[:, 4] > 0).all():      # priority block (prio_block_fast): words 4..6 = tree top + indices in LDS, siblings / TD / leaves done, ancestors done
            print("   prio phases (us; large batches: leaves + duplicate scan / sparse levels / dense top / rest): %.2f, %.2f, %.2f, %.2f" % tuple(float(x[0]) / TPU for x in (q[:, 4] - q[:, 2], q[:, 5] - q[:, 4], q[:, 6] - q[:, 5], q[:, 7] - q[:, 6])))
        if int(role) == 1 and (q[:, 4] > 0).all():      # dX (dx_units_body): words 4..6 = first tile staged, K loop done, unit sums combined
            ph = [(q[:, 4] - q[:, 2]) / TPU, (q[:, 5] - q[:, 4]) / TPU, (q[:, 6] - q[:, 5]) / TPU, (q[:, 7] - q[:, 6]) / TPU]
            print("   dX phases of wave 0 (median / max us): first tile staged %.2f / %.2f, K loop %.2f / %.2f, wait for the other units %.2f / %.2f, combine + store %.2f / %.2f" % tuple(x for p_ in ph for x in (np.median(p_), p_.max())))
        print(f"   {roles[int(role)]:5s} n={len(q):5d}  first entry {(q[:, 2].min() - t0) / TPU:6.2f}  last entry {(q[:, 2].max() - t0) / TPU:6.2f}  last exit {(q[:, 7].max() - t0) / TPU:6.2f}"
              f"  lifetime median {np.median(life):6.2f}  p10 {np.percentile(life, 10):6.2f}  p90 {np.percentile(life, 90):6.2f}  max {life.max():6.2f} us")
