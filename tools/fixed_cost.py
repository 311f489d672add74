"""Fixed cost of one dqn_train_steps(n) call on config 2 (the driver times --steps 20: a 3 ms region, so ~100 us of call overhead is 3-4 %).
Median wall time of train_steps(n) + sync for several n, a least-squares line through them (slope = us/step, intercept = per-call cost), and
the split of one n = 20 call into host enqueue time (the call returns after fetching the scalars) and the trailing sync."""
import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=False, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
eng, *_ = bench.build_workload(pkg, args, 0, 0)
eng.train_steps(200); eng.sync()
def t(n, reps=30):
    v = []
    for _ in range(reps):
        eng.sync(); t0 = time.perf_counter(); eng.train_steps(n); t1 = time.perf_counter(); eng.sync(); v.append((time.perf_counter() - t0, t1 - t0))
    v.sort(); return v[len(v) // 2][0] * 1e6, sorted(x[1] for x in v)[len(v) // 2] * 1e6
ns = (1, 2, 3, 5, 6, 9, 10, 20, 21, 40, 100, 400)
us = [t(n) for n in ns]
for n, (u, c) in zip(ns, us): print(f"train_steps({n:3d}): {u:9.1f} us = {u / n:7.1f} us/step   (call returned after {c:9.1f} us)")
A = np.vstack([np.array(ns[6:], float), np.ones(len(ns) - 6)]).T
slope, icpt = np.linalg.lstsq(A, np.array([u for u, _ in us[6:]]), rcond=None)[0]
print(f"fit over n >= 10: {slope:.2f} us/step + {icpt:.1f} us per call   (HSA_ENABLE_INTERRUPT={os.environ.get('HSA_ENABLE_INTERRUPT', 'unset')})")
# where the per-call cost sits: the same calls without fetching the scalars (null loss / grad_norm pointers: no fold launch, no D2H copy, no sync inside)
import ctypes as C
def t0(n, reps=30):
    v = []
    for _ in range(reps):
        eng.sync(); a = time.perf_counter(); eng._check(eng.f["train_steps"](eng._h, n, None, None)); eng.sync(); v.append(time.perf_counter() - a)
    v.sort(); return v[len(v) // 2] * 1e6
us0 = [t0(n) for n in ns[6:]]
s0, i0 = np.linalg.lstsq(A, np.array(us0), rcond=None)[0]
print(f"without the scalar fetch: {s0:.2f} us/step + {i0:.1f} us per call")
