import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=False, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
eng, *_ = bench.build_workload(pkg, args, 0, 0)
eng.train_steps(50); eng.sync()
def t(n, reps=20):
    best = []
    for _ in range(reps):
        eng.sync(); t0 = time.perf_counter(); eng.train_steps(n); eng.sync(); best.append(time.perf_counter() - t0)
    best.sort(); return best[len(best)//2] * 1e6
for n in (1, 2, 5, 20, 100):
    us = t(n); print(f"train_steps({n}): {us:8.1f} us  = {us/n:7.1f} us/step, fixed over 150.8*n: {us - 150.8*n:7.1f} us")
