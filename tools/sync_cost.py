import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=False, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
eng, *_ = bench.build_workload(pkg, args, 0, 0)
eng.train_steps(50); eng.sync(); torch.cuda.synchronize()
def med(f, reps=30):
    v = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); v.append(time.perf_counter() - t0)
    v.sort(); return v[len(v)//2] * 1e6
print("idle torch.cuda.synchronize: %.1f us" % med(torch.cuda.synchronize))
print("idle eng.sync: %.1f us" % med(eng.sync))
def a():
    eng.train_steps(20); eng.sync()
def b():
    eng.train_steps(20); eng.sync(); torch.cuda.synchronize()
def c():
    eng.train_steps(20); torch.cuda.synchronize()
for nm, f in (("steps+eng.sync", a), ("steps+eng.sync+torch.sync", b), ("steps+torch.sync", c), ("steps+eng.sync", a)):
    print(nm, "%.1f us" % med(f, 15))
import time as _t
for gap in (0.0, 0.0005, 0.002, 0.01, 0.05):
    v = []
    for _ in range(9):
        eng.sync(); torch.cuda.synchronize(); _t.sleep(gap)
        t0 = _t.perf_counter(); eng.train_steps(20); eng.sync(); torch.cuda.synchronize(); v.append(_t.perf_counter() - t0)
    v.sort(); print(f"idle gap {gap*1e3:.1f} ms before the timed 20 steps: {v[len(v)//2]*1e6:.1f} us (min {v[0]*1e6:.1f})")
