"""Is the first timed dqn_train_steps(20) after the warm-up slower than the following ones?  (bench.py --steps 20 --warmup 5 is what the driver runs.)"""
import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import torch
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=False, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
eng, *_ = bench.build_workload(pkg, args, 0, 0)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 5
eng.train_steps(W)
torch.cuda.synchronize(); eng.sync()
for i in range(6):
    t0 = time.perf_counter(); eng.train_steps(20); eng.sync(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"warmup {W}: timed call {i}: {dt*1e6:.1f} us = {dt/20*1e6:.2f} us/step")
