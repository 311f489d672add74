import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=True, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
eng, *_ = bench.build_workload(pkg, args, 0, 0)
eng.train_steps(300); eng.sync()
t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
print(f"u8 replay, B=32: {dt / 3000 * 1e6:7.1f} us/step  (DQN_NO_PREGATHER={os.environ.get('DQN_NO_PREGATHER')})")
