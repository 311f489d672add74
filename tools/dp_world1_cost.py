"""(env DQN_DP_OVERLAP=1: exchange on its own stream beside the conv backward; DQN_DP_NO_ONE_GRAPH=1: collectives enqueued eagerly between graph halves)
Compute side of the data-parallel step on ONE GPU: the engine's RCCL path forced on at world size 1 (DQN_FORCE_ALLREDUCE=1: step graph cut in
two around the collective, k_dp_pack, the wide dW over the gathered operands, Adam on the exchanged gradient) against the plain single-GPU step."""
import time, sys, os, importlib, argparse
sys.path.insert(0, os.getcwd())
import bench, __graft_entry__ as ge
pkg = ge.load_package(); pkg.nn = importlib.import_module(pkg.__name__ + ".nn"); pkg.envs = importlib.import_module(pkg.__name__ + ".envs")
args = argparse.Namespace(batch=32, u8=False, replay=10000, no_graph=False, no_mfma=False, conv_kc=0, fc_kc=0, envs_per_rank=32, device_fill=False)
def run(forced):
    if forced: os.environ["DQN_FORCE_ALLREDUCE"] = "1"
    else: os.environ.pop("DQN_FORCE_ALLREDUCE", None)
    eng, *_ = bench.build_workload(pkg, args, 0, 0)
    if forced: eng.comm_init(pkg.comm_unique_id(), 0, 1)
    eng.train_steps(300); eng.sync()
    t0 = time.perf_counter(); eng.train_steps(2000); eng.sync(); dt = time.perf_counter() - t0
    print(("forced RCCL path, world 1" if forced else "plain single-GPU step   "), f"{dt / 2000 * 1e6:7.1f} us/step")
    if "--profile" in sys.argv:
        for name, ms in eng.profile_step(): print(f"    {name:34s} {ms * 1e3:7.2f}")
run(False); run(True)
