"""PROBE: one field of the summation plan at a time on config 2 (train_steps rate + the per-launch table).  The plan only fixes rounding order, every case is a
valid engine.  usage (GPU box): PLAN_CASES="1.dx=4,2.dx=1,1.dx=4+2.dx=1" python tools/plan_probe.py     (layer.field=value; fields fwd, dx, dw; + joins edits of one case)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402
pkg = ge.load_package(); nn = importlib.import_module(pkg.__name__ + ".nn")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4)); layers, _ = nn.lower(net)
B = int(os.environ.get("PLAN_B", "32"))
hp = pkg.default_hparams(batch_size=B, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-4, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=2000, seed=1)
rng = np.random.default_rng(0)
S_ = rng.random((2048, 4, 84, 84), dtype=np.float32); A_ = rng.integers(0, 4, 2048).astype(np.int32); R_ = rng.standard_normal(2048).astype(np.float32); D_ = np.zeros(2048, np.uint8)
p = nn.glorot_params(net, seed=1)
base = pkg.default_plan(layers, hp)
print("default plan (fwd_kc, dx_kc, dw_kc):", base)
F = {"fwd": 0, "dx": 1, "dw": 2}
cases = [("default", [])] + [(c, [(int(x.split(".")[0]), F[x.split(".")[1].split("=")[0]], int(x.split("=")[1])) for x in c.split("+")]) for c in os.environ.get("PLAN_CASES", "").split(",") if c] + [("default", [])]
for rep in range(int(os.environ.get("PLAN_REPS", "2"))):
    for name, edits in cases:
        plan = [list(q) for q in base]
        for l, f, v in edits: plan[l][f] = v
        plan = [tuple(q) for q in plan]
        try:
            eng = pkg.Engine(layers, hp, plan=plan)
        except Exception as ex:
            print(f"{name:22s} refused: {str(ex)[:100]}"); continue
        eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p, pkg.NET_TARGET); eng.replay_add(S_, A_, R_, S_, D_)
        eng.train_steps(300); eng.sync()
        acc = {}
        for _ in range(30):
            for n, ms in eng.profile_step(steady=True): acc.setdefault(n, []).append(ms * 1e3)
        eng.train_steps(100); eng.sync()
        t0 = time.perf_counter(); eng.train_steps(3000); eng.sync(); dt = time.perf_counter() - t0
        print(f"{name:22s} {3000 / dt:8.1f} steps/s  {len(acc):2d} launches  " + "  ".join(f"{n.split('+')[0][:12]} {np.median(acc[n]):.1f}" for n in acc))
        eng.close()
