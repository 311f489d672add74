"""LONG SOAK (not a test: ~3 GPU-minutes): the in-launch hand-off of k_red_head (config 2 shape, B = 32) and k_head_cols4 (B = 128) against the two-launch / per-column schedules over
many more steps than tests/test_gpu_parity.py::test_fused_reduce_head_hand_off_soak: parameters, Adam state and priorities must be IDENTICAL at the end.
usage (GPU box): python tools/long_soak.py [steps_b32] [steps_b128]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nn = importlib.import_module(pkg.__name__ + ".nn")
net = nn.create_dueling_network(nn.nature_dqn(n_actions=4, in_channels=4))
layers, _ = nn.lower(net)
n32 = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
n128 = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
rng = np.random.default_rng(3)
S = rng.random((1024, 4, 84, 84), dtype=np.float32); A = rng.integers(0, 4, 1024).astype(np.int32); R = rng.standard_normal(1024).astype(np.float32); D = (rng.random(1024) < 0.1).astype(np.uint8)
p = nn.glorot_params(net, seed=1)
for B, steps, knob in ((32, n32, "DQN_NO_RED_HEAD"), (128, n128, "DQN_NO_HEAD_COLS4")):
    outs = []
    for off in (False, True):
        if off:
            os.environ[knob] = "1"
        hp = pkg.default_hparams(batch_size=B, n_actions=4, obs_c=4, obs_h=84, obs_w=84, learning_rate=1e-5, gamma=0.99, double_q=1, dueling=1, prioritized_replay=1, buffer_size=1024, seed=1)
        eng = pkg.Engine(layers, hp)
        os.environ.pop(knob, None)
        eng.set_params(p, pkg.NET_ONLINE); eng.set_params(p * np.float32(0.97), pkg.NET_TARGET)
        eng.replay_add(S, A, R, S[::-1].copy(), D)
        t0 = time.perf_counter(); done = 0
        for chunk in (1, 999, steps - 1000):
            loss, gn = eng.train_steps(chunk); done += chunk
            assert np.isfinite(loss) and np.isfinite(gn), (loss, gn)
        dt = time.perf_counter() - t0
        names = [n for n, _ in eng.profile_step()]
        outs.append((eng.get_params(0), eng.get_adam_state(), eng.replay_priorities(), (loss, gn)))
        print(f"B = {B}, {knob}={'1' if off else '0'}: {done} steps in {dt:.1f} s ({done / dt:.0f} steps/s), loss {loss:.6f}, head launch: {[n for n in names if 'head' in n]}", flush=True)
        eng.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    for x, y in zip(outs[0][1], outs[1][1]):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    assert outs[0][3] == outs[1][3]
    print(f"B = {B}: both schedules IDENTICAL after {steps} steps (parameters, Adam m / v / beta powers, priorities, last loss and grad norm)", flush=True)
