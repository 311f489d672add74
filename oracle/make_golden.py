"""
oracle/make_golden.py -- TEST INFRASTRUCTURE.  Run in the BUILD container only:

    python oracle/make_golden.py

Pins the NumPy oracle (oracle/dqn_oracle.py) with an INDEPENDENT torch-CPU
float64 autograd implementation of the same reference math
(src/solver.jl:191-236, src/dueling.jl:8-11, src/helpers.jl:14-19) and writes
small fixtures (inputs + expected outputs) to tests/golden/*.npz.  The
reference itself (Julia/Flux) cannot run in this image, so torch autograd is
the strongest independent check available; the fixtures travel to the GPU box,
torch-as-oracle does not need to.

The torch model below shares NO code with dqn_oracle.py: conv via
torch.nn.functional.conv2d on the flipped kernel (true convolution, NNlib
default), gradients via autograd, Adam via torch.optim.Adam.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import dqn_oracle as O  # noqa: E402

GOLD = os.path.join(HERE, "..", "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_default_dtype(torch.float64)
torch.set_num_threads(8)


def torch_act(y, act):
    return [lambda t: t, torch.relu, torch.tanh, torch.sigmoid][act](y)


def torch_chain(layers, ps, x):
    for i, l in enumerate(layers):
        W, b = ps[2 * i], ps[2 * i + 1]
        if l.kind == "dense":
            x = torch_act(x.reshape(x.shape[0], -1) @ W + b, l.act)
        else:
            x = torch_act(F.conv2d(x, W.flip(2, 3), b, stride=(l.sh, l.sw)), l.act)
    return x


def torch_q(net, ps, x):
    nb = 2 * len(net.base)
    xb = torch_chain(net.base, ps[:nb], x)
    if not net.dueling:
        return xb
    nv = 2 * len(net.val)
    v = torch_chain(net.val, ps[nb:nb + nv], xb)
    a = torch_chain(net.adv, ps[nb + nv:], xb)
    a2 = torch_chain(net.adv, ps[nb + nv:], xb)  # evaluated twice, dueling.jl:10
    return v + a - a2.mean(dim=1, keepdim=True)


def torch_huber(x):
    ab = x.abs()
    q = torch.minimum(ab, torch.ones_like(ab))
    return 0.5 * q * q + (ab - q)


def torch_step(net, p_on, p_tg, batch, gamma, double_q, lr):
    s, a, r, sp, done, w = [torch.tensor(np.asarray(t)) for t in batch]
    s, sp, r, done, w = s.double(), sp.double(), r.double(), done.double(), w.double()
    a = a.long()
    pon = [torch.tensor(np.asarray(p, np.float64), requires_grad=True) for p in p_on]
    ptg = [torch.tensor(np.asarray(p, np.float64)) for p in p_tg]
    B = s.shape[0]
    with torch.no_grad():
        q_tg_sp = torch_q(net, ptg, sp)
        if double_q:
            q_on_sp = torch_q(net, pon, sp)
            best = q_on_sp.argmax(dim=1)
            qmax = q_tg_sp[torch.arange(B), best]
        else:
            qmax = q_tg_sp.max(dim=1).values
        y = r + (1 - done) * gamma * qmax
    q = torch_q(net, pon, s)
    td = q[torch.arange(B), a] - y
    loss = torch_huber(w * td).sum() / B
    opt = torch.optim.Adam(pon, lr=float(np.float32(lr)), betas=(0.9, 0.999), eps=1e-8)
    loss.backward()
    grads = [p.grad.detach().numpy().copy() for p in pon]
    opt.step()
    return dict(loss=loss.item(), td=td.detach().numpy(), q=q.detach().numpy(), grads=grads,
                new_params=[p.detach().numpy().copy() for p in pon])


def make_case(name, net, B, seed, gamma, double_q, lr=1e-4, store_params=True, obs_scale=1.0):
    rng = np.random.default_rng(seed)
    p_on = O.init_params(net, seed=seed + 1)
    p_tg = O.init_params(net, seed=seed + 2)
    # non-zero biases so bias paths are exercised
    for i in range(1, len(p_on), 2):
        p_on[i] = (0.1 * rng.standard_normal(p_on[i].shape)).astype(np.float32)
        p_tg[i] = (0.1 * rng.standard_normal(p_tg[i].shape)).astype(np.float32)
    s = (obs_scale * rng.random((B,) + net.obs_shape)).astype(np.float32)
    sp = (obs_scale * rng.random((B,) + net.obs_shape)).astype(np.float32)
    a = rng.integers(0, net.n_actions, B).astype(np.int32)
    r = rng.standard_normal(B).astype(np.float32) * 3  # large enough to hit the linear Huber branch
    done = (rng.random(B) < 0.25).astype(np.float32)
    rng.random(B)   # (round <= 3 drew random IS weights here; the draw is kept so that every later draw -- there is none -- and the seeds stay put)
    # IS weights as get_batch computes them (src/prioritized_experience_replay.jl:93-102) for a replay that holds exactly these B transitions,
    # each added the way dqn_train! adds them: add_exp!(replay, exp, abs(exp.r)) (src/solver.jl:91-94) -> priority (|r| + eps)^alpha
    # (...replay.jl:67, eps = 1e-3, alpha = 0.6, beta = 0.4: :43-45).  An engine fed the same way returns the SAME w from get_batch(0..B-1),
    # so its loss / td / gradients / Adam step can be compared with the torch values below directly.
    prio = np.array([np.float32(float(np.float32(abs(float(x))) + np.float32(1e-3)) ** float(np.float32(0.6))) for x in r], np.float32)   # Float32^Float32 through Float64
    tot = np.float32(0)
    for q in prio:
        tot = np.float32(tot + q)
    w = np.array([np.float32(float(np.float32(B) * np.float32(q / tot)) ** (-float(np.float32(0.4)))) for q in prio], np.float32)
    np.testing.assert_array_equal(prio, O.priority_from_td(np.abs(r), np.float32(1e-3), np.float32(0.6)))
    np.testing.assert_allclose(w, O.is_weights(prio, prio, 0.4), rtol=3e-7)
    assert w.max() / w.min() > 1.5, "IS weights too uniform to pin anything"
    batch = (s, a, r, sp, done, w)

    adam = O.AdamState([np.asarray(p, np.float64) for p in p_on], lr)
    o = O.batch_train_step(net, p_on, p_tg, batch, gamma=gamma, double_q=double_q, adam=adam, dtype=np.float64)
    t = torch_step(net, p_on, p_tg, batch, gamma, double_q, lr)

    def rel(x, y):
        return float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / (1e-300 + np.max(np.abs(np.asarray(y)))))

    errs = dict(loss=rel(o["loss"], t["loss"]), td=rel(o["td"], t["td"]), q=rel(o["q"], t["q"]),
                grads=max(rel(g, h) for g, h in zip(o["grads"], t["grads"])),
                new_params=max(rel(g, h) for g, h in zip(o["new_params"], t["new_params"])))
    print(f"[{name}] oracle-vs-torch-autograd relative errors: {errs}")
    assert max(errs.values()) < 1e-9, errs

    out = dict(B=B, seed=seed, gamma=gamma, double_q=int(double_q), lr=lr,
               s=s, a=a, r=r, sp=sp, done=done, w=w, prio=prio, w_from_priorities=1,
               loss=np.float64(t["loss"]), td=t["td"], q=t["q"], grad_norm=np.float64(max(np.abs(g).max() for g in t["grads"])),
               grad_sums=np.array([g.sum() for g in t["grads"]]),
               grad_abs_sums=np.array([np.abs(g).sum() for g in t["grads"]]),
               newp_sums=np.array([p.sum() for p in t["new_params"]]))
    if store_params:
        out["p_on"] = O.Network.flatten(p_on)
        out["p_tg"] = O.Network.flatten(p_tg)
        out["grads"] = O.Network.flatten(t["grads"])
        out["new_params"] = O.Network.flatten(t["new_params"])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)


def nets():
    R, I, T = O.ACT_RELU, O.ACT_IDENTITY, O.ACT_TANH
    # config 1: README.md:38 Chain(Dense(2,32), Dense(32,4)) + dueling (nothing shared)
    mlp = [O.Dense(2, 32, I), O.Dense(32, 4, I)]
    b, v, a = O.create_dueling_network(mlp)
    yield "cfg1_gridworld_mlp_dueling", O.Network((2,), b, v, a), 32, 11, 0.95, True, True, 10.0
    # test/runtests.jl:45-61 style: flattenbatch, Dense(100,8,tanh), Dense(8,4); plain DQN
    yield "testmdp_mlp_tanh_plain", O.Network((4, 5, 5), [O.Dense(100, 8, T), O.Dense(8, 4, I)]), 32, 12, 0.99, False, True, 1.0
    # shrunk conv dueling net (exercises stride>1, true-convolution flip, rectangular maps)
    conv = [O.Conv(4, 3, 8, R, 2), O.Conv(3, 8, 16, R, 1), O.Dense(16 * 3 * 4, 32, R), O.Dense(32, 5, I)]
    b, v, a = O.create_dueling_network(conv)
    yield "small_conv_dueling", O.Network((3, 12, 14), b, v, a), 16, 13, 0.99, True, True, 1.0
    yield "small_conv_plain_single_q", O.Network((3, 12, 14), conv), 16, 14, 0.9, False, True, 1.0
    # config 2: Nature-DQN dueling on 84x84x4, B=4 (params regenerated from the seed in tests)
    nat = [O.Conv(8, 4, 32, R, 4), O.Conv(4, 32, 64, R, 2), O.Conv(3, 64, 64, R, 1),
           O.Dense(3136, 512, R), O.Dense(512, 4, I)]
    b, v, a = O.create_dueling_network(nat)
    yield "cfg2_nature_dueling_b4", O.Network((4, 84, 84), b, v, a), 4, 15, 0.99, True, False, 1.0


if __name__ == "__main__":
    for name, net, B, seed, gamma, dq, store, scale in nets():
        make_case(name, net, B, seed, gamma, dq, store_params=store, obs_scale=scale)
    print("fixtures written to", os.path.abspath(GOLD))


# ------------------------------------------------------------------ DRQN (src/solver.jl:239-287) cross-check
def torch_lstm_seq(net, ps, xs):
    sl = net.param_slices()
    B = xs[0].shape[0]
    state = {}
    for li, l in enumerate(net.base):
        if l.kind == "lstm":
            Wi, Wh, b, h0, c0 = ps[sl[li][0]:sl[li][1]]
            state[li] = (h0[None].expand(B, -1), c0[None].expand(B, -1))
    qs = []
    nb = len(net.base)
    for x in xs:
        for li, l in enumerate(net.base):
            a, b_ = sl[li]
            if l.kind == "lstm":
                Wi, Wh, b, h0, c0 = ps[a:b_]
                hp, cp = state[li]
                g = x.reshape(B, -1) @ Wi + hp @ Wh + b
                H = l.n_out
                i, f, gc, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
                c = f * cp + i * gc
                h = o * torch.tanh(c)
                state[li] = (h, c)
                x = h
            else:
                x = torch_chain([l], ps[a:b_], x)
        if net.dueling:
            nv = len(net.val)
            pv = [p for (a, b_) in sl[nb:nb + nv] for p in ps[a:b_]]
            pa = [p for (a, b_) in sl[nb + nv:] for p in ps[a:b_]]
            v = torch_chain(net.val, pv, x)
            ad = torch_chain(net.adv, pa, x)
            x = v + ad - ad.mean(dim=1, keepdim=True)
        qs.append(x)
    return qs


def make_drqn_case(name, net, B, T, seed, gamma, double_q, lr=1e-3):
    rng = np.random.default_rng(seed)
    p_on = O.init_params_recurrent(net, seed + 1)
    p_tg = O.init_params_recurrent(net, seed + 2)
    p_on = [(p + 0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in p_on]   # non-zero biases / initial states
    p_tg = [(p + 0.05 * rng.standard_normal(p.shape)).astype(np.float32) for p in p_tg]
    s = [rng.random((B,) + net.obs_shape).astype(np.float32) for _ in range(T)]
    sp = [rng.random((B,) + net.obs_shape).astype(np.float32) for _ in range(T)]
    a = [rng.integers(0, net.n_actions, B).astype(np.int32) for _ in range(T)]
    r = [(3 * rng.standard_normal(B)).astype(np.float32) for _ in range(T)]
    d = [(rng.random(B) < 0.2).astype(np.float32) for _ in range(T)]
    lens = rng.integers(0, T + 1, B)
    m = [(t < lens).astype(np.int32) for t in range(T)]
    batch = (s, a, r, sp, d, m)
    adam = O.AdamState([np.asarray(p, np.float64) for p in p_on], lr)
    o = O.drqn_train_step(net, p_on, p_tg, batch, gamma=gamma, double_q=double_q, adam=adam)
    # torch
    pon = [torch.tensor(np.asarray(p, np.float64), requires_grad=True) for p in p_on]
    ptg = [torch.tensor(np.asarray(p, np.float64)) for p in p_tg]
    ts = [torch.tensor(x).double() for x in s]
    tsp = [torch.tensor(x).double() for x in sp]
    with torch.no_grad():
        qt = torch_lstm_seq(net, ptg, tsp)
        qo = torch_lstm_seq(net, pon, tsp) if double_q else qt
        ys = []
        for t in range(T):
            best = qo[t].argmax(dim=1)
            qmax = qt[t][torch.arange(B), best] if double_q else qt[t].max(dim=1).values
            ys.append(torch.tensor(r[t]).double() + (1 - torch.tensor(d[t]).double()) * gamma * qmax)
    qs = torch_lstm_seq(net, pon, ts)
    loss = 0.0
    for t in range(T):
        td = qs[t][torch.arange(B), torch.tensor(a[t]).long()] - ys[t]
        loss = loss + torch_huber(torch.tensor(m[t]).double() * td).sum() / B
    loss = loss / T
    opt = torch.optim.Adam(pon, lr=float(np.float32(lr)), betas=(0.9, 0.999), eps=1e-8)
    loss.backward()
    tg = [p.grad.detach().numpy().copy() for p in pon]
    opt.step()

    def rel(x, y):
        return float(np.max(np.abs(np.asarray(x) - np.asarray(y))) / (1e-300 + np.max(np.abs(np.asarray(y)))))
    errs = dict(loss=rel(o["loss"], loss.item()), grads=max(rel(g, h) for g, h in zip(o["grads"], tg)),
                q=rel(o["q"], np.stack([q.detach().numpy() for q in qs])),
                new_params=max(rel(g, h.detach().numpy()) for g, h in zip(o["new_params"], pon)))
    print(f"[{name}] DRQN oracle-vs-torch-autograd relative errors: {errs}")
    assert max(errs.values()) < 1e-9, errs
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), B=B, T=T, seed=seed, gamma=gamma, double_q=int(double_q), lr=lr,
                        s=np.stack(s), a=np.stack(a), r=np.stack(r), sp=np.stack(sp), done=np.stack(d), mask=np.stack(m),
                        p_on=O.Network.flatten(p_on), p_tg=O.Network.flatten(p_tg), loss=np.float64(loss.item()),
                        q=np.stack([q.detach().numpy() for q in qs]), grads=O.Network.flatten(tg),
                        new_params=O.Network.flatten([p.detach().numpy() for p in pon]))


def drqn_nets():
    I = O.ACT_IDENTITY
    # BASELINE config 4 shape: TestMDP((5,5),1,6) obs 25, Chain(flattenbatch, LSTM(25,32), Dense(32,4)), trace_length 8 (benchmark/flux_dqn.jl:35-36)
    yield "drqn_cfg4_lstm_plain", O.RecurrentNetwork((1, 5, 5), [O.LSTM(25, 32), O.Dense(32, 4, I)]), 8, 8, 21, 0.99, True
    # test/runtests.jl:131-147 style: LSTM + dueling (base = LSTM, val = Dense(h,1), adv = Dense(h,nA)), Dense in front
    base = [O.Dense(6, 12, O.ACT_RELU), O.LSTM(12, 16)]
    yield "drqn_dense_lstm_dueling", O.RecurrentNetwork((6,), base, [O.Dense(16, 1, I)], [O.Dense(16, 5, I)]), 6, 5, 22, 0.95, True
    yield "drqn_lstm_single_q", O.RecurrentNetwork((6,), [O.LSTM(6, 8), O.Dense(8, 3, I)]), 4, 3, 23, 0.9, False


if __name__ == "__main__":
    for name, net, B, T, seed, gamma, dq in drqn_nets():
        make_drqn_case(name, net, B, T, seed, gamma, dq)
