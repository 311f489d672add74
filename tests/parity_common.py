"""Shared parity-test bodies (TEST INFRASTRUCTURE): each takes an engine factory, so the very same assertions run against the CPU twin
(`-m "not gpu"`) and against the HIP engine through the C ABI (`-m gpu`).

  * hand_derived_known_answer -- ONE batch_train! whose every number was worked out by hand from the cited reference lines (no oracle
    code involved): pins the COMPOSITION order that a misreading would get wrong -- huber(w*td) not w*huber(td) (src/solver.jl:223),
    the UNWEIGHTED td into update_priorities! (:231-233), first-max argmax ties (:212), done => y = r (:217), Adam's first step.
  * sampler_distribution -- inclusion frequencies of the sum-tree sampler against p / sum(p) (…replay.jl:85 draws with probability
    proportional to priority).
"""
import numpy as np

import ref

abi = ref.abi


def _dense_layers(n_in, n_out):
    d = abi.LayerDesc()
    d.kind, d.act, d.stream, d.n_in, d.n_out = abi.LAYER_DENSE, abi.ACT_IDENTITY, abi.STREAM_BASE, n_in, n_out
    return [d]


# --------------------------------------------------------------------------------------------------------------------
# Hand-derived known answer.  Net = Chain(Dense(2, 2)) (no dueling), double-Q, prioritized, B = 2, gamma = 0.5,
# alpha = 1, beta = 1, eps = 0.5, Adam eta = 0.25.
#   online  W = [1 2; 3 -1] (out, in), b = [0, 1]          target W = [2 0; 0 1], b = [1, 0]
#   replay (4 transitions; priorities p = td_err + eps = [1, 4, 2, 1], sum 8, n = 4):
#     0: s = (1, 0), a = 1 (1-based; 0 at the ABI), r = 1,  sp = (0, 1), done = false
#     1: s = (0, 2), a = 2 (1 at the ABI),          r = -2, sp = (1, 1), done = true
#   batch = indices [0, 1]:
#     IS weights (…replay.jl:101-102)  w = (n * p / sum)^-beta = (4*1/8)^-1, (4*4/8)^-1 = [2, 0.5]
#     online Q(s)   = W s + b = [1, 4], [4, -1]           -> Q[a]  = [1, -1]
#     online Q(sp)  = [2, 0], [3, 3]                       -> best  = [1st, 1st]   (tie in column 2: Julia argmax takes the FIRST max, :212)
#     target Q(sp)  = [1, 1], [3, 1]                       -> q_sp_max = [1, 3]
#     y  = r + (1 - done) * gamma * q_sp_max = [1 + 0.5*1, -2 + 0] = [1.5, -2]                       (:217)
#     td = Q[a] - y = [-0.5, 1]                                                                     (:222)
#     x  = w * td = [-1, 0.5];  huber(-1) = 0.5, huber(0.5) = 0.125;  loss = (0.5 + 0.125) / 2 = 0.3125   (:223-224, helpers.jl:14-19)
#        (the misreading w * huber(td) would give (2*0.125 + 0.5*0.5) / 2 = 0.25)
#     dL/dQ[a_b, b] = w * clamp(w*td, -1, 1) / B = [2 * -1 / 2, 0.5 * 0.5 / 2] = [-1, 0.125]
#     dW = dQ * s' = [-1 0; 0 0.25],  db = [-1, 0.125],  grad_norm = max|g| = 1                      (helpers.jl:38-46)
#     Adam, first step: m = 0.1 g, v = 0.001 g^2, update = eta * (m/0.1) / (sqrt(v/0.001) + 1e-8) = eta * g / (|g| + 1e-8)
#        -> W' = [1.25 2; 3 -1.25], b' = [0.25, 0.75]   (to ~1e-8 relative; entries with g = 0 do not move)
#     update_priorities!(idx, td) with the UNWEIGHTED td: p[idx] = |td| + eps = [1.0, 1.5]  (weighted would give [1.5, 1.0]); p[2:] untouched
# --------------------------------------------------------------------------------------------------------------------
def hand_derived_known_answer(Engine, **engine_kw):
    layers = _dense_layers(2, 2)
    hp = abi.default_hparams(batch_size=2, n_actions=2, obs_c=2, obs_h=1, obs_w=1, learning_rate=0.25, gamma=0.5, double_q=1, dueling=0,
                             prioritized_replay=1, buffer_size=4, prio_alpha=1.0, prio_beta=1.0, prio_eps=0.5)
    h = Engine(layers, hp, **engine_kw)
    # Flux.params order: W (out, in) column-major == float[in][out], then b
    h.set_params(np.array([1, 3, 2, -1, 0, 1], np.float32), 0)
    h.set_params(np.array([2, 0, 0, 1, 1, 0], np.float32), 1)
    s = np.array([[1, 0], [0, 2], [5, 5], [7, 7]], np.float32)
    sp = np.array([[0, 1], [1, 1], [6, 6], [8, 8]], np.float32)
    a = np.array([0, 1, 0, 1], np.int32)
    r = np.array([1, -2, 0, 0], np.float32)
    done = np.array([0, 1, 0, 0], np.uint8)
    h.replay_add(s, a, r, sp, done, td_err=np.array([0.5, 3.5, 1.5, 0.5], np.float32))
    np.testing.assert_array_equal(h.replay_priorities(), np.array([1, 4, 2, 1], np.float32))
    idx = np.array([0, 1], np.int64)
    w = h.get_batch(idx)[5]
    np.testing.assert_array_equal(w, np.array([2.0, 0.5], np.float32))
    loss, gn, td = h.train_step(idx)
    q = h.last_q()
    np.testing.assert_array_equal(q["q_on_s"], np.array([[1, 4], [4, -1]], np.float32))
    np.testing.assert_array_equal(q["q_on_sp"], np.array([[2, 0], [3, 3]], np.float32))
    np.testing.assert_array_equal(q["q_tg_sp"], np.array([[1, 1], [3, 1]], np.float32))
    np.testing.assert_array_equal(q["best_a"], np.array([0, 0], np.int32))
    np.testing.assert_array_equal(q["y"], np.array([1.5, -2.0], np.float32))
    np.testing.assert_array_equal(td, np.array([-0.5, 1.0], np.float32))
    assert loss == np.float32(0.3125), loss
    assert gn == np.float32(1.0), gn
    np.testing.assert_array_equal(h.get_grads(), np.array([-1, 0, 0, 0.25, -1, 0.125], np.float32))
    np.testing.assert_allclose(h.get_params(0), np.array([1.25, 3, 2, -1.25, 0.25, 0.75], np.float32), rtol=1e-6, atol=0)
    np.testing.assert_array_equal(h.get_params(1), np.array([2, 0, 0, 1, 1, 0], np.float32))       # the target net does not move
    np.testing.assert_array_equal(h.replay_priorities(), np.array([1.0, 1.5, 2, 1], np.float32))
    h.close()


def sampler_distribution(Engine, draws=2000, **engine_kw):
    """Inclusion frequencies of `sample(replay)` vs p / sum(p) on a skewed priority vector.  The reference draws B DISTINCT indices
    with probability proportional to priority (StatsBase A-ExpJ, …replay.jl:85); the engine's stratified sum-tree draws WITH replacement
    (DESIGN.md section 5): per call, stratum i of B takes one index with probability proportional to the priority mass inside
    [i, i+1) * total/B, so E[count_j] = draws * B * p_j / sum(p) exactly, and the variance is below the multinomial's."""
    n, B = 64, 16
    layers = _dense_layers(2, 2)
    hp = abi.default_hparams(batch_size=B, n_actions=2, obs_c=2, obs_h=1, obs_w=1, dueling=0, buffer_size=n, prio_alpha=1.0, prio_eps=0.5, seed=77)
    h = Engine(layers, hp, **engine_kw)
    rng = np.random.default_rng(5)
    p = (0.5 + 40.0 * rng.random(n) ** 4).astype(np.float32)          # three orders of magnitude between smallest and largest
    obs = rng.random((n, 2), dtype=np.float32)
    h.replay_add(obs, np.zeros(n, np.int32), np.zeros(n, np.float32), obs, np.zeros(n, np.uint8), td_err=p - np.float32(0.5))
    pr = h.replay_priorities().astype(np.float64)
    counts = np.zeros(n)
    for _ in range(draws):
        idx = h.replay_sample()
        assert idx.min() >= 0 and idx.max() < n
        np.add.at(counts, idx, 1)
    expect = draws * B * pr / pr.sum()
    chi2 = float(((counts - expect) ** 2 / expect).sum())
    dof = n - 1
    assert chi2 < dof + 5.0 * np.sqrt(2.0 * dof), (chi2, dof)          # 5 sigma of a chi-square with n-1 degrees of freedom
    heavy = expect > 200
    np.testing.assert_allclose(counts[heavy], expect[heavy], rtol=0.12)
    uniform = np.full(n, draws * B / n)
    assert ((counts - uniform) ** 2 / uniform).sum() > 50 * dof         # and the test can fail: a uniform sampler is nowhere near
    h.close()
    return chi2


def sampler_distinct(Engine, draws=400, **engine_kw):
    """hp.sample_distinct = 1 (...replay.jl:85: sample(rng, 1:n, Weights(p), B, replace=false)): B DISTINCT indices per call, always; a leaf whose
    priority spans several strata -- the stratified default returns it several times -- is returned once and the freed positions are redrawn on
    the RESIDUAL priorities (successive sampling): with one dominant leaf and equal light leaves the redraws must be uniform over the light ones."""
    n, B = 96, 16
    layers = _dense_layers(2, 2)
    hp = abi.default_hparams(batch_size=B, n_actions=2, obs_c=2, obs_h=1, obs_w=1, dueling=0, buffer_size=n, prio_alpha=1.0, prio_eps=0.5, seed=31, sample_distinct=1)
    h = Engine(layers, hp, **engine_kw)
    p = np.full(n, 1.0, np.float32); p[37] = 900.0                      # leaf 37 holds 90 % of the mass: 14-15 of the 16 strata
    obs = np.random.default_rng(2).random((n, 2), dtype=np.float32)
    h.replay_add(obs, np.zeros(n, np.int32), np.zeros(n, np.float32), obs, np.zeros(n, np.uint8), td_err=p - np.float32(0.5))
    counts = np.zeros(n)
    seen = []
    for _ in range(draws):
        idx = h.replay_sample()
        assert len(set(idx.tolist())) == B, idx                         # distinct, every time
        assert idx.min() >= 0 and idx.max() < n and 37 in idx           # the dominant leaf is always in the batch, once
        np.add.at(counts, idx, 1)
        seen.append(idx.copy())
    light = np.delete(counts, 37)
    expect = draws * (B - 1) / (n - 1)                                   # B - 1 of the n - 1 equal light leaves per call
    assert abs(light.sum() - draws * (B - 1)) < 1e-9
    chi2 = float(((light - expect) ** 2 / expect).sum()); dof = n - 2
    assert chi2 < dof + 6.0 * np.sqrt(2.0 * dof), (chi2, dof)           # uniform over the light leaves (a sampler that redraws "the next leaf" fails by far)
    # a skewed vector without a dominant leaf: still distinct, and heavier leaves are included more often
    pr = (0.5 + 40.0 * np.random.default_rng(5).random(n) ** 4).astype(np.float32)
    h.update_priorities(np.arange(n, dtype=np.int64), pr - np.float32(0.5))
    cnt2 = np.zeros(n)
    for _ in range(draws):
        idx = h.replay_sample()
        assert len(set(idx.tolist())) == B, idx
        np.add.at(cnt2, idx, 1)
    order = np.argsort(pr)
    assert cnt2[order[-10:]].mean() > 4 * cnt2[order[:40]].mean()
    assert cnt2.max() <= draws                                           # inclusion frequency <= 1 per call
    h.close()
    return np.array(seen)
