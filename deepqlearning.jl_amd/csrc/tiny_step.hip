// tiny_step.hip -- the WHOLE train step (batch_train!, src/solver.jl:191-236 + ...replay.jl:76-104) of a dense network that fits in LDS as ONE
// single-workgroup launch: sample -> get_batch -> forwards of both nets -> dueling / double-Q target / TD / Huber -> backward -> globalnorm + Adam ->
// update_priorities! + the next step's index draw.  BASELINE config 1 (SimpleGridWorld, Chain(Dense(2,32), Dense(32,4)) dueling, 357 parameters)
// ran as six dependent launches of one to four workgroups each: 1.95 us of launch + 3-10 us of a lone workgroup's chain per launch, 32.9 us per step.
// Here the launch is paid once and no phase waits for a kernel boundary.
//
// Every value follows the canonical order of the kernels this replaces, so the step is bit-identical to the multi-launch
// program and to the CPU twin:
//   forward   per plan chunk: acc = +0; k ascending: acc = fma(x[k], W[k][n], acc); chunk sums added ascending; + bias; activation
//   TD        k_head_td's per-column arithmetic (dueling (v + a) - mean, first-max argmax, r + ((1 - done) * gamma) * q, Huber, dL/dQ)
//   dW / db   per plan chunk over the samples: acc = +0; b ascending: acc = fma(x[k][b], dpre[n][b], acc)  (db: acc = acc + dpre[n][b]); chunk sums ascending
//   dX        acc = +0; n ascending: acc = fma(dpre[n][b], W[k][n], acc); at the dueling join dX_val + dX_adv; then act' of the producing layer
//   Adam      adam_upd (adam_body.h) per element, beta powers double-buffered by step parity; max |g| for globalnorm
//   priority  prio_block_run (common.h): the same block the multi-launch program carries in a backward launch
#include "common.h"
#include "adam_body.h"

// barrier for phases that exchange LDS data only: __syncthreads() also drains vmcnt -- every parity store (Q columns, td, gradients, ...) of the phase
// before would cost its round trip (1-2 us) at each of the step's ~8 barriers; here those stores stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NTH>
__global__ __launch_bounds__(NTH) void k_tiny_step(const TinyArgs* __restrict__ Ap, int sample, int stop) {
    extern __shared__ __align__(16) float sm[];
    __shared__ int act_s[64];
    __shared__ float rew_s[64], dn_s[64], w_s[64], hl_s[64], wmax_s[NTH / 64];
    // layer descriptors in LDS: a per-item `A.L[l].field` is a dependent GLOBAL load (the record lives in device memory); staged once, every later
    // access is an LDS read (probe, r03: forward + backward 22 us of a 39 us step with the loads inside the item loops)
    enum { F_K, F_N, F_ACT, F_SRC, F_W, F_B, F_FKC, F_DWKC, F_ON, F_TG, F_D, F_NF };
    __shared__ int LD[TINY_MAX_LAYERS][F_NF];
    __shared__ int LEV[TINY_MAX_LAYERS][3];
    const TinyArgs& A = *Ap;
    const int tid = threadIdx.x, B = A.B, nA = A.nA, E = A.E, ld0 = 2 * B, ncon = A.ncon;
    StepState* st = A.state;
    const int nlev = A.nlev, dueling = A.dueling, double_q = A.double_q, last_val = A.last_val, lq = A.dueling ? A.last_adv : A.last_base;
    const float gamma = A.gamma;
    float* Pon = sm + A.pon_off; float* Ptg = sm + A.ptg_off; float* Gs = sm + A.g_off; float* X0 = sm + A.x0_off;
    float* qs = sm + A.misc_off;                       // [B][3][nA]
    const int P = (int)A.P;
    auto qdiv = [](int x, float rcp) { return (int)(((float)x + 0.5f) * rcp); };      // x / d through one multiply, exact for x < 2^16 (nn_valu.hip)
    // ---- phase 0: ONE round trip for everything whose address is known at entry: counters, the pre-drawn indices, parameters of both nets, Adam's m and v
    const unsigned long long ctr0 = st->sample_ctr, step0 = st->step; const long long size = st->size; const int pv = st->pre_valid;
    const int bcol = tid < B ? tid : tid - B;          // threads [0, 2B): column tid of the arena (s_b | sp_b)
    long long r_pre = 0;
    if (tid < ld0) r_pre = sample ? (A.idx_pre ? A.idx_pre[bcol] : 0) : A.idx[bcol];
    constexpr int PR = 2;                              // parameters per thread kept in registers (m, v prefetch)
    const bool pre_mv = P <= PR * NTH;
    float m_r[PR], v_r[PR];
#pragma unroll
    for (int u = 0; u < PR; u++) { const int i = tid + u * NTH; m_r[u] = 0.0f; v_r[u] = 0.0f; if (pre_mv && i < P) { m_r[u] = A.m[i]; v_r[u] = A.v[i]; } }
    const int slot = (int)((step0 + 1) & 1ull);
    const double bp1 = st->bp[slot][0], bp2 = st->bp[slot][1];
    if (tid < A.nl) {
        const LayerDev& L = A.L[tid];
        LD[tid][F_K] = L.K; LD[tid][F_N] = L.N; LD[tid][F_ACT] = L.act; LD[tid][F_SRC] = L.src; LD[tid][F_W] = (int)L.w_off; LD[tid][F_B] = (int)L.b_off;
        LD[tid][F_FKC] = L.fwd_kc; LD[tid][F_DWKC] = L.dw_kc; LD[tid][F_ON] = A.on_off[tid]; LD[tid][F_TG] = A.tg_off[tid]; LD[tid][F_D] = A.d_off[tid];
    }
    if (tid >= 64 && tid < 64 + nlev) { const int v = tid - 64; LEV[v][0] = A.lev_n[v]; LEV[v][1] = A.lev_l[v][0]; LEV[v][2] = A.lev_l[v][1]; }
    for (int i = tid; i < P; i += NTH) { Pon[i] = A.p_on[i]; Ptg[i] = A.p_tg[i]; Gs[i] = 0.0f; }
    // hp.sample_distinct (r05): a sample() that is not pre-drawn is deduped like k_sample's (sample_distinct_block; the pre-drawn list already is) -- a uniform branch
    const bool draw_distinct = A.distinct && sample && !(A.idx_pre && pv);
    if (draw_distinct) {
        __shared__ long long t_list[64], t_taken[64]; __shared__ float t_tp[64]; __shared__ int t_any;
        if (tid < B) t_list[tid] = tree_descend(A.tree, A.cap2, size, A.seed, ctr0, tid, A.tree[1] / (float)B);
        __syncthreads();
        sample_distinct_block(A.tree, A.cap2, size, A.seed, ctr0, B, t_list, t_taken, t_tp, &t_any);
        if (tid < ld0) r_pre = t_list[bcol];
    }
    // ---- sample() + get_batch (...replay.jl:82-102): thread c < 2B owns arena column c; the first B also fetch the batch scalars and the IS weight
    if (tid < ld0) {
        long long r = r_pre;
        // the indices of this sample() were drawn in the tail of the previous step's priority block unless something changed the tree since
        if (sample && !(A.idx_pre && pv) && !draw_distinct) r = tree_descend(A.tree, A.cap2, size, A.seed, ctr0, bcol, A.tree[1] / (float)B);
        const bool first = tid < B;
        int a_ = 0; float rw_ = 0.0f, dn_ = 0.0f, leaf_ = 0.0f, tot_ = 1.0f;
        if (first) { a_ = A.ra[r]; rw_ = A.rr[r]; dn_ = (float)A.rdone[r]; leaf_ = A.tree[A.cap2 + r]; tot_ = A.tree[1]; if (sample) A.idx[tid] = r; }
        for (int f = 0; f < E; f++) {
            float v;
            if (A.obs_u8) v = u8_unit(reinterpret_cast<const unsigned char*>(first ? A.s_rows : A.sp_rows)[r * E + f]);      // byte / 255f0, test/test_env.jl:59
            else v = reinterpret_cast<const float*>(first ? A.s_rows : A.sp_rows)[r * E + f];
            X0[f * ld0 + tid] = v; A.x0[(size_t)f * ld0 + tid] = v;
        }
        if (first) {
            act_s[tid] = a_; rew_s[tid] = rw_; dn_s[tid] = dn_;
            const float p = leaf_ / tot_;                                    // p = prio ./ sum(prio[1:n]), ...replay.jl:101
            const float x = (float)size * p;                                 // n .* p
            const float w = (float)pow((double)x, -(double)A.beta);          // .^ (-beta), :102
            w_s[tid] = w; A.w_is[tid] = w;
        }
    }
    lds_barrier();
    if (stop == 2) return;
    // ---- phase 1: forward, level by level: online net on [s ; sp] (ncon columns), target net on sp
    const int cols = ncon + B; const float rcols = 1.0f / (float)cols;
    const bool v4 = (B & 3) == 0;
    for (int lv = 0; lv < nlev; lv++) {
        const int nlay = LEV[lv][0];
        // the level's one or two sibling layers share ONE item loop; their descriptors sit in registers (per-item selects, no LDS lookups)
        struct FD { int K, N, act, woff, boff, S, kc, ldon, ldtg; const float *Xon, *Xtg; float *Yon, *Ytg; } fd[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int l = LEV[lv][1 + (q < nlay ? q : 0)]; const int src = LD[l][F_SRC];
            fd[q].K = LD[l][F_K]; fd[q].N = LD[l][F_N]; fd[q].act = LD[l][F_ACT]; fd[q].woff = LD[l][F_W]; fd[q].boff = LD[l][F_B];
            fd[q].S = dqn_nchunks(fd[q].K, LD[l][F_FKC]); fd[q].kc = dqn_chunk_len(fd[q].K, LD[l][F_FKC]);
            fd[q].Xon = src < 0 ? X0 : sm + LD[src][F_ON]; fd[q].Xtg = src < 0 ? X0 + B : sm + LD[src][F_TG];
            fd[q].ldon = src < 0 ? ld0 : ncon; fd[q].ldtg = src < 0 ? ld0 : B;
            fd[q].Yon = sm + LD[l][F_ON]; fd[q].Ytg = sm + LD[l][F_TG];
        }
        if (v4) {
            // four consecutive columns per item (B % 4 == 0: a group never straddles the online / target boundary and every row is 16-B aligned): one CU
            // issues 64 lane-instructions per cycle, and at config 1 the scalar loop's ~60 instructions x 6144 items were 2.4 us of the level
            const int c4n = cols >> 2; const float rc4 = 1.0f / (float)c4n;
            const int cnt_a = fd[0].N * c4n, cnt_b = nlay > 1 ? fd[1].N * c4n : 0;
            for (int it0 = tid; it0 < cnt_a + cnt_b; it0 += NTH) {
                const bool sec = it0 >= cnt_a; const int it = sec ? it0 - cnt_a : it0;
                const int K = sec ? fd[1].K : fd[0].K, N = sec ? fd[1].N : fd[0].N, S = sec ? fd[1].S : fd[0].S, kc = sec ? fd[1].kc : fd[0].kc;
                const int n = qdiv(it, rc4), c = 4 * (it - n * c4n); const bool tg = c >= ncon;
                const float* Pn = tg ? Ptg : Pon;
                const float* W = Pn + (sec ? fd[1].woff : fd[0].woff) + n; const float bias = Pn[(sec ? fd[1].boff : fd[0].boff) + n];
                const float* X = tg ? (sec ? fd[1].Xtg : fd[0].Xtg) + (c - ncon) : (sec ? fd[1].Xon : fd[0].Xon) + c;
                const int ldx = tg ? (sec ? fd[1].ldtg : fd[0].ldtg) : (sec ? fd[1].ldon : fd[0].ldon);
                f32x4c tot = {0.f, 0.f, 0.f, 0.f};
                for (int s = 0; s < S; s++) {
                    const int k0 = s * kc, k1 = min(K, k0 + kc);
                    f32x4c acc = {0.f, 0.f, 0.f, 0.f}; int k = k0;
                    for (; k + 4 <= k1; k += 4) {
                        f32x4c xv[4]; float wv[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { xv[u] = *reinterpret_cast<const f32x4c*>(X + (k + u) * ldx); wv[u] = W[(k + u) * N]; }
#pragma unroll
                        for (int u = 0; u < 4; u++) { acc.x = fmaf(xv[u].x, wv[u], acc.x); acc.y = fmaf(xv[u].y, wv[u], acc.y); acc.z = fmaf(xv[u].z, wv[u], acc.z); acc.w = fmaf(xv[u].w, wv[u], acc.w); }
                    }
                    for (; k < k1; k++) { const f32x4c x = *reinterpret_cast<const f32x4c*>(X + k * ldx); const float w = W[k * N]; acc.x = fmaf(x.x, w, acc.x); acc.y = fmaf(x.y, w, acc.y); acc.z = fmaf(x.z, w, acc.z); acc.w = fmaf(x.w, w, acc.w); }
                    if (s == 0) tot = acc; else { tot.x = tot.x + acc.x; tot.y = tot.y + acc.y; tot.z = tot.z + acc.z; tot.w = tot.w + acc.w; }
                }
                const int act = sec ? fd[1].act : fd[0].act;
                const f32x4c y = {act_f(tot.x + bias, act), act_f(tot.y + bias, act), act_f(tot.z + bias, act), act_f(tot.w + bias, act)};
                if (tg) *reinterpret_cast<f32x4c*>((sec ? fd[1].Ytg : fd[0].Ytg) + n * B + (c - ncon)) = y; else *reinterpret_cast<f32x4c*>((sec ? fd[1].Yon : fd[0].Yon) + n * ncon + c) = y;
            }
            lds_barrier();
            continue;
        }
        const int cnt_a = fd[0].N * cols, cnt_b = nlay > 1 ? fd[1].N * cols : 0;
        for (int it0 = tid; it0 < cnt_a + cnt_b; it0 += NTH) {
            const bool sec = it0 >= cnt_a; const int it = sec ? it0 - cnt_a : it0;
            const int K = sec ? fd[1].K : fd[0].K, N = sec ? fd[1].N : fd[0].N, S = sec ? fd[1].S : fd[0].S, kc = sec ? fd[1].kc : fd[0].kc;
            const int n = qdiv(it, rcols), c = it - n * cols; const bool tg = c >= ncon;
            const float* Pn = tg ? Ptg : Pon;
            const float* W = Pn + (sec ? fd[1].woff : fd[0].woff) + n; const float bias = Pn[(sec ? fd[1].boff : fd[0].boff) + n];
            const float* X = tg ? (sec ? fd[1].Xtg : fd[0].Xtg) + (c - ncon) : (sec ? fd[1].Xon : fd[0].Xon) + c;
            const int ldx = tg ? (sec ? fd[1].ldtg : fd[0].ldtg) : (sec ? fd[1].ldon : fd[0].ldon);
            float tot = 0.0f;
            for (int s = 0; s < S; s++) {
                const int k0 = s * kc, k1 = min(K, k0 + kc);
                float acc = 0.0f; int k = k0;
                for (; k + 8 <= k1; k += 8) {      // 16 independent LDS reads in flight; the chain stays k-ascending
                    float xv[8], wv[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { xv[u] = X[(k + u) * ldx]; wv[u] = W[(k + u) * N]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) acc = fmaf(xv[u], wv[u], acc);
                }
                if (k + 2 == k1) { const float x0_ = X[k * ldx], x1_ = X[(k + 1) * ldx], w0_ = W[k * N], w1_ = W[(k + 1) * N]; acc = fmaf(x0_, w0_, acc); acc = fmaf(x1_, w1_, acc); k += 2; }
                for (; k < k1; k++) acc = fmaf(X[k * ldx], W[k * N], acc);
                tot = s == 0 ? acc : tot + acc;
            }
            const float y = act_f(tot + bias, sec ? fd[1].act : fd[0].act);
            if (tg) (sec ? fd[1].Ytg : fd[0].Ytg)[n * B + (c - ncon)] = y; else (sec ? fd[1].Yon : fd[0].Yon)[n * ncon + c] = y;
        }
        lds_barrier();
    }
    if (stop == 3) return;
    // ---- phase 2: Q columns, Bellman target, TD, Huber, dL/dQ (one lane per batch column)
    if (tid < B) {
        const int b = tid;
        const float* a_on = sm + LD[lq][F_ON]; const float* a_tg = sm + LD[lq][F_TG];
        float* q = qs + b * 3 * nA;
        for (int c = 0; c < 3; c++) {
            if (c == 1 && !double_q) continue;
            float* qg = c == 0 ? A.q_on_s : (c == 1 ? A.q_on_sp : A.q_tg_sp);
            auto ar = [&](int a) { return c == 2 ? a_tg[a * B + b] : a_on[a * ncon + (c == 1 ? B + b : b)]; };
            if (!dueling) { for (int a = 0; a < nA; a++) { const float v = ar(a); q[c * nA + a] = v; qg[(size_t)b * nA + a] = v; if (c == 2 && !double_q) A.q_on_sp[(size_t)b * nA + a] = v; } }
            else {
                const float vv = c == 2 ? sm[LD[last_val][F_TG] + b] : sm[LD[last_val][F_ON] + (c == 1 ? B + b : b)];
                float sum = ar(0);
                for (int a = 1; a < nA; a++) sum = sum + ar(a);
                const float mean = sum / (float)nA;                      // Q = (val .+ adv) .- mean(adv), src/dueling.jl:10
                for (int a = 0; a < nA; a++) { const float v = (vv + ar(a)) - mean; q[c * nA + a] = v; qg[(size_t)b * nA + a] = v; if (c == 2 && !double_q) A.q_on_sp[(size_t)b * nA + a] = v; }
            }
        }
        const float invB = 1.0f / (float)B;
        const int act = act_s[b]; const float rew = rew_s[b], dn = dn_s[b], w = w_s[b];
        const float* qsel = double_q ? q + nA : q + 2 * nA;          // argmax over the online net's Q(sp) (double-Q) or the target net's
        int best = 0; float bq = qsel[0];
        for (int a = 1; a < nA; a++) { const float v = qsel[a]; if (v > bq) { bq = v; best = a; } }      // first max (Julia argmax)
        const float qsp = q[2 * nA + best];
        A.best[b] = best;
        const float t1 = 1.0f - dn; const float t2 = t1 * gamma; const float t3 = t2 * qsp; const float y = rew + t3;
        A.ytarget[b] = y;
        const float qsa = q[act];
        const float td = qsa - y; A.td[b] = td;
        const float x = w * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
        hl_s[b] = (0.5f * qd) * qd + lin;
        const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        const float g = (invB * cl) * w;
        const int act_a = LD[lq][F_ACT];
        float* d_adv = sm + LD[lq][F_D];
        if (dueling) {
            sm[LD[last_val][F_D] + b] = dact_f(g, sm[LD[last_val][F_ON] + b], LD[last_val][F_ACT]);
            const float gm = g / (float)nA;
            for (int a = 0; a < nA; a++) d_adv[a * B + b] = dact_f((a == act ? g : 0.0f) - gm, a_on[a * ncon + b], act_a);
        } else {
            for (int a = 0; a < nA; a++) d_adv[a * B + b] = dact_f(a == act ? g : 0.0f, a_on[a * ncon + b], act_a);
        }
    }
    lds_barrier();
    if (tid == NTH - 1) {      // (a lane of the last wave: the first one has the TD columns behind it)
        float lsum = 0.0f;
        for (int b = 0; b < B; b++) lsum = lsum + hl_s[b];
        st->loss = lsum / (float)B;
        st->step = step0 + 1;
        if (sample) st->sample_ctr = ctr0 + 1;
        st->bp[slot ^ 1][0] = bp1 * A.b1; st->bp[slot ^ 1][1] = bp2 * A.b2;      // Flux: bp .= bp .* beta AFTER the update; double-buffered by step parity
    }
    if (stop == 4) return;
    // ---- phase 3: backward, level by level from the heads: dW / db of the level's layers and dX into the level below
    const float rB = 1.0f / (float)B;
    for (int lv = nlev - 1; lv >= 0; lv--) {
        const int nlay = LEV[lv][0];
        struct BD { int K, N, src, woff, S, kc, ldx, ndw, ndx, act_src; const float *D, *Xs, *Ys; float* Dsrc; } bd[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int l = LEV[lv][1 + (q < nlay ? q : 0)]; const int src = LD[l][F_SRC];
            bd[q].K = LD[l][F_K]; bd[q].N = LD[l][F_N]; bd[q].src = src; bd[q].woff = LD[l][F_W];
            bd[q].S = dqn_nchunks(B, LD[l][F_DWKC]); bd[q].kc = dqn_chunk_len(B, LD[l][F_DWKC]);
            bd[q].D = sm + LD[l][F_D];
            bd[q].Xs = src < 0 ? X0 : sm + LD[src][F_ON]; bd[q].ldx = src < 0 ? ld0 : ncon;      // the s columns are the first B of either
            bd[q].Ys = src < 0 ? nullptr : sm + LD[src][F_ON]; bd[q].Dsrc = src < 0 ? nullptr : sm + LD[src][F_D]; bd[q].act_src = src < 0 ? 0 : LD[src][F_ACT];
            bd[q].ndw = q < nlay ? (bd[q].K + 1) * bd[q].N : 0;
        }
        const bool join = nlay > 1 && bd[0].src >= 0 && bd[0].src == bd[1].src;      // the two streams meet at the base output: the value stream's items compute both
        bd[0].ndx = bd[0].src >= 0 ? bd[0].K * B : 0; bd[1].ndx = (nlay > 1 && bd[1].src >= 0 && !join) ? bd[1].K * B : 0;
        if (v4 && (bd[0].S == 1 || (bd[0].kc & 3) == 0) && (bd[1].S == 1 || (bd[1].kc & 3) == 0)) {
            // 16-byte LDS reads: a dW chain reads 4 samples of both operands per instruction (same b-ascending order), a dX item owns 4 consecutive columns
            const int B4 = B >> 2; const float rB4 = 1.0f / (float)B4;
            const int x0n = bd[0].ndx >> 2, x1n = bd[1].ndx >> 2;
            const int c1 = bd[0].ndw, c2 = c1 + bd[1].ndw, c3 = c2 + x0n, c4 = c3 + x1n;
            for (int it0 = tid; it0 < c4; it0 += NTH) {
                if (it0 < c2) {
                    const bool sec = it0 >= c1; const int it = sec ? it0 - c1 : it0;
                    const int K = sec ? bd[1].K : bd[0].K, N = sec ? bd[1].N : bd[0].N, S = sec ? bd[1].S : bd[0].S, kc = sec ? bd[1].kc : bd[0].kc, woff = sec ? bd[1].woff : bd[0].woff;
                    const int k = qdiv(it, 1.0f / (float)N), n = it - k * N;
                    const float* dr = (sec ? bd[1].D : bd[0].D) + n * B;
                    const float* xr = (sec ? bd[1].Xs : bd[0].Xs) + k * (sec ? bd[1].ldx : bd[0].ldx);
                    float tot = 0.0f;
                    for (int s = 0; s < S; s++) {
                        const int j0 = s * kc, j1 = min(B, j0 + kc);
                        float acc = 0.0f;
                        if (k < K) {
                            for (int j = j0; j < j1; j += 8) {
                                const f32x4c x0_ = *reinterpret_cast<const f32x4c*>(xr + j), d0_ = *reinterpret_cast<const f32x4c*>(dr + j);
                                f32x4c x1_ = {0.f, 0.f, 0.f, 0.f}, d1_ = {0.f, 0.f, 0.f, 0.f};
                                const bool two = j + 4 < j1;
                                if (two) { x1_ = *reinterpret_cast<const f32x4c*>(xr + j + 4); d1_ = *reinterpret_cast<const f32x4c*>(dr + j + 4); }
                                acc = fmaf(x0_.x, d0_.x, acc); acc = fmaf(x0_.y, d0_.y, acc); acc = fmaf(x0_.z, d0_.z, acc); acc = fmaf(x0_.w, d0_.w, acc);
                                if (two) { acc = fmaf(x1_.x, d1_.x, acc); acc = fmaf(x1_.y, d1_.y, acc); acc = fmaf(x1_.z, d1_.z, acc); acc = fmaf(x1_.w, d1_.w, acc); }
                            }
                        } else {
                            for (int j = j0; j < j1; j += 4) { const f32x4c d0_ = *reinterpret_cast<const f32x4c*>(dr + j); acc = acc + d0_.x; acc = acc + d0_.y; acc = acc + d0_.z; acc = acc + d0_.w; }
                        }
                        tot = s == 0 ? acc : tot + acc;
                    }
                    Gs[woff + it] = tot; A.grad[woff + it] = tot;
                } else {
                    const bool sec = it0 >= c3; const int e = sec ? it0 - c3 : it0 - c2;
                    const int N = sec ? bd[1].N : bd[0].N;
                    const int k = qdiv(e, rB4), b = 4 * (e - k * B4);
                    const float* W = Pon + (sec ? bd[1].woff : bd[0].woff) + k * N; const float* d = (sec ? bd[1].D : bd[0].D) + b;
                    f32x4c acc = {0.f, 0.f, 0.f, 0.f};
                    for (int n = 0; n < N; n++) { const f32x4c dv = *reinterpret_cast<const f32x4c*>(d + n * B); const float w = W[n]; acc.x = fmaf(dv.x, w, acc.x); acc.y = fmaf(dv.y, w, acc.y); acc.z = fmaf(dv.z, w, acc.z); acc.w = fmaf(dv.w, w, acc.w); }
                    if (join) {
                        const int Nb = bd[1].N;
                        const float* W2 = Pon + bd[1].woff + k * Nb; const float* d2p = bd[1].D + b;
                        f32x4c acc2 = {0.f, 0.f, 0.f, 0.f};
                        for (int n = 0; n < Nb; n++) { const f32x4c dv = *reinterpret_cast<const f32x4c*>(d2p + n * B); const float w = W2[n]; acc2.x = fmaf(dv.x, w, acc2.x); acc2.y = fmaf(dv.y, w, acc2.y); acc2.z = fmaf(dv.z, w, acc2.z); acc2.w = fmaf(dv.w, w, acc2.w); }
                        acc.x = acc.x + acc2.x; acc.y = acc.y + acc2.y; acc.z = acc.z + acc2.z; acc.w = acc.w + acc2.w;
                    }
                    const f32x4c y = *reinterpret_cast<const f32x4c*>((sec ? bd[1].Ys : bd[0].Ys) + k * ncon + b); const int as = sec ? bd[1].act_src : bd[0].act_src;
                    const f32x4c o = {dact_f(acc.x, y.x, as), dact_f(acc.y, y.y, as), dact_f(acc.z, y.z, as), dact_f(acc.w, y.w, as)};
                    *reinterpret_cast<f32x4c*>((sec ? bd[1].Dsrc : bd[0].Dsrc) + k * B + b) = o;
                }
            }
            lds_barrier();
            continue;
        }
        const int c1 = bd[0].ndw, c2 = c1 + bd[1].ndw, c3 = c2 + bd[0].ndx, c4 = c3 + bd[1].ndx;
        for (int it0 = tid; it0 < c4; it0 += NTH) {
            if (it0 < c2) {
                const bool sec = it0 >= c1; const int it = sec ? it0 - c1 : it0;
                const int K = sec ? bd[1].K : bd[0].K, N = sec ? bd[1].N : bd[0].N, S = sec ? bd[1].S : bd[0].S, kc = sec ? bd[1].kc : bd[0].kc, woff = sec ? bd[1].woff : bd[0].woff;
                const int k = qdiv(it, 1.0f / (float)N), n = it - k * N;
                const float* dr = (sec ? bd[1].D : bd[0].D) + n * B;
                const float* xr = (sec ? bd[1].Xs : bd[0].Xs) + k * (sec ? bd[1].ldx : bd[0].ldx);
                float tot = 0.0f;
                for (int s = 0; s < S; s++) {
                    const int j0 = s * kc, j1 = min(B, j0 + kc);
                    float acc = 0.0f; int j = j0;
                    if (k < K) {
                        for (; j + 8 <= j1; j += 8) {
                            float xv[8], dv[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) { xv[u] = xr[j + u]; dv[u] = dr[j + u]; }
#pragma unroll
                            for (int u = 0; u < 8; u++) acc = fmaf(xv[u], dv[u], acc);
                        }
                        for (; j < j1; j++) acc = fmaf(xr[j], dr[j], acc);
                    } else {
                        for (; j + 8 <= j1; j += 8) {
                            float dv[8];
#pragma unroll
                            for (int u = 0; u < 8; u++) dv[u] = dr[j + u];
#pragma unroll
                            for (int u = 0; u < 8; u++) acc = acc + dv[u];
                        }
                        for (; j < j1; j++) acc = acc + dr[j];
                    }
                    tot = s == 0 ? acc : tot + acc;
                }
                Gs[woff + it] = tot; A.grad[woff + it] = tot;
            } else {
                const bool sec = it0 >= c3; const int e = sec ? it0 - c3 : it0 - c2;
                const int N = sec ? bd[1].N : bd[0].N;
                const int k = qdiv(e, rB), b = e - k * B;
                const float* W = Pon + (sec ? bd[1].woff : bd[0].woff) + k * N; const float* d = (sec ? bd[1].D : bd[0].D) + b;
                float acc = 0.0f; int n = 0;
                for (; n + 4 <= N; n += 4) {
                    const float d0 = d[n * B], d1 = d[(n + 1) * B], d2 = d[(n + 2) * B], d3 = d[(n + 3) * B], w0 = W[n], w1 = W[n + 1], w2 = W[n + 2], w3 = W[n + 3];
                    acc = fmaf(d0, w0, acc); acc = fmaf(d1, w1, acc); acc = fmaf(d2, w2, acc); acc = fmaf(d3, w3, acc);
                }
                for (; n < N; n++) acc = fmaf(d[n * B], W[n], acc);
                if (join) {
                    const int Nb = bd[1].N;
                    const float* W2 = Pon + bd[1].woff + k * Nb; const float* d2p = bd[1].D + b;
                    float acc2 = 0.0f; int m = 0;
                    for (; m + 4 <= Nb; m += 4) {
                        const float d0 = d2p[m * B], d1 = d2p[(m + 1) * B], d2 = d2p[(m + 2) * B], d3 = d2p[(m + 3) * B], w0 = W2[m], w1 = W2[m + 1], w2 = W2[m + 2], w3 = W2[m + 3];
                        acc2 = fmaf(d0, w0, acc2); acc2 = fmaf(d1, w1, acc2); acc2 = fmaf(d2, w2, acc2); acc2 = fmaf(d3, w3, acc2);
                    }
                    for (; m < Nb; m++) acc2 = fmaf(d2p[m * B], W2[m], acc2);
                    acc = acc + acc2;
                }
                (sec ? bd[1].Dsrc : bd[0].Dsrc)[k * B + b] = dact_f(acc, (sec ? bd[1].Ys : bd[0].Ys)[k * ncon + b], sec ? bd[1].act_src : bd[0].act_src);
            }
        }
        lds_barrier();
    }
    if (stop == 5) return;
    // ---- phase 4: globalnorm (max |g|, helpers.jl:38-46) + Flux Adam (solver.jl:66,228)
    {
        float gmax = 0.0f;
        if (pre_mv) {
#pragma unroll
            for (int u = 0; u < PR; u++) {
                const int i = tid + u * NTH;
                if (i < P) { float mi = m_r[u], vi = v_r[u], pi = Pon[i]; gmax = fmaxf(gmax, adam_upd(Gs[i], mi, vi, pi, A.f64mode, A.lr, A.b1, A.b2, A.adam_eps, bp1, bp2, 1.0f)); A.m[i] = mi; A.v[i] = vi; A.p_on[i] = pi; }
            }
        } else {
            for (int i = tid; i < P; i += NTH) {
                float mi = A.m[i], vi = A.v[i], pi = Pon[i];
                gmax = fmaxf(gmax, adam_upd(Gs[i], mi, vi, pi, A.f64mode, A.lr, A.b1, A.b2, A.adam_eps, bp1, bp2, 1.0f));
                A.m[i] = mi; A.v[i] = vi; A.p_on[i] = pi;
            }
        }
        for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        if ((tid & 63) == 0) wmax_s[tid >> 6] = gmax;
    }
    __syncthreads();      // (full: the priority block below reads idx / td back from global memory)
    if (tid == 0) { float g = wmax_s[0]; for (int w = 1; w < NTH / 64; w++) g = fmaxf(g, wmax_s[w]); A.gmax_part[0] = g; }
    if (stop == 6) return;
    // ---- phase 5: update_priorities!(replay, idx, td) + the next sample()'s stratified draws (the tree is final, the Philox counter is known)
    PrioArgs pa; pa.n = B; pa.cap2 = A.cap2; pa.idx = A.idx; pa.td = A.td; pa.eps = A.prio_eps; pa.alpha = A.prio_alpha; pa.tree = A.tree;
    pa.idx_pre = A.idx_pre; pa.seed = A.seed; pa.B = B; pa.phase = 0; pa.distinct = A.distinct;
    prio_block_run(pa, st, reinterpret_cast<long long*>(sm), A.lds_bytes, nullptr, sample ? ctr0 + 1 : ctr0);
}
int launch_tiny_step(hipStream_t st, const TinyArgs* a_dev, unsigned lds_bytes, int sample, int stop) {
    // dynamic LDS beyond 64 KB has to be requested per function AND per device: asked for on every enqueue that needs it (an enqueue is a graph capture or an eager
    // launch, never the replay path), and a refusal is reported instead of leaving a launch that cannot run
    if (lds_bytes > 64 * 1024) {
        const hipError_t le = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tiny_step<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (le != hipSuccess) { (void)hipGetLastError(); return -1; }
    }
    hipLaunchKernelGGL((k_tiny_step<1024>), dim3(1), dim3(1024), lds_bytes, st, a_dev, sample, stop);
    return 0;
}
