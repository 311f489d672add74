// drqn_cols.hip -- the recurrent train step (batch_train!(..., ::EpisodeReplayBuffer), src/solver.jl:239-287; sample, src/episode_replay.jl:71-95) as ONE
// column-parallel launch + the Adam launch (BASELINE config 4; DrqnColsArgs in common.h).
//
// Batch columns never interact before the gradient sum: the two target passes (online and target network over the sp sequence from Flux.reset!), the
// online pass over the s sequence, TD / masked Huber, the head backward, BPTT and every dW / db contraction decompose by batch column.  Workgroup g owns the
// columns [g*cg, (g+1)*cg) for the whole step, holds the parameters of both networks (2 x ~31 KB at config 4) and all of its columns' activations in LDS,
// and writes ONE gradient slab (its chunk of every dW / db: the column-group chunks of the summation plan, dw_kc = -cg).  The step's
// second launch (k_adam, adam_body.h) adds the B / cg slabs in ascending order, folds the loss from the per-column Huber terms and applies Flux Adam.
// The multi-launch program this replaces ran 9 dependent launches (102 us at config 4).
//
// Canonical order (== the CPU twin, oracle/dqn_ref.c ref_train_step_drqn, bit for bit):
//   input projection   Gx[n] = chain_k (+0; k ascending) fma(x[k], Wi[k][n], .)                       (lstm_input_proj)
//   gate               g = (Gx[n] + chain_j fma(h[j], Wh[j][n], .)) + b[n]; sigm / tanh through Float64, rounded once
//   cell               c' = (f * c) + (i * g~);  h' = o * tanh(c')
//   head               per plan chunk of K = H: chain_k fma(h[k], W[k][n], .); chunk sums ascending; + bias; activation
//   TD                 k_td_drqn's per-column arithmetic (drqn.hip): dueling (v + a) - mean, first-max argmax, r + ((1 - done) * gamma) * q, Huber(mask * td)
//   head dX            chain_n fma(dpre[n], W[u][n], .); at the dueling join dX_val + dX_adv
//   BPTT               k_lstm_bwd_step's arithmetic; dh_{t-1}[u] = chain_n (n ascending over 4H) fma(dG[n], Wh[u][n], .)
//   dW / db / dstate0  chain over the group's columns (t ascending, then b ascending) from +0 (db, dstate0: plain adds from +0); groups added ascending by k_adam
#include "common.h"
#include <type_traits>

__device__ __forceinline__ float sigm_d(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
// (float)tanh((double)x) without ocml's 160-instruction double-double tanh on the serial path of every time step.  A double y with a known error bound decides:
// when every double within the bound rounds to the SAME float, that float IS the correctly rounded tanh -- which is also what (float)tanh((double)x) gives unless
// the true value sits within a double ulp of a float rounding boundary (the pre-existing 2^-29 event between any two libms).  Otherwise (about one wave in 10^4)
// the library function decides.  |x| >= 2^-5: y = (1 - t) / (1 + t), t = exp(-2|x|), absolute error < 1e-15; |x| < 2^-5: the odd Taylor series through x^11
// (truncation < 4e-21 relative), relative error < 1e-15.  Both branches are computed for every lane (no divergence: gate pre-activations near 0 are common).
__device__ __forceinline__ float tanh_d(float xf) {
    const double x = (double)xf, ax = fabs(x);
    const double t = exp(-2.0 * ax); const double num = 1.0 - t, den = 1.0 + t; const double yb = num / den;
    const double z = ax * ax;
    double p = -1382.0 / 155925.0; p = p * z + 62.0 / 2835.0; p = p * z - 17.0 / 315.0; p = p * z + 2.0 / 15.0; p = p * z - 1.0 / 3.0; p = p * z; const double ys = ax + ax * p;
    const bool small = ax < 0.03125;
    const double y = small ? ys : yb; const double d = small ? 4.0e-15 * ys : 2.0e-15;
    const float lo = (float)(y - d), hi = (float)(y + d);
    if (lo == hi && ax < 19.0) return copysignf(lo, xf);
    return (float)tanh(x);
}
__device__ __forceinline__ float sigm_fast(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { const float e = __expf(2.0f * x); return (e - 1.0f) / (e + 1.0f); }
__device__ __forceinline__ void ldsb() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }      // LDS-only phases: no vmcnt drain (tiny_step.hip)

// TT: compile-time bound on the trace length (the input projections of all time steps live in registers); HH: bound on H (a thread keeps its Wh column in registers)
template <int TT, int HH, int WK>
__global__ __launch_bounds__(1024) void k_drqn_cols(const DrqnColsArgs A, const int slot_i) {
    extern __shared__ __align__(16) float sm[];
    __shared__ int np_s[4]; __shared__ long long ep_s4[4];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int B = A.B, T = A.T, H = A.H, E = A.E, nA = A.nA, cg = A.cg, nset = A.nset, N = 4 * H, per = H * cg, Ep = (E + 3) & ~3;
    const int duel = A.dueling ? 1 : 0, no = nA + duel, Pint = (int)A.Pint;
    const int g = blockIdx.x, b0 = g * cg; const bool fastp = (A.probe & 1) != 0;
    unsigned long long* const stamps = A.stamps; int n_stamp = 0;
    auto stamp = [&]() { if (stamps && g == 0 && tid == 0) stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime(); };
    stamp();
    // ---- LDS layout
    float* Pon = sm; float* Ptg = Pon + Pint;
    float* Xs = Ptg + Pint;                     // [T][cg][Ep]   s sequence of the group's columns
    float* Xsp = Xs + T * cg * Ep;              // [T][cg][Ep]   sp sequence
    int* a_s = (int*)(Xsp + T * cg * Ep);       // [T][cg]
    float* r_s = (float*)(a_s + T * cg); float* dn_s = r_s + T * cg; float* m_s = dn_s + T * cg;
    float* Hout = m_s + T * cg;                 // [nset][T][cg][H]   h_t of every sequence set
    float* cst = Hout + nset * T * per;         // [nset][cg][H]      cell state
    float* g_s = cst + nset * per;              // [nset][4][cg][H]   activated gates of the current step
    float* GD = g_s + nset * 4 * per;           // [T][cg][4H]        set 0: activated gates, overwritten in place by dG during BPTT
    float* TC = GD + T * cg * N;                // [T][cg][H]         tanh(c_t)
    float* CP = TC + T * per;                   // [T][cg][H]         c_{t-1}
    float* QO = CP + T * per;                   // [nset][T][cg][no]  head outputs (advantage / plain Q first, the value stream last)
    // (QO and DQ are the only arrays whose extent need not be a multiple of 4 floats -- odd T x odd n_out: everything behind them is read 16 bytes at a time (WhP by BPTT), so
    // both extents are rounded up; a misaligned ds_read_b128 is replayed at 64 cycles or worse, MI355X guide / Guideline 17)
    float* DQ = QO + ((nset * T * cg * no + 3) & ~3);        // [T][cg][no]        dpre of the heads
    float* dH = DQ + ((T * cg * no + 3) & ~3);               // [T][cg][H]
    float* dhn = dH + T * per;                  // [cg][H]
    float* dcn = dhn + per;                     // [cg][H]
    float* WhP = dcn + per;                     // [H][4H + 4]        online Wh, padded rows (BPTT reads row u 16 B at a time: stride 4H would put every lane on one bank)
    const int MT = (T * cg + 15) / 16;          // 16-row M tiles of the input-projection GEMM (rows = the group's columns j = t * cg + c)
    float* GX = WhP + H * (N + 4);              // [nset][16 MT][4H]  input projections of every (set, column, gate output); the 16-column tiles of odd-kq rows swapped (conflict-free stores)
    // ---- thread role in the recurrence: (set, gate q, column c, unit u); sets: 0 = online net on s (kept for BPTT), [1 = online net on sp (double-Q)], last = target net on sp
    const bool on = tid < nset * 4 * per;
    const int set = on ? tid / (4 * per) : 0, rr = tid - set * 4 * per, q = rr / per, ee = rr - q * per, c = ee / H, u = ee - c * H, n = q * H + u;
    const bool tgt = set == nset - 1;
    // ---- phase 0: everything whose address is known at entry goes out in ONE round: the step's episode draws (mapped HOST memory: the longest latency, first),
    // this thread's Wh column, Wi column and bias (registers, straight from L2), the parameters of both networks (LDS)
    long long ep_v = 0; int np_v = 0;
    if (tid < cg) {                                                  // np: rows the prefix copy delivers (episode_replay.jl:82-92), computed by the host from (length, start)
        const size_t slot = (size_t)slot_i * B;
        ep_v = A.ring_idx[slot + b0 + tid]; np_v = A.ring_np[slot + b0 + tid];
    }
    // this thread's Wh column and bias go straight from L2 into registers: the serial path of the recurrence reads no weight from LDS
    const float* Pg = tgt ? A.p_tg : A.p_on;
    float gx[TT], wh[HH];
#pragma unroll
    for (int j = 0; j < HH; j++) wh[j] = (on && j < H) ? Pg[A.wh_off + (size_t)j * N + n] : 0.0f;
    const float bias_n = on ? Pg[A.b_off + n] : 0.0f;
    if (tid < cg) { ep_s4[tid] = ep_v; np_s[tid] = np_v; }          // waits for the draws only (the oldest loads of these lanes); everything above stays in flight
    ldsb();
    // the rows of the group's columns (prefix-copy quirk: always the episode PREFIX, zero beyond it), requested together with the parameters below
    auto pick = [&](int cc, long long& ep, int& np) { ep = ep_s4[cc]; np = np_s[cc]; };
    for (int i = tid; i < T * cg * Ep; i += NT) {
        const int f = i % Ep, cc = (i / Ep) % cg, t = i / (Ep * cg);
        long long ep; int np; pick(cc, ep, np);
        float vs = 0.0f, vp = 0.0f;
        if (f < E && t < np) { const size_t row = ((size_t)ep * T + t) * E + f; vs = A.ep_s[row]; vp = A.ep_sp[row]; }
        Xs[i] = vs; Xsp[i] = vp;
    }
    for (int i = tid; i < T * cg; i += NT) {
        const int cc = i % cg, t = i / cg; long long ep; int np; pick(cc, ep, np);
        const bool ok = t < np; const size_t slot = (size_t)ep * T + t;
        a_s[i] = ok ? A.ep_a[slot] : 0;                              // CartesianIndex(1,1) on masked rows: harmless, the mask multiplies inside huber
        r_s[i] = ok ? A.ep_r[slot] : 0.0f; dn_s[i] = ok ? (float)A.ep_done[slot] : 0.0f; m_s[i] = ok ? 1.0f : 0.0f;
    }
    {   // parameters of both networks into LDS -- without the Wh blocks: every thread holds its column of Wh in registers and BPTT reads the padded copy WhP
        const float4* po = reinterpret_cast<const float4*>(A.p_on); const float4* pt = reinterpret_cast<const float4*>(A.p_tg);
        const int s2 = (int)A.wh_off / 4, s3 = (int)(A.wh_off + (unsigned)(H * N)) / 4;
        for (int i = tid; i < Pint / 4; i += NT) {
            if (i >= s2 && i < s3) continue;
            reinterpret_cast<float4*>(Pon)[i] = po[i]; reinterpret_cast<float4*>(Ptg)[i] = pt[i];
        }
    }
    for (int i = tid; i < H * N / 4; i += NT) *reinterpret_cast<float4*>(WhP + (i / H) * (N + 4) + 4 * (i % H)) = reinterpret_cast<const float4*>(A.p_on + A.wh_off)[i];      // row = i / (N/4) = i / H
    if (g == 0 && tid == 0) A.st->step = A.st->step + 1;            // read by the Adam launch (beta-power slot)
    __syncthreads();
    stamp();
    const float* P = tgt ? Ptg : Pon;

    // ---- phase 1: input projections of ALL time steps (they do not depend on the recurrence) as fp32 MFMA tiles: rows = the group's 16 MT columns j = t * cg + c, 16 gate
    // outputs per tile, K = E in steps of 4 -- v_mfma_f32_16x16x4_f32 accumulates its four products in k order, one rounding each, so every element is the twin's chain
    // fma(x[k], Wi[k][n], .) over k ascending from +0 (padded k and padded rows contribute fma(0, 0, acc) = acc).  As VALU chains this phase was bound by
    // the LDS return path: 56 broadcast ds_read_b128 per thread for 200 fma (3.8 us); here a wave reads E/4 x 2 dwords per tile (r04).
    if (A.probe & 2) {      // A/B switch (DQN_DRQN_PROBE=2): the VALU form of the same chains (same bits), kept for same-box comparisons
#pragma unroll
        for (int t = 0; t < TT; t++) gx[t] = 0.0f;
        if (on) {
            const float* wi = P + A.wi_off + n; const float* xc = (set == 0 ? Xs : Xsp) + c * Ep;
            for (int k = 0; k < E; k++) {
                const float w0 = wi[(size_t)k * N];
#pragma unroll
                for (int t = 0; t < TT; t++) if (t < T) gx[t] = fmaf(xc[t * cg * Ep + k], w0, gx[t]);
            }
        }
    } else {
    {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        const int wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4, nwaves = NT >> 6;
        const int NTn = N / 16, ntiles = nset * MT * NTn, KS = Ep / 4, NJ1 = T * cg;
        for (int tile = wave; tile < ntiles; tile += nwaves) {
            const int st = tile / (MT * NTn), rem = tile - st * MT * NTn, mt = rem / NTn, nt = rem - mt * NTn;
            const float* Pn = (st == nset - 1 ? Ptg : Pon) + A.wi_off + 16 * nt + l15; const float* Xn = st == 0 ? Xs : Xsp;
            const int j = 16 * mt + l15;
            f32x4v acc = {0.f, 0.f, 0.f, 0.f};
            for (int ks0 = 0; ks0 < KS; ks0 += 8) {               // operands of eight k-steps requested before the first MFMA (E <= 32: all of them)
                float av[8], bv[8];
#pragma unroll
                for (int q8 = 0; q8 < 8; q8++) { const int k = 4 * (ks0 + q8) + kq; const bool okk = ks0 + q8 < KS;
                    av[q8] = (okk && j < NJ1) ? Xn[j * Ep + k] : 0.0f;      // the rows are zero-padded to Ep
                    bv[q8] = (okk && k < E) ? Pn[(size_t)k * N] : 0.0f; }
#pragma unroll
                for (int q8 = 0; q8 < 8; q8++) if (ks0 + q8 < KS) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q8], bv[q8], acc, 0, 0, 0);
            }
            float* G = GX + (size_t)(st * MT * 16 + 16 * mt + 4 * kq) * N + ((16 * nt + l15) ^ ((kq & 1) << 4));      // D: rows 4 kq + i, column l15
            G[0] = acc.x; G[N] = acc.y; G[2 * N] = acc.z; G[3 * N] = acc.w;
        }
    }
    ldsb();
    if (on) {
#pragma unroll
        for (int t = 0; t < TT; t++) { gx[t] = 0.0f; if (t < T) { const int j = t * cg + c; gx[t] = GX[(size_t)(set * MT * 16 + j) * N + (n ^ ((((j & 15) >> 2) & 1) << 4))]; } }
    }
    }
    stamp();
    // ---- phase 2: the recurrence, all sequence sets side by side
#pragma unroll
    for (int t = 0; t < TT; t++) {
        if (t >= T) break;
        if (on) {
            const float* hp = t == 0 ? P + A.h0_off : Hout + ((set * T + t - 1) * cg + c) * H;      // Flux.reset!: state0 broadcast over the batch
            float ch = 0.0f;
#pragma unroll
            for (int j = 0; j < HH; j += 4) if (j < H) {
                const float4 h4 = *reinterpret_cast<const float4*>(hp + j);
                ch = fmaf(h4.x, wh[j], ch); ch = fmaf(h4.y, wh[j + 1], ch); ch = fmaf(h4.z, wh[j + 2], ch); ch = fmaf(h4.w, wh[j + 3], ch);
            }
            const float gv = (gx[t] + ch) + bias_n;
            const float act = fastp ? (q == 2 ? tanh_fast(gv) : sigm_fast(gv)) : (q == 2 ? tanh_d(gv) : sigm_d(gv));
            g_s[(set * 4 + q) * per + ee] = act;
            if (set == 0) GD[(t * cg + c) * N + n] = act;
        }
        ldsb();
        if (on && q == 0) {
            const float* gs = g_s + set * 4 * per + ee;
            const float ig = gs[0], fg = gs[per], gg = gs[2 * per], og = gs[3 * per];
            const float cp = t == 0 ? P[A.c0_off + u] : cst[set * per + ee];
            const float t1 = fg * cp; const float t2 = ig * gg; const float cv = t1 + t2; const float tc = fastp ? tanh_fast(cv) : tanh_d(cv); const float h = og * tc;
            Hout[((set * T + t) * cg + c) * H + u] = h; cst[set * per + ee] = cv;
            if (set == 0) { TC[(t * cg + c) * H + u] = tc; CP[(t * cg + c) * H + u] = cp; }
        }
        ldsb();
    }
    stamp();
    // ---- phase 3: heads of every (set, t, column): advantage / plain Q outputs first, the value stream last
    for (int i = tid; i < nset * T * cg * no; i += NT) {
        const int o = i % no, col = i / no;                      // col = (set * T + t) * cg + c
        const int st = col / (T * cg); const int hd = o < nA ? 0 : 1, nn = o < nA ? o : 0, Nh = A.hN[hd];
        const float* Ph = st == nset - 1 ? Ptg : Pon; const float* W = Ph + A.hw_off[hd] + nn; const float* x = Hout + (size_t)col * H;
        const int S = A.h_S[hd], kc = A.h_kc[hd];
        float tot = 0.0f;
        if (S == 1) {                                                // one chain over K = H (a multiple of 8): operands requested eight at a time
            for (int k = 0; k < H; k += 8) {
                const float4 xa = *reinterpret_cast<const float4*>(x + k), xb = *reinterpret_cast<const float4*>(x + k + 4);
                float w[8];
#pragma unroll
                for (int j = 0; j < 8; j++) w[j] = W[(size_t)(k + j) * Nh];
                tot = fmaf(xa.x, w[0], tot); tot = fmaf(xa.y, w[1], tot); tot = fmaf(xa.z, w[2], tot); tot = fmaf(xa.w, w[3], tot);
                tot = fmaf(xb.x, w[4], tot); tot = fmaf(xb.y, w[5], tot); tot = fmaf(xb.z, w[6], tot); tot = fmaf(xb.w, w[7], tot);
            }
        } else
        for (int s = 0; s < S; s++) {
            float acc = 0.0f; const int k1 = (s + 1) * kc < H ? (s + 1) * kc : H;
            for (int k = s * kc; k < k1; k++) acc = fmaf(x[k], W[(size_t)k * Nh], acc);
            tot = s == 0 ? acc : tot + acc;
        }
        QO[i] = act_f(tot + Ph[A.hb_off[hd] + nn], A.hact[hd]);
    }
    ldsb();
    stamp();
    // ---- phase 4: targets, TD, masked Huber terms, dL/dQ (src/solver.jl:259-282); one thread per (t, column).  Loops are unrolled over the 16 action slots
    // (registers, no private-memory arrays); every LDS read of the item goes out before the arithmetic
    auto td_items = [&](auto na_c) {
    constexpr int NA = decltype(na_c)::value;                        // compile-time bound on the action slots of the unrolled loops
    for (int i = tid; i < T * cg; i += NT) {
        const int cc = i % cg, t = i / cg; const int k = t * B + b0 + cc;
        const float dn_v = dn_s[i], r_v = r_s[i], m = m_s[i]; const int act = a_s[i];
        float r0[NA + 1], r1[NA + 1], r2[NA + 1];                                // raw head outputs of the item: online s, online sp (double-Q), target sp
        const float* o0 = QO + (size_t)((0 * T + t) * cg + cc) * no; const float* o1 = QO + (size_t)((1 * T + t) * cg + cc) * no; const float* o2 = QO + (size_t)(((nset - 1) * T + t) * cg + cc) * no;
#pragma unroll
        for (int a = 0; a < NA + 1; a++) { r0[a] = a < no ? o0[a] : 0.0f; r1[a] = (a < no && A.double_q) ? o1[a] : 0.0f; r2[a] = a < no ? o2[a] : 0.0f; }
        // Q of a column from its raw outputs: dueling (v + a) - mean with mean = (a_0 + a_1 + ...) / nA (src/dueling.jl:10), else the outputs themselves
        auto qcol = [&](const float (&r)[NA + 1], float (&qo)[NA]) {
            float sum = r[0];
#pragma unroll
            for (int a = 1; a < NA; a++) if (a < nA) sum = sum + r[a];
            const float mean = sum / (float)nA;
            float v = 0.0f;
#pragma unroll
            for (int a = 0; a < NA + 1; a++) if (a == nA) v = r[a];
#pragma unroll
            for (int a = 0; a < NA; a++) qo[a] = duel ? (v + r[a]) - mean : r[a];
            return v;
        };
        float qt[NA], qp[NA], qs[NA];
        qcol(r2, qt);
        int best = 0;
        if (A.double_q) { qcol(r1, qp);
#pragma unroll
            for (int a = 1; a < NA; a++) { float qb = qp[0];
#pragma unroll
                for (int z = 1; z < NA; z++) if (z == best) qb = qp[z];
                if (a < nA && qp[a] > qb) best = a; } }              // first-max (Julia argmax)
        else {
#pragma unroll
            for (int a = 1; a < NA; a++) { float qb = qt[0];
#pragma unroll
                for (int z = 1; z < NA; z++) if (z == best) qb = qt[z];
                if (a < nA && qt[a] > qb) best = a; } }
        float qsp = qt[0];
#pragma unroll
        for (int a = 1; a < NA; a++) if (a == best) qsp = qt[a];
        const float t1 = 1.0f - dn_v; const float t2 = t1 * A.gamma; const float t3 = t2 * qsp; const float y = r_v + t3;
        const float vraw = qcol(r0, qs);
        float qsa = qs[0];
#pragma unroll
        for (int a = 1; a < NA; a++) if (a == act) qsa = qs[a];
        const float td = qsa - y; A.td[k] = td;
        const float x = m * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
        A.hl[k] = (0.5f * qd) * qd + lin;
        const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        const float invT = 1.0f / (float)T;
        const float gq = ((invT / (float)B) * cl) * m;
        float* dq = DQ + (size_t)i * no;
        const float gm = gq / (float)nA;
        if (duel) dq[nA] = dact_f(gq, vraw, A.hact[1]);
#pragma unroll
        for (int a = 0; a < NA; a++) if (a < nA) dq[a] = dact_f(duel ? (a == act ? gq : 0.0f) - gm : (a == act ? gq : 0.0f), r0[a], A.hact[0]);
    }
    };
    if (nA <= 4) td_items(std::integral_constant<int, 4>{}); else if (nA <= 8) td_items(std::integral_constant<int, 8>{}); else td_items(std::integral_constant<int, 16>{});
    ldsb();
    stamp();
    // ---- phase 5: dX of the heads = dH[t][column][u] (at the dueling join dX_val + dX_adv); online parameters
    for (int i = tid; i < T * per; i += NT) {
        const int uu = i % H, col = i / H;                       // col = t * cg + c
        const float* dq = DQ + (size_t)col * no; const float* Wa = Pon + A.hw_off[0] + (size_t)uu * nA;
        float xa = 0.0f;
        for (int a = 0; a < nA; a++) xa = fmaf(dq[a], Wa[a], xa);
        if (duel) { float xv = 0.0f; xv = fmaf(dq[nA], Pon[A.hw_off[1] + uu], xv); xa = xv + xa; }
        dH[i] = xa;
    }
    ldsb();
    stamp();
    // ---- phase 6: BPTT over the s sequence; thread (column, unit)
    const bool bw = tid < per; const int bc = tid / H, bu = tid - bc * H;
    for (int t = T - 1; t >= 0; t--) {
        if (bw) {
            float* gd = GD + (size_t)(t * cg + bc) * N;
            const float ig = gd[bu], fg = gd[H + bu], gg = gd[2 * H + bu], og = gd[3 * H + bu];
            const float tc = TC[(t * cg + bc) * H + bu], cprev = CP[(t * cg + bc) * H + bu];
            const float dhn_v = t == T - 1 ? 0.0f : dhn[tid], dcn_v = t == T - 1 ? 0.0f : dcn[tid];
            const float dh = dH[(t * cg + bc) * H + bu] + dhn_v;
            const float dov = dh * tc; const float t1 = dh * og; const float t2 = tc * tc; const float t3 = 1.0f - t2; const float t4 = t1 * t3; const float dc = dcn_v + t4;
            const float di = dc * gg, df = dc * cprev, dgc = dc * ig; dcn[tid] = dc * fg;
            const float a1 = di * ig, a2 = 1.0f - ig; const float b1 = df * fg, b2 = 1.0f - fg; const float c1 = gg * gg, c2 = 1.0f - c1; const float d1 = dov * og, d2 = 1.0f - og;
            gd[bu] = a1 * a2; gd[H + bu] = b1 * b2; gd[2 * H + bu] = dgc * c2; gd[3 * H + bu] = d1 * d2;
        }
        ldsb();
        if (bw) {                                                    // dh_{t-1}[u] = sum_n dG[n] Wh[u][n], n ascending
            const float* gd = GD + (size_t)(t * cg + bc) * N; const float* wr = WhP + (size_t)bu * (N + 4);
            float acc = 0.0f;
#pragma unroll 8
            for (int nn = 0; nn < N; nn += 4) {
                const float4 d4 = *reinterpret_cast<const float4*>(gd + nn); const float4 w4 = *reinterpret_cast<const float4*>(wr + nn);
                acc = fmaf(d4.x, w4.x, acc); acc = fmaf(d4.y, w4.y, acc); acc = fmaf(d4.z, w4.z, acc); acc = fmaf(d4.w, w4.w, acc);
            }
            dhn[tid] = acc;
        }
        ldsb();
    }
    stamp();
    // ---- phase 7: this group's chunk of every gradient: one fma chain per element over the group's columns (t ascending, then b ascending), from +0
    float* slab = A.slabs + (size_t)g * Pint;
    const unsigned wi0 = A.wi_off, wi1 = wi0 + (unsigned)(E * N), bb1 = A.b_off + (unsigned)N, wh0 = A.wh_off, wh1 = wh0 + (unsigned)(H * N);
    const int NJ = T * cg;                                           // the group's columns in chain order: j = t * cg + c
    auto elem = [&](unsigned ui) {                                   // scalar path: state0, heads, padding
        float acc = 0.0f;
        if (ui >= A.h0_off && ui < A.h0_off + (unsigned)H) {         // trainable state0: dh_{-1}, dc_{-1} summed over the group's columns
            const int uu = (int)(ui - A.h0_off); for (int cc = 0; cc < cg; cc++) acc = acc + dhn[cc * H + uu];
        } else if (ui >= A.c0_off && ui < A.c0_off + (unsigned)H) {
            const int uu = (int)(ui - A.c0_off); for (int cc = 0; cc < cg; cc++) acc = acc + dcn[cc * H + uu];
        } else {
            for (int hd = 0; hd <= duel; hd++) {
                const int Nh = A.hN[hd]; const unsigned w0 = A.hw_off[hd], w1 = w0 + (unsigned)(H * Nh), bo = A.hb_off[hd];
                const int o0 = hd == 0 ? 0 : nA;
                if (ui >= w0 && ui < w1) {                           // head dW[k][n]: x = h_t of the s sequence
                    const int k = (int)(ui - w0) / Nh, nn = (int)(ui - w0) % Nh;
#pragma unroll 4
                    for (int j = 0; j < NJ; j++) acc = fmaf(Hout[(size_t)j * H + k], DQ[(size_t)j * no + o0 + nn], acc);
                } else if (ui >= bo && ui < bo + (unsigned)Nh) {
                    const int nn = (int)(ui - bo);
                    for (int j = 0; j < NJ; j++) acc = acc + DQ[(size_t)j * no + o0 + nn];
                }
            }
        }
        return acc;                                                  // everything else (alignment padding, the junk bias row of Wh, the zero bias of the projection): +0
    };
    for (int i4 = tid; i4 < Pint / 4; i4 += NT) {                    // four consecutive elements per item: every LSTM array is 16-B aligned with a multiple of 4 elements per row
        const unsigned ui = 4u * (unsigned)i4; float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (ui >= wi0 && ui < wi1) {                                 // dWi[k][n..n+3]: x = s row, d = dG
            const int k = (int)(ui - wi0) / N, nn = (int)(ui - wi0) % N;
#pragma unroll 4
            for (int j = 0; j < NJ; j++) { const float x = Xs[j * Ep + k]; const float4 d = *reinterpret_cast<const float4*>(GD + (size_t)j * N + nn);
                acc.x = fmaf(x, d.x, acc.x); acc.y = fmaf(x, d.y, acc.y); acc.z = fmaf(x, d.z, acc.z); acc.w = fmaf(x, d.w, acc.w); }
        } else if (ui >= A.b_off && ui < bb1) {                      // db[n..n+3]
            const int nn = (int)(ui - A.b_off);
#pragma unroll 4
            for (int j = 0; j < NJ; j++) { const float4 d = *reinterpret_cast<const float4*>(GD + (size_t)j * N + nn); acc.x = acc.x + d.x; acc.y = acc.y + d.y; acc.z = acc.z + d.z; acc.w = acc.w + d.w; }
        } else if (ui >= wh0 && ui < wh1) {                          // dWh[j][n..n+3]: x = h_{t-1} (h0 at t = 0)
            const int jr = (int)(ui - wh0) / N, nn = (int)(ui - wh0) % N; const float x0 = Pon[A.h0_off + jr];
#pragma unroll 4
            for (int j = 0; j < NJ; j++) { const float x = j < cg ? x0 : Hout[(size_t)(j - cg) * H + jr]; const float4 d = *reinterpret_cast<const float4*>(GD + (size_t)j * N + nn);
                acc.x = fmaf(x, d.x, acc.x); acc.y = fmaf(x, d.y, acc.y); acc.z = fmaf(x, d.z, acc.z); acc.w = fmaf(x, d.w, acc.w); }
        } else { acc.x = elem(ui); acc.y = elem(ui + 1); acc.z = elem(ui + 2); acc.w = elem(ui + 3); }
        *reinterpret_cast<float4*>(slab + ui) = acc;
    }
    stamp();
}

static size_t drqn_cols_lds_floats(const DrqnColsArgs& a) {
    const size_t T = a.T, cg = a.cg, H = a.H, N = 4 * H, per = H * cg, Ep = (a.E + 3) & ~3, no = a.nA + (a.dueling ? 1 : 0), ns = a.nset;
    return 2 * (size_t)a.Pint + 2 * T * cg * Ep + 4 * T * cg + ns * T * per + ns * per + ns * 4 * per + T * cg * N + 2 * T * per + ((ns * T * cg * no + 3) & ~(size_t)3) + ((T * cg * no + 3) & ~(size_t)3) + T * per + 2 * per + H * (N + 4) + ns * ((T * cg + 15) / 16 * 16) * N;
}
int launch_drqn_cols(hipStream_t st, const DrqnColsArgs& a, int slot) {
    const size_t lds = drqn_cols_lds_floats(a) * sizeof(float);
    int nt = a.nset * 4 * a.H * a.cg; nt = (nt + 63) / 64 * 64; if (nt < a.H * a.cg) nt = a.H * a.cg; if (nt < 256) nt = 256;
    const int G = a.B / a.cg;
#define DRQN_COLS_LAUNCH(TTv, HHv) do { \
        if (lds > 64 * 1024) { const hipError_t le = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_drqn_cols<TTv, HHv, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (le != hipSuccess) { (void)hipGetLastError(); return -1; } } \
        hipLaunchKernelGGL((k_drqn_cols<TTv, HHv, 32>), dim3(G), dim3(nt), lds, st, a, slot); } while (0)
    if (a.H <= 32) { if (a.T <= 8) DRQN_COLS_LAUNCH(8, 32); else if (a.T <= 16) DRQN_COLS_LAUNCH(16, 32); else DRQN_COLS_LAUNCH(64, 32); }
    else { if (a.T <= 8) DRQN_COLS_LAUNCH(8, 64); else if (a.T <= 16) DRQN_COLS_LAUNCH(16, 64); else DRQN_COLS_LAUNCH(64, 64); }
#undef DRQN_COLS_LAUNCH
    return 0;
}
