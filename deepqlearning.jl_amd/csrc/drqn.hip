// drqn.hip -- DRQN on the device (BASELINE config 4):
//   EpisodeReplayBuffer sample  src/episode_replay.jl:71-95   -> k_gather_episodes (prefix-copy quirk reproduced)
//   recurrent batch_train!      src/solver.jl:239-287         -> k_lstm_step (x T), k_td_drqn, k_lstm_bwd_step (x T)
//   Flux LSTM (third-party; recalled): g = Wi*x .+ Wh*h .+ b, gates input/forget/cell/output,
//   c' = sigm(f).*c .+ sigm(i).*tanh(g), h' = sigm(o).*tanh(c'), trainable state0 (h0, c0).
// Columns are (time-major) t*B + b, so every feed-forward layer of the network runs ONCE over all T*B columns with the
// ordinary kernels (the LSTM's input projection Wi*x included); only the h/c recurrence is sequential: one small launch
// per time step, all three sequence sets (online s, online sp, target sp) batched in it.  Canonical order as in the CPU
// twin: gate pre-activation = ((chain_k Wi x) + (chain_j Wh h)) + b, sigm/tanh through double, rounded once.
#include "common.h"

__device__ __forceinline__ float sigm_f(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
__device__ __forceinline__ float tanh_f(float x) { return (float)tanh((double)x); }

// ------------------------------------------------------------------ sample(r::EpisodeReplayBuffer) for given draws
__global__ void k_gather_episodes(EpGatherArgs A) {
    const int TB = A.T * A.B, ld = 2 * TB;
    const size_t n = (size_t)A.E * ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ld); const int f = (int)(i / ld);
        const int sp = c >= TB; const int k = sp ? c - TB : c; const int t = k / A.B, b = k % A.B;
        const long long ep = A.ep_idx[b]; const int len = A.ep_len[ep];
        int np = min(len, A.T) - A.ep_start[b]; if (np < 0) np = 0;   // `for j = ep_start:min(len,T)` copies ep[1..] : the episode PREFIX (:82-92)
        float v = 0.0f;
        if (t < np) v = (sp ? A.ep_sp : A.ep_s)[((size_t)ep * A.T + t) * A.E + f];
        A.x0[i] = v;
    }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < TB; k += blockDim.x) {
            const int t = k / A.B, b = k % A.B; const long long ep = A.ep_idx[b];
            int np = min(A.ep_len[ep], A.T) - A.ep_start[b]; if (np < 0) np = 0;
            const bool ok = t < np; const size_t slot = (size_t)ep * A.T + t;
            A.a_out[k] = ok ? A.ep_a[slot] : 0;                        // CartesianIndex(1,1) on masked rows: harmless, the mask multiplies inside huber
            A.r_out[k] = ok ? A.ep_r[slot] : 0.0f; A.done_out[k] = ok ? (float)A.ep_done[slot] : 0.0f; A.mask_out[k] = ok ? 1.0f : 0.0f;
        }
}
void launch_gather_episodes(hipStream_t st, const EpGatherArgs& a) {
    const size_t n = (size_t)a.E * 2 * a.T * a.B; unsigned blocks = (unsigned)((n + 255) / 256); if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_gather_episodes, dim3(blocks), dim3(256), 0, st, a);
}

// ------------------------------------------------------------------ one LSTM time step for up to 3 sequence sets
__global__ void k_lstm_step(LstmStepArgs A, int t) {
    const int per = A.H * A.B;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * A.nseq) return;
    const LstmSeq& S = A.s[i / per];
    const int e = i % per, u = e / A.B, b = e % A.B, H = A.H, N = 4 * H;
    const int col = S.c0 + t * A.B + b;
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const int n = q * H + u; float ch = 0.0f;
        for (int j = 0; j < H; j++) ch = fmaf(S.hprev[(size_t)j * S.hp_ld + (size_t)b * S.hp_bs], S.Wh[(size_t)j * N + n], ch);
        g[q] = (S.Gx[(size_t)n * S.ld + col] + ch) + S.bias[n];
    }
    const float ig = sigm_f(g[0]), fg = sigm_f(g[1]), gg = tanh_f(g[2]), og = sigm_f(g[3]);
    const float cp = S.cprev[(size_t)u * S.cp_ld + (size_t)b * S.cp_bs];
    const float t1 = fg * cp; const float t2 = ig * gg; const float c = t1 + t2; const float tc = tanh_f(c); const float h = og * tc;
    S.Hout[(size_t)u * S.ld + col] = h; S.Cst[(size_t)u * S.ld + col] = c;
    if (S.gates) {
        const size_t k = (size_t)S.keep_c0 + t * A.B + b; const size_t kl = S.keep_ld;
        S.gates[(size_t)(0 * H + u) * kl + k] = ig; S.gates[(size_t)(1 * H + u) * kl + k] = fg; S.gates[(size_t)(2 * H + u) * kl + k] = gg; S.gates[(size_t)(3 * H + u) * kl + k] = og;
        S.tc[(size_t)u * kl + k] = tc; S.hprev_out[(size_t)u * kl + k] = S.hprev[(size_t)u * S.hp_ld + (size_t)b * S.hp_bs]; S.cprev_out[(size_t)u * kl + k] = cp;
    }
}
void launch_lstm_step_t(hipStream_t st, const LstmStepArgs& a, int t) {
    const int n = a.H * a.B * a.nseq;
    hipLaunchKernelGGL(k_lstm_step, dim3((n + 255) / 256), dim3(256), 0, st, a, t);
}

// ------------------------------------------------------------------ one BPTT step (single workgroup: dh_{t-1} needs all 4H gate gradients of step t)
__global__ __launch_bounds__(1024) void k_lstm_bwd_step(LstmBwdArgs A) {
    const int H = A.H, B = A.B, TB = A.TB, N = 4 * H, t = A.t, per = H * B;
    for (int e = threadIdx.x; e < per; e += blockDim.x) {
        const int u = e / B, b = e % B; const size_t k = (size_t)t * B + b;
        const float ig = A.gates[(size_t)u * TB + k], fg = A.gates[(size_t)(H + u) * TB + k], gg = A.gates[(size_t)(2 * H + u) * TB + k], og = A.gates[(size_t)(3 * H + u) * TB + k];
        const float tc = A.tc[(size_t)u * TB + k], cprev = A.cprev[(size_t)u * TB + k];
        const float dhn = t == A.T - 1 ? 0.0f : A.dhn[e], dcn = t == A.T - 1 ? 0.0f : A.dcn[e];
        const float dh = A.dH[(size_t)u * TB + k] + dhn;
        const float dov = dh * tc; const float t1 = dh * og; const float t2 = tc * tc; const float t3 = 1.0f - t2; const float t4 = t1 * t3; const float dc = dcn + t4;
        const float di = dc * gg, df = dc * cprev, dgc = dc * ig; A.dcn[e] = dc * fg;
        const float a1 = di * ig, a2 = 1.0f - ig; A.dG[(size_t)u * TB + k] = a1 * a2;
        const float b1 = df * fg, b2 = 1.0f - fg; A.dG[(size_t)(H + u) * TB + k] = b1 * b2;
        const float c1 = gg * gg, c2 = 1.0f - c1; A.dG[(size_t)(2 * H + u) * TB + k] = dgc * c2;
        const float d1 = dov * og, d2 = 1.0f - og; A.dG[(size_t)(3 * H + u) * TB + k] = d1 * d2;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < per; e += blockDim.x) {     // dh_{t-1}[j][b] = sum_n dG[n][t,b] Wh[j][n], n ascending
        const int j = e / B, b = e % B; const size_t k = (size_t)t * B + b;
        float acc = 0.0f;
        for (int n = 0; n < N; n++) acc = fmaf(A.dG[(size_t)n * TB + k], A.Wh[(size_t)j * N + n], acc);
        A.dhn[e] = acc;
    }
    if (t == 0) {                                              // trainable state0: gradient summed over the batch, ascending b
        __syncthreads();
        for (int u = threadIdx.x; u < H; u += blockDim.x) {
            float sh = 0.0f, sc = 0.0f;
            for (int b = 0; b < B; b++) { sh = sh + A.dhn[u * B + b]; sc = sc + A.dcn[u * B + b]; }
            A.g_h0[u] = sh; A.g_c0[u] = sc;
        }
    }
}
void launch_lstm_bwd_step(hipStream_t st, const LstmBwdArgs& a) {
    int bs = ((a.H * a.B + 63) / 64) * 64; if (bs > 1024) bs = 1024;
    hipLaunchKernelGGL(k_lstm_bwd_step, dim3(1), dim3(bs), 0, st, a);
}

// ------------------------------------------------------------------ whole-sequence kernels for small LSTMs (config 4: H = 32, B = 32, T = 8)
// The per-step launches above cost a dispatch and a cold walk over Wh per time step.  When Wh (H x 4H) and one step's state fit in
// LDS, ONE launch runs the whole recurrence: workgroup s owns sequence set s (online s, online sp, target sp), keeps Wh, the bias
// and the (h, c) state in LDS and walks t = 0..T-1; the arithmetic per output -- and therefore every bit -- is that of k_lstm_step.
// Batch columns are independent in the recurrence, so a sequence set is further split into groups of CB columns (one workgroup
// each, its own LDS copy of Wh): H*CB ~ 256 outputs per step keeps one wave per SIMD busy and the double-precision sigm/tanh
// (most of a step's instructions) spread over 4x more CUs.
static int lstm_cb(int H, int B) { int cb = 256 / H; if (cb < 1) cb = 1; if (cb > B) cb = B; while (B % cb) cb--; return cb; }
bool lstm_seq_fits(int H, int B, int T) {     // both kernels within 64 KB of dynamic LDS; the gate-parallel forward wants whole waves per gate
    const int cb = lstm_cb(H, B);
    const size_t fwd = (size_t)H * 4 * H + 4 * H + 7 * (size_t)H * cb, bwd = (size_t)H * (4 * H + 1) + 6 * (size_t)H * cb;
    return fwd <= 16384 && bwd <= 16384 && (H * cb) % 64 == 0 && H * cb <= 256 && T <= 64;
}

// r03: (1) the FOUR gates of an output advance on four threads (thread = (gate, unit, column): one 32-deep chain and ONE Float64 sigm/tanh each
// instead of four chains and five transcendentals in sequence), the gates meet in LDS and the (unit, column) thread of gate 0 finishes c, tanh(c), h;
// (2) the input projections Gx of ALL time steps are requested before the recurrence starts (they do not depend on it) -- one round trip instead of one
// per time step.  Per-element arithmetic unchanged (same chains, same association), so every bit is k_lstm_step's.  TT: compile-time bound on T.
template <int TT>
__global__ __launch_bounds__(1024) void k_lstm_seq(LstmSeqArgs A, int CB) {
    extern __shared__ float lds[];
    const int H = A.H, B = A.B, N = 4 * H, per = H * CB, T = A.T, nsplit = B / CB;
    float* Wh_s = lds;                 // [H][4H]
    float* bias_s = Wh_s + H * N;      // [4H]
    float* h_s = bias_s + N;           // [2][H*CB]
    float* c_s = h_s + 2 * per;        // [H*CB]
    float* g_s = c_s + per;            // [4][H*CB] activated gates of the current step
    const LstmSeqF& S = A.s[blockIdx.x / nsplit];
    const int b0 = (blockIdx.x % nsplit) * CB;
    for (int i = threadIdx.x; i < H * N; i += blockDim.x) Wh_s[i] = S.Wh[i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) bias_s[i] = S.bias[i];
    for (int e = threadIdx.x; e < per; e += blockDim.x) { const int u = e / CB; h_s[e] = S.h0[u]; c_s[e] = S.c0v[u]; }      // Flux.reset!: state0 broadcast over the batch
    const int q = threadIdx.x / per, e = threadIdx.x - q * per;      // gate, (unit, column) element; per is a multiple of 64, so a wave has one gate
    const bool on = q < 4;
    const int u = e / CB, bl = e - u * CB, b = b0 + bl;
    float gxr[TT];
#pragma unroll
    for (int t = 0; t < TT; t++) gxr[t] = (on && t < T) ? S.Gx[(size_t)(q * H + u) * S.ld + S.c0 + t * B + b] : 0.0f;
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int t = 0; t < TT; t++) {
        if (t >= T) break;
        const float* hp = h_s + cur * per; float* hn = h_s + (cur ^ 1) * per;
        const int col = S.c0 + t * B + b;
        if (on) {
            float ch = 0.0f;
            const float* wr = Wh_s + q * H + u;
#pragma unroll 8
            for (int j = 0; j < H; j++) ch = fmaf(hp[j * CB + bl], wr[j * N], ch);
            const float g = (gxr[t] + ch) + bias_s[q * H + u];
            const float act = q == 2 ? tanh_f(g) : sigm_f(g);
            g_s[q * per + e] = act;
            if (S.gates) S.gates[(size_t)(q * H + u) * S.keep_ld + (size_t)S.keep_c0 + t * B + b] = act;
        }
        __syncthreads();
        if (q == 0) {
            const float ig = g_s[e], fg = g_s[per + e], gg = g_s[2 * per + e], og = g_s[3 * per + e];
            const float cp = c_s[e];
            const float t1 = fg * cp; const float t2 = ig * gg; const float c = t1 + t2; const float tc = tanh_f(c); const float h = og * tc;
            S.Hout[(size_t)u * S.ld + col] = h; S.Cst[(size_t)u * S.ld + col] = c;
            if (S.gates) { const size_t k = (size_t)S.keep_c0 + t * B + b; const size_t kl = S.keep_ld; S.tc[(size_t)u * kl + k] = tc; S.hprev_out[(size_t)u * kl + k] = hp[e]; S.cprev_out[(size_t)u * kl + k] = cp; }
            hn[e] = h; c_s[e] = c;
        }
        __syncthreads();
        cur ^= 1;
    }
}
void launch_lstm_seq(hipStream_t st, const LstmSeqArgs& a) {
    const int cb = lstm_cb(a.H, a.B);
    const size_t lds = ((size_t)a.H * 4 * a.H + 4 * a.H + 7 * (size_t)a.H * cb) * sizeof(float);
    const int bs = 4 * a.H * cb;                      // 4 gates x (unit, column) elements; lstm_seq_fits: H * cb is a multiple of 64 and <= 256
    if (a.T <= 8) hipLaunchKernelGGL((k_lstm_seq<8>), dim3(a.nseq * (a.B / cb)), dim3(bs), lds, st, a, cb);
    else if (a.T <= 32) hipLaunchKernelGGL((k_lstm_seq<32>), dim3(a.nseq * (a.B / cb)), dim3(bs), lds, st, a, cb);
    else hipLaunchKernelGGL((k_lstm_seq<64>), dim3(a.nseq * (a.B / cb)), dim3(bs), lds, st, a, cb);      // lstm_seq_fits: T <= 64
}

// BPTT over the whole s-sequence, one workgroup per group of CB columns (same arithmetic as T calls of k_lstm_bwd_step); the
// trainable state0's gradient (a sum over ALL columns, ascending b) is folded by k_state0_grad afterwards.
// PF: the seven stashed values of EVERY time step are requested before the loop (T <= 8: 56 registers) instead of one step ahead -- their round trip
// was longer than a step's arithmetic (r03: 4.6 us per time step)
template <bool PF>
__global__ __launch_bounds__(1024) void k_lstm_bwd_seq(LstmBwdArgs A, int CB) {
    extern __shared__ float lds[];
    const int H = A.H, B = A.B, TB = A.TB, N = 4 * H, per = H * CB, b0 = blockIdx.x * CB;
    const int NP = N + 1;              // padded row stride: lanes of one wave hold different rows j of Wh at the same n -- stride 4H put all of them on ONE bank (8-way conflict on every read of the 128-deep chain)
    float* Wh_s = lds;                 // [H][4H + 1]
    float* dG_s = Wh_s + H * NP;       // [4H][CB]
    float* dhn_s = dG_s + N * CB;      // [H*CB]
    float* dcn_s = dhn_s + per;        // [H*CB]
    for (int i = threadIdx.x; i < H * N; i += blockDim.x) Wh_s[(i / N) * NP + i % N] = A.Wh[i];
    for (int e = threadIdx.x; e < per; e += blockDim.x) { dhn_s[e] = 0.0f; dcn_s[e] = 0.0f; }
    __syncthreads();
    // one (u, column) element per thread (per <= blockDim by construction); the seven stashed values of step t-1 are requested
    // while step t's dh chain runs, so their latency is off the serial path
    const int e = threadIdx.x; const bool on = e < per;
    const int u = on ? e / CB : 0, bl = on ? e - u * CB : 0;
    struct St { float ig, fg, gg, og, tc, cprev, dH; };
    auto fetch = [&](int t) { St s; const size_t k = (size_t)t * B + b0 + bl;
        s.ig = A.gates[(size_t)u * TB + k]; s.fg = A.gates[(size_t)(H + u) * TB + k]; s.gg = A.gates[(size_t)(2 * H + u) * TB + k]; s.og = A.gates[(size_t)(3 * H + u) * TB + k];
        s.tc = A.tc[(size_t)u * TB + k]; s.cprev = A.cprev[(size_t)u * TB + k]; s.dH = A.dH[(size_t)u * TB + k]; return s; };
    St all[PF ? 8 : 1];
    if constexpr (PF) {
#pragma unroll
        for (int t = 0; t < 8; t++) if (t < A.T) all[t] = fetch(t);
    }
    St nx; if constexpr (!PF) nx = fetch(A.T - 1);
#pragma unroll
    for (int tt = 0; tt < (PF ? 8 : 1 << 30); tt++) {
        const int t = (PF ? 7 : A.T - 1) - tt;
        if (t < 0) break;
        if (PF && t >= A.T) continue;
        St c; if constexpr (PF) c = all[PF ? t : 0]; else c = nx;
        if (on) {
            const size_t k = (size_t)t * B + b0 + bl;
            const float ig = c.ig, fg = c.fg, gg = c.gg, og = c.og, tc = c.tc, cprev = c.cprev;
            const float dhn = t == A.T - 1 ? 0.0f : dhn_s[e], dcn = t == A.T - 1 ? 0.0f : dcn_s[e];
            const float dh = c.dH + dhn;
            const float dov = dh * tc; const float t1 = dh * og; const float t2 = tc * tc; const float t3 = 1.0f - t2; const float t4 = t1 * t3; const float dc = dcn + t4;
            const float di = dc * gg, df = dc * cprev, dgc = dc * ig; dcn_s[e] = dc * fg;
            const float a1 = di * ig, a2 = 1.0f - ig, v0 = a1 * a2;
            const float b1 = df * fg, b2 = 1.0f - fg, v1 = b1 * b2;
            const float c1 = gg * gg, c2 = 1.0f - c1, v2 = dgc * c2;
            const float d1 = dov * og, d2 = 1.0f - og, v3 = d1 * d2;
            A.dG[(size_t)u * TB + k] = v0; A.dG[(size_t)(H + u) * TB + k] = v1; A.dG[(size_t)(2 * H + u) * TB + k] = v2; A.dG[(size_t)(3 * H + u) * TB + k] = v3;
            dG_s[u * CB + bl] = v0; dG_s[(H + u) * CB + bl] = v1; dG_s[(2 * H + u) * CB + bl] = v2; dG_s[(3 * H + u) * CB + bl] = v3;
        }
        if constexpr (!PF) { if (t > 0) nx = fetch(t - 1); }
        __syncthreads();
        if (on) {                                                  // dh_{t-1}[j][b] = sum_n dG[n][t,b] Wh[j][n], n ascending  (j == u)
            float acc = 0.0f;
#pragma unroll 8
            for (int n = 0; n < N; n++) acc = fmaf(dG_s[n * CB + bl], Wh_s[u * NP + n], acc);
            dhn_s[e] = acc;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < per; e += blockDim.x) { const int u = e / CB, bl = e - u * CB; A.dhn[u * B + b0 + bl] = dhn_s[e]; A.dcn[u * B + b0 + bl] = dcn_s[e]; }
}
__global__ void k_state0_grad(int H, int B, const float* __restrict__ dhn, const float* __restrict__ dcn, float* __restrict__ g_h0, float* __restrict__ g_c0) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= H) return;
    float sh = 0.0f, sc = 0.0f;
    for (int b = 0; b < B; b++) { sh = sh + dhn[u * B + b]; sc = sc + dcn[u * B + b]; }      // ascending b
    g_h0[u] = sh; g_c0[u] = sc;
}
void launch_lstm_bwd_seq(hipStream_t st, const LstmBwdArgs& a) {
    const int cb = lstm_cb(a.H, a.B);
    const size_t lds = ((size_t)a.H * (4 * a.H + 1) + 6 * (size_t)a.H * cb) * sizeof(float);
    int bs = ((a.H * cb + 63) / 64) * 64; if (bs > 1024) bs = 1024;
    if (a.T <= 8) hipLaunchKernelGGL((k_lstm_bwd_seq<true>), dim3(a.B / cb), dim3(bs), lds, st, a, cb);
    else hipLaunchKernelGGL((k_lstm_bwd_seq<false>), dim3(a.B / cb), dim3(bs), lds, st, a, cb);
    hipLaunchKernelGGL(k_state0_grad, dim3((a.H + 63) / 64), dim3(64), 0, st, a.H, a.B, a.dhn, a.dcn, a.g_h0, a.g_c0);
}

// ------------------------------------------------------------------ recurrent TD: targets, masked Huber / B / T, dL/dQ  (src/solver.jl:259-282)
__device__ __forceinline__ float head_at(const HeadSrc& h, int n, int col) { return h.p[(size_t)n * h.ld + col]; }
__device__ __forceinline__ void q_col(int nA, int dueling, const HeadSrc& val, const HeadSrc& adv, int col, float* q, float* vout, float* araw) {
    for (int a = 0; a < nA; a++) araw[a] = head_at(adv, a, col);
    if (!dueling) { for (int a = 0; a < nA; a++) q[a] = araw[a]; *vout = 0.0f; return; }
    const float v = head_at(val, 0, col); *vout = v;
    float sum = araw[0];
    for (int a = 1; a < nA; a++) sum = sum + araw[a];
    const float mean = sum / (float)nA;
    for (int a = 0; a < nA; a++) q[a] = (v + araw[a]) - mean;
}
__global__ __launch_bounds__(1024) void k_td_drqn(TdDrqnArgs A) {
    extern __shared__ float hl[];   // T*B Huber terms
    const int B = A.B, T = A.T, TB = T * B, nA = A.nA;
    const float invT = 1.0f / (float)T;
    for (int k = threadIdx.x; k < TB; k += blockDim.x) {
        float q[DQN_MAX_ACTIONS], qt[DQN_MAX_ACTIONS], araw[DQN_MAX_ACTIONS], vraw;
        q_col(nA, A.dueling, A.tg_val, A.tg_adv, k, qt, &vraw, araw);
        int best = 0;
        if (A.double_q) { q_col(nA, A.dueling, A.on_val, A.on_adv, TB + k, q, &vraw, araw); for (int a = 1; a < nA; a++) if (q[a] > q[best]) best = a; }
        else for (int a = 1; a < nA; a++) if (qt[a] > qt[best]) best = a;
        float qsp = qt[0]; for (int a = 1; a < nA; a++) if (a == best) qsp = qt[a];
        const float t1 = 1.0f - A.done[k]; const float t2 = t1 * A.gamma; const float t3 = t2 * qsp; const float y = A.r[k] + t3;
        q_col(nA, A.dueling, A.on_val, A.on_adv, k, q, &vraw, araw);
        const int act = A.a[k]; float qsa = q[0];
        for (int a = 1; a < nA; a++) if (a == act) qsa = q[a];
        const float td = qsa - y; A.td[k] = td; const float m = A.mask[k];
        const float x = m * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
        hl[k] = (0.5f * qd) * qd + lin;
        const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        const float g = ((invT / (float)B) * cl) * m;
        if (A.dueling) {
            A.d_val[k] = dact_f(g, vraw, A.on_val.act);
            const float gm = g / (float)nA;
            for (int a = 0; a < nA; a++) A.d_adv[(size_t)a * TB + k] = dact_f((a == act ? g : 0.0f) - gm, araw[a], A.on_adv.act);
        } else
            for (int a = 0; a < nA; a++) A.d_adv[(size_t)a * TB + k] = dact_f(a == act ? g : 0.0f, araw[a], A.on_adv.act);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float loss = 0.0f;
        for (int t = 0; t < T; t++) { float lsum = 0.0f; for (int b = 0; b < B; b++) lsum = lsum + hl[t * B + b]; loss = loss + lsum / (float)B; }
        A.st->loss = loss / (float)T;
        A.st->step = A.st->step + 1;
    }
}
void launch_td_drqn(hipStream_t st, const TdDrqnArgs& a) {
    int bs = ((a.T * a.B + 63) / 64) * 64; if (bs > 1024) bs = 1024;
    hipLaunchKernelGGL(k_td_drqn, dim3(1), dim3(bs), (size_t)a.T * a.B * sizeof(float), st, a);
}

// policy state helper: dst[u][b] = src[u]  (Flux.reset!: state <- state0 broadcast over the streams)
__global__ void k_bcast_state(const float* __restrict__ src, int H, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H * n) dst[i] = src[i / n];
}
void launch_bcast_state(hipStream_t st, const float* src, int H, int n, float* dst) {
    hipLaunchKernelGGL(k_bcast_state, dim3((H * n + 255) / 256), dim3(256), 0, st, src, H, n, dst);
}
