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
