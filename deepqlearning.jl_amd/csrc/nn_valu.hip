// nn_valu.hip -- canonical-order VALU kernels of the batch_train! inner loop (src/solver.jl:191-236).
// Each output element is ONE k-ascending fp32 fmaf chain per plan chunk -- the numerics gfx950's fp32 MFMA
// produces -- so these kernels are bit-identical to nn_mfma.hip and serve (a) every shape the MFMA tiles do not
// cover (N or batch not a multiple of 16, tiny heads) and (b) as the on-device cross-check of the MFMA path.
// Threads are laid out with the batch column fastest, so activation loads are coalesced 256-B lines and the
// weight is a wave-uniform (scalar) load.
#include "common.h"
#include "valu_tasks.h"
#include "adam_body.h"
#include "gather_body.h"

// ------------------------------------------------------------------ split-K reduction epilogues
// mode 0: forward      Y = act(sum_s part + bias[n])             (n = e / per_n)
// mode 1: dX           out = dact(sum_s part (+ addend), ysrc)   (element e=(feat,b); ysrc has leading dim ldy)
// mode 2: dW / db      out = sum_s part
__global__ void k_reduce(const float* __restrict__ part, int S, size_t elems, int mode, const float* __restrict__ bias, int per_n, int act,
                         const float* __restrict__ addend, const float* __restrict__ ysrc, int B, int ldy, float* __restrict__ out) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= elems) return;
    float tot = slab_sum(part + e, elems, S);
    if (mode == 0) tot = act_f(tot + bias[e / per_n], act);
    else if (mode == 1) {
        if (addend) tot = addend[e] + tot;
        if (ysrc) tot = dact_f(tot, ysrc[(e / B) * ldy + (e % B)], act);
    }
    out[e] = tot;
}
// several reductions in one launch (segment table in device memory).  mode 1 with S2 > 0 is the dueling JOIN:
// out = dact( (sum of the first S slabs) + (sum of the next S2 slabs) ) == dX_val + dX_adv, then act' of the base output.
__global__ __launch_bounds__(256) void k_reduce_multi(const RSeg* __restrict__ segs, int nseg) {
    int si = 0;
    while (si + 1 < nseg && blockIdx.x >= segs[si + 1].first_block) si++;
    const RSeg& R = segs[si];
    const size_t e = (size_t)(blockIdx.x - R.first_block) * 256 + threadIdx.x;
    if (e >= R.elems) return;
    float tot = slab_sum(R.part + e, (size_t)R.elems, R.S);
    if (R.S2 > 0) {
        const float t2 = slab_sum(R.part + (size_t)R.S * R.elems + e, (size_t)R.elems, R.S2);
        tot = tot + t2;
    }
    // element-index decodes: through reciprocals while the segment is small (fdiv_*, common.h: exact below 2^21 -- the 64-bit divisions written here before cost
    // ~100 instructions each, three per element of a launch that is all prologue), plain 64-bit division otherwise
    const bool small = R.elems < (1ull << 21);
    auto dv = [&](size_t x, int d, size_t& q, size_t& r) { if (small) { int qi, ri; fdiv_qr((int)x, fdiv_of(d), qi, ri); q = (size_t)qi; r = (size_t)ri; } else { q = x / (size_t)d; r = x - q * (size_t)d; } };
    if (R.mode == 0) { size_t q, r; dv(e, R.per_n, q, r); tot = act_f(tot + R.bias[q], R.act); }
    else if (R.mode == 1) {
        if (R.addend) tot = R.addend[e] + tot;
        if (R.ysrc) { size_t q, r; dv(e, R.B, q, r); tot = dact_f(tot, R.ysrc[q * R.ldy + r], R.act); }
    }
    R.out[e] = tot;
    if (R.outT) { size_t feat, col; dv(e, R.ncolsT, feat, col); size_t nf, rr; dv((size_t)R.elems, R.ncolsT, nf, rr); R.outT[col * nf + feat] = tot; }
}
void launch_reduce_multi(hipStream_t st, const RSeg* segs_dev, int nseg, unsigned total_blocks) {
    hipLaunchKernelGGL(k_reduce_multi, dim3(total_blocks), dim3(256), 0, st, segs_dev, nseg);
}
static void launch_reduce(hipStream_t st, const float* part, int S, size_t elems, int mode, const float* bias, int per_n, int act,
                          const float* addend, const float* ysrc, int B, int ldy, float* out) {
    hipLaunchKernelGGL(k_reduce, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, part, S, elems, mode, bias, per_n, act, addend, ysrc, B, ldy, out);
}
// exported for the MFMA path
void launch_reduce_pub(hipStream_t st, const float* part, int S, size_t elems, int mode, const float* bias, int per_n, int act,
                       const float* addend, const float* ysrc, int B, int ldy, float* out) {
    launch_reduce(st, part, S, elems, mode, bias, per_n, act, addend, ysrc, B, ldy, out);
}

__global__ void k_valu_fwd(LayerDev L, const float* __restrict__ P, const float* __restrict__ X, int ldx, int col0, int ncols, int S, int kc,
                           float* __restrict__ out) {
    valu_fwd_body(L, P, X, ldx, col0, ncols, S, kc, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
void launch_valu_fwd(hipStream_t st, const LayerDev& L, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, float* partials) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    const size_t per_s = (size_t)L.N * L.npos * ncols, tot = per_s * S;
    hipLaunchKernelGGL(k_valu_fwd, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, P, X, ldx, col0, ncols, S, kc, S == 1 ? Y : partials);
    if (S > 1) launch_reduce(st, partials, S, per_s, 0, P + L.b_off, L.npos * ncols, L.act, nullptr, nullptr, 0, 0, Y);
}

__global__ void k_valu_dw(LayerDev L, const float* __restrict__ X, int ldx, const float* __restrict__ dpre, int B, int S, int kc, float* __restrict__ out) {
    valu_dw_body(L, X, ldx, dpre, B, S, kc, out, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// G layout: dW at w_off ([K][N]) immediately followed by db at b_off ([N]) == rows 0..K of a (K+1) x N matrix.
void launch_valu_dw(hipStream_t st, const LayerDev& L, const float* X, int ldx, const float* dpre, int B, float* G, float* partials) {
    const int KK = L.npos * B, S = dqn_nchunks(KK, L.dw_kc), kc = dqn_chunk_len(KK, L.dw_kc);
    const size_t per_s = (size_t)(L.K + 1) * L.N, tot = per_s * S;
    float* dst = G + L.w_off;
    hipLaunchKernelGGL(k_valu_dw, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, X, ldx, dpre, B, S, kc, S == 1 ? dst : partials);
    if (S > 1) launch_reduce(st, partials, S, per_s, 2, nullptr, 1, 0, nullptr, nullptr, 0, 0, dst);
}

__global__ void k_valu_dx(LayerDev L, const float* __restrict__ P, const float* __restrict__ dpre, int B, int S, int kc, float* __restrict__ out,
                          const float* __restrict__ addend, const float* __restrict__ ysrc, int ldy, int act_src) {
    valu_dx_body(L, P, dpre, B, S, kc, out, addend, ysrc, ldy, act_src, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
// One launch for a TABLE of independent small tasks (head forwards of both nets, head dW + dX, ...): the per-launch floor
// (~4.6 us on MI355X) dominates these tiny kernels, so they are batched.  Tasks live in device memory (static schedule).
__global__ __launch_bounds__(256) void k_valu_multi(const VTask* __restrict__ tasks, int ntasks) {
    valu_task_run(tasks, ntasks, blockIdx.x);
}
unsigned valu_task_blocks(const VTask& T) {
    size_t n;
    if (T.kind == 0) n = (size_t)T.L.N * T.L.npos * T.ncols * T.S;
    else if (T.kind == 1) n = (size_t)(T.L.K + 1) * T.L.N * T.S;
    else if (T.kind == 2) n = (size_t)T.L.in_feat * T.B * T.S;
    else n = 1;
    return (unsigned)((n + 255) / 256);
}
// ... with the step's priority block as workgroup 0: its dependent tree levels hide under the task table instead of inside the Adam launch,
// which leaves the Adam launch free to gather the next batch (PreGather)
__global__ __launch_bounds__(256) void k_valu_multi_prio(const VTask* __restrict__ tasks, int ntasks, PrioArgs P, StepState* state) {
    __shared__ __attribute__((aligned(16))) long long sidx[1024 + 128];      // 7808 B of path state + the top 256 nodes (small replays: most of the tree)
    if (blockIdx.x == 0) { prio_block_run(P, state, sidx, (unsigned)sizeof sidx); return; }
    valu_task_run(tasks, ntasks, blockIdx.x - 1);
}
void launch_valu_multi(hipStream_t st, const VTask* tasks_dev, int ntasks, unsigned total_blocks, const PrioArgs* prio, StepState* state) {
    if (prio) hipLaunchKernelGGL(k_valu_multi_prio, dim3(total_blocks + 1), dim3(256), 0, st, tasks_dev, ntasks, *prio, state);
    else hipLaunchKernelGGL(k_valu_multi, dim3(total_blocks), dim3(256), 0, st, tasks_dev, ntasks);
}
void launch_valu_dx(hipStream_t st, const LayerDev& L, const float* P, const float* dpre, int B, float* out, float* partials,
                    const float* addend, const float* ysrc, int ldy, int act_src) {
    const int S = L.kind == DQN_LAYER_DENSE ? dqn_nchunks(L.N, L.dx_kc) : 1, kc = dqn_chunk_len(L.N, L.dx_kc);
    const size_t per_s = (size_t)L.in_feat * B, tot = per_s * S;
    hipLaunchKernelGGL(k_valu_dx, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, L, P, dpre, B, S, kc, S == 1 ? out : partials, addend, ysrc, ldy, act_src);
    if (S > 1) launch_reduce(st, partials, S, per_s, 1, nullptr, 1, act_src, addend, ysrc, B, ldy, out);
}

// ------------------------------------------------------------------ dueling reduce + argmax + Bellman target + TD + Huber + dL/dQ  (K4+K5+K6)
//   Q = (val .+ adv) .- mean(adv, dims=1)                      src/dueling.jl:10
//   best = argmax(Qonline(sp)[:, b]) (first max)               src/solver.jl:212
//   y = r + (1 - done) * gamma * Qtarget(sp)[best]             src/solver.jl:213-217
//   td = Q(s)[a] - y ; loss = sum(huber(w .* td)) / B          src/solver.jl:220-224, src/helpers.jl:14-19
// One workgroup; the loss is summed by one lane in ascending b (canonical order).
__device__ __forceinline__ void q_column(int nA, int dueling, const float* val, const float* adv, int ld, int col, float* q) {
    if (!dueling) { for (int a = 0; a < nA; a++) q[a] = adv[(size_t)a * ld + col]; return; }
    const float v = val[col];
    float sum = adv[col];
    for (int a = 1; a < nA; a++) sum = sum + adv[(size_t)a * ld + col];
    const float mean = sum / (float)nA;
    for (int a = 0; a < nA; a++) q[a] = (v + adv[(size_t)a * ld + col]) - mean;
}
// NMAX is a compile-time bound on n_actions so that the per-lane Q arrays live in registers (runtime-bounded loops over a
// local array put it in scratch: 784 B/lane and ~8 us of this single-workgroup kernel before the change)
template <int NMAX>
__device__ __forceinline__ void q_from_lds(int nA, int dueling, const float* v, const float* a, int ld, int col, float (&q)[NMAX], float* vout, float (&araw)[NMAX]) {
#pragma unroll
    for (int i = 0; i < NMAX; i++) araw[i] = i < nA ? a[i * ld + col] : 0.0f;
    if (!dueling) {
#pragma unroll
        for (int i = 0; i < NMAX; i++) q[i] = araw[i];
        *vout = 0.0f; return;
    }
    const float vv = v[col]; *vout = vv;
    float sum = araw[0];
#pragma unroll
    for (int i = 1; i < NMAX; i++) if (i < nA) sum = sum + araw[i];
    const float mean = sum / (float)nA;
#pragma unroll
    for (int i = 0; i < NMAX; i++) q[i] = (vv + araw[i]) - mean;
}
template <int NMAX>
__global__ __launch_bounds__(1024) void k_td(TdArgs A) {
    extern __shared__ float hl[];   // [B] Huber terms | [B] long long indices | head outputs: on_val[ncon] on_adv[nA][ncon] tg_val[B] tg_adv[nA][B]
    const int B = A.B, nA = A.nA, ncon = A.ncon;
    float* hv_on_val = hl + B;
    float* hv_on_adv = hv_on_val + ncon;
    float* hv_tg_val = hv_on_adv + nA * ncon;
    float* hv_tg_adv = hv_tg_val + B;
    // the batch metadata of this thread's column (blockDim >= B: one column per thread) is requested FIRST -- two dependent round trips
    // (idx -> a, r, done, priority) and the double-precision pow of the IS weight ride under the head reduction below
    const int b_ = threadIdx.x; const bool hasb = b_ < B;
    long long j_ = 0; int act_ = 0; float rew_ = 0.0f, dn_ = 0.0f, w_ = 0.0f;
    if (hasb) {
        const float total = A.tree[1]; const long long size = A.st->size;
        j_ = A.take_pre ? A.idx_pre[b_] : A.idx[b_];
        if (A.take_pre) { A.idx_mut[b_] = j_; if (b_ == 0 && A.st->pre_valid != 2) A.st->err = 3; }      // publish the pre-drawn indices (priority update, parity API)
        act_ = A.a[j_]; rew_ = A.r[j_]; dn_ = (float)A.done[j_];
        const float p = A.tree[A.cap2 + j_] / total; const float xw = (float)size * p;
        w_ = (float)pow((double)xw, -(double)A.prio_beta);               // IS weight, ...replay.jl:101-102
    }
    // phase 1: every lane of the workgroup finishes head outputs (split-K slabs of the head layers are reduced here)
    {
        const int n_on = (A.dueling ? 1 : 0) * ncon, n_oa = nA * ncon, n_tv = (A.dueling ? 1 : 0) * B, n_ta = nA * B;
        for (int e = threadIdx.x; e < n_on + n_oa + n_tv + n_ta; e += blockDim.x) {
            if (e < n_on) hv_on_val[e] = head_val(A.on_val, 0, e);
            else if (e < n_on + n_oa) { const int i = e - n_on; hv_on_adv[i] = head_val(A.on_adv, i / ncon, i % ncon); }
            else if (e < n_on + n_oa + n_tv) { const int i = e - n_on - n_oa; hv_tg_val[i] = head_val(A.tg_val, 0, i); }
            else { const int i = e - n_on - n_oa - n_tv; hv_tg_adv[i] = head_val(A.tg_adv, i / B, i % B); }
        }
    }
    __syncthreads();
    const float invB = 1.0f / (float)B;
    if (hasb) {
        const int b = b_; const int act = act_; const float rew = rew_, dn = dn_, w = w_;
        A.w_is[b] = w;
        float q[NMAX], qt[NMAX], araw[NMAX], vraw;
        q_from_lds<NMAX>(nA, A.dueling, hv_tg_val, hv_tg_adv, B, b, qt, &vraw, araw);
#pragma unroll
        for (int a = 0; a < NMAX; a++) if (a < nA) A.q_tg_sp[(size_t)b * nA + a] = qt[a];
        int best = 0; float qsp;
        if (A.double_q) {
            q_from_lds<NMAX>(nA, A.dueling, hv_on_val, hv_on_adv, ncon, B + b, q, &vraw, araw);
#pragma unroll
            for (int a = 0; a < NMAX; a++) if (a < nA) A.q_on_sp[(size_t)b * nA + a] = q[a];
            float bq = q[0];
#pragma unroll
            for (int a = 1; a < NMAX; a++) if (a < nA && q[a] > bq) { bq = q[a]; best = a; }
        } else {
#pragma unroll
            for (int a = 0; a < NMAX; a++) if (a < nA) A.q_on_sp[(size_t)b * nA + a] = qt[a];
            float bq = qt[0];
#pragma unroll
            for (int a = 1; a < NMAX; a++) if (a < nA && qt[a] > bq) { bq = qt[a]; best = a; }
        }
        qsp = qt[0];
#pragma unroll
        for (int a = 1; a < NMAX; a++) if (a == best) qsp = qt[a];
        A.best[b] = best;
        const float t1 = 1.0f - dn; const float t2 = t1 * A.gamma; const float t3 = t2 * qsp; const float y = rew + t3;
        A.ytarget[b] = y;
        q_from_lds<NMAX>(nA, A.dueling, hv_on_val, hv_on_adv, ncon, b, q, &vraw, araw);
        float qsa = q[0];
#pragma unroll
        for (int a = 0; a < NMAX; a++) if (a < nA) { A.q_on_s[(size_t)b * nA + a] = q[a]; if (a == act) qsa = q[a]; }
        const float td = qsa - y; A.td[b] = td;
        const float x = w * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
        hl[b] = (0.5f * qd) * qd + lin;
        const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        const float g = (invB * cl) * w;
        if (A.dueling) {
            A.d_val[b] = dact_f(g, vraw, A.on_val.act);
            const float gm = g / (float)nA;
#pragma unroll
            for (int a = 0; a < NMAX; a++) if (a < nA) A.d_adv[(size_t)a * B + b] = dact_f((a == act ? g : 0.0f) - gm, araw[a], A.on_adv.act);
        } else {
#pragma unroll
            for (int a = 0; a < NMAX; a++) if (a < nA) A.d_adv[(size_t)a * B + b] = dact_f(a == act ? g : 0.0f, araw[a], A.on_adv.act);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lsum = 0.0f;
        for (int b = 0; b < B; b++) lsum = lsum + hl[b];
        A.st->loss = lsum / (float)B;
        A.st->step = A.st->step + 1;                 // read by k_adam (beta-power slot) later in this step
        if (A.bump_sample_ctr) A.st->sample_ctr = A.st->sample_ctr + 1;   // the fused sample+gather kernel cannot bump it itself
    }
    // update_priorities! runs as k_update_priorities on a forked graph branch (it only needs td; it overlaps the backward pass)
}
void launch_td(hipStream_t st, const TdArgs& a) {
    int bs = ((a.B + 63) / 64) * 64; if (bs < 512) bs = 512; if (bs > 1024) bs = 1024;
    const size_t lds = (size_t)a.B * sizeof(float) + (size_t)(1 + a.nA) * (a.ncon + a.B) * sizeof(float);
    if (a.nA <= 8) hipLaunchKernelGGL((k_td<8>), dim3(1), dim3(bs), lds, st, a);
    else if (a.nA <= 32) hipLaunchKernelGGL((k_td<32>), dim3(1), dim3(bs), lds, st, a);
    else hipLaunchKernelGGL((k_td<DQN_MAX_ACTIONS>), dim3(1), dim3(bs), lds, st, a);
}

// ------------------------------------------------------------------ small batches: head forwards + TD + head dX in ONE launch, a workgroup per batch column
// Replaces three dependent launches (head forwards of both nets as a task table, the single-workgroup k_td, the head dX tasks): 19.7 us -> one
// launch at B = 32.  Every value follows the canonical order of the kernels it replaces:
//   head forward   per plan chunk s: acc = +0; k ascending: acc = fma(x[k], W[k][n], acc); chunk sums added in ascending order; + bias; activation
//   TD             exactly k_td's per-column arithmetic (dueling (v + a) - mean, first-max argmax, r + ((1 - done) * gamma) * q, Huber, dL/dQ)
//   head dX        acc = +0; n ascending: acc = fma(dpre[n][b], W[k][n], acc); at a join dX_val + dX_adv; then act' of the producing layer
// Workgroup b stages its three input columns (s_b, sp_b of the online net; sp_b of the target net) of each head in LDS, so the strided 4-byte
// column reads happen once.  The loss is folded later from the per-column Huber terms (`hl`, valu_tasks.h kind 3); block 0 ticks the step counters.
// LDS index padding: 4 floats per 32.  The lanes of phase 2 differ in the chunk index, i.e. by multiples of kc (32) elements -- unpadded
// they would all hit one bank; with a 36-float pitch 16 lanes reading 16 bytes each cover all 64 banks once, and 32-chunks stay 16-byte aligned.
#define PADI(i) ((i) + (((i) >> 5) << 2))
// x / d for small non-negative ints through one multiply: floor((x + 0.5) * (1 / d)) is exact while x * 2^-23 << 0.5 / d (x < 2^16 here);
// a runtime integer division costs ~40 instructions, and a lone wave pays every instruction in full
__device__ __forceinline__ int qdiv(int x, float rcp) { return (int)(((float)x + 0.5f) * rcp); }
// phase 1 of one head: request the three input columns (k = tid, tid + 256) and the first three weight float4s of this lane
__device__ __forceinline__ void head_issue(const HeadLayer& L, int B, int b, int double_q, int stage_w, int tid, float (&xv)[6], f32x4v (&wv)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int net = c == 2 ? 1 : 0, col = L.c0[net] + b + (c == 1 ? B : 0);
        const bool on = !(c == 1 && !double_q);
        const float* xt = L.XT[net];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int k = tid + 256 * j; xv[2 * c + j] = 0.0f;
            if (on && k < L.K) xv[2 * c + j] = xt ? xt[(size_t)col * L.K + k] : L.X[net][(size_t)k * L.ldx[net] + col];
        }
    }
    const int half = (L.K * L.N) >> 2;                     // float4s per net
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int q = tid + 256 * u; wv[u] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        if (stage_w && q < 2 * half) { const int net = q >= half ? 1 : 0; wv[u] = reinterpret_cast<const f32x4v*>(L.W[net])[q - net * half]; }
    }
}
// the weights are staged TRANSPOSED, wt[net][n][k] (k contiguous), so that a chain reads its 32 weights as eight 16-byte LDS loads
__device__ __forceinline__ void head_put_w4(const HeadLayer& L, float rN, int q, int half, const f32x4v& w, float* wt) {
    const int net = q >= half ? 1 : 0; const int e0 = 4 * (q - net * half);                      // first of 4 consecutive elements k * N + n
    const float v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; i++) { const int e = e0 + i, k = qdiv(e, rN), n = e - k * L.N; const int d = (net * L.N + n) * L.K + k; wt[PADI(d)] = v[i]; }
}
__device__ __forceinline__ void head_commit(const HeadLayer& L, int B, int b, int double_q, int stage_w, int tid, const float (&xv)[6], const f32x4v (&wv)[3], float* xs, float* wt) {
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int j = 0; j < 2; j++) { const int k = tid + 256 * j; if (k < L.K) { const int i = c * L.K + k; xs[PADI(i)] = xv[2 * c + j]; } }
    const int half = (L.K * L.N) >> 2; const float rN = 1.0f / (float)L.N;
#pragma unroll
    for (int u = 0; u < 3; u++) { const int q = tid + 256 * u; if (stage_w && q < 2 * half) head_put_w4(L, rN, q, half, wv[u], wt); }
    // shapes beyond the straight-line slots (K > 512 or more than 768 weight float4s): plain rounds
    for (int c = 0; c < 3; c++) {
        if (c == 1 && !double_q) continue;
        const int net = c == 2 ? 1 : 0, col = L.c0[net] + b + (c == 1 ? B : 0);
        for (int k = tid + 512; k < L.K; k += 256) { const int i = c * L.K + k; xs[PADI(i)] = L.XT[net] ? L.XT[net][(size_t)col * L.K + k] : L.X[net][(size_t)k * L.ldx[net] + col]; }
    }
    if (stage_w) for (int q = tid + 768; q < 2 * half; q += 256) { const int net = q >= half ? 1 : 0; head_put_w4(L, rN, q, half, reinterpret_cast<const f32x4v*>(L.W[net])[q - net * half], wt); }
}
// The argument record lives in device memory (built once per engine): 350 bytes of kernel argument kept in SGPRs for the whole kernel
// spilled ~700 lane moves; fields are fetched with scalar loads where they are used instead.  The kernel is written for a SHORT instruction
// stream: its 32 workgroups are lone waves on cold instruction caches, so every instruction is paid in full -- the two heads share one copy of
// each phase (per-lane selects instead of two inlined bodies), indices are decoded with multiplies, LDS is read 16 bytes at a time.
__global__ __launch_bounds__(256) void k_head_td(const HeadTdArgs* __restrict__ Ap, int bump_sample_ctr, int take_pre) {
    extern __shared__ float hs[];
    const HeadTdArgs& A = *Ap;
    const int B = A.B, nA = A.nA, b = blockIdx.x, tid = threadIdx.x;
    const int double_q = A.double_q, stage_w = A.stage_w;
    const bool has_val = A.dueling != 0;                   // plain network: only the `adv` record is populated
    const int Kv = has_val ? A.val.K : 0, Nv = has_val ? A.val.N : 0, Sv = has_val ? A.val.S : 0, Ka = A.adv.K, Na = A.adv.N, Sa = A.adv.S;
    // LDS carve-up (padded, see PADI): input columns | transposed weights of both nets | chunk sums | head outputs | Q columns | head gradients
    float* p = hs;
    float* const xs_v = p; p += PADI(3 * Kv);
    float* const xs_a = p; p += PADI(3 * Ka);
    float* const wt_v = p; p += stage_w ? PADI(2 * Kv * Nv) : 0;
    float* const wt_a = p; p += stage_w ? PADI(2 * Ka * Na) : 0;
    float* const part_v = p; p += 3 * Nv * Sv;
    float* const part_a = p; p += 3 * Na * Sa;
    float* const hv_v = p; p += 3 * Nv;
    float* const hv_a = p; p += 3 * Na;
    float* const qs = p; p += 3 * nA;                      // Q[slot][a]
    float* const dq = p;                                   // [1 + nA]: head pre-activation gradients of this column (val first)
    if (A.dbg == 9) return;
    if (take_pre && b == 0) {      // the batch in the arena was gathered by the previous step's Adam launch: publish its indices (priority block, parity API)
        for (int i = tid; i < B; i += 256) A.idx[i] = A.idx_pre[i];
        if (tid == 0 && A.st->pre_valid != 2) A.st->err = 3;
    }
    int act_ = 0; float rew_ = 0.0f, dn_ = 0.0f, w_ = 0.0f;
    if (tid == 0) { act_ = A.bm_a[b]; rew_ = A.bm_r[b]; dn_ = A.bm_done[b]; w_ = A.bm_w[b]; }     // get_batch scalars + IS weight (gather launch)
    // biases of the outputs this lane finishes (o = slot * N + n; adv outputs first, then val): requested now, consumed after phase 2
    const int no_a = 3 * Na, no_v = 3 * Nv;
    float bias_ = 0.0f;
    if (tid < no_a) bias_ = A.adv.bias[tid >= 2 * Na ? 1 : 0][tid % Na];
    else if (tid < no_a + no_v) { const int o = tid - no_a; bias_ = A.val.bias[o >= 2 * Nv ? 1 : 0][o % Nv]; }
    // phase 1: input columns (slot c: 0 = online s_b, 1 = online sp_b, 2 = target sp_b; contiguous runs of the transposed copy when there is
    // one) and the contiguous weights: every load of both heads is issued before the first LDS store -- one round trip at K <= 512
    {
        float xv[6], xa[6]; f32x4v wv[3], wa[3];
        if (has_val) head_issue(A.val, B, b, double_q, stage_w, tid, xv, wv);
        head_issue(A.adv, B, b, double_q, stage_w, tid, xa, wa);
        if (has_val) head_commit(A.val, B, b, double_q, stage_w, tid, xv, wv, xs_v, wt_v);
        head_commit(A.adv, B, b, double_q, stage_w, tid, xa, wa, xs_a, wt_a);
    }
    __syncthreads();
    if (A.dbg == 1) return;
    // phase 2: chunk chains, one (head, column slot c, output n, chunk s) per lane; adv items first: it = (c * N + n) * S + s
    {
        const int ia = 3 * Na * Sa, iv = 3 * Nv * Sv, kcv = has_val ? A.val.kc : 1, kca = A.adv.kc;
        const float rSa = 1.0f / (float)Sa, rNa = 1.0f / (float)Na, rSv = has_val ? 1.0f / (float)Sv : 1.0f, rNv = has_val ? 1.0f / (float)Nv : 1.0f;
        for (int it = tid; it < ia + iv; it += 256) {
            const bool v = it >= ia; const int i2 = v ? it - ia : it;
            const int K = v ? Kv : Ka, N = v ? Nv : Na, S = v ? Sv : Sa, kc = v ? kcv : kca;
            const float* xs = v ? xs_v : xs_a; const float* wt = v ? wt_v : wt_a; float* part = v ? part_v : part_a;
            const int cn = qdiv(i2, v ? rSv : rSa), s = i2 - cn * S, c = qdiv(cn, v ? rNv : rNa), n = cn - c * N;
            if (c == 1 && !double_q) { part[i2] = 0.0f; continue; }
            const int net = c == 2 ? 1 : 0;
            const int k0 = s * kc, k1 = min(K, k0 + kc);
            const int xi = c * K + k0, wi = (net * N + n) * K + k0;
            float acc = 0.0f;
            if (stage_w && k1 - k0 == 32 && ((xi | wi) & 31) == 0) {      // a whole aligned chunk: sixteen 16-byte LDS loads, then the k-ascending chain
                const f32x4v* xp = reinterpret_cast<const f32x4v*>(xs + PADI(xi)); const f32x4v* wp = reinterpret_cast<const f32x4v*>(wt + PADI(wi));
                f32x4v xq[8], wq[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { xq[u] = xp[u]; wq[u] = wp[u]; }
#pragma unroll
                for (int u = 0; u < 8; u++) { acc = fmaf(xq[u].x, wq[u].x, acc); acc = fmaf(xq[u].y, wq[u].y, acc); acc = fmaf(xq[u].z, wq[u].z, acc); acc = fmaf(xq[u].w, wq[u].w, acc); }
            } else if (stage_w) {
                for (int k = 0; k < k1 - k0; k++) acc = fmaf(xs[PADI(xi + k)], wt[PADI(wi + k)], acc);
            } else {
                const float* W = (v ? A.val.W[net] : A.adv.W[net]) + n;
                for (int k = k0; k < k1; k++) acc = fmaf(xs[PADI(c * K + k)], W[(size_t)k * N], acc);
            }
            part[i2] = acc;
        }
    }
    __syncthreads();
    // chunk sums added in ascending order, + bias, activation: output o = c * N + n (adv outputs first)
    for (int o0 = tid; o0 < no_a + no_v; o0 += 256) {
        const bool v = o0 >= no_a; const int o = v ? o0 - no_a : o0;
        const int N = v ? Nv : Na, S = v ? Sv : Sa;
        const float* pp = (v ? part_v : part_a) + o * S;
        float tot = pp[0];
        for (int s = 1; s < S; s++) tot = tot + pp[s];
        float bias = bias_;
        if (o0 >= 256) { const int c = o / N, n = o - c * N; bias = (v ? A.val.bias[c == 2 ? 1 : 0] : A.adv.bias[c == 2 ? 1 : 0])[n]; }
        (v ? hv_v : hv_a)[o] = act_f(tot + bias, v ? A.val.act : A.adv.act);
    }
    __syncthreads();
    if (A.dbg == 2) return;
    // phase 3a: the three Q columns (slot c per lane): Q = (val .+ adv) .- mean(adv), src/dueling.jl:10; parity copies go out from here
    if (tid < 3) {
        const int c = tid; const float* ar = hv_a + c * Na;
        float* qg = c == 0 ? A.q_on_s : (c == 1 ? A.q_on_sp : A.q_tg_sp);
        if (c != 1 || double_q) {
            if (!has_val) { for (int a = 0; a < nA; a++) { const float q = ar[a]; qs[c * nA + a] = q; qg[(size_t)b * nA + a] = q; if (c == 2 && !double_q) A.q_on_sp[(size_t)b * nA + a] = q; } }
            else {
                const float vv = hv_v[c];
                float sum = ar[0];
                for (int a = 1; a < nA; a++) sum = sum + ar[a];
                const float mean = sum / (float)nA;
                for (int a = 0; a < nA; a++) { const float q = (vv + ar[a]) - mean; qs[c * nA + a] = q; qg[(size_t)b * nA + a] = q; if (c == 2 && !double_q) A.q_on_sp[(size_t)b * nA + a] = q; }
            }
        }
    }
    __syncthreads();
    // phase 3b: this column's TD (one lane; k_td's arithmetic)
    if (tid == 0) {
        const float invB = 1.0f / (float)B;
        const int act = act_; const float rew = rew_, dn = dn_, w = w_;
        A.w_is[b] = w;
        const float* qsel = double_q ? qs + nA : qs + 2 * nA;          // argmax over the online net's Q(sp) (double-Q) or the target net's
        int best = 0; float bq = qsel[0];
        for (int a = 1; a < nA; a++) { const float q = qsel[a]; if (q > bq) { bq = q; best = a; } }      // first max (Julia argmax)
        const float qsp = qs[2 * nA + best];
        A.best[b] = best;
        const float t1 = 1.0f - dn; const float t2 = t1 * A.gamma; const float t3 = t2 * qsp; const float y = rew + t3;
        A.ytarget[b] = y;
        const float qsa = qs[act];
        const float td = qsa - y; A.td[b] = td;
        const float x = w * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
        A.hl[b] = (0.5f * qd) * qd + lin;
        const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        const float g = (invB * cl) * w;
        const int act_a = A.adv.act;
        if (has_val) {
            const float dv = dact_f(g, hv_v[0], A.val.act); dq[0] = dv; A.val.dpre[b] = dv;
            const float gm = g / (float)nA;
            for (int a = 0; a < nA; a++) { const float d = dact_f((a == act ? g : 0.0f) - gm, hv_a[a], act_a); dq[1 + a] = d; A.adv.dpre[(size_t)a * B + b] = d; }
        } else {
            for (int a = 0; a < nA; a++) { const float d = dact_f(a == act ? g : 0.0f, hv_a[a], act_a); dq[1 + a] = d; A.adv.dpre[(size_t)a * B + b] = d; }
        }
        if (b == 0) {
            A.st->step = A.st->step + 1;                 // read by k_adam (beta-power slot) later in this step
            if (bump_sample_ctr) A.st->sample_ctr = A.st->sample_ctr + 1;
        }
    }
    __syncthreads();
    if (A.dbg == 3) return;
    // phase 4: dX of the head layers for column b (acc = +0; n ascending: acc = fma(dpre[n], W[k][n], acc)), the join (dX_val + dX_adv), then act'
    // of the producing layer, whose activation y[k][b] is the staged input column of slot 0
    if (A.join) {
        const int act_src = A.adv.act_src; float* dsrc = A.adv.dsrc;
        const float* Wv = A.val.W[0]; const float* Wa = A.adv.W[0];
        for (int k = tid; k < Ka; k += 256) {
            float xv = 0.0f, xa = 0.0f;
            for (int n = 0; n < Nv; n++) xv = fmaf(dq[n], stage_w ? wt_v[PADI(n * Kv + k)] : Wv[(size_t)k * Nv + n], xv);
            for (int n = 0; n < Na; n++) xa = fmaf(dq[1 + n], stage_w ? wt_a[PADI(n * Ka + k)] : Wa[(size_t)k * Na + n], xa);
            const float v = xv + xa;
            dsrc[(size_t)k * B + b] = dact_f(v, xs_a[PADI(k)], act_src);
        }
    } else {
        float* dsa = A.adv.dsrc; float* dsv = has_val ? A.val.dsrc : nullptr;
        const int na = dsa ? Ka : 0, nv = dsv ? Kv : 0;
        for (int i = tid; i < na + nv; i += 256) {
            const bool v = i >= na; const int k = v ? i - na : i;
            const int K = v ? Kv : Ka, N = v ? Nv : Na;
            const float* wt = v ? wt_v : wt_a; const float* d = v ? dq : dq + 1;
            float acc = 0.0f;
            if (stage_w) for (int n = 0; n < N; n++) acc = fmaf(d[n], wt[PADI(n * K + k)], acc);
            else { const float* W = v ? A.val.W[0] : A.adv.W[0]; for (int n = 0; n < N; n++) acc = fmaf(d[n], W[(size_t)k * N + n], acc); }
            (v ? dsv : dsa)[(size_t)k * B + b] = dact_f(acc, (v ? xs_v : xs_a)[PADI(k)], v ? A.val.act_src : A.adv.act_src);
        }
    }
}
size_t head_td_lds_bytes(const HeadTdArgs& a) {
    size_t f = 0;
    const HeadLayer* H[2] = {&a.val, &a.adv};
    for (int h = a.dueling ? 0 : 1; h < 2; h++) f += PADI((size_t)3 * H[h]->K) + (a.stage_w ? PADI((size_t)2 * H[h]->K * H[h]->N) : 0) + (size_t)3 * H[h]->N * H[h]->S + (size_t)3 * H[h]->N;
    return (f + 3 * a.nA + 1 + a.nA + 8) * sizeof(float);
}
void launch_head_td(hipStream_t st, const HeadTdArgs& a, const HeadTdArgs* a_dev, int bump_sample_ctr, int take_pre) {
    const size_t lds = head_td_lds_bytes(a);
    hipLaunchKernelGGL(k_head_td, dim3(a.B), dim3(256), lds, st, a_dev, bump_sample_ctr, take_pre);
}

// Q columns for the policy path (src/policy.jl:38-64): q_out[n][nA], argmax (first max)
__global__ void k_q_columns(int n, int nA, int dueling, const float* __restrict__ val, const float* __restrict__ adv, float* __restrict__ q_out, int* __restrict__ amax) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    float q[DQN_MAX_ACTIONS];
    q_column(nA, dueling, val, adv, n, b, q);
    int best = 0;
    for (int a = 0; a < nA; a++) { if (q_out) q_out[(size_t)b * nA + a] = q[a]; if (q[a] > q[best]) best = a; }
    if (amax) amax[b] = best;
}
void launch_q_columns(hipStream_t st, int n, int nA, int dueling, const float* val, const float* adv, float* q_out, int* argmax_out) {
    hipLaunchKernelGGL(k_q_columns, dim3((n + 63) / 64), dim3(64), 0, st, n, nA, dueling, val, adv, q_out, argmax_out);
}

// ------------------------------------------------------------------ globalnorm (helpers.jl:38-46) + Flux Adam (solver.jl:66,228), fused, HBM-bound:
// per element 16 B read (p,m,v,g) + 12 B written; the job body lives in adam_body.h (shared with the backward launches' tails)
__global__ __launch_bounds__(256) void k_adam(AdamJob J) {
    __shared__ __attribute__((aligned(16))) long long sidx[3072]; __shared__ float wmax[4];      // 7808 B of path state + the top 4096 nodes of the sum-tree (prio_block_fast)
    adam_job_run(J, (int)blockIdx.x, sidx, wmax, false, (unsigned)sizeof sidx);
}
// jobs without a priority block: no LDS to speak of, so their workgroups fit beside the LDS-heavy GEMM workgroups of a concurrent launch
__global__ __launch_bounds__(256) void k_adam_stream(AdamJob J) {
    __shared__ float wmax[4]; __shared__ float fold_buf[256];      // fold_buf: the loss fold of the fused recurrent step (J.fold_hl)
    adam_job_run(J, (int)blockIdx.x, nullptr, wmax, false, 0, fold_buf);
}
// the Adam launch of a step that is followed by another sampled step: its FIRST workgroups gather the next batch (PreGather); their dependent
// round trips (row index -> 256-B row segments -> LDS -> arena lines) hide under the parameter stream of the remaining workgroups
__global__ __launch_bounds__(256) void k_adam_pg(AdamJob J, PreGather G) {
    __shared__ float tile[64][65]; __shared__ long long rows[64]; __shared__ float wmax[4];
    const int npg = G.gx * G.gy;
    if ((int)blockIdx.x < npg) {
        const int bx = (int)blockIdx.x % G.gx, by = (int)blockIdx.x / G.gx;
        gather_fb_body(G.s_rows, G.sp_rows, 0, G.E, G.B, G.idx_pre, G.x0, 1, G.cap2, G.tree, G.seed, J.state, G.meta, G.idx_pre, bx, by, tile, rows);
        if (blockIdx.x == 0 && threadIdx.x == 0) J.state->pre_valid = 2;      // unconditionally: without pre-drawn indices the gather body descended itself and wrote the same batch (+ idx_pre)
        return;
    }
    adam_job_run(J, (int)blockIdx.x - npg, nullptr, wmax);
}
// ... and for u8 replays on the byte arena (33 KB tile: a kernel of its own, so that the f32 variant keeps its smaller LDS footprint)
__global__ __launch_bounds__(256) void k_adam_pg_u8(AdamJob J, PreGather G) {
    __shared__ uint32_t tile32[128 * 65]; __shared__ long long rows[128]; __shared__ float wmax[4];
    const int npg = G.gx * G.gy;
    if ((int)blockIdx.x < npg) {
        const int bx = (int)blockIdx.x % G.gx, by = (int)blockIdx.x / G.gx;
        gather_u8b_body((const unsigned char*)G.s_rows, (const unsigned char*)G.sp_rows, G.E, G.B, G.idx_pre, (unsigned char*)G.x0, 1, G.cap2, G.tree, G.seed, J.state, G.meta,
                        G.idx_pre, bx, by, tile32, rows);
        if (blockIdx.x == 0 && threadIdx.x == 0) J.state->pre_valid = 2;      // unconditionally: without pre-drawn indices the gather body descended itself and wrote the same batch (+ idx_pre)
        return;
    }
    adam_job_run(J, (int)blockIdx.x - npg, nullptr, wmax);
}
int adam_blocks(size_t P) { size_t blocks = (P + 255) / 256; if (blocks > 2048) blocks = 2048; return (int)blocks; }
void launch_adam(hipStream_t st, const AdamJob& job, const PreGather* pg) {
    if (pg && pg->on && job.prio.n == 0) {
        if (pg->u8b) hipLaunchKernelGGL(k_adam_pg_u8, dim3(pg->gx * pg->gy + adam_job_blocks(job)), dim3(256), 0, st, job, *pg);
        else hipLaunchKernelGGL(k_adam_pg, dim3(pg->gx * pg->gy + adam_job_blocks(job)), dim3(256), 0, st, job, *pg);
        return;
    }
    if (job.prio.n > 0) hipLaunchKernelGGL(k_adam, dim3(adam_job_blocks(job)), dim3(256), 0, st, job);
    else hipLaunchKernelGGL(k_adam_stream, dim3(adam_job_blocks(job)), dim3(256), 0, st, job);
}

// ------------------------------------------------------------------ parameter layout conversion (Flux.params order <-> internal [K][N], conv kernels flipped)
__global__ void k_convert_params(const LayerDev* __restrict__ layers, int nl, const float* __restrict__ src, float* __restrict__ dst, int to_internal, size_t P) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // i indexes the EXTERNAL vector
    if (i >= P) return;
    size_t j = i;
    for (int l = 0; l < nl; l++) {
        const LayerDev& L = layers[l];
        const size_t wn = (size_t)L.K * L.N;
        if (i >= L.ew_off && i < L.ew_off + wn) {
            size_t e = i - L.ew_off;
            if (L.kind == DQN_LAYER_CONV) {      // external ((co*cin + ci)*kh + yy)*kw + xx, yy/xx un-flipped
                const int xx = (int)(e % L.kw); e /= L.kw; const int yy = (int)(e % L.kh); e /= L.kh; const int ci = (int)(e % L.cin); const int co = (int)(e / L.cin);
                const int ky = L.kh - 1 - yy, kx = L.kw - 1 - xx;
                e = ((size_t)(ci * L.kh + ky) * L.kw + kx) * L.cout + co;
            }
            j = L.w_off + e; break;
        }
        if (i >= L.eb_off && i < L.eb_off + (size_t)L.N) { j = L.b_off + (i - L.eb_off); break; }
        if (L.kind == DQN_LAYER_LSTM) {          // Flux.params order Wi, Wh, b, h0, c0 -> internal [Wi][b][Wh][junk][h0][c0][zeros]
            if (i >= L.ewh_off && i < L.ewh_off + (size_t)L.H * L.N) { j = L.wh_off + (i - L.ewh_off); break; }
            if (i >= L.eh0_off && i < L.eh0_off + (size_t)L.H) { j = L.h0_off + (i - L.eh0_off); break; }
            if (i >= L.ec0_off && i < L.ec0_off + (size_t)L.H) { j = L.c0_off + (i - L.ec0_off); break; }
        }
    }
    if (to_internal) dst[j] = src[i]; else dst[i] = src[j];
}
void launch_convert_params(hipStream_t st, const LayerDev* layers_dev, int nl, const float* src, float* dst, int to_internal, size_t P) {
    hipLaunchKernelGGL(k_convert_params, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, layers_dev, nl, src, dst, to_internal, P);
}
