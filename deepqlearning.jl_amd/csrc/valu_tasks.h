// valu_tasks.h -- device bodies of the canonical-order VALU contractions (one k-ascending fp32 fmaf chain per plan chunk) and the task-table
// dispatcher built on them.  Included by nn_valu.hip (k_valu_*, k_valu_multi) and by nn_gemm.hip, whose LDS-tiled launches can carry a TAIL of
// small independent tasks (head dW/db, the loss fold) in their last workgroups instead of paying a launch for them.
#pragma once
#include "common.h"
typedef float f32x4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ forward: Y[n][pos][col] = act(sum_k X[xb(pos)+koff(k)][col] W[k][n] + b[n])
__device__ __forceinline__ void valu_fwd_body(const LayerDev& L, const float* __restrict__ P, const float* __restrict__ X, int ldx, int col0, int ncols,
                                              int S, int kc, float* __restrict__ out, size_t t) {
    const size_t per_s = (size_t)L.N * L.npos * ncols;
    if (t >= per_s * S) return;
    const int s = (int)(t / per_s); const size_t e = t % per_s;
    const int col = (int)(e % ncols); const int pos = (int)((e / ncols) % L.npos); const int n = (int)(e / ((size_t)ncols * L.npos));
    const float* W = P + L.w_off;
    int xb = 0;
    if (L.kind == DQN_LAYER_CONV) { const int oy = pos / L.ow, ox = pos % L.ow; xb = oy * L.sh * L.iw + ox * L.sw; }
    const int k0 = s * kc, k1 = min(L.K, k0 + kc);
    float acc = 0.0f;
    if (L.kind == DQN_LAYER_CONV) {
        const int khw = L.kh * L.kw;
        int ci = k0 / khw, ky = (k0 / L.kw) % L.kh, kx = k0 % L.kw;
        for (int k = k0; k < k1; k++) {
            const int koff = (ci * L.ih + ky) * L.iw + kx;
            acc = fmaf(X[(size_t)(xb + koff) * ldx + col0 + col], W[(size_t)k * L.N + n], acc);
            if (++kx == L.kw) { kx = 0; if (++ky == L.kh) { ky = 0; ++ci; } }
        }
    } else {
        const float* xp = X + col0 + col; const float* wp = W + n;
        int k = k0;
        for (; k + 32 <= k1; k += 32) {   // a whole 32-deep head chunk in one round of 64 independent loads; the fma chain stays k-ascending
            float xv[32], wv[32];
#pragma unroll
            for (int u = 0; u < 32; u++) { xv[u] = xp[(size_t)(k + u) * ldx]; wv[u] = wp[(size_t)(k + u) * L.N]; }
#pragma unroll
            for (int u = 0; u < 32; u++) acc = fmaf(xv[u], wv[u], acc);
        }
        for (; k + 8 <= k1; k += 8) {     // 16 independent loads in flight; the fma chain stays k-ascending
            float xv[8], wv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { xv[u] = xp[(size_t)(k + u) * ldx]; wv[u] = wp[(size_t)(k + u) * L.N]; }
#pragma unroll
            for (int u = 0; u < 8; u++) acc = fmaf(xv[u], wv[u], acc);
        }
        for (; k < k1; k++) acc = fmaf(xp[(size_t)k * ldx], wp[(size_t)k * L.N], acc);
    }
    if (S == 1) out[e] = act_f(acc + P[L.b_off + n], L.act);
    else out[(size_t)s * per_s + e] = acc;
}
// ------------------------------------------------------------------ dW[k][n] = sum_{(pos,b)} X[xb(pos)+koff(k)][b] dpre[n][pos][b];  db[n] = sum dpre
// thread = (chunk, k, n) for k < K, plus a virtual row k == K that accumulates the bias gradient.
__device__ __forceinline__ void valu_dw_body(const LayerDev& L, const float* __restrict__ X, int ldx, const float* __restrict__ dpre, int B, int S, int kc,
                                             float* __restrict__ out, size_t t) {
    const size_t per_s = (size_t)(L.K + 1) * L.N;
    if (t >= per_s * S) return;
    const int s = (int)(t / per_s); const size_t e = t % per_s;
    const int n = (int)(e % L.N), k = (int)(e / L.N);
    const int KK = L.npos * B, j0 = s * kc, j1 = min(KK, j0 + kc);
    int koff = k;
    if (L.kind == DQN_LAYER_CONV && k < L.K) { const int khw = L.kh * L.kw; const int ci = k / khw, ky = (k / L.kw) % L.kh, kx = k % L.kw; koff = (ci * L.ih + ky) * L.iw + kx; }
    float acc = 0.0f;
    int pos = j0 / B, b = j0 % B;
    if (L.kind != DQN_LAYER_CONV) {      // dense: one "position"; operands are two contiguous rows -> 16 loads in flight, chain order unchanged
        const float* dr = dpre + (size_t)n * B; const float* xr = k < L.K ? X + (size_t)k * ldx : nullptr;
        int j = j0;
        // rounds of 16 samples (4 + 4 float4 loads in flight), chain order unchanged.  One round of 32 or 64 samples held 64-128 registers here and,
        // through the tail tasks, set the register budget -- and the occupancy -- of every LDS-tiled backward launch.
        if (((j1 - j0) & 15) == 0 && (((size_t)dr | (size_t)(xr ? xr : dr)) & 15) == 0 && (j0 & 3) == 0) {
#pragma unroll 1
            for (; j < j1; j += 16) {
                f32x4v dq[4], xq[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { dq[u] = *reinterpret_cast<const f32x4v*>(dr + j + 4 * u); xq[u] = xr ? *reinterpret_cast<const f32x4v*>(xr + j + 4 * u) : (f32x4v){1.f, 1.f, 1.f, 1.f}; }
                if (xr) {
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc = fmaf(xq[u].x, dq[u].x, acc); acc = fmaf(xq[u].y, dq[u].y, acc); acc = fmaf(xq[u].z, dq[u].z, acc); acc = fmaf(xq[u].w, dq[u].w, acc); }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) { acc = acc + dq[u].x; acc = acc + dq[u].y; acc = acc + dq[u].z; acc = acc + dq[u].w; }
                }
            }
        }
        for (; j + 8 <= j1; j += 8) {
            float dv[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { dv[u] = dr[j + u]; xv[u] = xr ? xr[j + u] : 1.0f; }
            if (xr) {
#pragma unroll
                for (int u = 0; u < 8; u++) acc = fmaf(xv[u], dv[u], acc);
            } else {
#pragma unroll
                for (int u = 0; u < 8; u++) acc = acc + dv[u];
            }
        }
        for (; j < j1; j++) { if (xr) acc = fmaf(xr[j], dr[j], acc); else acc = acc + dr[j]; }
        out[(size_t)s * per_s + e] = acc;
        return;
    }
    for (int j = j0; j < j1; j++) {
        const float d = dpre[((size_t)n * L.npos + pos) * B + b];
        if (k < L.K) {
            int xb = 0; if (L.kind == DQN_LAYER_CONV) { const int oy = pos / L.ow, ox = pos % L.ow; xb = oy * L.sh * L.iw + ox * L.sw; }
            acc = fmaf(X[(size_t)(xb + koff) * ldx + b], d, acc);
        } else acc = acc + d;
        if (++b == B) { b = 0; ++pos; }
    }
    out[(size_t)s * per_s + e] = acc;
}
// ------------------------------------------------------------------ dX[feat][b]  (then dact of the producing layer, optionally + addend at the dueling join)
__device__ __forceinline__ void valu_dx_body(const LayerDev& L, const float* __restrict__ P, const float* __restrict__ dpre, int B, int S, int kc,
                                             float* __restrict__ out, const float* __restrict__ addend, const float* __restrict__ ysrc, int ldy, int act_src, size_t t) {
    const size_t per_s = (size_t)L.in_feat * B;
    if (t >= per_s * S) return;
    const int s = (int)(t / per_s); const size_t e = t % per_s;
    const int b = (int)(e % B); const int feat = (int)(e / B);
    const float* W = P + L.w_off;
    float acc = 0.0f;
    if (L.kind == DQN_LAYER_DENSE) {
        const int n0 = s * kc, n1 = min(L.N, n0 + kc);
        for (int n = n0; n < n1; n++) acc = fmaf(dpre[(size_t)n * B + b], W[(size_t)feat * L.N + n], acc);
    } else {
        const int hw = L.ih * L.iw; const int ci = feat / hw, iy = (feat % hw) / L.iw, ix = feat % L.iw;
        // plan.dx_kc: RAW taps (ky*kw + kx) per chunk; a chunk = one chain from +0 over its valid taps, chunk sums added in ascending order
        const int tc = DQN_CONV_TAP_CHUNK(L); int cur = -1; bool have = false; float tot = 0.0f;
        for (int ky = 0; ky < L.kh; ky++) {
            const int ty = iy - ky; if (ty < 0 || ty % L.sh) continue; const int oy = ty / L.sh; if (oy >= L.oh) continue;
            for (int kx = 0; kx < L.kw; kx++) {
                const int tx = ix - kx; if (tx < 0 || tx % L.sw) continue; const int ox = tx / L.sw; if (ox >= L.ow) continue;
                const int cid = (ky * L.kw + kx) / tc;
                if (cid != cur) { if (cur >= 0) { tot = have ? tot + acc : acc; have = true; acc = 0.0f; } cur = cid; }
                const float* wr = W + (size_t)((ci * L.kh + ky) * L.kw + kx) * L.N; const int pos = oy * L.ow + ox;
                for (int co = 0; co < L.N; co++) acc = fmaf(dpre[((size_t)co * L.npos + pos) * B + b], wr[co], acc);
            }
        }
        if (have) acc = tot + acc;
    }
    if (S == 1) {
        if (addend) acc = addend[e] + acc;
        if (ysrc) acc = dact_f(acc, ysrc[(size_t)feat * ldy + b], act_src);
        out[e] = acc;
    } else out[(size_t)s * per_s + e] = acc;
}
// kind 3: loss fold.  loss = (sum of the B per-column Huber terms, ascending b) / B  (src/solver.jl:223-224); one lane; `dpre` = the terms, `out` = &state->loss
__device__ __forceinline__ void valu_loss_fold(const float* __restrict__ hl, int B, float* __restrict__ out, size_t t) {
    if (t != 0) return;
    float lsum = 0.0f;
    for (int b = 0; b < B; b++) lsum = lsum + hl[b];
    *out = lsum / (float)B;
}
// run workgroup `blk` (256 threads) of a task table: task i owns blocks [first_block_i, first_block_{i+1})
// FWD = false: the table holds no forward task (the TAIL tables of the backward launches: heads' dW / dX, the loss fold) -- the forward body is the
// register-hungriest of the four and, compiled into an LDS-tiled kernel, cost that kernel a third of its occupancy (123 vs 80 VGPRs)
template <bool FWD = true>
__device__ __forceinline__ void valu_task_run(const VTask* __restrict__ tasks, int ntasks, unsigned blk) {
    int ti = 0;
    while (ti + 1 < ntasks && blk >= tasks[ti + 1].first_block) ti++;
    const VTask& T = tasks[ti];
    const size_t t = (size_t)(blk - T.first_block) * 256 + threadIdx.x;
    if (T.kind == 0) { if constexpr (FWD) valu_fwd_body(T.L, T.P, T.X, T.ldx, T.col0, T.ncols, T.S, T.kc, T.out, t); }
    else if (T.kind == 1) valu_dw_body(T.L, T.X, T.ldx, T.dpre, T.B, T.S, T.kc, T.out, t);
    else if (T.kind == 2) valu_dx_body(T.L, T.P, T.dpre, T.B, T.S, T.kc, T.out, T.addend, T.ysrc, T.ldy, T.act_src, t);
    else valu_loss_fold(T.dpre, T.B, T.out, t);
}
