// nn_mfma.hip -- fp32 MFMA kernels (v_mfma_f32_16x16x4_f32, gfx950) for the contractions of batch_train!
// (src/solver.jl:210,211,219-225): Q-network forward (implicit-GEMM conv + dense), dX and dW/db.
//
// Why 16x16x4 f32: the parity target (TD loss bit-exact, Q within 1e-5) rules out bf16; the f32 MFMA is a k-ordered
// fp32 fma chain (MI355X guide), i.e. bit-identical to the canonical order of the CPU twin and of nn_valu.hip.
// The batch-innermost activation layout Y[feature][column] makes a 16-row MFMA M-tile = 16 samples of one feature:
//   A operand  lane (i = l&15, kq = l>>4)  <- 16 consecutive floats of row k   (one 64-B segment per kq)
//   B operand  lane (j = l&15, kq = l>>4)  <- 16 consecutive floats of W[k][.]  (one 64-B segment per kq)
//   C/D        lane (col = l&15, rows 4*(l>>4)+r)  -> one float4 store of 4 consecutive samples
// so forward operands stream straight from L2 with no LDS transpose.  One wave = one (MT x NT) register tile of
// 16x16 MFMA tiles; a workgroup is 4 independent waves.  Chains are never reordered: split-K follows the layer plan
// and partial sums are combined in ascending chunk order by k_reduce (nn_valu.hip).
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

void launch_reduce_pub(hipStream_t st, const float* part, int S, size_t elems, int mode, const float* bias, int per_n, int act,
                       const float* addend, const float* ysrc, int B, int ldy, float* out);

// ------------------------------------------------------------------ forward
struct FwdProb { const float* P; const float* X; int ldx, col0, ncols; float* out; };

template <int MT, int NT>
__global__ __launch_bounds__(256) void k_mfma_fwd(LayerDev L, FwdProb p, int S, int kc, int ntasks) {
    extern __shared__ int koff_lds[];
    if (L.kind == DQN_LAYER_CONV) {
        const int khw = L.kh * L.kw;
        for (int k = threadIdx.x; k < L.K; k += 256) { const int ci = k / khw, ky = (k / L.kw) % L.kh, kx = k % L.kw; koff_lds[k] = (ci * L.ih + ky) * L.iw + kx; }
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int task = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int ngroups = L.N / (16 * NT), mgroups = p.ncols / (16 * MT);
    const int ng = task % ngroups; task /= ngroups;
    const int mg = task % mgroups; task /= mgroups;
    const int pos = task % L.npos; const int s = task / L.npos;
    const int n0 = ng * 16 * NT, c0 = mg * 16 * MT;
    int xb = 0;
    if (L.kind == DQN_LAYER_CONV) { const int oy = pos / L.ow, ox = pos % L.ow; xb = oy * L.sh * L.iw + ox * L.sw; }
    const int k0 = s * kc, k1 = min(L.K, k0 + kc);
    const float* Wp = p.P + L.w_off + n0 + l15;
    const float* Xp = p.X + p.col0 + c0 + l15;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool conv = L.kind == DQN_LAYER_CONV;
    // Latency hiding: U MFMA steps of operands are fetched as one batch (U*(MT+NT) independent loads in flight) into a
    // register double buffer while the previous batch's MFMAs issue.  The fma chain order (k ascending) is unchanged.
    constexpr int U = 16;
    const unsigned ldx = (unsigned)p.ldx, ldn = (unsigned)L.N;
    auto fetch = [&](int k, float (&a)[U][MT], float (&b)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int kk = k + 4 * u;
            const int ko = conv ? koff_lds[kk] : kk;
#pragma unroll
            for (int m = 0; m < MT; m++) a[u][m] = Xp[(unsigned)(xb + ko) * ldx + 16 * m];
#pragma unroll
            for (int t = 0; t < NT; t++) b[u][t] = Wp[(unsigned)kk * ldn + 16 * t];
        }
    };
    auto issue = [&](float (&a)[U][MT], float (&b)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[m][t] = MFMA(a[u][m], b[u][t], acc[m][t]);
    };
    float a0[U][MT], b0[U][NT], a1[U][MT], b1[U][NT];
    const int nb = (k1 - k0) / (4 * U);
    int k = k0 + kq;
    if (nb > 0) fetch(k, a0, b0);
    for (int i = 0; i < nb; i += 2) {
        if (i + 1 < nb) fetch(k + 4 * U, a1, b1);
        issue(a0, b0);
        if (i + 1 < nb) { if (i + 2 < nb) fetch(k + 8 * U, a0, b0); issue(a1, b1); }
        k += 8 * U;
    }
    for (k = k0 + kq + nb * 4 * U; k < k1; k += 4) {
        const int ko = conv ? koff_lds[k] : k;
        float a[MT], b[NT];
#pragma unroll
        for (int m = 0; m < MT; m++) a[m] = Xp[(unsigned)(xb + ko) * ldx + 16 * m];
#pragma unroll
        for (int t = 0; t < NT; t++) b[t] = Wp[(unsigned)k * ldn + 16 * t];
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[m][t] = MFMA(a[m], b[t], acc[m][t]);
    }
    const size_t per_s = (size_t)L.N * L.npos * p.ncols;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + l15;
        const float bias = p.P[L.b_off + n];
#pragma unroll
        for (int m = 0; m < MT; m++) {
            f32x4 v = acc[m][t];
            if (S == 1) { v.x = act_f(v.x + bias, L.act); v.y = act_f(v.y + bias, L.act); v.z = act_f(v.z + bias, L.act); v.w = act_f(v.w + bias, L.act); }
            float* dst = p.out + (size_t)s * per_s + ((size_t)n * L.npos + pos) * p.ncols + c0 + 16 * m + 4 * kq;
            *reinterpret_cast<f32x4*>(dst) = v;
        }
    }
}

static int pick_tile(long tiles_m, long tiles_n, long other, int* MT, int* NT, int max_mt, int max_nt) {
    // largest register tile that still leaves >= ~1024 wave tasks (256 CUs x 4 SIMDs); prefers sharing the weight
    // operand (MT) first, then the activation operand (NT)
    int mt = 1, nt = 1;
    const int cand_m[3] = {4, 2, 1}, cand_n[3] = {4, 2, 1};
    long best = -1;
    for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) {
        const int m = cand_m[a], n = cand_n[b];
        if (m > max_mt || n > max_nt || tiles_m % m || tiles_n % n) continue;
        const long tasks = (tiles_m / m) * (tiles_n / n) * other;
        const long score = tasks >= 1024 ? 1000000L + m * n * 10 + m : tasks;   // enough parallelism -> biggest tile
        if (score > best) { best = score; mt = m; nt = n; }
    }
    *MT = mt; *NT = nt; return 0;
}

template <int MT, int NT>
static void fwd_launch(hipStream_t st, const LayerDev& L, const FwdProb& p, int S, int kc) {
    const int ntasks = (L.N / (16 * NT)) * (p.ncols / (16 * MT)) * L.npos * S;
    const size_t lds = L.kind == DQN_LAYER_CONV ? (size_t)L.K * sizeof(int) : 0;
    hipLaunchKernelGGL((k_mfma_fwd<MT, NT>), dim3((ntasks + 3) / 4), dim3(256), lds, st, L, p, S, kc, ntasks);
}
bool mfma_fwd_ok(const LayerDev& L, int ncols) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    return !(L.N % 16 || ncols % 16 || L.K % 4 || (S > 1 && kc % 4) || L.K > 16384);
}
bool launch_mfma_fwd(hipStream_t st, const LayerDev& L, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, float* partials, bool reduce) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    if (!mfma_fwd_ok(L, ncols)) return false;
    FwdProb p{P, X, ldx, col0, ncols, S == 1 ? Y : partials};
    int MT, NT; pick_tile(ncols / 16, L.N / 16, (long)L.npos * S, &MT, &NT, 4, 2);
#define FWD_CASE(m, n) if (MT == m && NT == n) fwd_launch<m, n>(st, L, p, S, kc)
    FWD_CASE(1, 1); else FWD_CASE(2, 1); else FWD_CASE(4, 1); else FWD_CASE(1, 2); else FWD_CASE(2, 2); else FWD_CASE(4, 2);
#undef FWD_CASE
    if (S > 1 && reduce) launch_reduce_pub(st, partials, S, (size_t)L.N * L.npos * ncols, 0, P + L.b_off, L.npos * ncols, L.act, nullptr, nullptr, 0, 0, Y);
    return true;
}

// ------------------------------------------------------------------ dX (then act' of the producing layer, + addend at the dueling join)
//   dense: dX[f][b] = sum_n dpre[n][b] W[f][n]                        (chunks over n per plan.dx_kc)
//   conv : dX[ci][iy][ix][b] = sum_{valid taps (ky,kx)} sum_co dpre[co][oy][ox][b] W[(ci,ky,kx)][co]   (chunks of RAW taps per plan.dx_kc)
template <int MT>
__global__ __launch_bounds__(256) void k_mfma_dx(LayerDev L, const float* __restrict__ P, const float* __restrict__ dpre, int B, int S, int kc,
                                                 float* __restrict__ out, const float* __restrict__ addend, const float* __restrict__ ysrc, int ldy,
                                                 int act_src, int ntasks) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int task = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int mgroups = B / (16 * MT);
    const int mg = task % mgroups; task /= mgroups;
    const int b0 = mg * 16 * MT;
    const float* W = P + L.w_off;
    f32x4 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    size_t feat; int s = 0;
    if (L.kind == DQN_LAYER_DENSE) {
        const int ftiles = L.K / 16; const int ft = task % ftiles; s = task / ftiles;
        feat = (size_t)ft * 16 + l15;
        const int n0 = s * kc, n1 = min(L.N, n0 + kc);
        const float* wr = W + feat * L.N; const float* dp = dpre + b0 + l15;
        constexpr int U = 16;
        int n = n0 + kq;
        for (; n + 4 * (U - 1) < n1; n += 4 * U) {      // U steps of operands in flight
            float av[U][MT], bv[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                bv[u] = wr[n + 4 * u];
#pragma unroll
                for (int m = 0; m < MT; m++) av[u][m] = dp[(unsigned)(n + 4 * u) * (unsigned)B + 16 * m];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int m = 0; m < MT; m++) acc[m] = MFMA(av[u][m], bv[u], acc[m]);
        }
        for (; n < n1; n += 4) {
            const float b = wr[n];
#pragma unroll
            for (int m = 0; m < MT; m++) acc[m] = MFMA(dp[(size_t)n * B + 16 * m], b, acc[m]);
        }
    } else {
        const int ctiles = L.cin / 16; const int ct = task % ctiles; const int ip = task / ctiles;
        const int iy = ip / L.iw, ix = ip % L.iw; const int ci = ct * 16 + l15;
        feat = (size_t)ci * L.ih * L.iw + ip;
        // plan.dx_kc: RAW taps per chunk; a chunk = one chain from +0 over its valid taps, chunk sums added in ascending order
        const int tc = DQN_CONV_TAP_CHUNK(L); int cur = -1; bool have = false; f32x4 tot[MT];
#pragma unroll
        for (int m = 0; m < MT; m++) tot[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int ky = 0; ky < L.kh; ky++) {
            const int ty = iy - ky; if (ty < 0 || ty % L.sh) continue; const int oy = ty / L.sh; if (oy >= L.oh) continue;
            for (int kx = 0; kx < L.kw; kx++) {
                const int tx = ix - kx; if (tx < 0 || tx % L.sw) continue; const int ox = tx / L.sw; if (ox >= L.ow) continue;
                const int cid = (ky * L.kw + kx) / tc;
                if (cid != cur) {
                    if (cur >= 0) {
#pragma unroll
                        for (int m = 0; m < MT; m++) { if (have) { tot[m].x = tot[m].x + acc[m].x; tot[m].y = tot[m].y + acc[m].y; tot[m].z = tot[m].z + acc[m].z; tot[m].w = tot[m].w + acc[m].w; } else tot[m] = acc[m]; acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                        have = true;
                    }
                    cur = cid;
                }
                const float* wr = W + (size_t)((ci * L.kh + ky) * L.kw + kx) * L.N;
                const float* dp = dpre + (size_t)(oy * L.ow + ox) * B + b0 + l15;
                constexpr int U = 16;
                const unsigned cstride = (unsigned)L.npos * (unsigned)B;
                int co = kq;
                for (; co + 4 * (U - 1) < L.N; co += 4 * U) {
                    float av[U][MT], bv[U];
#pragma unroll
                    for (int u = 0; u < U; u++) {
                        bv[u] = wr[co + 4 * u];
#pragma unroll
                        for (int m = 0; m < MT; m++) av[u][m] = dp[(unsigned)(co + 4 * u) * cstride + 16 * m];
                    }
#pragma unroll
                    for (int u = 0; u < U; u++)
#pragma unroll
                        for (int m = 0; m < MT; m++) acc[m] = MFMA(av[u][m], bv[u], acc[m]);
                }
                for (; co < L.N; co += 4) {
                    const float b = wr[co];
#pragma unroll
                    for (int m = 0; m < MT; m++) acc[m] = MFMA(dp[(size_t)co * L.npos * B + 16 * m], b, acc[m]);
                }
            }
        }
        if (have) {
#pragma unroll
            for (int m = 0; m < MT; m++) { acc[m].x = tot[m].x + acc[m].x; acc[m].y = tot[m].y + acc[m].y; acc[m].z = tot[m].z + acc[m].z; acc[m].w = tot[m].w + acc[m].w; }
        }
    }
    const size_t per_s = (size_t)L.in_feat * B;
#pragma unroll
    for (int m = 0; m < MT; m++) {
        f32x4 v = acc[m];
        const int bcol = b0 + 16 * m + 4 * kq;
        const size_t e = feat * B + bcol;
        if (S == 1) {
            if (addend) { const f32x4 ad = *reinterpret_cast<const f32x4*>(addend + e); v.x = ad.x + v.x; v.y = ad.y + v.y; v.z = ad.z + v.z; v.w = ad.w + v.w; }
            if (ysrc) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(ysrc + feat * ldy + bcol);
                v.x = dact_f(v.x, y.x, act_src); v.y = dact_f(v.y, y.y, act_src); v.z = dact_f(v.z, y.z, act_src); v.w = dact_f(v.w, y.w, act_src);
            }
        }
        *reinterpret_cast<f32x4*>(out + (size_t)s * per_s + e) = v;
    }
}
bool mfma_dx_ok(const LayerDev& L, int B, int ldy) {
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int S = dense ? dqn_nchunks(L.N, L.dx_kc) : 1, kc = dqn_chunk_len(L.N, L.dx_kc);
    if (B % 16 || L.N % 4 || (S > 1 && kc % 4) || ldy % 4) return false;
    return dense ? (L.K % 16 == 0) : (L.cin % 16 == 0);
}
bool launch_mfma_dx(hipStream_t st, const LayerDev& L, const float* P, const float* dpre, int B, float* out, float* partials,
                    const float* addend, const float* ysrc, int ldy, int act_src, bool reduce) {
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int S = dense ? dqn_nchunks(L.N, L.dx_kc) : 1, kc = dqn_chunk_len(L.N, L.dx_kc);
    if (!mfma_dx_ok(L, B, ldy)) return false;
    const long ftiles = dense ? L.K / 16 : (long)(L.cin / 16) * L.ih * L.iw;
    int MT, NT; pick_tile(B / 16, 1, ftiles * S, &MT, &NT, 2, 1);
    const int ntasks = (int)((B / (16 * MT)) * ftiles * S);
    float* dst = S == 1 ? out : partials;
    if (MT == 2) hipLaunchKernelGGL((k_mfma_dx<2>), dim3((ntasks + 3) / 4), dim3(256), 0, st, L, P, dpre, B, S, kc, dst, addend, ysrc, ldy, act_src, ntasks);
    else hipLaunchKernelGGL((k_mfma_dx<1>), dim3((ntasks + 3) / 4), dim3(256), 0, st, L, P, dpre, B, S, kc, dst, addend, ysrc, ldy, act_src, ntasks);
    if (S > 1 && reduce) launch_reduce_pub(st, partials, S, (size_t)L.in_feat * B, 1, nullptr, 1, act_src, addend, ysrc, B, ldy, out);
    return true;
}

// ------------------------------------------------------------------ dW / db
//   G[k][n] = sum_{(pos,b)} X[xb(pos)+koff(k)][b] dpre[n][pos][b]   for k < K;   row k == K is the bias gradient
//   (A operand == 1: fma(1, d, acc) == acc + d exactly).  M = K+1 rows (padded to 16), N = n, contraction over
//   (pos, b) with b innermost; chunks of plan.dw_kc samples.
template <int NT>
__global__ __launch_bounds__(256) void k_mfma_dw(LayerDev L, const float* __restrict__ X, int ldx, const float* __restrict__ dpre, int B, int S, int kc,
                                                 float* __restrict__ out, int ntasks) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    int task = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
    if (task >= ntasks) return;
    const int ngroups = L.N / (16 * NT), mtiles = (L.K + 1 + 15) / 16;
    const int ng = task % ngroups; task /= ngroups;
    const int mt = task % mtiles; const int s = task / mtiles;
    const int n0 = ng * 16 * NT;
    const int krow = mt * 16 + l15;                       // this lane's A row
    int koff = krow;
    if (L.kind == DQN_LAYER_CONV && krow < L.K) { const int khw = L.kh * L.kw; const int ci = krow / khw, ky = (krow / L.kw) % L.kh, kx = krow % L.kw; koff = (ci * L.ih + ky) * L.iw + kx; }
    const bool real = krow < L.K; const float aconst = krow == L.K ? 1.0f : 0.0f;
    const int KK = L.npos * B, j0 = s * kc, j1 = min(KK, j0 + kc);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int bsteps = B / 4;
    for (int pos = j0 / B; pos * B < j1; pos++) {          // chunks cover whole positions (dw_kc % B == 0) or the single dense "position"
        int xb = 0;
        if (L.kind == DQN_LAYER_CONV) { const int oy = pos / L.ow, ox = pos % L.ow; xb = oy * L.sh * L.iw + ox * L.sw; }
        const float* xr = X + (size_t)(xb + koff) * ldx + kq;
        const float* dr = dpre + ((size_t)(n0 + l15) * L.npos + pos) * B + kq;
        constexpr int U = 8;
        const unsigned nstride = 16u * (unsigned)L.npos * (unsigned)B;
        int bs = 0;
        for (; bs + U <= bsteps; bs += U) {
            float av[U], bv[U][NT];
#pragma unroll
            for (int u = 0; u < U; u++) {
                av[u] = real ? xr[4 * (bs + u)] : aconst;
#pragma unroll
                for (int t = 0; t < NT; t++) bv[u][t] = dr[t * nstride + 4 * (bs + u)];
            }
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t] = MFMA(av[u], bv[u][t], acc[t]);
        }
        for (; bs < bsteps; bs++) {
            const float a = real ? xr[4 * bs] : aconst;
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = MFMA(a, dr[(size_t)16 * t * L.npos * B + 4 * bs], acc[t]);
        }
    }
    const size_t per_s = (size_t)(L.K + 1) * L.N;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + l15;
        const float v[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = mt * 16 + 4 * kq + r;
            if (k <= L.K) out[(size_t)s * per_s + (size_t)k * L.N + n] = v[r];
        }
    }
}
bool mfma_dw_ok(const LayerDev& L, int B) {
    const int KK = L.npos * B, S = dqn_nchunks(KK, L.dw_kc), kc = dqn_chunk_len(KK, L.dw_kc);
    return !(L.N % 16 || B % 4 || (S > 1 && kc % B));
}
bool launch_mfma_dw(hipStream_t st, const LayerDev& L, const float* X, int ldx, const float* dpre, int B, float* G, float* partials, bool reduce) {
    const int KK = L.npos * B, S = dqn_nchunks(KK, L.dw_kc), kc = dqn_chunk_len(KK, L.dw_kc);
    if (!mfma_dw_ok(L, B)) return false;
    const long mtiles = (L.K + 1 + 15) / 16;
    int MT, NT; pick_tile(1, L.N / 16, mtiles * S, &MT, &NT, 1, 4);
    const int ntasks = (int)((L.N / (16 * NT)) * mtiles * S);
    float* dst = S == 1 ? G + L.w_off : partials;
    if (NT == 4) hipLaunchKernelGGL((k_mfma_dw<4>), dim3((ntasks + 3) / 4), dim3(256), 0, st, L, X, ldx, dpre, B, S, kc, dst, ntasks);
    else if (NT == 2) hipLaunchKernelGGL((k_mfma_dw<2>), dim3((ntasks + 3) / 4), dim3(256), 0, st, L, X, ldx, dpre, B, S, kc, dst, ntasks);
    else hipLaunchKernelGGL((k_mfma_dw<1>), dim3((ntasks + 3) / 4), dim3(256), 0, st, L, X, ldx, dpre, B, S, kc, dst, ntasks);
    if (S > 1 && reduce) launch_reduce_pub(st, partials, S, (size_t)(L.K + 1) * L.N, 2, nullptr, 1, 0, nullptr, nullptr, 0, 0, G + L.w_off);
    return true;
}
