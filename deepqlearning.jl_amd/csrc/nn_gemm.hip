// nn_gemm.hip -- LDS-tiled fp32-MFMA GEMM kernels (gfx950) for the heavy contractions of batch_train!
// (src/solver.jl:210,211,219-225).  Same numerics as nn_mfma.hip / nn_valu.hip / the CPU twin: every output is a
// k-ascending fp32 fma chain per plan chunk (v_mfma_f32_16x16x4_f32 accumulates k0..k3 in order), so swapping
// kernels never changes a bit.
//
// Design (MI355X): a workgroup = 4 waves (one per SIMD).  Operand tiles are fetched with 16-B/lane coalesced
// global_load_dwordx4 (full 64..256-B row segments of the batch-innermost activations / [K][N] weights) into
// registers, written to a padded LDS tile with ds_write_b128, and read back as MFMA fragments with conflict-free
// ds_read_b32.  Tiles are double-buffered in LDS; the next tile's global loads are in flight while the current
// tile's 8 MFMA steps issue, one barrier per 32-deep K tile.  ~40 KB LDS per workgroup => 4 workgroups (16 waves)
// per CU hide the rest of the latency.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

void launch_reduce_pub(hipStream_t st, const float* part, int S, size_t elems, int mode, const float* bias, int per_n, int act,
                       const float* addend, const float* ysrc, int B, int ldy, float* out);

// =====================================================================================================================
// forward:  Y[n][pos][col] = act( sum_k X[xb(pos)+koff(k)][col] * W[k][n] + bias[n] )
// workgroup tile = 4 M-tiles (16 columns each, drawn from consecutive (pos, column-tile) pairs; wave w owns M-tile w)
//                  x NT N-tiles (16*NT output channels), K walked in tiles of 32.
// =====================================================================================================================
struct GFwdProb { const float* W; const float* bias; const float* X; int ldx, col0, ncols; float* out; int mtiles, mgroups; };
struct GFwdProbs { GFwdProb p[4]; int wg_end[4]; };   // up to 4 problems of one geometry per launch: {val,adv} x {online,target}

constexpr int F_KT = 32;        // K tile depth
constexpr int F_SA = 80;        // A tile row stride (64 columns + 16 pad): ds_read_b32 of lanes (i, kq) hits banks 16*kq + i
template <int NT> struct FwdCfg { static constexpr int NW = 16 * NT; static constexpr int SB = (NW % 32 == 0) ? NW + 16 : NW + 32; };

template <int NT>
__global__ __launch_bounds__(256) void k_fwd_lds(LayerDev L, GFwdProbs pr, int S, int kc) {
    constexpr int NW = FwdCfg<NT>::NW, SB = FwdCfg<NT>::SB;
    extern __shared__ float lds[];
    float* As = lds;                                   // [2][F_KT][F_SA]
    float* Bs = lds + 2 * F_KT * F_SA;                 // [2][F_KT][SB]
    int* koff_lds = (int*)(Bs + 2 * F_KT * SB);        // [K] (conv only)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const bool conv = L.kind == DQN_LAYER_CONV;
    if (conv) {
        const int khw = L.kh * L.kw;
        for (int k = tid; k < L.K; k += 256) { const int ci = k / khw, ky = (k / L.kw) % L.kh, kx = k % L.kw; koff_lds[k] = (ci * L.ih + ky) * L.iw + kx; }
    }
    int pi = 0;
    while (pi < 3 && (int)blockIdx.x >= pr.wg_end[pi]) pi++;
    const GFwdProb& p = pr.p[pi];
    const int wg_begin = pi == 0 ? 0 : pr.wg_end[pi - 1];
    int w = xcd_remap(blockIdx.x - wg_begin, pr.wg_end[pi] - wg_begin);
    const int ngroups = L.N / NW;
    const int ng = w % ngroups; w /= ngroups;
    const int mgrp = w % p.mgroups; const int s = w / p.mgroups;
    const int n0 = ng * NW, ctiles = p.ncols >> 4;
    const int k0 = s * kc, k1 = min(L.K, k0 + kc), nkt = (k1 - k0) / F_KT;

    // ---- this thread's slice of the A tile: column group (t & 15) -> M-tile j = (t&15)>>2, float4 (t&3); rows t>>4, +16
    const int aj = (tid & 15) >> 2;
    const int amt = mgrp * 4 + aj;
    const bool a_ok = amt < p.mtiles;
    int a_xb = 0, a_ct = 0;
    if (a_ok) {
        const int pos = amt / ctiles; a_ct = amt % ctiles;
        if (conv) { const int oy = pos / L.ow, ox = pos % L.ow; a_xb = oy * L.sh * L.iw + ox * L.sw; }
    }
    const float* Xa = p.X + p.col0 + a_ct * 16 + (tid & 3) * 4;
    const int arow = tid >> 4;                         // 0..15 (second float4: +16)
    const unsigned ldx = (unsigned)p.ldx;
    // ---- B tile slice: NW/4 float4 per row
    constexpr int BF4 = NW / 4;                        // float4 per B row
    constexpr int BQ = (F_KT * BF4 + 255) / 256;       // float4 per thread (1 or 2)
    const float* Wp = p.W + n0;

    if (conv) __syncthreads();                         // koff table ready
    // Register staging with HAND-COUNTED waits.  hipcc's waitcnt pass drains vmcnt(0) before every prefetch issue in a
    // pipelined loop with conditional loads (seen in the ISA), which exposes the full L2/HBM latency once per K tile.
    // So the staging loads are inline asm (invisible to that pass), ALWAYS issued (tile index clamped, so the number of
    // loads in flight is a compile-time constant) and retired with explicit s_waitcnt vmcnt(N) naming their registers
    // (cdna_hip_programming.md section 5.7 form (ii)).
    constexpr int LPS = 2 + BQ;                        // loads per stage
    struct Stage { f32x4 a0, a1, b0, b1; };
    const int bq0 = tid < F_KT * BF4 ? tid : F_KT * BF4 - 1;          // clamped B slots (threads beyond the tile re-load the last one)
    const int bq1 = tid + 256 < F_KT * BF4 ? tid + 256 : F_KT * BF4 - 1;
    const float* Wb0 = Wp + (unsigned)(bq0 / BF4) * (unsigned)L.N + 4 * (bq0 % BF4);
    const float* Wb1 = Wp + (unsigned)(bq1 / BF4) * (unsigned)L.N + 4 * (bq1 % BF4);
    auto gld = [](const float* ptr) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(ptr) : "memory"); return v; };
    auto gload = [&](int kt, Stage& r) {
        kt = min(kt, nkt - 1);
        const int kb = k0 + kt * F_KT;
        const int ka = kb + arow;
        const int ko0 = conv ? koff_lds[ka] : ka, ko1 = conv ? koff_lds[ka + 16] : ka + 16;
        r.a0 = gld(Xa + (unsigned)(a_xb + ko0) * ldx);
        r.a1 = gld(Xa + (unsigned)(a_xb + ko1) * ldx);
        r.b0 = gld(Wb0 + (unsigned)kb * (unsigned)L.N);
        if (BQ > 1) r.b1 = gld(Wb1 + (unsigned)kb * (unsigned)L.N);
    };
    auto lstore = [&](int buf, const Stage& r) {
        *reinterpret_cast<f32x4*>(As + (buf * F_KT + arow) * F_SA + (tid & 15) * 4) = r.a0;
        *reinterpret_cast<f32x4*>(As + (buf * F_KT + arow + 16) * F_SA + (tid & 15) * 4) = r.a1;
        if (tid < F_KT * BF4) *reinterpret_cast<f32x4*>(Bs + (buf * F_KT + tid / BF4) * SB + 4 * (tid % BF4)) = r.b0;
        if (BQ > 1) *reinterpret_cast<f32x4*>(Bs + (buf * F_KT + (tid + 256) / BF4) * SB + 4 * ((tid + 256) % BF4)) = r.b1;
    };
#define STAGE_WAIT(N, r) do { if (BQ > 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0), "+v"(r.b1) : "n"(N) : "memory"); \
                              else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0) : "n"(N) : "memory"); } while (0)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto compute = [&](int buf) {
        const float* Ab = As + buf * F_KT * F_SA + 16 * wave + l15;
        const float* Bb = Bs + buf * F_KT * SB + l15;
#pragma unroll
        for (int st = 0; st < F_KT / 4; st++) {
            const float a = Ab[(4 * st + kq) * F_SA];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = MFMA(a, Bb[(4 * st + kq) * SB + 16 * t], acc[t]);
        }
    };
    // prefetch distance 2: tile kt computes from LDS while tile kt+1 lands in one register stage and tile kt+2's loads
    // are issued into the other; one barrier per K tile.
    Stage r0, r1;
    r0.b1 = r1.b1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    gload(0, r0); STAGE_WAIT(0, r0); lstore(0, r0); __syncthreads();
    gload(1, r0);
    for (int kt = 0; kt < nkt; kt += 2) {
        gload(kt + 2, r1);
        compute(0);
        STAGE_WAIT(LPS, r0); lstore(1, r0);
        __syncthreads();
        gload(kt + 3, r0);
        if (kt + 1 < nkt) compute(1);
        STAGE_WAIT(LPS, r1); lstore(0, r1);
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the clamped tail loads before the registers are reused
#undef STAGE_WAIT
    // ---- epilogue: wave w's M-tile
    const int mt = mgrp * 4 + wave;
    if (mt >= p.mtiles) return;
    const int pos = mt / ctiles, ct = mt % ctiles;
    const size_t per_s = (size_t)L.N * L.npos * p.ncols;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + l15;
        f32x4 v = acc[t];
        if (S == 1) {
            const float bias = p.bias[n];
            v.x = act_f(v.x + bias, L.act); v.y = act_f(v.y + bias, L.act); v.z = act_f(v.z + bias, L.act); v.w = act_f(v.w + bias, L.act);
        }
        *reinterpret_cast<f32x4*>(p.out + (size_t)s * per_s + ((size_t)n * L.npos + pos) * p.ncols + ct * 16 + 4 * kq) = v;
    }
}

static int fwd_pick_nt(const LayerDev& L, long mgroups_total, int S) {
    // widest N tile (most reuse of the im2col'd A tile) that still yields >= ~400 workgroups; tiny-M problems (dense
    // layers at B=32) take the widest tile regardless and get their parallelism from split-K
    const int cands[3] = {4, 2, 1};
    int best = 1; long best_wgs = -1;
    for (int c = 0; c < 3; c++) {
        const int nt = cands[c];
        if (L.N % (16 * nt)) continue;
        const long wgs = mgroups_total * (L.N / (16 * nt)) * S;
        if (wgs >= 400 || mgroups_total <= 8) return nt;
        if (wgs > best_wgs) { best_wgs = wgs; best = nt; }
    }
    return best;
}

// one launch for up to four problems sharing the layer geometry (sibling layers x {online net on [s;sp], target net on sp});
// split-K partial slabs are left for the caller to reduce (k_reduce_multi, or folded into the consumer)
bool gemm_fwd_eligible(const LayerDev& L, int nprob, const int* ldx, const int* col0, const int* ncols) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    if (L.N % 16 || L.K % F_KT || (S > 1 && kc % F_KT) || L.K > 8192 || L.w_off % 4) return false;
    for (int i = 0; i < nprob; i++) if (ncols[i] % 16 || ldx[i] % 4 || col0[i] % 4) return false;
    return true;
}
void launch_gemm_fwd(hipStream_t st, const LayerDev& L, int nprob, const float* const* W, const float* const* bias, const float* const* X,
                     const int* ldx, const int* col0, const int* ncols, float* const* out) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    GFwdProbs pr; long mg_total = 0;
    for (int i = 0; i < 4; i++) {
        const int j = i < nprob ? i : 0;
        GFwdProb& q = pr.p[i];
        q.W = W[j]; q.bias = bias[j]; q.X = X[j]; q.ldx = ldx[j]; q.col0 = col0[j]; q.ncols = ncols[j]; q.out = out[j];
        q.mtiles = L.npos * (ncols[j] / 16); q.mgroups = (q.mtiles + 3) / 4;
        if (i < nprob) mg_total += q.mgroups;
    }
    const int NT = fwd_pick_nt(L, mg_total, S);
    const int ngroups = L.N / (16 * NT);
    int end = 0;
    for (int i = 0; i < 4; i++) { if (i < nprob) end += pr.p[i].mgroups * ngroups * S; pr.wg_end[i] = end; }
    const int SB = NT == 4 ? FwdCfg<4>::SB : (NT == 2 ? FwdCfg<2>::SB : FwdCfg<1>::SB);
    const size_t lds = (size_t)(2 * F_KT * F_SA + 2 * F_KT * SB) * 4 + (L.kind == DQN_LAYER_CONV ? (size_t)L.K * 4 : 0);
    if (NT == 4) hipLaunchKernelGGL((k_fwd_lds<4>), dim3(end), dim3(256), lds, st, L, pr, S, kc);
    else if (NT == 2) hipLaunchKernelGGL((k_fwd_lds<2>), dim3(end), dim3(256), lds, st, L, pr, S, kc);
    else hipLaunchKernelGGL((k_fwd_lds<1>), dim3(end), dim3(256), lds, st, L, pr, S, kc);
}
