// common.h -- shared host/device definitions of the MI355X DQN engine (gfx950 only).
// Layout conventions:
//   activations  Y[feature][column]  (batch-innermost, "FB"): 64-lane waves read 64 samples of one
//                feature as one 256-B line; a 16-row MFMA M-tile is 16 samples.
//   weights      W[k][n] (k = cin*kh*kw over the FLIPPED kernel, or n_in), n contiguous; bias[n].
//   replay       rows s[cap][obs], sp[cap][obs] (f32 or u8), a int32, r f32, done u8,
//                sum-tree tree[2*cap2] (root at 1, leaves at cap2+i).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/dqn_mi355x.h"

#define DQN_MAX_LAYERS 32
#define DQN_MAX_ACTIONS 64

struct LayerDev {
    int kind, act, stream, src;       // src: producing layer, -1 = observation batch
    int K, N;                          // forward contraction: K inputs per output, N output channels/units
    int cin, cout, kh, kw, sh, sw, ih, iw, oh, ow;
    int npos;                          // oh*ow (1 for dense)
    int in_feat, out_feat;
    int fwd_kc, dx_kc, dw_kc;          // summation-order plan (0 = unsplit)
    unsigned long long w_off, b_off;   // offsets into the INTERNAL flat parameter vector (w_off 16-B aligned; b_off = w_off + K*N)
    unsigned long long ew_off, eb_off; // offsets into the EXTERNAL (Flux.params order, unpadded) vector
    // LSTM (Flux Recur(LSTMCell)): K = n_in, N = 4H.  internal block [Wi K x 4H][b 4H][Wh H x 4H][junk 4H][h0 H][c0 H][zeros 4H]:
    // Wi|b and Wh|junk are (K+1) x N blocks for the dW kernels; `zeros` is the bias of the bias-free input projection.
    int H; unsigned long long wh_off, h0_off, c0_off, z_off, ewh_off, eh0_off, ec0_off;
    int opt;                           // per-ENGINE experiment switches the kernel launchers look at (DQN_LOPT_*, set at dqn_engine_create from EngineOpts): no process-wide state
    int xu8;                           // this layer reads the observation arena and the arena holds BYTES (u8 replay): value = byte / 255f0, converted in the tile load
};

// device-resident mutable state of one engine (one instance in HBM)
struct StepState {
    unsigned long long sample_ctr;     // Philox counter: number of sample() calls so far
    long long size;                    // _curr_size
    unsigned long long step;           // train steps started so far (bumped by k_td); Adam reads bp[step & 1], writes bp[(step+1) & 1]
    double bp[2][2];                   // Adam beta powers (Flux keeps them per array; identical for all), double-buffered
    unsigned int gnorm_bits;           // max |g| as uint bits (non-negative floats order like uints)
    float loss;
    float gnorm;
    int err;                           // sticky device-side error code
    int pre_valid;                     // 1: the NEXT sample()'s indices are already in idx_pre (computed in the tail of the priority block with the
                                       // final tree); cleared by everything that changes the tree, the size or the counters in between.
                                       // 2: ... and their rows, batch scalars and IS weights are already in the batch arena (PreGather)
};

// (loss, grad_norm) of a train step as the step's LAST launch publishes them into mapped pinned HOST memory (k_publish_scalars): the host reads
// them without a fold launch, a D2H copy or a stream synchronize.  A ring of DQN_MAIL_SLOTS records indexed by the publish sequence number.
#define DQN_MAIL_SLOTS 64
#define DQN_DRAW_SLOTS 32      /* episode-draw slots of the fused recurrent step (DrqnColsArgs): [0,8) / [8,16) the two alternating 8-step graphs, 16 / 17 the two alternating single-step graphs */
struct StepMail {
    float loss, gnorm; int err, pad;
    unsigned long long step;           // StepState::step when published (train steps started so far)
    unsigned long long seq;            // publish sequence number, written LAST (release, system scope): seq == ticket <=> the record is complete
};

static inline __host__ __device__ int dqn_nchunks(int K, int kc) { return (kc <= 0 || kc >= K) ? 1 : (K + kc - 1) / kc; }
static inline __host__ __device__ int dqn_chunk_len(int K, int kc) { return (kc <= 0 || kc >= K) ? K : kc; }
// conv dX: RAW kernel taps (index ky*kw + kx, ascending) per summation chunk (plan.dx_kc; 0 / >= kh*kw = one chunk)
#define DQN_CONV_TAP_CHUNK(L) (((L).dx_kc > 0 && (L).dx_kc < (L).kh * (L).kw) ? (L).dx_kc : (L).kh * (L).kw)

// ---- device math shared by VALU and MFMA epilogues (compiled with -ffp-contract=off: every fused
//      multiply-add is an explicit fmaf / MFMA, never a compiler contraction)
// tanh / sigmoid through Float64 (Flux's Float32 activations round once): OUT OF LINE.  Inlined at every use -- four per epilogue vector, in every kernel -- these two bodies
// were most of the code of the forward kernels (k_fwd_lds<2>: 1185 instructions without them, ~10 000 with) and stood, as branch targets nobody takes at relu / identity
// layers, between the instructions that run (r05: same-box A/B of a build without them: +1.4 % steps/s at config 2).  Same arithmetic, same bits.
__device__ __attribute__((noinline)) static float act_slow(float y, int act) {
    if (act == DQN_ACT_TANH) return (float)tanh((double)y);
    return (float)(1.0 / (1.0 + exp(-(double)y)));
}
__device__ __forceinline__ float act_f(float y, int act) {
#ifdef DQN_PROBE_NO_TRANS      /* TIMING PROBE (r05): tanh / sigmoid compiled out */
    return act == DQN_ACT_RELU ? (y > 0.0f ? y : 0.0f) : y;
#else
    if (act == DQN_ACT_RELU) return y > 0.0f ? y : 0.0f;
    if (act == DQN_ACT_TANH || act == DQN_ACT_SIGMOID) return act_slow(y, act);
    return y;
#endif
}
__device__ __forceinline__ float dact_f(float dy, float y, int act) {
    switch (act) {
    case DQN_ACT_RELU: return y > 0.0f ? dy : 0.0f;
    case DQN_ACT_TANH: { float t = y * y; float u = 1.0f - t; return dy * u; }
    case DQN_ACT_SIGMOID: { float u = 1.0f - y; float t = y * u; return dy * t; }
    default: return dy;
    }
}
// four values at once: ONE decode of the activation code per vector (the element-wise helpers above cost a switch -- three or four scalar branches -- per element, and a
// lone wave pays every branch: r04).  V = any 4-component vector type with .x .y .z .w
template <class V> __device__ __forceinline__ void act_v4(V& v, float bias, int act) {
    const float a = v.x + bias, b = v.y + bias, c = v.z + bias, d = v.w + bias;
    if (act == DQN_ACT_RELU) { v.x = a > 0.0f ? a : 0.0f; v.y = b > 0.0f ? b : 0.0f; v.z = c > 0.0f ? c : 0.0f; v.w = d > 0.0f ? d : 0.0f; }
    else if (act == DQN_ACT_TANH || act == DQN_ACT_SIGMOID) { v.x = act_f(a, act); v.y = act_f(b, act); v.z = act_f(c, act); v.w = act_f(d, act); }
    else { v.x = a; v.y = b; v.z = c; v.w = d; }
}
template <class V> __device__ __forceinline__ void dact_v4(V& v, const V& y, int act) {
    if (act == DQN_ACT_RELU) { v.x = y.x > 0.0f ? v.x : 0.0f; v.y = y.y > 0.0f ? v.y : 0.0f; v.z = y.z > 0.0f ? v.z : 0.0f; v.w = y.w > 0.0f ? v.w : 0.0f; }
    else if (act == DQN_ACT_TANH || act == DQN_ACT_SIGMOID) { v.x = dact_f(v.x, y.x, act); v.y = dact_f(v.y, y.y, act); v.z = dact_f(v.z, y.z, act); v.w = dact_f(v.w, y.w, act); }
}
// (float)b / 255.0f for a byte b (u8 observations, test/test_env.jl:59) in TWO operations beside the conversion instead of the ~10-instruction IEEE division:
// 1/255 = r_hi + r_lo with r_hi = 0x1.01p-8 (9 significant bits: b * r_hi is exact for b < 256) and r_lo = fl(1/255 - r_hi), so fma(b, r_lo, b * r_hi) rounds
// b / 255 (1 + 2^-40) once -- and b / 255 = 0.bbb... in base 256 is never that close to a rounding boundary.  Equal to the division for all 256 bytes
// (tests/test_abi_cpu.py checks the identity in numpy; the u8 parity tests compare with the twin's plain division).  On gfx950 fp32 MFMA and VALU work do
// not overlap (tools/micro/mfma_mix.cpp), so every instruction of this conversion is paid in matrix time: r03's multiply + Newton step was one more.
__device__ __forceinline__ float u8_unit(unsigned b) {
    const float x = (float)b;
    return fmaf(x, 0x1.010102p-24f, x * 0x1.01p-8f);
}
__device__ __forceinline__ float prio_f(float td_abs, float eps, float alpha) {
    float base = td_abs + eps;  // (td + eps)^alpha through Float64 (prioritized_experience_replay.jl:67,77)
    return (float)pow((double)base, (double)alpha);
}

__device__ __forceinline__ void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

// XCD-aware workgroup remap (bijective for any n): hardware block b runs on XCD b % 8 (observed dispatch order; used for
// L2 locality only, never for correctness).  Logical ids are handed out so that each XCD owns one CONTIGUOUS range of
// logical workgroups == a contiguous band of output positions, whose activations then stay in that XCD's private 4 MB L2
// instead of being re-fetched over the fabric by all eight.
// x / d and x % d for the small non-negative ints of the workgroup prologues (tile / position / tap decodes): floor((x + 0.5) * (1 / d)) is exact while
// x < 2^21 even with the 1-ulp hardware reciprocal (error x / d * 2^-22 against the 0.5 / d margin; tests/test_abi_cpu.py checks every multiple +-1 for the divisors in use).  An integer division compiles to ~25-40 instructions; a
// forward workgroup issued ~500 instructions before its first operand load, most of them these (r04), and at B = 32 a wave is alone on its SIMD and pays each.
struct FDiv { int d; float r; };
__device__ __forceinline__ FDiv fdiv_of(int d) { FDiv f; f.d = d; f.r = __builtin_amdgcn_rcpf((float)d); return f; }
__device__ __forceinline__ int fdiv_q(int x, const FDiv& f) { return (int)(((float)x + 0.5f) * f.r); }
__device__ __forceinline__ void fdiv_qr(int x, const FDiv& f, int& q, int& r) { q = fdiv_q(x, f); r = x - q * f.d; }
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// update_priorities!(r, idx, td) (src/prioritized_experience_replay.jl:76-80) executed by ONE workgroup: leaves
// p = (|td| + eps)^alpha (duplicates: last write wins, :79; assert p > 0, :78), then the ancestors level by level (one
// barrier per level; equal parents are written with equal values).  `sidx` is >= n long longs of LDS.
// dense_floats > 0: LDS floats available BEHIND sidx[n] for the dense top (see below).
__device__ __forceinline__ void prio_update_block_levels(int n, long long cap2, const long long* __restrict__ idx, const float* __restrict__ td, float eps,
                                                  float alpha, float* tree, StepState* state, long long* sidx, long long dense_floats = 0, unsigned long long* ktr = nullptr) {
    // 32-bit copies of the indices behind the 64-bit list (the region the dense top uses later): the "last occurrence wins" scan (:79) reads them four
    // at a time.  (r03 ktrace, 512 paths: the scan over the 64-bit list took 57 us of the block's 89 -- one dependent LDS round trip per comparison.)
    int* s32 = reinterpret_cast<int*>(sidx + ((n + 1) / 2) * 2);
    const int n4 = (n + 3) / 4; const bool fast_scan = dense_floats >= (long long)n4 * 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const long long v = idx[i]; sidx[i] = v; if (fast_scan) s32[i] = (int)v; }
    if (fast_scan) for (int i = n + threadIdx.x; i < 4 * n4; i += blockDim.x) s32[i] = -1;
    __syncthreads();
    if (ktr) ktr[6] = __builtin_amdgcn_s_memtime();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long long mine = sidx[i]; int dupes = 0;
        if (fast_scan) {
            const int m32 = (int)mine; const int4* q4 = reinterpret_cast<const int4*>(s32);
#pragma unroll 4
            for (int j4 = (i + 1) / 4; j4 < n4; j4++) { const int4 v = q4[j4]; const int jb = 4 * j4; dupes += (jb > i && v.x == m32) + (jb + 1 > i && v.y == m32) + (jb + 2 > i && v.z == m32) + (jb + 3 > i && v.w == m32); }
        } else {
#pragma unroll 8
            for (int j = 0; j < n; j++) dupes += (j > i && sidx[j] == mine) ? 1 : 0;
        }
        const bool last = dupes == 0;
        if (ktr && i < (int)blockDim.x) ktr[5] = __builtin_amdgcn_s_memtime();
        const float p = prio_f(fabsf(td[i]), eps, alpha);
        if (!(p > 0.0f)) state->err = 2;
        if (last) tree[cap2 + sidx[i]] = p;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) sidx[i] = (cap2 + sidx[i]) >> 1;
    __syncthreads();
    if (ktr) ktr[4] = __builtin_amdgcn_s_memtime();
    // The TOP of the tree is recomputed DENSELY out of LDS: with hundreds of paths nearly every node of the upper levels is an ancestor of an
    // updated leaf, and walking them path by path costs a dependent global round trip per level (r03: 120 us for 512 paths x 20 levels, 6 us per
    // level).  So the sparse walk stops at the level of W nodes; those W values are then read back in one round trip, every level above is
    // rebuilt as node = left + right (for an untouched node that reproduces the stored bits: every internal node always was the fp32 sum of its
    // current children) and stored in one coalesced pass.
    long long W = 0;
    if (dense_floats >= 128) { W = 64; while (4 * W <= dense_floats && 2 * W <= cap2 / 2) W *= 2; if (W > cap2 / 2) W = 0; }
    float* top = reinterpret_cast<float*>(sidx + ((n + 1) / 2) * 2);      // 16-B aligned behind the n indices; top[id], ids in [1, 2W)
    for (long long width = cap2; width > 1 && (W == 0 || width >= 2 * W); width >>= 1) {
        // up to 4 nodes per thread (n <= 1024 at 256 threads): all child loads are issued before any store, so a level costs one
        // memory round trip however many nodes a thread owns
        long long nd[4]; float vs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = threadIdx.x + u * blockDim.x; nd[u] = -1; if (i < n) { nd[u] = sidx[i]; vs[u] = tree[2 * nd[u]] + tree[2 * nd[u] + 1]; } }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int i = threadIdx.x + u * blockDim.x; if (nd[u] >= 0) { tree[nd[u]] = vs[u]; sidx[i] = nd[u] >> 1; } }
        for (int i = threadIdx.x + 4 * blockDim.x; i < n; i += blockDim.x) { const long long node = sidx[i]; tree[node] = tree[2 * node] + tree[2 * node + 1]; sidx[i] = node >> 1; }
        __syncthreads();
    }
    if (W > 0) {
        for (long long i = threadIdx.x; i < W; i += blockDim.x) top[W + i] = tree[W + i];       // the level of W nodes: final in memory
        __syncthreads();
        for (long long w = W / 2; w >= 1; w >>= 1) {
            for (long long i = threadIdx.x; i < w; i += blockDim.x) top[w + i] = top[2 * (w + i)] + top[2 * (w + i) + 1];
            __syncthreads();
        }
        for (long long i = threadIdx.x + 1; i < W; i += blockDim.x) tree[i] = top[i];
        __syncthreads();
    }
}
// The same update for SMALL batches (n <= 64, <= 22 levels) in TWO memory round trips instead of one per level: every path's siblings are
// requested up front (they are independent of the new leaf values), then the ancestors are recomputed level by level in LDS -- a sibling that
// is itself on an updated path takes that path's freshly computed value instead of the prefetched one -- and stored on the way.  Same node
// arithmetic (node = f32(left + right) of its current children), so the resulting tree is identical.  `lds`: >= 8 KB (7808 bytes used).
__device__ __forceinline__ void prio_update_block(int n, long long cap2, const long long* __restrict__ idx, const float* __restrict__ td, float eps,
                                                  float alpha, float* tree, StepState* state, long long* lds, unsigned lds_bytes = 0, unsigned long long* ktr = nullptr) {
    int L = 0; for (long long w = cap2; w > 1; w >>= 1) L++;
    if (n > 64 || L > 22) {
        const long long used = (long long)((n + 1) / 2) * 16;      // bytes of the index list (padded to 16)
        prio_update_block_levels(n, cap2, idx, td, eps, alpha, tree, state, lds, (long long)lds_bytes > used ? ((long long)lds_bytes - used) / 4 : 0, ktr); return;
    }
    long long* node = lds;                                 // [64]  leaf node id of path i
    float* val = reinterpret_cast<float*>(lds + 64);       // [64]  value of path i's node at the current level
    float* sib = val + 64;                                 // [L][64] prefetched sibling values
    signed char* sj = reinterpret_cast<signed char*>(sib + 22 * 64);   // [L][64] path whose node is path i's sibling at level l, or -1
    const int t = threadIdx.x;
    if (t < n) node[t] = cap2 + idx[t];
    __syncthreads();
    // siblings of every level of every path: n * L independent loads, one round trip
    const float rn = 1.0f / (float)n;
    for (int q = t; q < n * L; q += blockDim.x) { const int l = (int)(((float)q + 0.5f) * rn), i = q - l * n; sib[l * 64 + i] = tree[(node[i] >> l) ^ 1]; }
    if (t < n) {
        // duplicates: the LAST occurrence decides the leaf (last write wins, :79); every occurrence carries that value up
        int last = t;
        for (int l = 0; l < L; l++) sj[l * 64 + t] = -1;
        const long long mine = node[t];
        for (int j = 0; j < n; j++) {
            const long long x = node[j] ^ mine;
            if (x == 0) { if (j > last) last = j; continue; }
            // paths i and j are siblings exactly at the level of their highest differing bit (above it they are the same node); several j with
            // the same level have merged with each other by then, so any of them carries the sibling's value
            sj[(63 - __clzll(x)) * 64 + t] = (signed char)j;
        }
        const float p = prio_f(fabsf(td[last]), eps, alpha);
        if (!(prio_f(fabsf(td[t]), eps, alpha) > 0.0f)) state->err = 2;      // assert all(new_priorities .> 0) (:78)
        val[t] = p;
        if (last == t) tree[mine] = p;
    }
    __syncthreads();
    for (int l = 0; l < L; l++) {
        float parent = 0.0f; long long c = 0;
        if (t < n) {
            c = node[t] >> l;
            const int j = sj[l * 64 + t];
            const float sv = j >= 0 ? val[j] : sib[l * 64 + t];                    // an updated sibling subtree: its fresh value
            parent = (c & 1) ? sv + val[t] : val[t] + sv;                           // left + right
        }
        __syncthreads();
        if (t < n) { val[t] = parent; tree[c >> 1] = parent; }
        __syncthreads();
    }
}
#ifdef __HIPCC__
typedef float f32x4c __attribute__((ext_vector_type(4)));
// one stratified sum-tree descent (sample(), src/prioritized_experience_replay.jl:82-87): stratum i of B, Philox4x32-10 keyed by (seed, call counter, i)
__device__ __forceinline__ long long tree_descend(const float* __restrict__ tree, long long cap2, long long size, unsigned long long seed,
                                                  unsigned long long ctr, int i, float seg) {
    uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)i, 0x5A4D504Cu};
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
    const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);
    float t = ((float)i + u) * seg;
    long long node = 1;
    // three levels per memory round trip: the 2 children, 4 grandchildren and 8 great-grandchildren of a heap node are three
    // contiguous runs (2n.., 4n.., 8n..), fetched with independent 8/16-byte loads; the three left/right decisions then use exactly
    // the values (and the comparisons) of the one-level walk below, so the chosen leaf is identical.
    while (8 * node < 2 * cap2) {
        const float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
        const f32x4c g = *reinterpret_cast<const f32x4c*>(tree + 4 * node);
        const f32x4c h0 = *reinterpret_cast<const f32x4c*>(tree + 8 * node), h1 = *reinterpret_cast<const f32x4c*>(tree + 8 * node + 4);
        int b1 = 0, b2 = 0, b3 = 0;
        if (!(t < c.x || !(c.y > 0.0f))) { t -= c.x; b1 = 1; }
        const float gl = b1 ? g.z : g.x, gr = b1 ? g.w : g.y;
        if (!(t < gl || !(gr > 0.0f))) { t -= gl; b2 = 1; }
        const f32x4c hh = b1 ? h1 : h0;
        const float hl = b2 ? hh.z : hh.x, hr = b2 ? hh.w : hh.y;
        if (!(t < hl || !(hr > 0.0f))) { t -= hl; b3 = 1; }
        node = 8 * node + 4 * b1 + 2 * b2 + b3;
    }
    while (node < cap2) {
        const float l = tree[2 * node], rg = tree[2 * node + 1];
        if (t < l || !(rg > 0.0f)) node = 2 * node; else { t -= l; node = 2 * node + 1; }
    }
    long long leaf = node - cap2; if (leaf >= size) leaf = size - 1;
    return leaf;
}
#endif
#ifdef __HIPCC__
// hp.sample_distinct (...replay.jl:85, replace=false): after the stratified draws the positions are visited in ascending order and every index an earlier
// position already took is redrawn by successive sampling on the residual priorities -- u * (total - taken mass) walked down the tree, a child's mass being its
// stored sum minus the priorities of the taken leaves below it (in the order they were taken), 0 when no untaken leaf is left below it; Philox lane B + i,
// word 3 offset by the attempt; after 8 attempts the first untaken leaf in index order.  Sequential by construction (each redraw excludes the ones before it).
//
// r06: executed by ONE WAVE instead of one lane, so that large batches can keep distinct draws on the fast path (the priority workgroup of a backward launch):
//   * taken[i] / tp[i] arrive FILLED with every position's stratified draw and its priority, tp[i] NEGATED where the draw repeats an earlier position's DRAW
//     (sample_distinct_block computes that in parallel).  Position i is a duplicate iff its draw equals an earlier draw, or an earlier REDRAWN leaf: an earlier position
//     with the same draw either kept it or was redrawn because someone before it holds it -- so until the first redraw the flag alone decides, afterwards the wave scans the
//     final leaves of the positions before i (64 per step);
//   * the serial parts of a redraw keep their order but only over the entries that matter: "which taken leaves lie below this child" is a ballot over the taken list, the
//     subtractions then run over the set bits in ascending order (level l sees ~nt / 2^l of them: ~2 nt per descent instead of 2 L nt).
// Every lane computes the same scalars (wave-uniform control flow); same arithmetic, same order, same Philox words as the one-lane form the CPU twin restates: identical lists.
__device__ __forceinline__ bool sdw_taken_before(const long long* taken, int n, long long leaf, int lane) {      // leaf in taken[0, n)?  (one wave)
    bool hit = false;
    for (int j0 = 0; j0 < n; j0 += 64) { const int j = j0 + lane; hit = hit || (j < n && taken[j] == leaf); }
    return __ballot(hit) != 0ull;
}
__device__ __forceinline__ void sample_distinct_fix(const float* __restrict__ tree, long long cap2, long long size, unsigned long long seed, unsigned long long ctr,
                                                    int B, long long* idx, long long* taken, float* tp) {
    int L = 0; for (long long w = cap2; w > 1; w >>= 1) L++;
    const int lane = threadIdx.x & 63;
    if (size < B) { for (int i = lane; i < B; i += 64) tp[i] = fabsf(tp[i]); return; }
    int nr = 0;                                        // redraws so far
    for (int i = 0; i < B; i++) {
        const float f = tp[i]; const long long x = taken[i];
        bool dup = f < 0.0f;
        if (!dup && nr > 0) dup = sdw_taken_before(taken, i, x, lane);
        if (!dup) continue;                            // taken[i] / tp[i] already hold this position's leaf and priority
        const int nt = i;
        long long leaf = x; bool ok = false;
        for (int att = 0; att < 8 && !ok; att++) {
            float R = tree[1]; for (int j = 0; j < nt; j++) R = R - tp[j];
            uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(B + i), 0x5A4D504Cu + (uint32_t)(att + 1)};
            philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
            float t = (float)(c[0] >> 8) * (1.0f / 16777216.0f) * R;
            long long node = 1;
            for (int lev = 0; lev < L; lev++) {
                float m[2];
#pragma unroll
                for (int ch = 0; ch < 2; ch++) {
                    const long long cn = 2 * node + ch; const int sh = L - lev - 1;
                    long long lo = (cn << sh) - cap2, hi = ((cn + 1) << sh) - cap2; if (hi > size) hi = size;
                    long long cnt = hi > lo ? hi - lo : 0; float v = tree[cn];
                    for (int j0 = 0; j0 < nt; j0 += 64) {      // the taken leaves below cn, in the order they were taken
                        const int j = j0 + lane;
                        unsigned long long mb = __ballot(j < nt && ((taken[j] + cap2) >> sh) == cn);
                        while (mb) { const int b = __ffsll((long long)mb) - 1; mb &= mb - 1; v = v - tp[j0 + b]; cnt--; }
                    }
                    m[ch] = (cnt > 0 && v > 0.0f) ? v : 0.0f;
                }
                if (t < m[0] || !(m[1] > 0.0f)) node = 2 * node; else { t -= m[0]; node = 2 * node + 1; }
            }
            leaf = node - cap2; if (leaf >= size) leaf = size - 1;
            ok = !sdw_taken_before(taken, nt, leaf, lane);
        }
        if (!ok) for (leaf = 0; leaf < size; leaf++) if (!sdw_taken_before(taken, nt, leaf, lane)) break;
        __builtin_amdgcn_wave_barrier();               // every lane is past its reads of taken / tp for this position
        if (lane == 0) { idx[i] = leaf; taken[i] = leaf; tp[i] = tree[cap2 + leaf]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // LDS operations of a wave execute in order: the next position reads these
        nr++;
    }
}
#endif
#ifdef __HIPCC__
// hp.sample_distinct, by a whole workgroup: `list` holds the B stratified draws; every lane tests its draws against the ones before them (out of `taken`, O(B) per lane),
// fetches their priorities, and only when some draw repeats an earlier one does wave 0 run the sequential redraw -- without a duplicate it would change nothing.  Every thread
// of the workgroup calls; taken: B long longs, tp: B floats, any: one int of scratch (LDS).  k_sample, the priority block's pre-draw, the fused sample + gather workgroups and
// the single-launch step all go through here: the same list whichever of them draws it (the reference's replace=false semantics on the fast path, ...replay.jl:85).
__device__ __forceinline__ void sample_distinct_block(const float* __restrict__ tree, long long cap2, long long size, unsigned long long seed, unsigned long long ctr,
                                                      int B, long long* list, long long* taken, float* tp, int* any) {
    if (threadIdx.x == 0) *any = 0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) taken[i] = list[i];
    __syncthreads();
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const long long v = taken[i]; bool d = false;
        int j = 0;
        for (; j + 8 <= i; j += 8) {                   // eight independent LDS reads per round (one read per iteration was a dependent ~100-cycle round trip each: 21 us of a priority block at B = 512)
            long long w[8];
#pragma unroll
            for (int u = 0; u < 8; u++) w[u] = taken[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) d = d || w[u] == v;
        }
        for (; j < i; j++) d = d || taken[j] == v;
        if (d) *any = 1;
        tp[i] = d ? -1.0f : 1.0f;                      // (the sign is the flag; the priorities are only needed -- and only fetched -- when something repeats)
    }
    __syncthreads();
    if (*any) {
        for (int i = threadIdx.x; i < B; i += blockDim.x) tp[i] = tp[i] * tree[cap2 + taken[i]];      // priorities are > 0: the sign survives
        __syncthreads();
        if (threadIdx.x < 64) sample_distinct_fix(tree, cap2, size, seed, ctr, B, list, taken, tp);
    }
    __syncthreads();
}
#endif
struct PrioArgs { int n; long long cap2; const long long* idx; const float* td; float eps, alpha; float* tree;
                  long long* idx_pre; unsigned long long seed; int B;       // idx_pre != nullptr: also draw the next step's B indices (see prio_block_run)
                  int phase;                                             // prio_block_fast only: 0 = update + draw, 1 = update, 2 = draw (split over two launches of one step)
                  int distinct; };                                       // hp.sample_distinct: the pre-drawn list is deduped like sample()'s (sample_distinct_block)
#ifdef __HIPCC__
// the priority workgroup of a step: update_priorities!(replay, idx, td), then -- the tree is final and the Philox counter of the next sample()
// is known (k_td / k_head_td bumped it earlier in this step) -- the NEXT step's B stratified descents, so that the next gather launch starts
// with its row loads instead of ~5 dependent round trips per workgroup.  Anything that touches the tree, the size or the counters before
// that gather clears state->pre_valid and the gather descends itself, exactly as before.
// continue a descent from `node` with residual mass t: three levels per memory round trip, exactly tree_descend's decisions
__device__ __forceinline__ long long tree_descend_from(const float* __restrict__ tree, long long cap2, long long node, float t) {
    while (8 * node < 2 * cap2) {
        const float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
        const f32x4c g = *reinterpret_cast<const f32x4c*>(tree + 4 * node);
        const f32x4c h0 = *reinterpret_cast<const f32x4c*>(tree + 8 * node), h1 = *reinterpret_cast<const f32x4c*>(tree + 8 * node + 4);
        int b1 = 0, b2 = 0, b3 = 0;
        if (!(t < c.x || !(c.y > 0.0f))) { t -= c.x; b1 = 1; }
        const float gl = b1 ? g.z : g.x, gr = b1 ? g.w : g.y;
        if (!(t < gl || !(gr > 0.0f))) { t -= gl; b2 = 1; }
        const f32x4c hh = b1 ? h1 : h0;
        const float hl = b2 ? hh.z : hh.x, hr = b2 ? hh.w : hh.y;
        if (!(t < hl || !(hr > 0.0f))) { t -= hl; b3 = 1; }
        node = 8 * node + 4 * b1 + 2 * b2 + b3;
    }
    if (4 * node < 2 * cap2) {      // two levels left: one round trip
        const float2 c = *reinterpret_cast<const float2*>(tree + 2 * node);
        const f32x4c g = *reinterpret_cast<const f32x4c*>(tree + 4 * node);
        int b1 = 0, b2 = 0;
        if (!(t < c.x || !(c.y > 0.0f))) { t -= c.x; b1 = 1; }
        const float gl = b1 ? g.z : g.x, gr = b1 ? g.w : g.y;
        if (!(t < gl || !(gr > 0.0f))) { t -= gl; b2 = 1; }
        node = 4 * node + 2 * b1 + b2;
    }
    while (node < cap2) {
        const float l = tree[2 * node], rg = tree[2 * node + 1];
        if (t < l || !(rg > 0.0f)) node = 2 * node; else { t -= l; node = 2 * node + 1; }
    }
    return node;
}
// The priority workgroup in FEWER DEPENDENT ROUND TRIPS (small batches: n <= 64 paths, B <= blockDim draws): the top TOPN nodes of the sum-tree
// are copied into LDS while the index list is fetched, the update patches that copy as it rewrites the ancestors (one wave, no workgroup
// barrier per level), and the B descents of the next sample() walk the copy -- 11-12 of the 14 levels at 10 000 transitions -- before ONE global
// round trip finishes them.  Same node arithmetic, same comparisons, same Philox draws as prio_update_block + tree_descend: identical tree,
// identical indices.  r03 ktrace: the two-phase block lived 10-13 us (9 dependent round trips at 0.7-1.5 us) and bounded whichever backward
// launch carried it.  lds: 7808 bytes of path state + TOPN floats.
// ctr_in: the Philox counter of the NEXT sample() when the caller bumped it itself earlier in the SAME kernel (k_tiny_step: a uniform re-read of
// state->sample_ctr may be served by the scalar cache, which does not see the kernel's own vector stores); ~0: read it from the state
__device__ __forceinline__ bool prio_block_fast(const PrioArgs& P, StepState* state, long long* lds, unsigned lds_bytes, unsigned long long* ktr = nullptr, unsigned long long ctr_in = ~0ull) {
    int L = 0; for (long long w = P.cap2; w > 1; w >>= 1) L++;
    const int n = P.n, B = P.B;
    if (n < 1 || n > 64 || L > 22 || L < 3 || !P.idx_pre || B > (int)blockDim.x || lds_bytes < 7808 + 64 * 4 || P.cap2 < 64) return false;
    long long* node = lds;                                 // [64]  leaf node id of path i
    float* val = reinterpret_cast<float*>(lds + 64);       // [64]  value of path i's node at the current level
    float* sib = val + 64;                                 // [L][64] prefetched sibling values
    signed char* sj = reinterpret_cast<signed char*>(sib + 22 * 64);   // [L][64]
    float* top = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + 7808);          // 16-B aligned (lds is): node ids [0, TOPN)
    long long topn = 64; { const long long avail = (long long)(lds_bytes - 7808) / 4; while (2 * topn <= avail && 2 * topn <= P.cap2 && 2 * topn <= 8192) topn *= 2; }
    const int t = threadIdx.x;
    float* tree = P.tree;
    const int phase = P.phase;                             // 0: update + draw; 1: update only; 2: draw only (the tree is final: a later launch of the same step)
    // ---- round trip 1: the index list, the TD errors, the step counters, the top of the tree (all independent).  The Float64 pow of the new
    // priorities (~1.7 us of dependent arithmetic per lane) runs while the tree copy is in flight.
    long long my_idx = 0; float my_td = 0.0f; if (t < n && phase != 2) { my_idx = P.idx[t]; my_td = P.td[t]; }
    const unsigned long long ctr = ctr_in != ~0ull ? ctr_in : state->sample_ctr; const long long size = state->size;
    f32x4c tq[8];                                          // <= 8192 nodes at 256 threads
    const int nq = (int)((topn / 4 + blockDim.x - 1) / blockDim.x);
#pragma unroll
    for (int q = 0; q < 8; q++) if (q < nq) { const long long i = 4 * ((long long)q * blockDim.x + t); if (i < topn) tq[q] = *reinterpret_cast<const f32x4c*>(tree + i); }
    float my_p = 0.0f;
    if (t < n && phase != 2) { my_p = prio_f(fabsf(my_td), P.eps, P.alpha); if (!(my_p > 0.0f)) state->err = 2; val[t] = my_p; node[t] = P.cap2 + my_idx; }      // assert all(new_priorities .> 0) (:78)
#pragma unroll
    for (int q = 0; q < 8; q++) if (q < nq) { const long long i = 4 * ((long long)q * blockDim.x + t); if (i < topn) *reinterpret_cast<f32x4c*>(top + i) = tq[q]; }
    __syncthreads();
    if (ktr) ktr[4] = __builtin_amdgcn_s_memtime();
    if (phase != 2) {
    // ---- round trip 2: siblings below the LDS copy; everything above comes out of the copy
    const float rn = 1.0f / (float)n;
    for (int q = t; q < n * L; q += blockDim.x) { const int l = (int)(((float)q + 0.5f) * rn), i = q - l * n; const long long sid = (node[i] >> l) ^ 1; sib[l * 64 + i] = sid < topn ? top[sid] : tree[sid]; }
    float leaf_p = 0.0f; int last = t;
    if (t < n) {
        for (int l = 0; l < L; l++) sj[l * 64 + t] = -1;
        const long long mine = node[t];
        for (int j = 0; j < n; j++) {
            const long long x = node[j] ^ mine;
            if (x == 0) { if (j > last) last = j; continue; }
            sj[(63 - __clzll(x)) * 64 + t] = (signed char)j;
        }
        leaf_p = val[last];                                // duplicates: the LAST occurrence decides the leaf (:79); every occurrence carries that value up
    }
    __syncthreads();
    if (t < n) { val[t] = leaf_p; if (last == t) tree[node[t]] = leaf_p; }
    __syncthreads();
    if (ktr) ktr[5] = __builtin_amdgcn_s_memtime();
    // ---- the ancestors, level by level, by ONE wave (LDS operations of a wave execute in order: no barrier between levels)
    if (t < 64) {
        for (int l = 0; l < L; l++) {
            float parent = 0.0f; long long c = 0;
            if (t < n) {
                c = node[t] >> l;
                const int j = sj[l * 64 + t];
                const float sv = j >= 0 ? val[j] : sib[l * 64 + t];
                parent = (c & 1) ? sv + val[t] : val[t] + sv;                           // left + right
            }
            __builtin_amdgcn_wave_barrier();      // every lane has read this level's val[] before any lane overwrites it
            if (t < n) { val[t] = parent; tree[c >> 1] = parent; if ((c >> 1) < topn) top[c >> 1] = parent; }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    }
    if (ktr) ktr[6] = __builtin_amdgcn_s_memtime();
    if (phase == 1) return true;
    // ---- the next sample(): B stratified descents; the LDS copy serves the levels whose children it holds
    if (t < B) {
        const float seg = top[1] / (float)B;
        uint32_t c4[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)t, 0x5A4D504Cu};
        philox4x32_10((uint32_t)P.seed, (uint32_t)(P.seed >> 32), c4);
        const float u = (float)(c4[0] >> 8) * (1.0f / 16777216.0f);
        float m = ((float)t + u) * seg;
        long long nd = 1;
        while (2 * nd + 1 < topn) {
            const float l = top[2 * nd], rg = top[2 * nd + 1];
            if (m < l || !(rg > 0.0f)) nd = 2 * nd; else { m -= l; nd = 2 * nd + 1; }
        }
        nd = tree_descend_from(tree, P.cap2, nd, m);
        long long leaf = nd - P.cap2; if (leaf >= size) leaf = size - 1;
        if (P.distinct) node[t] = leaf; else P.idx_pre[t] = leaf;      // (the path state is free by now: the draws of a distinct list stay in LDS until they are deduped)
    }
    __syncthreads();
    if (P.distinct) {
        sample_distinct_block(tree, P.cap2, size, P.seed, ctr, B, node, lds + 64, reinterpret_cast<float*>(lds + 128), reinterpret_cast<int*>(lds + 160));
        if (t < B) P.idx_pre[t] = node[t];
        __syncthreads();
    }
    if (threadIdx.x == 0) state->pre_valid = 1;
    return true;
}
// the priority workgroup of a step: update_priorities!(replay, idx, td), then -- the tree is final and the Philox counter of the next sample()
// is known (k_td / k_head_td bumped it earlier in this step) -- the NEXT step's B stratified descents, so that the next gather launch starts
// with its row loads instead of ~5 dependent round trips per workgroup.  Anything that touches the tree, the size or the counters before
// that gather clears state->pre_valid and the gather descends itself, exactly as before.  lds_bytes: LDS behind `sidx` (0: unknown -- the
// two-phase form, which needs 8 KB).
__device__ __forceinline__ void prio_block_run(const PrioArgs& P, StepState* state, long long* sidx, unsigned lds_bytes = 0, unsigned long long* ktr = nullptr, unsigned long long ctr_in = ~0ull) {
    if (lds_bytes && prio_block_fast(P, state, sidx, lds_bytes, ktr, ctr_in)) return;
    if (P.phase != 2) prio_update_block(P.n, P.cap2, P.idx, P.td, P.eps, P.alpha, P.tree, state, sidx, lds_bytes, ktr);
    if (!P.idx_pre || P.phase == 1) return;
    __syncthreads();
    const unsigned long long ctr = ctr_in != ~0ull ? ctr_in : state->sample_ctr; const long long size = state->size;
    const float seg = P.tree[1] / (float)P.B;
    if (P.distinct) {      // draws -> LDS, dedupe, publish; without room for the scratch (20 bytes per draw) nothing is pre-drawn: the next sample() draws itself
        const unsigned have = lds_bytes ? lds_bytes : 8192u;
        if ((unsigned)P.B * 20u + 8u > have) { if (threadIdx.x == 0) state->pre_valid = 0; __syncthreads(); return; }      // (a stale pre_valid would let the next gather reuse the batch it already consumed)
        long long* list = sidx; long long* taken = sidx + P.B; float* tp = reinterpret_cast<float*>(sidx + 2 * P.B); int* any = reinterpret_cast<int*>(tp + P.B);
        for (int i = threadIdx.x; i < P.B; i += blockDim.x) list[i] = tree_descend(P.tree, P.cap2, size, P.seed, ctr, i, seg);
        __syncthreads();
        sample_distinct_block(P.tree, P.cap2, size, P.seed, ctr, P.B, list, taken, tp, any);
        for (int i = threadIdx.x; i < P.B; i += blockDim.x) P.idx_pre[i] = list[i];
    } else
    for (int i = threadIdx.x; i < P.B; i += blockDim.x) P.idx_pre[i] = tree_descend(P.tree, P.cap2, size, P.seed, ctr, i, seg);
    __syncthreads();
    if (threadIdx.x == 0) state->pre_valid = 1;
}
#endif
// deferred dW split-K slabs reduced INSIDE the Adam launch by dedicated blocks (single-GPU path): element ranges [beg, end) of the
// gradient vector, each the ascending sum of S slabs
struct AdamSegs { int n; unsigned long long beg[8], end[8]; const float* part[8]; int S[8]; unsigned blocks; unsigned long long stride[8]; };   // stride 0: slabs `len` apart
// one Adam job (adam_body.h): element ranges streamed 16 B at a time, slab ranges reduced on the fly, optionally the priority update and
// the beta-power tick.  Run by k_adam or by tail workgroups of a backward launch.
struct AdamJob {
    float *p, *m, *v; const float* g; float* g_out; StepState* state; float* gmax_part; int slot0;
    int f64mode; float lr; double b1, b2, eps; float gscale;
    int nr; unsigned long long beg[4], end[4];
    AdamSegs segs; PrioArgs prio; int tick; unsigned sblocks;
    // recurrent fused step: the tick thread also folds the loss from the per-column Huber terms, loss = (sum_t (sum_b hl[t*B + b]) / B) / T (src/solver.jl:276-281)
    const float* fold_hl; int fold_T, fold_B;
    int wt;      // m, v stored write-through (small-batch engines: DQN_LOPT_ST_WT)
};
static inline __host__ __device__ unsigned adam_job_blocks(const AdamJob& j) { return (j.prio.n > 0 ? 1u : 0u) + j.segs.blocks + j.sblocks; }

// ---- kernel launchers (defined in the .hip files; all enqueue on `st` and never synchronise)
// a head tensor as seen by k_td: finished activation (S <= 1) or split-K partial slabs to be reduced on the fly
struct HeadSrc { const float* p; int ld; int S; unsigned long long per_s; const float* bias; int act; };
struct TdArgs {
    int B, nA, ncon, dueling, double_q, prioritized, bump_sample_ctr;
    float gamma, prio_beta, prio_eps, prio_alpha;
    long long cap2;
    const long long* idx; const int* a; const float* r; const unsigned char* done; float* tree;
    HeadSrc on_val, on_adv, tg_val, tg_adv;            // last-layer outputs (adv doubles as the plain Q head)
    float *d_val, *d_adv;                              // dpre of the last layers, [*][B]
    float *w_is, *td, *q_on_s, *q_on_sp, *q_tg_sp, *ytarget; int* best;
    StepState* st;
    long long* idx_mut; const long long* idx_pre; int take_pre;      // take_pre: this step's batch was drawn and gathered by the previous step (PreGather)
};

// ---- fused head kernel (small batches): head forwards of both nets + dueling reduce + argmax + Bellman target + TD + Huber + dL/dQ +
// the head layers' dX, ONE workgroup per batch column (nn_valu.hip: k_head_td).  A head layer reads its input column(s) straight from the
// producing layer's finished activations (or the observation arena): online net columns c0 + b (s) and c0 + B + b (sp), target net column c0 + b.
struct HeadLayer {
    int K, N, S, kc, act;                       // forward contraction K -> N in S chunks of kc (plan fwd_kc), head activation
    const float *W[2], *bias[2];                // [net]: online, target
    const float* X[2]; int ldx[2], c0[2];       // [net]
    const float* XT[2];                         // [net] optional transposed copy [column][K] of X (see RSeg::outT); column index as for X
    float* dpre;                                // [N][B]: dL/d(pre-activation) of the head (read by the head's dW task)
    float* dsrc; const float* ysrc; int ldy, act_src;   // dX: dsrc[K][B] = dact(W * dpre, ysrc[k][b]); null when the head reads the observation
};
struct HeadTdArgs {
    int B, nA, dueling, double_q, bump_sample_ctr, join;     // join: both heads read the SAME producer (dX_val + dX_adv, src/dueling.jl:10 backward)
    int stage_w;                                             // the head weights of both nets are staged in LDS (they fit; K*N % 4 == 0)
    int dbg;                                                 // timing experiments only (env DQN_HEAD_DBG): return after phase dbg (0 = run everything)
    float gamma, prio_beta; long long cap2;
    const int* bm_a; const float *bm_r, *bm_done, *bm_w;      // batch scalars of the B columns (written by the gather launch)
    HeadLayer val, adv;                         // adv doubles as the plain Q head when !dueling
    float *w_is, *td, *q_on_s, *q_on_sp, *q_tg_sp, *ytarget, *hl; int* best;
    StepState* st;
    long long* idx; const long long* idx_pre;   // take_pre launches: this step's indices were drawn AND gathered by the previous step (PreGather)
};
size_t head_td_lds_bytes(const HeadTdArgs& a);
void launch_head_td(hipStream_t st, const HeadTdArgs& a, const HeadTdArgs* a_dev, int bump_sample_ctr, int take_pre = 0);   // a_dev: the same record in device memory

// ---- split-K reduce of the last hidden layer + the head level in ONE chip-filling launch (red_head.hip, r05): workgroup = (4 batch columns, stream, plan chunk of 32 hidden
// rows); the last arriver of a column group does the TD arithmetic and the heads' dX.  Stream 0 = advantage / plain Q head, stream 1 = value head.
struct RedHeadStream {
    const float* part[2];              // [net] split-K slabs [S][K][ncols(net)] of the hidden layer's forward (net 0: ncon columns, net 1: B columns)
    const float* pbias[2]; int pact;   // [net] hidden layer bias [K], its activation
    const float* W[2]; const float* hbias[2]; int N, hact;      // [net] head weights [K][N], bias [N]; head activation
    const float* partT[2];             // [net] S == 1 only, optional: the transposed copy [column][K] of the finished activation (written by the dense forward's epilogue)
    float* y_on;                       // online hidden activations [K][ncon]: columns 0..B-1 (the s columns) are written -- what the backward pass reads
    float* dpre;                       // [N][B]  dL/d(pre-activation) of the head
    float* dsrc;                       // [K][B]  dL/d(pre-activation) of the hidden layer
};
struct RedHeadArgs {
    int B, nA, K, S, ncon, nstream, NO, double_q;      // K hidden rows (= head inputs), S slabs, NO = head outputs of both streams (nA, + 1 with a value stream)
    int pm;                                            // the slabs are PIECE-MAJOR: [S][column quad][K][4] (written so by the forward launch, GFwdProb::pm)
    float gamma;
    RedHeadStream st[2];
    const int* bm_a; const float *bm_r, *bm_done, *bm_w;      // batch scalars of the B columns (written by the gather launch)
    float *w_is, *td, *q_on_s, *q_on_sp, *q_tg_sp, *ytarget, *hl; int* best;
    StepState* stt; long long* idx; const long long* idx_pre;
    float* partials;                   // [B/4][3 slots][4 columns][NO][K/32] chunk sums of the head outputs
    float* ypm;                        // k_red_head: [B/4][nstream][K][4] piece-major copy of the online hidden activations of the s columns (the last arriver's act' operand); null: read y_on
    unsigned* tickets;                 // [B/4] arrival counters (zero between launches: the last arriver re-arms its group's)
    unsigned long long* stamps;        // timing probe (DQN_DRQN_STAMPS at create, tools/red_head_phases.py), else null
};
size_t red_head_lds_bytes(const RedHeadArgs& a);
bool red_head_ok(int B, int K, int S, int nA, int nstream, int N0, int N1);
void launch_red_head(hipStream_t st, const RedHeadArgs& a, const RedHeadArgs* a_dev, int bump_sample_ctr, int take_pre = 0);

// task / segment tables of the batched small kernels (device-resident, built once per engine)
struct VTask {
    int kind;                          // 0 forward, 1 dW/db, 2 dX, 3 loss fold (dpre = B Huber terms, out = &state->loss)
    LayerDev L; const float* P; const float* X; int ldx, col0, ncols; const float* dpre; int B; float* out; int S, kc;
    const float* addend; const float* ysrc; int ldy, act_src; unsigned first_block;
};
struct RSeg {
    const float* part; int S, S2; unsigned long long elems; int mode; const float* bias; int per_n, act;
    const float* addend; const float* ysrc; int B, ldy; float* out; unsigned first_block;
    float* outT; int ncolsT;           // mode 0, optional: a TRANSPOSED copy outT[col][feature] (feature contiguous) of the [feature][ncolsT] activation,
                                       // so that the fused head kernel reads a batch column as one contiguous run instead of one 128-B line per element
};
unsigned valu_task_blocks(const VTask& T);
void launch_valu_multi(hipStream_t st, const VTask* tasks_dev, int ntasks, unsigned total_blocks, const PrioArgs* prio = nullptr /* workgroup 0 = priority block */, StepState* state = nullptr);
void launch_reduce_multi(hipStream_t st, const RSeg* segs_dev, int nseg, unsigned total_blocks);

// get_batch scalars of the sampled transitions (a, r, done, IS weight: ...replay.jl:93-102), written by ONE workgroup of the gather launch
// for the fused head kernel (a_out == nullptr: not wanted)
struct BatchMeta { const int* a; const float* r; const unsigned char* done; float beta; int* a_out; float* r_out; float* done_out; float* w_out;
                   int distinct; /* hp.sample_distinct (B <= 64): a gather workgroup that draws the indices itself dedupes the list like sample() does */ };
// The NEXT step's get_batch, run by the first workgroups of this step's Adam launch (inside dqn_train_steps(n): the host knows that a sampled
// step follows and that nothing touches the replay in between).  By then every reader of the arena, of the batch scalars and of the index list
// in this step is done, the tree is final (the priority block ran in an earlier backward launch and drew idx_pre), and the Philox counter of
// the next sample() was set by k_head_td.  The next step then runs WITHOUT its gather launch; its k_head_td copies idx_pre -> idx and checks
// StepState::pre_valid == 2.  f32 observations, or u8 observations on the byte arena.
struct PreGather { int on; const void *s_rows, *sp_rows; int E, B; long long* idx_pre; float* x0; long long cap2; const float* tree; unsigned long long seed;
                   BatchMeta meta; int gx, gy; int u8b; /* u8 rows into the BYTE arena (gather_u8b_body); else f32 rows into the fp32 arena */ };
void launch_gather_fb(hipStream_t st, const void* s_rows, const void* sp_rows, int obs_u8, int E, int B,
                      long long* idx, float* x0 /*[E][2B]*/, int do_sample, long long cap2, const float* tree, unsigned long long seed,
                      const StepState* state, const BatchMeta& meta, const long long* idx_pre /* or null */, int arena_u8 = 0 /* x0 is unsigned char[E][2B] */);
void launch_gather_rows(hipStream_t st, const void* rows, int obs_u8, int E, int n, const long long* idx, float* out /*[n][E]*/);
void launch_transpose_obs(hipStream_t st, const float* obs /*[n][E]*/, int E, int n, float* x /*[E][n]*/);
void launch_replay_commit(hipStream_t st, int n, long long start, long long cap, long long cap2, const int* a_in, const float* r_in,
                          const unsigned char* done_in, const float* td_in, float eps, float alpha, int* a, float* r,
                          unsigned char* done, float* tree, StepState* state);
void launch_tree_rebuild(hipStream_t st, float* tree, long long cap2);
void launch_sample(hipStream_t st, int B, long long cap2, const float* tree, unsigned long long seed, long long* idx, StepState* state, int bump, int distinct = 0);
void launch_batch_meta(hipStream_t st, int B, long long cap2, const long long* idx, const int* a, const float* r,
                       const unsigned char* done, const float* tree, float beta, const StepState* state,
                       int* a_out, float* r_out, float* done_out, float* w_out);
void launch_update_priorities(hipStream_t st, int n, long long cap2, const long long* idx, const float* td, float eps, float alpha,
                              float* tree, StepState* state, int tick_adam, double beta1, double beta2, const float* gmax_part, int n_gmax,
                              long long* idx_pre = nullptr /* also draw the NEXT sample()'s pre_B indices (prio_block_run's second half) */, unsigned long long seed = 0, int pre_B = 0);

void launch_publish_scalars(hipStream_t st, StepState* state, const float* gmax_part, int n_gmax, unsigned long long* pub_ctr, StepMail* mail /* device view of the mapped host ring */);

void launch_valu_fwd(hipStream_t st, const LayerDev& L, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, float* partials);
void launch_valu_dw(hipStream_t st, const LayerDev& L, const float* X, int ldx, const float* dpre, int B, float* G, float* partials);
void launch_valu_dx(hipStream_t st, const LayerDev& L, const float* P, const float* dpre, int B, float* out, float* partials,
                    const float* addend, const float* ysrc, int ldy, int act_src);
void launch_td(hipStream_t st, const TdArgs& a);
// the whole train step of a network that fits in LDS as ONE single-workgroup launch (tiny_step.hip; BASELINE config 1)
#define TINY_MAX_LAYERS 8
struct TinyArgs {
    int nl, nlev, B, nA, E, ncon, dueling, double_q, obs_u8, distinct;
    int last_base, last_val, last_adv;
    float gamma, beta, prio_eps, prio_alpha;
    long long cap2; unsigned long long seed, P;
    LayerDev L[TINY_MAX_LAYERS];
    int lev_n[TINY_MAX_LAYERS], lev_l[TINY_MAX_LAYERS][2];      // level -> its (one or two sibling) layers
    int on_off[TINY_MAX_LAYERS], tg_off[TINY_MAX_LAYERS], d_off[TINY_MAX_LAYERS];      // LDS float offsets: act_on [N][ncon], act_tg [N][B], dact [N][B]
    int pon_off, ptg_off, g_off, x0_off, misc_off; unsigned lds_bytes;
    const float* p_tg; float *p_on, *m, *v, *grad; StepState* state; float* gmax_part; float* tree;
    long long *idx, *idx_pre; const void *s_rows, *sp_rows; const int* ra; const float* rr; const unsigned char* rdone;
    float* x0; float *w_is, *td, *q_on_s, *q_on_sp, *q_tg_sp, *ytarget; int* best;
    int f64mode; float lr; double b1, b2, adam_eps;
};
int launch_tiny_step(hipStream_t st, const TinyArgs* a_dev, unsigned lds_bytes, int sample, int stop /* timing probe: return after phase n */);      // -1: the LDS attribute was refused

// ---- the recurrent train step (batch_train!(..., ::EpisodeReplayBuffer), src/solver.jl:239-287) as a COLUMN-PARALLEL launch (drqn_cols.hip; BASELINE config 4):
// batch columns never interact before the gradient sum, so workgroup g owns the batch columns [g*cg, (g+1)*cg) for the WHOLE step -- episode gather (prefix quirk),
// both target passes and the online pass of the LSTM, heads, TD / masked Huber, BPTT, and its chunk of every dW / db (the column-group chunks of the summation plan,
// dw_kc = -cg) -- with the parameters of both networks and all of its columns' activations in LDS; the step's second launch (the Adam launch) adds the G = B / cg
// slabs in ascending order, folds the loss and updates.  Networks: Chain(flattenbatch, LSTM(E, H), Dense(H, nA)) with or without the dueling split of the head.
struct DrqnColsArgs {
    int B, T, H, E, nA, dueling, double_q, cg, nset; float gamma;
    unsigned Pint;                                              // floats of the internal parameter vector (a multiple of 4)
    unsigned wi_off, b_off, wh_off, h0_off, c0_off;             // LSTM block [Wi E x 4H][b 4H][Wh H x 4H][junk 4H][h0 H][c0 H][zeros 4H]
    unsigned hw_off[2], hb_off[2]; int hN[2], hact[2], h_S[2], h_kc[2];      // heads: [0] = advantage stream (or the plain Q head), [1] = value stream; forward plan chunks of K = H
    const float *p_on, *p_tg;
    const float *ep_s, *ep_sp; const int* ep_a; const float* ep_r; const unsigned char* ep_done; const int* ep_len;
    // the step's episode draws: slot `slot` (a launch parameter, fixed per graph node) of a mapped pinned HOST buffer written by dqn_train_step_drqn -- no H2D copy
    // launch per step, no device-side sequence number to chase
    const long long* ring_idx; const int* ring_np;              // ring_np: rows the prefix copy delivers = max(0, min(len, T) - start)
    float* slabs;                                               // [B / cg][Pint] per-workgroup gradient chunks
    float *hl, *td;                                             // [T*B] Huber terms (folded by the Adam launch), TD errors
    StepState* st;
    int probe;                                                  // TIMING PROBES (DQN_DRQN_PROBE at creation; wrong numbers, right schedule): bit 0 = single-precision hardware exp instead of the Float64 sigm / tanh
    unsigned long long* stamps;                                 // timing probe (DQN_DRQN_STAMPS at creation, tools/drqn_phases.py): 100 MHz s_memrealtime at the phase boundaries of workgroup 0, else null
};
// the column-group size of the fused recurrent step for this network, 0 = not covered (the twin restates this rule: oracle/dqn_ref.c fused_cg)
static inline int drqn_fused_cg(const LayerDev* L, int nl, int E, int B, int T, int nA, int dueling, int double_q, int recurrence) {
    if (!recurrence) return 0;
    const int duel = dueling ? 1 : 0;
    if (nl != (duel ? 3 : 2) || L[0].kind != DQN_LAYER_LSTM || L[0].stream != DQN_STREAM_BASE || L[0].src >= 0) return 0;
    const int H = L[0].H, N = 4 * H;
    if (L[0].K != E || H < 8 || H > 64 || H % 8 || T < 1 || T > 64 || nA > 16 || E > 512) return 0;
    if (!duel) { if (L[1].kind != DQN_LAYER_DENSE || L[1].stream != DQN_STREAM_BASE || L[1].K != H || L[1].N != nA) return 0; }
    else if (L[1].kind != DQN_LAYER_DENSE || L[1].stream != DQN_STREAM_VAL || L[1].K != H || L[1].N != 1 ||
             L[2].kind != DQN_LAYER_DENSE || L[2].stream != DQN_STREAM_ADV || L[2].K != H || L[2].N != nA) return 0;
    const int nset = double_q ? 3 : 2, Ep = (E + 3) / 4 * 4, no = nA + duel;
    const long pp = (long)(E + 1) * N + (long)(H + 1) * N + 2 * H + N + (long)(H + 1) * no + 32;
    for (int c = 4; c >= 1; c >>= 1) {
        if (B % c || nset * 4 * H * c > 1024) continue;
        const long lds = 2 * pp + 2L * T * c * Ep + 4L * T * c + (long)nset * T * H * c + (long)nset * 7 * H * c + (long)T * N * c + 3L * T * H * c +
                         (long)(nset + 1) * T * c * (no + 1) + (long)T * H * c + 2L * H * c + (long)H * (N + 4) + (long)nset * ((T * c + 15) / 16 * 16) * N + 64;
        if (lds <= 36000) return c;
    }
    return 0;
}
int launch_drqn_cols(hipStream_t st, const DrqnColsArgs& a, int slot);      // 0 = launched; -1 = the LDS attribute could not be raised
int adam_blocks(size_t P);
static inline int gmax_slots(size_t) { return 65536; }   // per-block max |g| of every Adam job of a step (each job owns a slot range)
void launch_adam(hipStream_t st, const AdamJob& job, const PreGather* pg = nullptr /* see PreGather */);
void launch_q_columns(hipStream_t st, int n, int nA, int dueling, const float* val, const float* adv, float* q_out /*[n][nA]*/, int* argmax_out);
void launch_convert_params(hipStream_t st, const LayerDev* layers_dev, int nl, const float* src, float* dst, int to_internal, size_t P_ext);

// LDS-tiled MFMA path (nn_gemm.hip): up to two problems (online / target net) of one layer per launch
bool gemm_fwd_eligible(const LayerDev& L, int nprob, const int* ldx, const int* col0, const int* ncols);
int gemm_set_ktrace(unsigned long long* p);   // debug (-DDQN_KTRACE builds): per-workgroup timestamps of the LDS-tiled kernels (nullptr = off); -1 = not a trace build
// LayerDev::opt bits
#define DQN_LOPT_FWD_M32 1      /* DQN_FWD_M32=1: 32x32x2 MFMA blocks for the 64-channel forward tiles in EVERY launch (default: only the large ones, >= 1024 workgroups, where they measure
                                   conv3 forward 58.3 -> 55.7 us at config 5 since the r04 instruction diet; no faster before it) */
#define DQN_LOPT_NO_FWD_M32 32  /* DQN_FWD_M32=0: never */
#define DQN_LOPT_NO_DX_WIDE 4   /* DQN_NO_DX_WIDE: large batches take the 32-sample dX tiles instead of the 128-sample ones */
#define DQN_LOPT_ST_WT 64       /* small-batch engines (<= 64 columns per sequence set; DQN_NO_ST_WT=1 turns it off): the GEMM launches store their outputs write-through, nn_gemm.hip st_out4 */
#define DQN_LOPT_NO_FWD_WRES 8  /* DQN_NO_FWD_WRES: large-batch forwards of a narrow layer take the per-tile kernel instead of the weights-resident persistent one (A/B) */
void launch_gemm_fwd(hipStream_t st, const LayerDev& L, int nprob, const float* const* W, const float* const* bias, const float* const* X,
                     const int* ldx, const int* col0, const int* ncols, float* const* out /* Y, or split-K partial slabs */,
                     float* const* outT = nullptr /* optional, dense unsplit layers: a TRANSPOSED copy [column][feature] of Y per problem (k_head_td's input columns) */,
                     int piece_major = 0 /* dense split-K layers whose slabs only k_red_head reads: slabs laid out [S][column quad][N][4] (GFwdProb::pm, nn_gemm.hip) */);

// ---- DRQN (drqn.hip): EpisodeReplayBuffer gather, LSTM recurrence / BPTT steps, recurrent TD
struct LstmSeq {          // one sequence set advancing one time step: B columns starting at column c0 (+ t*B) of [*][ld] arrays
    const float* Gx; float* Hout; float* Cst; int ld, c0;
    const float *Wh, *bias;
    const float* hprev; int hp_ld, hp_bs;     // h_{t-1}(j,b) = hprev[j*hp_ld + b*hp_bs]   (h0 broadcast: ld 1, bs 0)
    const float* cprev; int cp_ld, cp_bs;
    float *gates, *tc, *hprev_out, *cprev_out; int keep_ld, keep_c0;   // BPTT stash (online s-sequence) or null
};
struct LstmStepArgs { LstmSeq s[3]; int nseq, H, B; };
void launch_lstm_step_t(hipStream_t st, const LstmStepArgs& a, int t);
struct LstmBwdArgs {
    int t, T, H, B, TB; const float *gates, *tc, *cprev, *Wh; const float* dH; float* dG; float *dhn, *dcn; float *g_h0, *g_c0;
};
void launch_lstm_bwd_step(hipStream_t st, const LstmBwdArgs& a);
// whole-sequence variants (small LSTMs: Wh and one step's state in LDS)
struct LstmSeqF {
    const float* Gx; float* Hout; float* Cst; int ld, c0;
    const float *Wh, *bias, *h0, *c0v;
    float *gates, *tc, *hprev_out, *cprev_out; int keep_ld, keep_c0;
};
struct LstmSeqArgs { LstmSeqF s[3]; int nseq, H, B, T; };
bool lstm_seq_fits(int H, int B, int T);
void launch_lstm_seq(hipStream_t st, const LstmSeqArgs& a);
void launch_lstm_bwd_seq(hipStream_t st, const LstmBwdArgs& a);   // a.t ignored
struct EpGatherArgs {
    const float *ep_s, *ep_sp; const int* ep_a; const float* ep_r; const unsigned char* ep_done; const int* ep_len;
    const long long* ep_idx; const int* ep_start; int E, B, T; float* x0; int* a_out; float *r_out, *done_out, *mask_out;
};
void launch_gather_episodes(hipStream_t st, const EpGatherArgs& a);
struct TdDrqnArgs {
    int B, T, nA, ncon, dueling, double_q; float gamma;
    HeadSrc on_val, on_adv, tg_val, tg_adv; float *d_val, *d_adv;
    const int* a; const float *r, *done, *mask; float* td; StepState* st;
};
void launch_td_drqn(hipStream_t st, const TdDrqnArgs& a);
void launch_bcast_state(hipStream_t st, const float* src, int H, int n, float* dst);

#ifdef __HIPCC__
// head output (n, col): either the finished activation, or -- when the head's forward ran split-K and its reduction is
// folded into this kernel -- act(sum_s partial + bias)
// ascending sum of S split-K slabs p[0], p[stride], ..., p[(S-1)*stride]: loads go out in rounds of up to 32 (then 8) INDEPENDENT requests,
// the last round predicated instead of a one-load-at-a-time tail; the additions stay strictly in slab order (the canonical chunk order)
__device__ __forceinline__ float slab_sum(const float* __restrict__ p, size_t stride, int S) {
    float tot = p[0];
    int s = 1;
    for (; s + 32 <= S; s += 32) {
        float v[32];
#pragma unroll
        for (int u = 0; u < 32; u++) v[u] = p[(size_t)(s + u) * stride];
#pragma unroll
        for (int u = 0; u < 32; u++) tot = tot + v[u];
    }
    for (; s < S; s += 16) {      // (r04: rounds of 16 -- the 16 column-group slabs of the fused recurrent step are ONE round trip instead of three)
        const int m = S - s < 16 ? S - s : 16;
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = u < m ? p[(size_t)(s + u) * stride] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; u++) if (u < m) tot = tot + v[u];
    }
    return tot;
}
__device__ __forceinline__ float head_val(const HeadSrc& h, int n, int col) {
    const size_t e = (size_t)n * h.ld + col;
    if (h.S <= 1) return h.p[e];
    const float tot = slab_sum(h.p + e, (size_t)h.per_s, h.S);
    return act_f(tot + h.bias[n], h.act);
}
__device__ __forceinline__ void q_column_h(int nA, int dueling, const HeadSrc& val, const HeadSrc& adv, int col, float* q, float* vout, float* araw) {
    for (int a = 0; a < nA; a++) araw[a] = head_val(adv, a, col);
    if (!dueling) { for (int a = 0; a < nA; a++) q[a] = araw[a]; *vout = 0.0f; return; }
    const float v = head_val(val, 0, col); *vout = v;
    float sum = araw[0];
    for (int a = 1; a < nA; a++) sum = sum + araw[a];
    const float mean = sum / (float)nA;
    for (int a = 0; a < nA; a++) q[a] = (v + araw[a]) - mean;
}
#endif

// ---- vectorised environments on the device (envs.hip)
struct EnvDev {
    int kind, n, E, H, W, nA, max_episode_length, prioritized, eval_mode; unsigned long long seed;
    // TestMDP
    const unsigned char* images; signed char *tm_s, *tm_prev; int* tm_t; int max_time;
    // SimpleGridWorld
    int *gw_pos, *gw_prev; int size_x, size_y, n_reward; float tprob; int reward_xy[8][2]; float reward_val[8];
    // per-env loop state / outputs
    int* actions; float* rewards; unsigned char* dones; unsigned char* pending; float* ep_reward; int* ep_step; long long* fin_eps; double* fin_reward;
};
struct RolloutDev { long long t, widx; float eps_start, eps_stop, eps_steps; int pad; };     // t, widx: values of the LAST completed vector step
struct ReplayMeta { long long cap, cap2; int* a; float* r; unsigned char* done; float* tree; StepState* state; float eps, alpha; };
void launch_env_observe(hipStream_t st, const EnvDev& V, void* rows, int rows_u8, float* x);
struct ActHeads { HeadSrc val, adv; int dueling; float* q_out; int* amax; };     // last-layer outputs of the acting forward (adv doubles as the plain Q head)
void launch_env_step(hipStream_t st, const EnvDev& V, RolloutDev* rs, const ActHeads& Hd, const ReplayMeta& R);
// grouped: rs is an array with one record per group of four copies (ticked by k_act_head's last arrivers); tree != nullptr: workgroup 0 rebuilds the sum-tree ancestors of the n new leaves
void launch_env_observe2(hipStream_t st, const EnvDev& V, const RolloutDev* rs, int rows_u8, void* s_rows, void* sp_rows, long long cap, float* x, int grouped = 0, const ReplayMeta* tree = nullptr);
// ---- the acting step's tail in one launch (act_head.hip): split-K reduce of the last hidden layer + heads + Q / argmax + eps-greedy + act! + add_exp!'s per-experience part
#define DQN_ROLL_RECORDS 257           // RolloutDev records per env set: [g] = group g of four copies (n <= 1024); record 0 doubles as "the" record of the four-launch tail
struct ActHeadStream { const float* part; const float* pbias; int pact; const float* W; const float* hbias; int N, hact; };      // hidden layer slabs [S][K][n] / bias / activation; head weights [K][N], bias, activation
struct ActHeadArgs {
    int n, nA, K, S, nstream, NO, pm;  // n copies, K hidden rows (= head inputs), S slabs, NO head outputs of both streams; pm: piece-major slabs [S][n/4][K][4]
    ActHeadStream st[2];               // 0 = advantage / plain Q head, 1 = value head
    float* partials;                   // [n/4][4][NO][K/32] chunk sums of the head outputs
    unsigned* tickets;                 // [n/4] arrival counters (zero between launches)
    float* q_out; int* amax;           // Q columns [n][nA], greedy actions (parity / inspection)
    RolloutDev* rs;                    // [n/4] records
    EnvDev V; ReplayMeta R;
};
bool act_head_ok(int n, int K, int S, int nA, int nstream, int N0, int N1);
void launch_act_head(hipStream_t st, const ActHeadArgs& a);
void launch_env_reset_pending(hipStream_t st, const EnvDev& V, const RolloutDev* rs, int force_all);

// ---- data-parallel exchange (dp.hip)
#define DP_MAX_REGIONS 48
struct DpRegion { const float* src; unsigned long long dst, n; int B, ld; int S; unsigned long long per_s; };   // S > 1: the value is the ascending sum of S split-K slabs
struct DpPackArgs { int n; DpRegion r[DP_MAX_REGIONS]; float* send; };
struct DpRange { unsigned long long src, dst, n; };
struct DpSumArgs { int n; DpRange r[DP_MAX_REGIONS]; const float* recv; unsigned long long stride; int world; float* grad; };
void launch_dp_pack(hipStream_t st, const DpPackArgs& a);
void launch_dp_unpack_sum(hipStream_t st, const DpSumArgs& a);

// MFMA path (nn_mfma.hip): returns false if the shape is not eligible (caller falls back to the VALU kernel,
// which computes bit-identical values)
bool gemm_dw_eligible(const LayerDev& L, int B, int ldx);
// a TAIL riding in the last workgroups of an LDS-tiled launch: small independent VALU tasks (valu_tasks.h), then (has_adam) an Adam job
struct GemmTail { const VTask* tasks; int n; unsigned blocks; int has_adam; AdamJob adam; unsigned lds_bytes; /* dynamic LDS of the launch (set by the launcher): the priority block's tree cache */
                  int probe; /* trace builds: timing-probe bits (env DQN_PROBE), see nn_gemm.hip */ };
static inline __host__ __device__ unsigned gemm_tail_blocks(const GemmTail& t) { return t.blocks + (t.has_adam ? adam_job_blocks(t.adam) : 0u); }
static inline GemmTail gemm_no_tail() { GemmTail t; memset(&t, 0, sizeof t); return t; }

void launch_gemm_dw(hipStream_t st, const LayerDev& L, int nprob, const float* const* X, int ldx, const float* const* dpre, int B, float* const* out,
                    int ldd = 0, int tpr = 0, int rstride = 0, GemmTail tail = gemm_no_tail());   // ldd 0 = plain layout; else gathered rank blocks (see DwStride in nn_gemm.hip)

bool gemm_dx_eligible(const LayerDev& L, int B, int ldy, int nsrc = 1);
bool gemm_dx_internal_chunks(const LayerDev& L, int B, int ldy, int nsrc = 1);   // plan chunks combined inside the launch: no partial slabs
void launch_gemm_dx(hipStream_t st, const LayerDev& L, int nsrc, const float* const* W, const float* const* dpre, int B, float* out /* dact or partial slabs */,
                    const float* ysrc, int ldy, int act_src, GemmTail tail = gemm_no_tail());

void launch_gemm_dwdx(hipStream_t st, const LayerDev& Lw, int nprob, const float* const* X, int ldx, const float* const* dpre_w, int B, float* const* out_w,
                      const LayerDev& Lx, int nsrc, const float* const* W, const float* const* dpre_x, float* out_x, const float* ysrc, int ldy, int act_src,
                      GemmTail tail = gemm_no_tail());

// (reduce == false leaves split-K partial slabs in `partials` for the caller's batched k_reduce_multi)
bool mfma_fwd_ok(const LayerDev& L, int ncols);
bool mfma_dw_ok(const LayerDev& L, int B);
bool mfma_dx_ok(const LayerDev& L, int B, int ldy);
bool launch_mfma_fwd(hipStream_t st, const LayerDev& L, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, float* partials, bool reduce = true);
bool launch_mfma_dw(hipStream_t st, const LayerDev& L, const float* X, int ldx, const float* dpre, int B, float* G, float* partials, bool reduce = true);
bool launch_mfma_dx(hipStream_t st, const LayerDev& L, const float* P, const float* dpre, int B, float* out, float* partials,
                    const float* addend, const float* ysrc, int ldy, int act_src, bool reduce = true);
