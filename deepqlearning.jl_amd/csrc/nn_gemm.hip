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
#include <type_traits>

#include "common.h"
#include "valu_tasks.h"
#include "adam_body.h"
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ---- debug aid (dqn_debug_ktrace, builds with -DDQN_KTRACE only: `DQN_EXTRA_DEF=DQN_KTRACE python __graft_entry__.py --force`): per-workgroup
// timestamps of the LDS-tiled kernels.  g_ktrace[0] = record counter, then 8-word records {gridDim.x, blockIdx.x, s_memtime at: entry, tables
// ready | role, first tile in LDS, K loop done, chunks combined, stores issued}.  The product build carries none of it: reading the trace
// pointer is a global load + wait at the top of every workgroup.
#ifdef DQN_KTRACE
__device__ unsigned long long* g_ktrace = nullptr;
// timing probes (WRONG numbers, right schedule; env DQN_PROBE): bit 0 = dW workgroups of the fused backward launches return at once, bit 1 = the
// priority block returns at once, bit 2 = dW workgroups skip their stores, bit 3 = dX workgroups return at once.  A kernel argument
// (GemmTail::probe): a __device__ variable set before the module's first launch is reset by the lazy module load.
static int host_probe() { static const int v = getenv("DQN_PROBE") ? atoi(getenv("DQN_PROBE")) : 0; return v; }
#define HOST_PROBE() host_probe()
int gemm_set_ktrace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ktrace), &p, sizeof p); return 0; }
// stamps are taken into REGISTERS (s_memtime is a scalar instruction: no memory traffic) and the record -- slot from an atomic counter, eight
// stores -- is written when the workgroup is done, so tracing does not perturb what it times (the first version fetched the trace pointer, ran the
// atomic and stored as it went: 1-3 us of round trips inside the timed prologue).  The probe word is read BEFORE the entry stamp.
// Word 0 / 1 carry, in their upper halves, the low 32 bits of s_memrealtime (the 100 MHz constant clock, common to all XCDs -- s_memtime is not) at
// entry / exit: the only stamps that can be compared ACROSS workgroups.
#define KTRACE_BEGIN() unsigned long long kts_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; kts_[0] = __builtin_amdgcn_s_memrealtime() << 32; kts_[2] = __builtin_amdgcn_s_memtime();
#define KTRACE(i) do { kts_[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define KTRACE_SET(i, v) do { kts_[i] = (v); } while (0)
#define KTRACE_REC() kts_
#define KTRACE_PROBE() tail.probe
#define KTRACE_END() do { unsigned long long* tr_ = g_ktrace; if (tr_ && threadIdx.x == 0) { const unsigned long long rt_ = __builtin_amdgcn_s_memrealtime(); const unsigned long long o_ = 1 + 8 * atomicAdd(tr_, 1ull); if (o_ + 8 < 1 + 8 * 65536ull) { \
    kts_[0] |= gridDim.x; kts_[1] = (rt_ << 32) | blockIdx.x; for (int i_ = 0; i_ < 8; i_++) tr_[o_ + i_] = kts_[i_]; } } } while (0)
#else
int gemm_set_ktrace(unsigned long long*) { return -1; }     // not a trace build
#define HOST_PROBE() 0
#define KTRACE_BEGIN()
#define KTRACE_END() do {} while (0)
#define KTRACE(i) do {} while (0)
#define KTRACE_SET(i, v) do {} while (0)
#define KTRACE_REC() nullptr
#define KTRACE_PROBE() 0
#endif

// Output tiles of the GEMM launches (activations, split-K slabs, dX tiles, dW slabs / gradients) are stored WRITE-THROUGH (sc1) by the small-batch engines
// (LayerDev::opt & DQN_LOPT_ST_WT, set at dqn_engine_create for <= 64 columns per sequence set).  A kernel boundary writes back what its launch left dirty in the eight L2s
// (MI355X guide, "boundary" row: + B / 6 TB/s behind B dirty bytes) and at B = 32 the step is ten short launches that each leave 1-13 MB behind; write-through stores
// stream out while the launch still runs.  r05, same box, alternating (profiles/history/r05_l_store_ab.txt): config 2 7960 -> 8115 steps/s with every output write-through (dW only:
// 8050; activations / slabs / dX only: 7970); config 5 (B = 512: launches of 45-105 us) 1642 -> 1632, so large batches keep plain / non-temporal stores.
// INVARIANT (inline asm: these stores are invisible to hipcc's vmcnt bookkeeping): NOTHING in the same kernel may read data stored through st_out4 / st_grad4 / adam_st4.
// A tail or last-arriver consumer added later needs an explicit `s_waitcnt vmcnt(0)` in the producer BEFORE its release (fence + ticket), as red_head.hip does --
// __threadfence() alone is not enough, the compiler may drop its vmcnt wait when it sees no pending stores.  (A wave's stores complete before it ends: a kernel boundary is safe.)
__device__ __forceinline__ void st_out4(f32x4* p, const f32x4& v, int wt) {
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ void st_grad4(f32x4* p, const f32x4& v, int wt) {      // dW slabs / gradients: read once, by the Adam launch
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
    else __builtin_nontemporal_store(v, p);
}
void launch_reduce_pub(hipStream_t st, const float* part, int S, size_t elems, int mode, const float* bias, int per_n, int act,
                       const float* addend, const float* ysrc, int B, int ldy, float* out);

// Kernel arguments are fetched with scalar loads where they are first USED; with kilobyte-sized by-value records (LayerDev, GemmTail, ...) and
// branchy prologues that is a chain of 6-10 DEPENDENT round trips to a cold scalar cache before the first operand load goes out (ISA of
// k_dwdx_lds / k_fwd_lds; ktrace r03: 2.7-3.8 us from workgroup entry to the first global_load of a dX workgroup, the loads themselves back in
// 0.15 us).  karg_warm touches one dword of every 64-byte line of the kernarg segment at the top of the kernel: independent loads, issued back to
// back, retired behind ONE wait -- afterwards every argument load hits the scalar cache.
template <int NBYTES> __device__ __forceinline__ void karg_warm() {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const uint32_t __attribute__((address_space(4))) karg_u32;
    karg_u32* p = (karg_u32*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < (NBYTES + 63) / 64; i++) x ^= p[16 * i];
    asm volatile("" ::"s"(x));
#endif
}
// a wave-uniform pointer the compiler cannot prove uniform (derived from threadIdx >> 6), moved to scalar registers for the scalar-base form of global loads
__device__ __forceinline__ const float* uniform_ptr(const float* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float*)(((unsigned long long)hi << 32) | lo);
}
// tail workgroups of a backward launch: small VALU tasks (blk counts from the first tail workgroup).  The record's AdamJob carries only the priority block these days (workgroup 0
// of the launch, GEMM_TAIL_PROLOGUE); the Adam STREAM as tail workgroups (DQN_ADAM_MODE=1, r02-r05) was removed in r06 after losing four rounds running
__device__ __forceinline__ void gemm_tail_run(const GemmTail& tail, unsigned blk) {
    if (blk < tail.blocks) valu_task_run<false>(tail.tasks, tail.n, blk);
}
// the priority block of a carried Adam job is workgroup 0 of the launch: its ~14 dependent tree levels need the whole launch to hide under
#define GEMM_TAIL_PROLOGUE(tail, bid_var, main_var)                                                                                     \
    const int pre_ = (tail.has_adam && tail.adam.prio.n > 0) ? 1 : 0;                                                                   \
    if (pre_ && blockIdx.x == 0) { extern __shared__ float lds[]; prio_block_run(tail.adam.prio, tail.adam.state, reinterpret_cast<long long*>(lds), tail.lds_bytes); return; } \
    const int bid_var = (int)blockIdx.x - pre_;                                                                                         \
    const int main_var = (int)gridDim.x - (int)gemm_tail_blocks(tail);                                                                  \
    if (bid_var >= main_var) { gemm_tail_run(tail, (unsigned)(bid_var - main_var)); return; }
// =====================================================================================================================
// forward:  Y[n][pos][col] = act( sum_k X[xb(pos)+koff(k)][col] * W[k][n] + bias[n] )
// workgroup tile = 4 M-tiles (16 columns each, drawn from consecutive (pos, column-tile) pairs; wave w owns M-tile w)
//                  x NT N-tiles (16*NT output channels), K walked in tiles of 32.
// =====================================================================================================================
struct GFwdProb { const float* W; const float* bias; const float* X; int ldx, col0, ncols; float* out; int mtiles, mgroups; float* outT;      // outT: optional transposed copy [column][N] (dense, one chunk)
                  int pm; };   // pm (dense, split-K slabs read by k_red_head only): PIECE-MAJOR slabs [S][column quad][N][4] instead of [S][N][columns] -- the 32 hidden rows x 16 bytes a
                               // (column group, chunk) workgroup of k_red_head reads per slab are then 512 contiguous bytes instead of 32 pieces 256 bytes apart
struct GFwdProbs { GFwdProb p[4]; int wg_end[4]; };   // up to 4 problems of one geometry per launch: {val,adv} x {online,target}

constexpr int F_KT_DEF = 32;    // K tile depth of the small-batch launches; launches of >= 1024 workgroups use 16-deep tiles (see launch_gemm_fwd).  64-deep tiles measured slower (r03) and were removed
constexpr int F_SA = 80;        // A tile row stride (64 columns + 16 pad): ds_read_b32 of lanes (i, kq) hits banks 16*kq + i
template <int NT> struct FwdCfg { static constexpr int NW = 16 * NT; static constexpr int SB = (NW % 32 == 0) ? NW + 16 : NW + 32; };

// XU8: the A operand is the BYTE observation arena (u8 replay): a lane fetches 4 bytes = 4 columns and converts them (byte / 255f0, exactly) on
// the way into the LDS tile -- a quarter of the operand bytes of the float arena, and the gather wrote a quarter as well
typedef float f32x16 __attribute__((ext_vector_type(16)));
// M32 (NT == 4 only; experiment, DQN_FWD_M32=1): the 64-column x 64-channel tile as 2 x 2 blocks of v_mfma_f32_32x32x2_f32, one per wave, instead of four
// 16x16x4 accumulators per wave -- 32 instead of 40 fragment reads and 16 instead of 32 MFMA instructions per wave and K tile
// one unit of a forward launch: problem p, channels n0 .. n0 + 16 NT - 1, M-group mgrp (64 columns), chunk s
template <int NT, bool XU8, int KT, bool M32>
__device__ __forceinline__ void fwd_lds_body(const LayerDev& L, const GFwdProb& p, int S, int kc, int n0, int mgrp, int s) {
    constexpr int NW = FwdCfg<NT>::NW, SB = FwdCfg<NT>::SB, F_KT = KT;
    using AT = std::conditional_t<XU8, uint32_t, f32x4>;
    extern __shared__ float lds[];
    float* As = lds;                                   // [2][F_KT][F_SA]
    float* Bs = lds + 2 * F_KT * F_SA;                 // [2][F_KT][SB]
    int* koff_lds = (int*)(Bs + 2 * F_KT * SB);        // [K] (conv only)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    KTRACE_BEGIN()
    const bool conv = L.kind == DQN_LAYER_CONV;
    constexpr unsigned ESZ = XU8 ? 1u : 4u;            // bytes per arena element
    const unsigned ldb = (unsigned)p.ldx * ESZ;        // bytes per input row
    const int ctiles = p.ncols >> 4;
    const int k0 = s * kc, k1 = min(L.K, k0 + kc), nkt = (k1 - k0) / F_KT;

    // ---- this thread's slice of the A tile: column group (t & 15) -> M-tile j = (t&15)>>2, float4 (t&3); rows t>>4, +16
    const int aj = (tid & 15) >> 2;
    const int amt = mgrp * 4 + aj;
    const bool a_ok = amt < p.mtiles;
    int a_xb = 0, a_ct = 0;
    if (a_ok) {
        int pos; fdiv_qr(amt, fdiv_of(ctiles), pos, a_ct);
        if (conv) { int oy, ox; fdiv_qr(pos, fdiv_of(L.ow), oy, ox); a_xb = oy * L.sh * L.iw + ox * L.sw; }
    }
    const int arow = tid >> 4;                         // 0..15 (second float4: +16)
    // r04: operand address = SCALAR base (+ the K tile's offset where it is uniform: dense rows, weight rows) + a 32-bit per-thread byte offset fixed for the whole
    // workgroup (+ the row offset from the table for conv): at most one VALU add per load -- every VALU instruction is paid in fp32 MFMA time on gfx950
    // (tools/micro/mfma_mix.cpp).  One layer's operands are < 4 GB.
    const unsigned char* Xbase = reinterpret_cast<const unsigned char*>(p.X) + (size_t)p.col0 * ESZ;
    const unsigned a_fix = (unsigned)a_xb * ldb + (unsigned)(a_ct * 16 + (tid & 3) * 4) * ESZ;
    // ---- B tile slice: NW/4 float4 per row
    constexpr int BF4 = NW / 4;                        // float4 per B row
    constexpr int BQ = (F_KT * BF4 + 255) / 256;       // float4 per thread (1 or 2)
    const float* Wp = p.W + n0;

    // Register staging with HAND-COUNTED waits.  hipcc's waitcnt pass drains vmcnt(0) before every prefetch issue in a
    // pipelined loop with conditional loads (seen in the ISA), which exposes the full L2/HBM latency once per K tile.
    // So the staging loads are inline asm (invisible to that pass), ALWAYS issued (tile index clamped, so the number of
    // loads in flight is a compile-time constant) and retired with explicit s_waitcnt vmcnt(N) naming their registers
    // (cdna_hip_programming.md section 5.7 form (ii)).
    constexpr int AQ = F_KT / 16;                      // A float4 per thread
    constexpr int LPS = AQ + BQ;                       // loads per stage
    struct Stage { AT a[AQ]; f32x4 b[BQ]; };
    bool bok[BQ]; int brow[BQ], bc4[BQ]; unsigned b_fix[BQ];
#pragma unroll
    for (int i = 0; i < BQ; i++) {                                   // clamped B slots (threads beyond the tile re-load the last one)
        const int q = tid + 256 * i; bok[i] = q < F_KT * BF4; const int qc = bok[i] ? q : F_KT * BF4 - 1;
        brow[i] = qc / BF4; bc4[i] = qc % BF4; b_fix[i] = 4u * ((unsigned)brow[i] * (unsigned)L.N + 4u * bc4[i]);
    }
    unsigned a_den[AQ];                                                // dense: the thread's rows of a K tile
#pragma unroll
    for (int q = 0; q < AQ; q++) a_den[q] = a_fix + (unsigned)(arow + 16 * q) * ldb;
    auto gld = [](unsigned off, const void* base) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
    auto gld1 = [](unsigned off, const void* base) { uint32_t v; asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
    // conv: the row offsets of a tile are read from the table one call AHEAD of the loads that use them (gload is called for tiles 0, 1, 2, ... in order, clamped
    // at the last): no LDS round trip -- and no drain of the fragment reads in flight -- in front of a load issue
    // (kn holds the RAW table value until the call that uses it: the empty asm pins the wait for the table read there, a whole tile after its issue -- with the
    // add written next to the read the compiler waits for the LDS round trip on the spot, once per tile of a lone wave)
    // The FIRST tile's row offsets are computed by each thread for itself (two values), its loads go out, and only then is the table built: the table's ~100 instructions
    // and its barrier ride under the first tile's global latency instead of standing in front of it (ktrace r04, B = 32: 1400-1640 cycles from entry to the table, 1700-1950
    // more to the first tile in LDS, of workgroup lifetimes of 16-18 k cycles).
    unsigned kn[AQ];
    {
        const FDiv fkhw = fdiv_of(conv ? L.kh * L.kw : 1), fkw = fdiv_of(conv ? L.kw : 1);
#pragma unroll
        for (int q = 0; q < AQ; q++) {
            if (conv) { const int k = k0 + arow + 16 * q; int ci, rem, ky, kx; fdiv_qr(k, fkhw, ci, rem); fdiv_qr(rem, fkw, ky, kx); kn[q] = (unsigned)((ci * L.ih + ky) * L.iw + kx) * ldb; }
            else kn[q] = a_den[q] - a_fix;
        }
    }
    auto kn_next = [&](int kt) {                          // (kt already clamped) the NEXT tile's row offsets, from the table
        if (conv) {
            const int kbn = k0 + min(kt + 1, nkt - 1) * F_KT;
#pragma unroll
            for (int q = 0; q < AQ; q++) kn[q] = (unsigned)koff_lds[kbn + arow + 16 * q];
        }
    };
    auto gissue = [&](int kt, Stage& r) {
        const int kb = k0 + kt * F_KT;
        const unsigned char* pa = conv ? Xbase : Xbase + (size_t)((unsigned)kb * ldb);
#pragma unroll
        for (int q = 0; q < AQ; q++) { asm volatile("" : "+v"(kn[q])); const unsigned off = a_fix + kn[q]; if constexpr (XU8) r.a[q] = gld1(off, pa); else r.a[q] = gld(off, pa); }
        const float* pb = Wp + (size_t)((unsigned)kb * (unsigned)L.N);
#pragma unroll
        for (int i = 0; i < BQ; i++) r.b[i] = gld(b_fix[i], pb);
    };
    auto gload = [&](int kt, Stage& r) { kt = min(kt, nkt - 1); gissue(kt, r); kn_next(kt); };
    auto gload_first = [&](Stage& r) {
        gissue(0, r);
        if (conv) {
        // BYTE offset of contraction index k = (ci, ky, kx)'s input row, tabulated once per workgroup.  The two divisions go through reciprocals
        // (floor((x + 0.5) / d) is exact for these small ints): with integer divisions this table cost 0.85-2.0 us at the head of every conv
        // launch (ktrace, r02) before the first operand load could be issued.
        const float r_khw = __builtin_amdgcn_rcpf((float)(L.kh * L.kw)), r_kw = __builtin_amdgcn_rcpf((float)L.kw); const int khw = L.kh * L.kw;      // (1-ulp reciprocals: exact for these ranges, fdiv_of in common.h)
        for (int k = tid; k < L.K; k += 256) { const int ci = (int)(((float)k + 0.5f) * r_khw); const int rem = k - ci * khw; const int ky = (int)(((float)rem + 0.5f) * r_kw); koff_lds[k] = (int)((unsigned)((ci * L.ih + ky) * L.iw + (rem - ky * L.kw)) * ldb); }
        }
        if (conv) __syncthreads();                     // koff table ready
        KTRACE(3);
        kn_next(0);
    };
    auto lstore = [&](int buf, const Stage& r) {
#pragma unroll
        for (int q = 0; q < AQ; q++) {
            f32x4 av;
            if constexpr (XU8) { const uint32_t w4 = r.a[q]; av = (f32x4){u8_unit(w4 & 0xffu), u8_unit((w4 >> 8) & 0xffu), u8_unit((w4 >> 16) & 0xffu), u8_unit(w4 >> 24)}; } else av = r.a[q];
            *reinterpret_cast<f32x4*>(As + (buf * F_KT + arow + 16 * q) * F_SA + (tid & 15) * 4) = av;
        }
#pragma unroll
        for (int i = 0; i < BQ; i++) if (bok[i]) *reinterpret_cast<f32x4*>(Bs + (buf * F_KT + brow[i]) * SB + 4 * bc4[i]) = r.b[i];
    };
    auto stage_wait = [&](auto N, Stage& r) {
        constexpr int n = decltype(N)::value;
        if constexpr (AQ == 1 && BQ == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.a[0]), "+v"(r.b[0]) : "n"(n) : "memory");
        else if constexpr (AQ == 2 && BQ == 1) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.b[0]) : "n"(n) : "memory");
        else if constexpr (AQ == 2 && BQ == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.b[0]), "+v"(r.b[1]) : "n"(n) : "memory");
        else if constexpr (AQ == 4 && BQ == 1) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]) : "n"(n) : "memory");
        else if constexpr (AQ == 4 && BQ == 2) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]) : "n"(n) : "memory");
        else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3]) : "n"(n) : "memory");
    };
#define STAGE_WAIT(N, r) stage_wait(std::integral_constant<int, N>{}, r)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the epilogue's bias values are requested now: their latency (a cold line, ~1-2 us) rides under the K loop instead of
    // stalling every workgroup at its end.  (An older outstanding load only makes the counted vmcnt waits conservative.)
    float bias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bias_r[t] = S == 1 ? p.bias[n0 + 16 * t + l15] : 0.0f;
    // Software pipeline of one wave (it may be alone on its SIMD at config 2): the fragments of tile kt+1 are read from LDS (into a second
    // register set) right after the barrier that publishes them, i.e. in the MIDDLE of tile kt's MFMA chain, and the wait for tile kt+1's global
    // loads + its LDS stores sit between the two halves of that chain -- instead of LDS-read latency, global wait, LDS store and barrier each
    // being exposed once per tile with the MFMA pipe idle.  The chain order per accumulator is unchanged (tiles ascending, st = 0..7).
    struct Frag { float a[F_KT / 4]; float b[F_KT / 4][NT]; };
    auto fread = [&](int buf, Frag& f) {
        const float* Ab = As + buf * F_KT * F_SA + 16 * wave + l15;
        const float* Bb = Bs + buf * F_KT * SB + l15;
#pragma unroll
        for (int st = 0; st < F_KT / 4; st++) {
            f.a[st] = Ab[(4 * st + kq) * F_SA];
#pragma unroll
            for (int t = 0; t < NT; t++) f.b[st][t] = Bb[(4 * st + kq) * SB + 16 * t];
        }
    };
    auto mma = [&](const Frag& f, int s0, int s1) {
#pragma unroll
        for (int st = s0; st < s1; st++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = MFMA(f.a[st], f.b[st][t], acc[t]);
    };
    if constexpr (!M32) {
        // r05: a wave whose M-tile lies beyond the problem -- the target net's 32 columns are 2 of a dense workgroup's 4 M-tiles, so waves 2 and 3 of those workgroups
        // contracted padding: 4/3 of the algorithmic MFMAs in the FC forward (SQ_VALU_MFMA_BUSY_CYCLES 12.85 M vs 9.6 M, profiles/history/r04_y_pmc_sq.txt), issued on SIMDs the
        // co-resident workgroup's waves need.  Such a wave still moves its share of both operand tiles -- the loads, LDS stores and barriers of the pipeline below, in the same
        // order -- but reads no fragments and issues no MFMAs.  The productive waves' instruction stream is unchanged (a separate path, not a predicate in the loop).
        if (!__builtin_amdgcn_readfirstlane((int)(mgrp * 4 + wave < p.mtiles))) {
            Stage r0, r1;
            gload_first(r0); STAGE_WAIT(0, r0); lstore(0, r0); __syncthreads();
            gload(1, r0);
            for (int kt = 0; kt < nkt; kt += 2) {
                gload(kt + 2, r1);
                STAGE_WAIT(LPS, r0); lstore(1, r0);
                __syncthreads();
                gload(kt + 3, r0);
                STAGE_WAIT(LPS, r1); lstore(0, r1);
                __syncthreads();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            return;
        }
    }
    if constexpr (M32) {
        static_assert(NT == 4 || !M32, "M32 needs the 64-channel tile");
        const int wm = wave & 1, wn = wave >> 1, l31 = lane & 31, kh = lane >> 5;
        constexpr int NS = F_KT / 2;                   // 32x32x2 steps per K tile
        struct F32 { float a[NS], b[NS]; };
        auto fread32 = [&](int buf, F32& f) {
            const float* Ab = As + (buf * F_KT + kh) * F_SA + 32 * wm + l31;
            const float* Bb = Bs + (buf * F_KT + kh) * SB + 32 * wn + l31;
#pragma unroll
            for (int st = 0; st < NS; st++) { f.a[st] = Ab[2 * st * F_SA]; f.b[st] = Bb[2 * st * SB]; }
        };
        f32x16 c16;
#pragma unroll
        for (int i = 0; i < 16; i++) c16[i] = 0.0f;
        auto mma32 = [&](const F32& f, int s0, int s1) {
#pragma unroll
            for (int st = s0; st < s1; st++) c16 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[st], f.b[st], c16, 0, 0, 0);
        };
        Stage r0, r1; F32 g0, g1;
        gload_first(r0); STAGE_WAIT(0, r0); lstore(0, r0); __syncthreads();
        fread32(0, g0);
        gload(1, r0);
        for (int kt = 0; kt < nkt; kt += 2) {
            gload(kt + 2, r1);
            mma32(g0, 0, NS / 2);
            STAGE_WAIT(LPS, r0); lstore(1, r0);
            __syncthreads();
            if (kt + 1 < nkt) fread32(1, g1);
            mma32(g0, NS / 2, NS);
            gload(kt + 3, r0);
            if (kt + 1 < nkt) mma32(g1, 0, NS / 2);
            STAGE_WAIT(LPS, r1); lstore(0, r1);
            __syncthreads();
            if (kt + 2 < nkt) fread32(0, g0);
            if (kt + 1 < nkt) mma32(g1, NS / 2, NS);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // epilogue: lane = channel n0 + 32 wn + l31; register 4g + r = column 32 wm + 8g + 4 kh + r of the workgroup's 64
        const int n = n0 + 32 * wn + l31;
        const float bias = S == 1 ? p.bias[n] : 0.0f;
        const size_t per_s = (size_t)L.N * L.npos * p.ncols;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int cc = 32 * wm + 8 * g + 4 * kh;       // first of 4 consecutive columns, inside M-tile cc >> 4
            const int mt = mgrp * 4 + (cc >> 4);
            if (mt >= p.mtiles) continue;
            const int pos = mt / ctiles, ct = mt % ctiles;
            f32x4 v = {c16[4 * g], c16[4 * g + 1], c16[4 * g + 2], c16[4 * g + 3]};
            if (S == 1) { act_v4(v, bias, L.act); }
            st_out4(reinterpret_cast<f32x4*>(p.out + (size_t)s * per_s + ((size_t)n * L.npos + pos) * p.ncols + ct * 16 + (cc & 15)), v, L.opt & DQN_LOPT_ST_WT);
        }
        KTRACE(7); KTRACE_END();
        return;
    }
    constexpr int HS = F_KT / 8;                       // MFMA steps per half tile
    Stage r0, r1; Frag f0, f1;
    gload_first(r0); STAGE_WAIT(0, r0); lstore(0, r0); __syncthreads();
    KTRACE(4);
    fread(0, f0);
    gload(1, r0);
    for (int kt = 0; kt < nkt; kt += 2) {
        gload(kt + 2, r1);
        mma(f0, 0, HS);
        STAGE_WAIT(LPS, r0); lstore(1, r0);
        __syncthreads();
        if (kt + 1 < nkt) fread(1, f1);
        mma(f0, HS, 2 * HS);
        gload(kt + 3, r0);
        if (kt + 1 < nkt) mma(f1, 0, HS);
        STAGE_WAIT(LPS, r1); lstore(0, r1);
        __syncthreads();
        if (kt + 2 < nkt) fread(0, f0);
        if (kt + 1 < nkt) mma(f1, HS, 2 * HS);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the clamped tail loads before the registers are reused
    KTRACE(5); KTRACE(6);
#undef STAGE_WAIT
    // ---- epilogue: wave w's M-tile
    const int mt = mgrp * 4 + wave;
    if (mt >= p.mtiles) return;
    int pos, ct; fdiv_qr(mt, fdiv_of(ctiles), pos, ct);
    const size_t per_s = (size_t)L.N * L.npos * p.ncols;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + l15;
        f32x4 v = acc[t];
        if (S == 1) {
            const float bias = bias_r[t];
            if constexpr (KT == 16) act_v4(v, bias, L.act);      // (large launches; for the lone waves of the 32-deep form the per-element switch measures 1.9 % FASTER per step, same box)
            else { v.x = act_f(v.x + bias, L.act); v.y = act_f(v.y + bias, L.act); v.z = act_f(v.z + bias, L.act); v.w = act_f(v.w + bias, L.act); }
        }
        const size_t o_el = p.pm ? ((size_t)(ct * 4 + kq) * L.N + n) * 4 : ((size_t)n * L.npos + pos) * p.ncols + ct * 16 + 4 * kq;
        st_out4(reinterpret_cast<f32x4*>(p.out + (size_t)s * per_s + o_el), v, L.opt & DQN_LOPT_ST_WT);
        if (p.outT) {      // (dense, one chunk) the same activations with a batch column's features contiguous: k_head_td reads its columns as runs instead of one 64-byte sector per element
            float* o = p.outT + (size_t)(ct * 16 + 4 * kq) * L.N + n;
            o[0] = v.x; o[L.N] = v.y; o[2 * (size_t)L.N] = v.z; o[3 * (size_t)L.N] = v.w;
        }
    }
    KTRACE(7); KTRACE_END();
}
// the kernel: workgroup -> (problem, N-group, M-group, chunk), XCD-contiguous within a problem.
// (r06, measured and dropped: the launch's last units as two halves along N, as the dW sections do -- conv2 / conv3 forward at B = 512 76.9 / 55.3 -> 76.6 / 54.8 us with 320 units cut,
//  78.1 / 58.0 with 640: the 16-channel-per-wave half body is slower per FLOP than the 32x32x2 one and these launches' second round is short; profiles/r06_g_split_ab.txt)
template <int NT, bool XU8 = false, int KT = F_KT_DEF, bool M32 = false>
__global__ __launch_bounds__(256, (NT == 4 && KT == 16 && M32) ? 6 : 1)      // (the large 32x32x2 launches: six waves per SIMD, accumulators in plain VGPRs -- r06 same box: conv2 / conv3 forward 78.1 / 56.2 -> 76.9 / 55.3 us)
void k_fwd_lds(LayerDev L, GFwdProbs pr, int S, int kc) {
    karg_warm<sizeof(LayerDev) + sizeof(GFwdProbs) + 8>();
    int pi = 0;
    while (pi < 3 && (int)blockIdx.x >= pr.wg_end[pi]) pi++;
    const GFwdProb& p = pr.p[pi];
    const int wg_begin = pi == 0 ? 0 : pr.wg_end[pi - 1];
    const int w = xcd_remap(blockIdx.x - wg_begin, pr.wg_end[pi] - wg_begin);
    const int ngroups = L.N / (16 * NT);               // (the decodes go through reciprocals: fdiv_*, common.h)
    int ng, mgrp, s;
    // (r06, measured and dropped: a BLOCKED decode for the wide dense layers at large batches -- an XCD's share covering nb N-groups x a run of M-groups instead of every N-group
    //  x two M-groups, so that its L2 streams a fraction of the weights instead of all of them (FETCH_SIZE 240 MB for 45 MB of operands, profiles/r06_q_cfg5_pmc_fetch.txt):
    //  nb = 2 / 4 / 8 leave the 3136 -> 512 pair's forward at 95.2-96.5 us, as it was (profiles/r06_m_fwd_nb.txt) -- fabric traffic is not what bounds this launch)
    { int w2; fdiv_qr(w, fdiv_of(ngroups), w2, ng); fdiv_qr(w2, fdiv_of(p.mgroups), s, mgrp); }
    fwd_lds_body<NT, XU8, KT, M32>(L, p, S, kc, ng * 16 * NT, mgrp, s);
}

// =====================================================================================================================
// forward with the layer's WEIGHTS RESIDENT in LDS (r04; large-batch launches of a narrow layer whose [K][N] block fits beside the A tiles -- the first
// convolution of the Nature net at B = 512: 256 x 32 floats = 32 KB).  Long-lived workgroups (2-3 per CU) fetch the weights ONCE and then walk their share of
// the output in ONE continuous software pipeline.  Each WAVE owns MT adjacent 16-column M-tiles of one output position and stages ITS OWN A rows (16 MT columns
// per input row, through registers into a wave-private LDS tile): the stream has no workgroup barrier, the waves of a SIMD drift apart and fill each other's
// LDS / conversion / wait phases with MFMA work, and a wave carries MT x NT independent accumulator chains through every latency of its own.
// D register stages keep D - 1 tiles of loads in flight; the first tiles of the next macro-tile are requested while the last MFMA steps of the current one issue.
// Same chains as k_fwd_lds (k ascending per accumulator, one chunk): bit-identical outputs.
// =====================================================================================================================
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int NT, int MT, bool XU8>
__global__ __launch_bounds__(256) void k_fwd_wres(LayerDev L, GFwdProbs pr) {
    constexpr int NW = 16 * NT, F_KT = 32, AQ = F_KT / 16, HS = F_KT / 8, SWZ = NW >= 32 ? 16 : 0, D = 4;
    constexpr int ASTR = F_KT * 16 + 4;                // floats per M-tile region [32 k][16 columns]; the + 4 rotates the banks of the regions a lane quad writes in one instruction
    constexpr unsigned ESZ = XU8 ? 1u : 4u;            // bytes per arena element
    constexpr int NLD = XU8 ? 1 : MT;                  // load instructions per row slice: u8 = one load of MT dwords, f32 = MT loads of 16 bytes
    using UT = std::conditional_t<MT == 1, uint32_t, std::conditional_t<MT == 2, u32x2, u32x4>>;
    static_assert(AQ == 2 && (MT == 1 || MT == 2 || MT == 4), "the counted waits name these registers");
    extern __shared__ float lds[];
    float* As = lds;                                   // [4 waves][MT][ASTR]: every wave stages its own columns -- no workgroup barrier in the stream
    float* Bres = lds + 4 * MT * ASTR;                 // [K][NW]; element (k, c) at column c ^ 16 (k & 1): the 16-lane fragment reads of rows k, k + 1 hit disjoint banks without padding
    int* koff_lds = (int*)(Bres + L.K * NW);           // [K]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    karg_warm<sizeof(LayerDev) + sizeof(GFwdProbs)>();
    KTRACE_BEGIN()
    const bool conv = L.kind == DQN_LAYER_CONV;
    int pi = 0;
    while (pi < 3 && (int)blockIdx.x >= pr.wg_end[pi]) pi++;
    const GFwdProb& p = pr.p[pi];
    const unsigned ldb = (unsigned)p.ldx * ESZ;        // bytes per input row
    // BYTE offset of contraction index k's input row, tabulated once per workgroup (dense: k itself, so that the stream below has no layer-kind branch)
    if (conv) {
        const float r_khw = __builtin_amdgcn_rcpf((float)(L.kh * L.kw)), r_kw = __builtin_amdgcn_rcpf((float)L.kw); const int khw = L.kh * L.kw;      // (1-ulp reciprocals: exact for these ranges, fdiv_of in common.h)
        for (int k = tid; k < L.K; k += 256) { const int ci = (int)(((float)k + 0.5f) * r_khw); const int rem = k - ci * khw; const int ky = (int)(((float)rem + 0.5f) * r_kw); koff_lds[k] = (int)((unsigned)((ci * L.ih + ky) * L.iw + (rem - ky * L.kw)) * ldb); }
    } else for (int k = tid; k < L.K; k += 256) koff_lds[k] = (int)((unsigned)k * ldb);
    const int wg_begin = pi == 0 ? 0 : pr.wg_end[pi - 1];
    const int cnt = pr.wg_end[pi] - wg_begin;          // workgroups of this problem; this one walks workgroup tiles li, li + cnt, ...  (at any time an XCD works on one contiguous band)
    const int li = xcd_remap(blockIdx.x - wg_begin, cnt);
    const int ctiles = p.ncols >> 4, nkt = L.K / F_KT;
    const int wgt = (p.mtiles + 4 * MT - 1) / (4 * MT);        // workgroup tiles: 4 waves x MT M-tiles (ctiles % MT == 0: a wave's M-tiles share their output position)
    const int ngm = li < wgt ? (wgt - li + cnt - 1) / cnt : 0;
    {
        constexpr int BF4 = NW / 4;
        for (int q = tid; q < L.K * BF4; q += 256) {
            const int k = q / BF4, c4 = q % BF4;
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.W + (size_t)k * L.N + 4 * c4);
            *reinterpret_cast<f32x4*>(Bres + k * NW + ((4 * c4) ^ (SWZ * (k & 1)))) = v;
        }
    }
    float bias_r[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) bias_r[t] = p.bias[16 * t + l15];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the counted waits below assume only staging loads in flight
    __syncthreads();
    if (ngm == 0) return;
    // ---- this lane's slice of the wave's A tile (32 k x 16 MT columns): rows lane >> 2 and + 16, columns 4 MT (lane & 3) .. + 4 MT - 1
    const int arow = lane >> 2, aq3 = lane & 3;
    float* Aw = As + wave * (MT * ASTR);
    // operand address = scalar base (arena + col0) + 32-bit byte offset {row offset from the table + this lane's column bytes (kn) + the wave's tile offset (scalar)}:
    // one VALU add per load, no 64-bit address arithmetic in the stream (the arena is < 4 GB)
    const unsigned char* Xbase = reinterpret_cast<const unsigned char*>(p.X) + (size_t)p.col0 * ESZ;
    const unsigned lane_b = (unsigned)(aq3 * 4 * MT) * ESZ;
    const FDiv fct = fdiv_of(ctiles), fow = fdiv_of(conv ? L.ow : 1);
    auto tile_of = [&](int gi, unsigned& go) {         // byte offset of this wave's macro-tile of the gi-th workgroup tile: input-row base + column (a missing one re-reads tile 0: never stored)
        const int amt = __builtin_amdgcn_readfirstlane(((li + gi * cnt) * 4 + wave) * MT);
        go = 0;
        if (amt < p.mtiles) {
            int pos, ct; fdiv_qr(amt, fct, pos, ct); go = (unsigned)ct * 16u * ESZ;
            if (conv) { int oy, ox; fdiv_qr(pos, fow, oy, ox); go += (unsigned)(oy * L.sh * L.iw + ox * L.sw) * ldb; }
        }
    };
    struct Stage { UT u[AQ]; f32x4 v[AQ][MT]; };      // (one of the two is live)
    // the input-row offsets of a tile are fetched from the table ONE STEP AHEAD of the loads that use them (kn): no LDS round trip in front of a load issue
    unsigned kn[AQ];
    auto koff_of = [&](int kt) {
#pragma unroll
        for (int q = 0; q < AQ; q++) kn[q] = (unsigned)koff_lds[kt * F_KT + arow + 16 * q];      // RAW: the adds wait for the read where it is USED (next step)
    };
    auto gload = [&](unsigned go, Stage& r) {             // ALWAYS AQ * NLD loads (hand-counted waits, see k_fwd_lds); rows from kn
#pragma unroll
        for (int q = 0; q < AQ; q++) {
            asm volatile("" : "+v"(kn[q]));
            const unsigned off = kn[q] + (go + lane_b);
            if constexpr (XU8) {
                if constexpr (MT == 1) asm volatile("global_load_dword %0, %1, %2" : "=v"(r.u[q]) : "v"(off), "s"(Xbase) : "memory");
                else if constexpr (MT == 2) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r.u[q]) : "v"(off), "s"(Xbase) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.u[q]) : "v"(off), "s"(Xbase) : "memory");
            } else {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r.v[q][0]) : "v"(off), "s"(Xbase) : "memory");
                if constexpr (MT >= 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(r.v[q][1]) : "v"(off), "s"(Xbase) : "memory");
                if constexpr (MT == 4) { asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(r.v[q][2]) : "v"(off), "s"(Xbase) : "memory");
                                         asm volatile("global_load_dwordx4 %0, %1, %2 offset:48" : "=v"(r.v[q][3]) : "v"(off), "s"(Xbase) : "memory"); }
            }
        }
    };
    auto lstore = [&](const Stage& r) {
#pragma unroll
        for (int q = 0; q < AQ; q++)
#pragma unroll
            for (int u = 0; u < MT; u++) {
                f32x4 av;
                if constexpr (XU8) {
                    uint32_t w4;
                    if constexpr (MT == 1) w4 = r.u[q]; else w4 = r.u[q][u];
                    av = (f32x4){u8_unit(w4 & 0xffu), u8_unit((w4 >> 8) & 0xffu), u8_unit((w4 >> 16) & 0xffu), u8_unit(w4 >> 24)};
                } else av = r.v[q][u];
                const int idx = aq3 * MT + u;          // column quad idx of the wave's 4 MT: M-tile idx >> 2, columns 4 (idx & 3) ..
                *reinterpret_cast<f32x4*>(Aw + (idx >> 2) * ASTR + (arow + 16 * q) * 16 + (idx & 3) * 4) = av;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();      // the wave's LDS operations execute in order: its own later fragment reads see these stores
    };
    // everything older than the newest N operations has landed: the stage `r`, and an epilogue's stores issued after it (loads and stores retire in order)
    auto stage_wait = [&](auto N, Stage& r) {
        constexpr int n = decltype(N)::value;
        if constexpr (XU8) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.u[0]), "+v"(r.u[1]) : "n"(n) : "memory");
        else if constexpr (MT == 1) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r.v[0][0]), "+v"(r.v[1][0]) : "n"(n) : "memory");
        else if constexpr (MT == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.v[0][0]), "+v"(r.v[0][1]), "+v"(r.v[1][0]), "+v"(r.v[1][1]) : "n"(n) : "memory");
        else asm volatile("s_waitcnt vmcnt(%8)" : "+v"(r.v[0][0]), "+v"(r.v[0][1]), "+v"(r.v[0][2]), "+v"(r.v[0][3]), "+v"(r.v[1][0]), "+v"(r.v[1][1]), "+v"(r.v[1][2]), "+v"(r.v[1][3]) : "n"(n) : "memory");
    };
#define WRES_WAIT(N, r) stage_wait(std::integral_constant<int, (N)>{}, r)
    constexpr int LPS = AQ * NLD;                      // loads per stage
    struct Frag { float a[F_KT / 4][MT]; float b[F_KT / 4][NT]; };
    const float* Afr = Aw + kq * 16 + l15;             // rows kq = 0, 1 of a 32-lane group: banks 0-15 / 16-31 (+ the region's rotation)
    const float* Bfr[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) Bfr[t] = Bres + kq * NW + ((16 * t) ^ (SWZ * (kq & 1))) + l15;
    auto fread = [&](int kt, Frag& f) {                // per-lane bases + compile-time offsets: no address arithmetic per read
#pragma unroll
        for (int st = 0; st < F_KT / 4; st++) {
#pragma unroll
            for (int m = 0; m < MT; m++) f.a[st][m] = Afr[m * ASTR + 4 * st * 16];
#pragma unroll
            for (int t = 0; t < NT; t++) f.b[st][t] = (Bfr[t] + kt * F_KT * NW)[4 * st * NW];
        }
    };
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int t = 0; t < NT; t++) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](const Frag& f, int s0, int s1) {
#pragma unroll
        for (int st = s0; st < s1; st++)
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int t = 0; t < NT; t++) acc[m][t] = MFMA(f.a[st][m], f.b[st][t], acc[m][t]);
    };
    unsigned go0, go1;
    tile_of(0, go0); tile_of(1, go1);
    // The A tile has ONE LDS buffer: its fragments are read into the second register set a full step before they are used, and a wave's LDS operations
    // execute in order, so the stores of tile t + 1 may follow the reads of tile t directly.
    Stage r[D]; Frag f[2];
#pragma unroll
    for (int j = 0; j < D; j++) { koff_of(j); gload(go0, r[j]); }
    koff_of(D % nkt);
    WRES_WAIT((D - 1) * LPS, r[0]); lstore(r[0]);
    fread(0, f[0]);
    // one tile step; invariant at its top: tile kt + J in f[J & 1], its stage r[J] free, tiles kt + J + 1 .. kt + J + D - 1 in flight.  Straight-line code: selects, no branches
#define WRES_STEP(J, XTRA)                                                                                                                   \
    {                                                                                                                                  \
        const int tn = kt + (J) + D, tn2 = tn + 1;                                                                                     \
        const bool nx = tn >= nkt && more;          /* past the macro-tile's last tile the stream continues with the NEXT one's */     \
        gload(nx ? go1 : go0, r[J]);                                                                                                   \
        koff_of(tn2 < nkt ? tn2 : (more ? tn2 - nkt : nkt - 1));                                                                       \
        mma(f[(J) & 1], 0, HS);                                                                                                        \
        WRES_WAIT((D - 1) * LPS + (XTRA), r[((J) + 1) % D]); lstore(r[((J) + 1) % D]);                                                 \
        fread(kt + (J) + 1 < nkt ? kt + (J) + 1 : 0, f[((J) + 1) & 1]);                                                                \
        mma(f[(J) & 1], HS, 2 * HS);                                                                                                   \
    }
    bool stored = false;                                // the previous epilogue issued its stores (a ragged macro-tile does not)
    for (int gi = 0; gi < ngm; gi++) {
        const bool more = gi + 1 < ngm;
        // (an epilogue's MT * NT stores sit in the memory queue between the stages in flight and the loads issued after it: the first three steps of the next
        // macro-tile count them in, so that a wave never waits for a store to be acknowledged; the fourth wait is three steps later)
        { const int kt = 0; if (stored) { WRES_STEP(0, MT * NT) WRES_STEP(1, MT * NT) WRES_STEP(2, MT * NT) WRES_STEP(3, 0) } else { WRES_STEP(0, 0) WRES_STEP(1, 0) WRES_STEP(2, 0) WRES_STEP(3, 0) } }
        for (int kt = D; kt < nkt; kt += D) { WRES_STEP(0, 0) WRES_STEP(1, 0) WRES_STEP(2, 0) WRES_STEP(3, 0) }      // nkt % D == 0
        // ---- epilogue of workgroup tile gi: this wave's MT M-tiles
        const int mt0 = ((li + gi * cnt) * 4 + wave) * MT;
        stored = mt0 < p.mtiles;
        if (stored) {
            int pos, ct0; fdiv_qr(mt0, fct, pos, ct0);
#pragma unroll
            for (int m = 0; m < MT; m++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const int n = 16 * t + l15;
                    const float bias = bias_r[t];
                    f32x4 v = acc[m][t];
                    act_v4(v, bias, L.act);
                    *reinterpret_cast<f32x4*>(p.out + ((size_t)n * L.npos + pos) * p.ncols + (ct0 + m) * 16 + 4 * kq) = v;
                }
        }
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        go0 = go1; tile_of(gi + 2, go1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the dummy tail loads
    KTRACE_SET(3, 100); KTRACE(7); KTRACE_END();
#undef WRES_WAIT
#undef WRES_STEP
}
// launch_gemm_fwd takes this form for: one chunk, all N channels in one tile of 16 or 32, a multiple of four K tiles, weights + A tiles within half of a CU's LDS, and >= 2048 M-groups
static size_t fwd_wres_lds(const LayerDev& L, int MT) { return (size_t)(4 * MT * (32 * 16 + 4) + L.K * L.N + L.K) * 4; }
static bool fwd_wres_ok(const LayerDev& L, long mg_total, int S) {
    if ((L.opt & DQN_LOPT_NO_FWD_WRES) || S != 1 || (L.N != 16 && L.N != 32) || L.K % 128 || mg_total < 2048) return false;
    return fwd_wres_lds(L, 1) <= (size_t)(160 * 1024 / 3 - 256);
}

// (r03 experiment, removed in r05: an LDS-DMA form of the forward -- global_load_lds_dwordx4 into unpadded, globally swizzled tiles, ring depth 3 -- was bit-exact and no faster:
//  conv2 / conv3 forward at config 5 88.4 / 63.3 us vs 86.1 / 63.7 us for the register-staged kernel; DESIGN.md section 8.1 keeps the numbers)
static int fwd_pick_nt(const LayerDev& L, long mgroups_total, int S) {
    // widest N tile (most reuse of the im2col'd A tile) that still yields >= ~400 workgroups.  Dense layers at B=32 stream
    // their weights once whatever the tile, so they too prefer more, narrower workgroups (measured: FC1 forward 19.2 -> 16.3 us
    // at NT=2 in the train step, 15.6 -> ~10 us at NT=1 for the 32-column acting forward)
    const int cands[3] = {4, 2, 1};
    int best = 1; long best_wgs = -1;
    for (int c = 0; c < 3; c++) {
        const int nt = cands[c];
        if (L.N % (16 * nt)) continue;
        const long wgs = mgroups_total * (L.N / (16 * nt)) * S;
        // (r03: 1024-2048 measured slower for the conv layers -- the A tile is re-read more often.  r06: a DENSE layer re-reads nothing -- its weights stream once whatever the
        //  tile -- and the 3136 -> 512 pair's forward at config 2 runs 14.9 -> 13.2 us on 896 workgroups of 16 channels instead of 448 of 32; the first convolution on 1200
        //  instead of 600: 13.4 -> 14.1; profiles/r06_zo_fwd_nt_probe.txt)
#ifndef DQN_FWD_MIN_WGS_CONV
#define DQN_FWD_MIN_WGS_CONV 400      /* (probe builds: tools/build_variant.sh <name> -DDQN_FWD_MIN_WGS_CONV=...) */
#endif
        const long min_wgs = (L.kind == DQN_LAYER_DENSE && S > 1) ? 800 : DQN_FWD_MIN_WGS_CONV;      // (S > 1: the split-K forward of small batches; the unsplit large-batch launches keep their tiles)
        if (wgs >= min_wgs) return nt;
        if (wgs > best_wgs) { best_wgs = wgs; best = nt; }
    }
    return best;
}

// one launch for up to four problems sharing the layer geometry (sibling layers x {online net on [s;sp], target net on sp});
// split-K partial slabs are left for the caller to reduce (k_reduce_multi, or folded into the consumer)
bool gemm_fwd_eligible(const LayerDev& L, int nprob, const int* ldx, const int* col0, const int* ncols) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    if (L.N % 16 || L.K % F_KT_DEF || (S > 1 && kc % F_KT_DEF) || L.K > 8192 || L.w_off % 4) return false;
    for (int i = 0; i < nprob; i++) if (ncols[i] % 16 || ldx[i] % 4 || col0[i] % 4) return false;
    return true;
}
void launch_gemm_fwd(hipStream_t st, const LayerDev& L, int nprob, const float* const* W, const float* const* bias, const float* const* X,
                     const int* ldx, const int* col0, const int* ncols, float* const* out, float* const* outT, int piece_major) {
    const int S = dqn_nchunks(L.K, L.fwd_kc), kc = dqn_chunk_len(L.K, L.fwd_kc);
    if (piece_major && (L.kind != DQN_LAYER_DENSE || S <= 1)) piece_major = 0;
    GFwdProbs pr; long mg_total = 0;
    for (int i = 0; i < 4; i++) {
        const int j = i < nprob ? i : 0;
        GFwdProb& q = pr.p[i];
        q.W = W[j]; q.bias = bias[j]; q.X = X[j]; q.ldx = ldx[j]; q.col0 = col0[j]; q.ncols = ncols[j]; q.out = out[j]; q.outT = (outT && S == 1 && L.kind == DQN_LAYER_DENSE) ? outT[j] : nullptr;
        q.mtiles = L.npos * (ncols[j] / 16); q.mgroups = (q.mtiles + 3) / 4; q.pm = piece_major;
        if (i < nprob) mg_total += q.mgroups;
    }
    const bool want_t = pr.p[0].outT != nullptr;
    if (!want_t && fwd_wres_ok(L, mg_total, S)) {
        // M-tiles per wave: 4 (64-column row slices: a quarter of the L1 line visits, weight fragments shared by four accumulators; two workgroups per CU) when every
        // problem's column count allows and the LDS holds it, else 2, else 1
        int MT = L.xu8 ? 4 : 2;                          // (float operands: 4 would need > 256 registers)
        for (int i = 0; i < nprob; i++) while (MT > 1 && (ncols[i] / 16) % MT) MT >>= 1;
        while (MT > 1 && fwd_wres_lds(L, MT) > (size_t)(160 * 1024 / 2 - 256)) MT >>= 1;
        const size_t lds_b = fwd_wres_lds(L, MT);
        // persistent workgroups, shared out between the problems in proportion to their tiles
        const int per_cu = (int)((size_t)(160 * 1024) / (lds_b + 256)) < 3 ? (int)((size_t)(160 * 1024) / (lds_b + 256)) : 3;
        const int slots = 256 * per_cu; int end = 0; long wgt[4], wgt_total = 0;
        for (int i = 0; i < nprob; i++) { wgt[i] = (pr.p[i].mtiles + 4 * MT - 1) / (4 * MT); wgt_total += wgt[i]; }
        // (r06, measured and dropped: EQUAL tile counts -- t = ceil(tiles / slots) tiles per workgroup, ceil(tiles_i / t) workgroups per problem, 480 instead of 512 -- because the ktrace
        //  shows workgroups walking 4 or 5 tiles and the launch ending a whole tile after its median workgroup: 105.6 vs 105.7 us, profiles/r06_h_ab.txt; fewer M-tiles per wave
        //  (finer tiles: DQN_WRES_MT = 2 / 1) cost 3 us, profiles/r06_g_split_ab.txt)
        for (int i = 0; i < 4; i++) {
            if (i < nprob) { long w = (slots * wgt[i] + wgt_total / 2) / wgt_total; if (w < 1) w = 1; if (w > wgt[i]) w = wgt[i]; end += (int)w; }
            pr.wg_end[i] = end;
        }
#define WRES_LAUNCH(NT_, MT_, U8_) do { if (lds_b > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_fwd_wres<NT_, MT_, U8_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b); \
        hipLaunchKernelGGL((k_fwd_wres<NT_, MT_, U8_>), dim3(end), dim3(256), lds_b, st, L, pr); } while (0)
#define WRES_PICK(NT_) do { if (L.xu8) { if (MT == 4) WRES_LAUNCH(NT_, 4, true); else if (MT == 2) WRES_LAUNCH(NT_, 2, true); else WRES_LAUNCH(NT_, 1, true); } \
                            else { if (MT == 4) WRES_LAUNCH(NT_, 4, false); else if (MT == 2) WRES_LAUNCH(NT_, 2, false); else WRES_LAUNCH(NT_, 1, false); } } while (0)
        if (L.N == 32) WRES_PICK(2); else WRES_PICK(1);
#undef WRES_PICK
#undef WRES_LAUNCH
        return;
    }
    const int NT = fwd_pick_nt(L, mg_total, S);
    const int ngroups = L.N / (16 * NT);
    int end = 0;
    for (int i = 0; i < 4; i++) { if (i < nprob) end += pr.p[i].mgroups * ngroups * S; pr.wg_end[i] = end; }
    const int SB = NT == 4 ? FwdCfg<4>::SB : (NT == 2 ? FwdCfg<2>::SB : FwdCfg<1>::SB);
    // K tile depth: 32 while a launch is a few hundred workgroups (B = 32: each pays its barriers in full), 16 once it is >= 1024 of them (large
    // batches): half the LDS per workgroup doubles the resident waves (3-4 -> 6-8 per SIMD), which hides more of the operand latency than the extra
    // barrier rounds cost -- measured at config 5 (r03): conv forwards 129.7 / 93.4 / 67.0 -> 120.8 / 85.5 / 63.0 us; at config 2 the same tiles LOSE 1.5 us per launch
    const int kt16_min = 1024;
    const bool k16 = end >= kt16_min;
    const int kt = k16 ? 16 : F_KT_DEF;
    const size_t lds = (size_t)(2 * kt * F_SA + 2 * kt * SB) * 4 + (L.kind == DQN_LAYER_CONV ? (size_t)L.K * 4 : 0);
#define FWD_LAUNCH(NT_, U8_, KT_) hipLaunchKernelGGL((k_fwd_lds<NT_, U8_, KT_>), dim3(end), dim3(256), lds, st, L, pr, S, kc)
#define FWD_PICK(U8_, KT_) do { if (NT == 4) FWD_LAUNCH(4, U8_, KT_); else if (NT == 2) FWD_LAUNCH(2, U8_, KT_); else FWD_LAUNCH(1, U8_, KT_); } while (0)
    // 32x32x2 MFMA blocks for the 64-channel tiles: large launches by default (r04), everywhere with DQN_FWD_M32=1, nowhere with =0 (read at dqn_engine_create)
    const int m32 = ((L.opt & DQN_LOPT_NO_FWD_M32) || want_t || piece_major) ? 0 : ((L.opt & DQN_LOPT_FWD_M32) || k16) ? 1 : 0;
    if (m32 && NT == 4 && !L.xu8) { if (k16) hipLaunchKernelGGL((k_fwd_lds<4, false, 16, true>), dim3(end), dim3(256), lds, st, L, pr, S, kc); else hipLaunchKernelGGL((k_fwd_lds<4, false, F_KT_DEF, true>), dim3(end), dim3(256), lds, st, L, pr, S, kc); }
    else if (L.xu8) { if (k16) FWD_PICK(true, 16); else FWD_PICK(true, F_KT_DEF); }
    else { if (k16) FWD_PICK(false, 16); else FWD_PICK(false, F_KT_DEF); }
#undef FWD_PICK
#undef FWD_LAUNCH
}

// =====================================================================================================================
// dW / db:  G[k][n] = sum_{(pos,b)} X[xb(pos)+koff(k)][b] * dpre[n][pos][b]      db[n] = sum_{(pos,b)} dpre[n][pos][b]
// Both operands have the CONTRACTION index (the sample b) contiguous in memory, so tiles are fetched as full 128-B row
// segments (32 samples) and transposed through LDS: A tile = 64 weight rows x 32 samples, B tile = NW channels x 32 samples.
// workgroup = 64 weight rows (wave w owns rows 16w..16w+15) x NW channels x one split-K chunk; the chunk is walked in
// K tiles of 32 samples (position-major, sample-minor == the canonical chain order).  The workgroup of weight-row tile 0
// also accumulates the bias gradient from the B tile (ascending sample order).
// =====================================================================================================================
struct GDwProb { const float* X; const float* dpre; float* out; };   // out: G + w_off (S == 1) or the partial slab base
struct GDwProbs { GDwProb p[2]; };
// operand layout: ldd = row stride of dpre ([n][pos][b]: npos*B).  The sample axis of a position is either one plain run (tpr = 0) or the
// concatenation of gathered per-rank blocks (dp.hip): tpr 32-sample tiles per rank, consecutive ranks rstride floats apart (X and dpre alike)
struct DwStride { int ldd, tpr, rstride; };
// LDS row stride of the tiles whose rows are read as MFMA fragments ALONG the row (lane (i, kq) reads dword kq + 4*st of row i): 32 + 2.
// ds_read_b32 banks are (dword address) mod 32 per 32-lane half, so row stride 34 puts lane (i, kq) on bank 2*i + kq (+ 4*st): all 32
// lanes of a half on distinct banks.  (Stride 36 -- 16-B aligned rows for ds_write_b128 -- maps rows i and i + 8 to one bank: every
// fragment read 2-way conflicted, r02 PMC: 29-37 % of the LDS cycles of the backward launches.)  Rows are only 8-B aligned, so the
// staging stores are ds_write_b64 pairs (lds_st4): conflict-free too (16-lane groups: two rows x 8 float4 = 32 distinct banks).
constexpr int W_ST = 34;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lds_st4(float* p, f32x4 v) {
    *reinterpret_cast<f32x2*>(p) = (f32x2){v.x, v.y};
    *reinterpret_cast<f32x2*>(p + 2) = (f32x2){v.z, v.w};
}

struct DwUnit { int pi, ng, mr, s; };      // problem (sibling layer), channel group of 16 NT, 64-row tile, split-K chunk
// (r06, built, bit-exact, measured and removed -- VERDICT r05 item 1a: the 64 x 64 dW tile as 2 x 2 blocks of v_mfma_f32_32x32x2_f32, one per wave, on K-MAJOR LDS tiles ([32 samples][65]:
//  conflict-free fragment reads and ds_write_b32 staging stores): 32 instead of 40 fragment reads and 16 instead of 32 MFMA instructions per wave and K tile.  Config 5, same box, three
//  alternations: FC / conv2 / conv3 pairs 74.4 / 68.2 / 46.9 -> 74.1 / 69.6 / 47.3 us (599.1-601.0 vs 598.2-601.1 us per step); config 2 125.9 -> 127.4 us (its epilogue stored single
//  dwords).  Fragment reads and MFMA issue do not bound this body either, as r05's "a quarter of the fragment reads changes nothing" said.  profiles/r06_s_dw_m32_ab.txt)
template <int NT, bool XU8 = false>
__device__ __forceinline__ void dw_lds_body(const LayerDev& L, const GDwProbs& pr, int ldx, int B, int S, int kc, const DwUnit& u, DwStride ds, int probe = 0) {
    constexpr int NW = 16 * NT;
    using AT = std::conditional_t<XU8, uint32_t, f32x4>;
    constexpr int BQ = (NW * 8 + 255) / 256;          // float4 per thread for the B tile (1 or 2)
    extern __shared__ float lds[];
    float* As = lds;                                   // [2][64][W_ST]
    float* Bs = lds + 2 * 64 * W_ST;                   // [2][NW][W_ST]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const bool conv = L.kind == DQN_LAYER_CONV;
    const int pi = u.pi, ng = u.ng, mr = u.mr, s = u.s;
    const GDwProb& p = pr.p[pi];
    const int n0 = ng * NW;
    const int KK = L.npos * B, j0 = s * kc, j1 = min(KK, j0 + kc);
    const int nsub = B / 32, nkt = (j1 - j0) / 32, kt0 = j0 / 32;      // a chunk may start inside a position (kc % 32 == 0, B % 32 == 0: a K tile never straddles two)
    // ---- A tile slice of this thread: rows tid>>3 and +32, float4 (tid & 7)
    const int f4 = tid & 7;
    int koff0, koff1;
    {
        const int k0r = min(mr * 64 + (tid >> 3), L.K - 1), k1r = min(mr * 64 + (tid >> 3) + 32, L.K - 1);
        if (conv) {
            const FDiv fkhw = fdiv_of(L.kh * L.kw), fkw = fdiv_of(L.kw);
            int ci, rem, ky, kx;
            fdiv_qr(k0r, fkhw, ci, rem); fdiv_qr(rem, fkw, ky, kx); koff0 = (ci * L.ih + ky) * L.iw + kx;
            fdiv_qr(k1r, fkhw, ci, rem); fdiv_qr(rem, fkw, ky, kx); koff1 = (ci * L.ih + ky) * L.iw + kx;
        } else { koff0 = k0r; koff1 = k1r; }
    }
    // r04: operand address = SCALAR base of the K tile (arena / dpre + tile offset, scalar ALU) + a 32-bit per-thread byte offset fixed for the whole workgroup: no VALU
    // address arithmetic in the loop (every VALU instruction is paid in fp32 MFMA time on gfx950, tools/micro/mfma_mix.cpp).  One layer's operands are < 4 GB.
    constexpr unsigned ESZ = XU8 ? 1u : 4u;
    const unsigned char* Xbytes = reinterpret_cast<const unsigned char*>(p.X);
    const unsigned a_off0 = ((unsigned)koff0 * (unsigned)ldx + 4u * f4) * ESZ, a_off1 = ((unsigned)koff1 * (unsigned)ldx + 4u * f4) * ESZ;
    // ---- B tile slice: channel rows q>>3 (clamped), float4 (q & 7)
    const int bq0 = tid < NW * 8 ? tid : NW * 8 - 1, bq1 = tid + 256 < NW * 8 ? tid + 256 : NW * 8 - 1;
    const unsigned b_off0 = 4u * ((unsigned)(n0 + (bq0 >> 3)) * (unsigned)ds.ldd + 4u * (bq0 & 7));
    const unsigned b_off1 = 4u * ((unsigned)(n0 + (bq1 >> 3)) * (unsigned)ds.ldd + 4u * (bq1 & 7));
    constexpr int LPS = 2 + BQ;
    struct Stage { AT a0, a1; f32x4 b0, b1; };
    auto gld = [](unsigned off, const void* base) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
    auto gld1 = [](unsigned off, const void* base) { uint32_t v; asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
    auto cvt = [](const AT& a) { if constexpr (XU8) { const uint32_t w4 = a; return (f32x4){u8_unit(w4 & 0xffu), u8_unit((w4 >> 8) & 0xffu), u8_unit((w4 >> 16) & 0xffu), u8_unit(w4 >> 24)}; } else return a; };
    // cursor of the staging loads (r04): gload is called for K tiles 0, 1, 2, ... in order (clamped at the last), so the 32-sample block `sub` of position
    // `pos` = (oy, ox) and the block's place in a gathered rank layout advance by carries -- the five integer divisions per tile this replaces were ~5 scalar
    // instructions per MFMA in the first convolution's dW launch (PMC r04_o)
    int g_kt = 0, g_pos, g_sub, g_oy = 0, g_ox = 0, g_rk = 0, g_st;
    fdiv_qr(kt0, fdiv_of(nsub), g_pos, g_sub); g_st = g_sub;
    if (conv) fdiv_qr(g_pos, fdiv_of(L.ow), g_oy, g_ox);
    if (ds.tpr > 0) fdiv_qr(g_sub, fdiv_of(ds.tpr), g_rk, g_st);
    auto gload = [&](int, Stage& r) {
        const unsigned so = ds.tpr > 0 ? (unsigned)g_rk * (unsigned)ds.rstride + (unsigned)g_st * 32u : (unsigned)g_sub * 32u;
        const unsigned ao = so, bo = (unsigned)g_pos * (unsigned)B + so;
        const int xb = conv ? g_oy * L.sh * L.iw + g_ox * L.sw : 0;
        if (g_kt + 1 < nkt) {
            g_kt++; g_sub++; g_st++;
            if (ds.tpr > 0 && g_st == ds.tpr) { g_st = 0; g_rk++; }
            if (g_sub == nsub) { g_sub = 0; g_st = 0; g_rk = 0; g_pos++; g_ox++; if (g_ox == L.ow) { g_ox = 0; g_oy++; } }
        }
        const unsigned char* pa = Xbytes + (size_t)((unsigned)xb * (unsigned)ldx + ao) * ESZ;
        const float* pb = p.dpre + bo;
        if constexpr (XU8) { r.a0 = gld1(a_off0, pa); r.a1 = gld1(a_off1, pa); }
        else { r.a0 = gld(a_off0, pa); r.a1 = gld(a_off1, pa); }
        r.b0 = gld(b_off0, pb);
        if (BQ > 1) r.b1 = gld(b_off1, pb);
    };
    auto lstore = [&](int buf, const Stage& r) {
        lds_st4(As + (buf * 64 + (tid >> 3)) * W_ST + 4 * f4, cvt(r.a0));
        lds_st4(As + (buf * 64 + (tid >> 3) + 32) * W_ST + 4 * f4, cvt(r.a1));
        if (tid < NW * 8) lds_st4(Bs + (buf * NW + (tid >> 3)) * W_ST + 4 * f4, r.b0);
        if (BQ > 1) lds_st4(Bs + (buf * NW + ((tid + 256) >> 3)) * W_ST + 4 * f4, r.b1);
    };
#define STAGE_WAIT(N, r) do { if (BQ > 1) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0), "+v"(r.b1) : "n"(N) : "memory"); \
                              else asm volatile("s_waitcnt vmcnt(%3)" : "+v"(r.a0), "+v"(r.a1), "+v"(r.b0) : "n"(N) : "memory"); } while (0)
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float dbacc = 0.0f;
    const bool do_bias = mr == 0 && tid < NW;
    auto compute = [&](int buf) {
        const float* Ab = As + (buf * 64 + 16 * wave + l15) * W_ST + kq;
        const float* Bb = Bs + (buf * NW + l15) * W_ST + kq;
        float af[8], bf[8][NT];
#pragma unroll
        for (int st = 0; st < 8; st++) {
            af[st] = Ab[4 * st];
#pragma unroll
            for (int t = 0; t < NT; t++) bf[st][t] = Bb[16 * t * W_ST + 4 * st];
        }
#pragma unroll
        for (int st = 0; st < 8; st++)
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t] = MFMA(af[st], bf[st][t], acc[t]);
        if (do_bias) {
            // two samples per read: ds_read_b64 banks are (dword address) mod 64 per 32-lane group and 34 * lane takes 32 distinct even values there -- conflict-free, where the
            // one-dword reads put lanes l and l + 16 on one bank (r04; the additions stay in ascending sample order)
            const float* br = Bs + (buf * NW + tid) * W_ST;
#pragma unroll
            for (int b = 0; b < 32; b += 2) { const f32x2 v2 = *reinterpret_cast<const f32x2*>(br + b); dbacc = dbacc + v2.x; dbacc = dbacc + v2.y; }
        }
    };
    Stage r0, r1;
    r0.b1 = r1.b1 = (f32x4){0.f, 0.f, 0.f, 0.f};
    gload(0, r0); STAGE_WAIT(0, r0); lstore(0, r0); __syncthreads();
    if (nkt == 1) {
        // a chunk of a single K tile (conv layers at B = 32: one position x 32 samples; the wide dense layers): nothing to pipeline, and
        // the clamped prefetches below would only add three serialized round trips before the workgroup may exit
        compute(0);
    } else {
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef STAGE_WAIT
    const size_t per_s = (size_t)(L.K + 1) * L.N;
    float* out = p.out + (size_t)s * per_s;
    if (probe & 4) return;
    if ((((size_t)out) & 15) == 0 && NT >= 2) {
        // epilogue through LDS: the accumulators hold 4 rows x 1 column per lane (16 scattered 64-byte pieces per wave and N tile); transposed through LDS each store
        // instruction writes whole 16*NT-float row segments -- NT instead of 4*NT store instructions per wave, full lines instead of halves (r03_i PMC: the non-temporal
        // half-line stores moved 21.7 MB for the 12.8 MB dense gradient).  Non-temporal: the gradient is read once, by the Adam launch.
        // r04: the transposed tile is [64 rows][NW] with NO padding and the 16-column tiles of rows 4..7 (mod 8) swapped pairwise (column ^ 16 for odd kq):
        //   stores  (ds_write_b32: banks mod 32 per 32-lane group): lanes (i, kq = 0 | 1) hit banks (16 t + i) and (16 (t ^ 1) + i) -- disjoint;
        //   reads   (ds_read_b128: banks mod 64 per 16-lane group {0-3,12-15,20-27} / {4-11,16-19,28-31} / +32): each group reads 64 CONTIGUOUS floats
        //           (one row of NW = 64, two adjacent rows of NW = 32) -- every bank once.
        // The round-3 layout (the wave's own rows of the A tile, stride 34) needed no barrier but conflicted 2-way on both sides: 12-14 % of the backward
        // launches' LDS cycles (profiles/history/r03_q_pmc_sq.txt).  One barrier: every wave is past its last fragment read before the tile is overwritten.
        __syncthreads();
        float* T = lds;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            float* Tt = T + (16 * wave + 4 * kq) * NW + ((16 * t + l15) ^ ((kq & 1) << 4));
            Tt[0] = acc[t].x; Tt[NW] = acc[t].y; Tt[2 * NW] = acc[t].z; Tt[3 * NW] = acc[t].w;
        }
        // lane -> (16-lane hardware group g, index j in the group): rows of this wave only (LDS operations of a wave execute in order: no second barrier)
        const int l5 = lane & 31; int g, j;
        if (l5 < 4) { g = 0; j = l5; } else if (l5 < 12) { g = 1; j = l5 - 4; } else if (l5 < 16) { g = 0; j = l5 - 8; } else if (l5 < 20) { g = 1; j = l5 - 8; } else if (l5 < 28) { g = 0; j = l5 - 12; } else { g = 1; j = l5 - 16; }
        g += (lane >> 5) << 1;
        constexpr int F = 4 * NT, RPG = 16 / F, RPI = 4 * RPG;      // float4 per row, rows per 16-lane group (1 or 2), rows per store instruction (4 or 8)
        const int rg = g * RPG + j / F, p4 = j % F;                  // row within the instruction, PHYSICAL float4 slot of that row
#pragma unroll
        for (int i = 0; i < 16 / RPI; i++) {
            const int row = i * RPI + rg, k = mr * 64 + 16 * wave + row;
            const f32x4 v = *reinterpret_cast<const f32x4*>(T + (16 * wave + row) * NW + 4 * p4);
            const int n4 = p4 ^ (((row >> 2) & 1) << 2);            // the logical float4 this slot holds (rows 4kq + r: tiles swapped where kq is odd)
            if (k < L.K) st_grad4(reinterpret_cast<f32x4*>(out + (size_t)k * L.N + n0 + 4 * n4), v, L.opt & DQN_LOPT_ST_WT);
        }
    } else {
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = n0 + 16 * t + l15;
        const float v[4] = {acc[t].x, acc[t].y, acc[t].z, acc[t].w};
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int k = mr * 64 + 16 * wave + 4 * kq + r;
            if (k < L.K) __builtin_nontemporal_store(v[r], &out[(size_t)k * L.N + n]);
        }
    }
    }
    if (do_bias) out[(size_t)L.K * L.N + n0 + tid] = dbacc;
}
// The dW section of a launch: `nblocks` workgroups = nfull whole units (64 rows x 16 NT channels x one chunk) followed by 2 * nsplit HALF units -- the launch's last
// nsplit units cut in two along N (8 NT channels each; NT >= 2).  r06 ktrace at B = 512 (profiles/r06_b_ktrace_bwd_b512.txt): a backward launch ends in a drain as long as
// one dW workgroup lives (16-19 us of a 45-77 us launch) at ~40 % occupancy -- the dispatcher runs out of workgroups while the last round is still whole; halves at the end
// halve the drain.  Every output element keeps its chain (a half contracts the same samples in the same order for half of the channels): bit-identical.
template <int NT, bool XU8>
__device__ __forceinline__ void dw_section(const LayerDev& L, const GDwProbs& pr, int nprob, int ldx, int B, int S, int kc, int bid, int nblocks, int nsplit, DwStride ds, int probe = 0) {
    const int mrows = (L.K + 63) / 64, ngroups = L.N / (16 * NT), nfull = nblocks - 2 * nsplit;
    int w, half = -1;
    if (NT == 1 || bid < nfull) w = xcd_remap(bid, nfull);
    else { const int hw = xcd_remap(bid - nfull, 2 * nsplit); w = nfull + (hw >> 1); half = hw & 1; }
    DwUnit u;                                          // (decodes through reciprocals: fdiv_*, common.h)
    { int w2, w3; fdiv_qr(w, fdiv_of(nprob), w2, u.pi); fdiv_qr(w2, fdiv_of(ngroups), w3, u.ng); fdiv_qr(w3, fdiv_of(mrows), u.s, u.mr); }
    // (r06, measured and dropped: the bias-carrying units (mr == 0: 10-20 % longer, profiles/r06_c_ktrace_groups_b512.txt) leading the section -- conv2's pair 67.6 -> 66.8 us, nothing elsewhere)
    if constexpr (NT >= 2) { if (half >= 0) { u.ng = 2 * u.ng + half; dw_lds_body<NT / 2, XU8>(L, pr, ldx, B, S, kc, u, ds, probe); return; } }
    dw_lds_body<NT, XU8>(L, pr, ldx, B, S, kc, u, ds, probe);
}
// halves at the end of a dW section: only where a unit is long enough to matter (>= 4 K tiles: large batches) and the tile can be cut (NT >= 2);
// DQN_DW_SPLIT (LayerDev::opt bits 8..15, in units of 16) = how many of the last units are cut, at most half of them
static int dw_split_units(const LayerDev& L, int NT, int units, int B) {
    const int KK = L.npos * B, kc = dqn_chunk_len(KK, L.dw_kc), v = 16 * ((L.opt >> 8) & 0xff);
    if (NT < 4 || kc < 128 || v == 0) return 0;      // (NT = 2 -> 1: the 16-channel body has no transposed epilogue and conv1's 1024 workgroups are all resident at once -- measured 43.9 -> 54.6 us)
    return v < units / 2 ? v : units / 2;
}
template <int NT, bool XU8 = false>
__global__ __launch_bounds__(256) void k_dw_lds(LayerDev L, GDwProbs pr, int nprob, int ldx, int B, int S, int kc, DwStride ds, int nsplit, GemmTail tail) {
    // dispatch order: [priority block][tail: VALU tasks, Adam job][dW workgroups] -- the bandwidth-bound tail starts at once and the short dW
    // workgroups fill the slots beside it (at the END of the grid the tail would wait for LDS: every workgroup of a launch reserves the tile size)
    karg_warm<sizeof(LayerDev) + sizeof(GDwProbs) + 32 + sizeof(DwStride) + sizeof(GemmTail)>();
    KTRACE_BEGIN()      // record: {grid, block, entry, role (0 prio, 2 tail, 3 dW), -, -, -, exit}
    const int pre_ = (tail.has_adam && tail.adam.prio.n > 0) ? 1 : 0;
    if (pre_ && blockIdx.x == 0) { extern __shared__ float lds[]; prio_block_run(tail.adam.prio, tail.adam.state, reinterpret_cast<long long*>(lds), tail.lds_bytes); KTRACE_SET(3, 0); KTRACE(7); KTRACE_END(); return; }
    const int bid = (int)blockIdx.x - pre_, ntail = (int)gemm_tail_blocks(tail) - pre_;
    if (bid < ntail) { gemm_tail_run(tail, (unsigned)bid); KTRACE_SET(3, 2); KTRACE(7); KTRACE_END(); return; }
    dw_section<NT, XU8>(L, pr, nprob, ldx, B, S, kc, bid - ntail, (int)gridDim.x - pre_ - ntail, nsplit, ds);
    KTRACE_SET(3, 3);
    KTRACE(7); KTRACE_END();
}
bool gemm_dw_eligible(const LayerDev& L, int B, int ldx) {
    const int KK = L.npos * B, S = dqn_nchunks(KK, L.dw_kc), kc = dqn_chunk_len(KK, L.dw_kc);
    return !(L.N % 16 || B % 32 || ldx % 4 || L.K < 16 || (S > 1 && kc % 32));
}
// nprob (<= 2) sibling layers of identical geometry in one launch; out[i] = gradient base (S == 1) or partial slab base
void launch_gemm_dw(hipStream_t st, const LayerDev& L, int nprob, const float* const* X, int ldx, const float* const* dpre, int B, float* const* out, int ldd, int tpr, int rstride, GemmTail tail) {
    const int KK = L.npos * B, S = dqn_nchunks(KK, L.dw_kc), kc = dqn_chunk_len(KK, L.dw_kc);
    const DwStride ds = {ldd > 0 ? ldd : L.npos * B, tpr, rstride};
    GDwProbs pr;
    for (int i = 0; i < 2; i++) { const int j = i < nprob ? i : 0; pr.p[i].X = X[j]; pr.p[i].dpre = dpre[j]; pr.p[i].out = out[j]; }
    const int NT = L.N % 64 == 0 ? 4 : (L.N % 32 == 0 ? 2 : 1);
    const int units = ((L.K + 63) / 64) * (L.N / (16 * NT)) * S * nprob, nsplit = tpr > 0 ? 0 : dw_split_units(L, NT, units, B);
    const int grid = units + nsplit + (int)gemm_tail_blocks(tail);
    const size_t lds = (size_t)(2 * 64 * W_ST + 2 * 16 * NT * W_ST) * 4;
    tail.lds_bytes = (unsigned)lds; tail.probe = HOST_PROBE();
    if (L.xu8) {
        if (NT == 4) hipLaunchKernelGGL((k_dw_lds<4, true>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
        else if (NT == 2) hipLaunchKernelGGL((k_dw_lds<2, true>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
        else hipLaunchKernelGGL((k_dw_lds<1, true>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
    }
    else if (NT == 4) hipLaunchKernelGGL((k_dw_lds<4>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
    else if (NT == 2) hipLaunchKernelGGL((k_dw_lds<2>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
    else hipLaunchKernelGGL((k_dw_lds<1>), dim3(grid), dim3(256), lds, st, L, pr, nprob, ldx, B, S, kc, ds, nsplit, tail);
}

// =====================================================================================================================
// dX (+ act' of the producing layer, + the dueling join):
//   dense: dX[f][b] = sum_n dpre[n][b] W[f][n]                 conv: dX[ci][iy][ix][b] = sum_{valid taps} sum_co dpre[co][oy][ox][b] W[(ci,tap)][co]
// workgroup = 32 input features (dense: rows f; conv: channels ci at ONE input position) x 32 samples; wave w owns
// feature tile (w & 1) and sample tile (w >> 1).  K tiles are 32 deep: A tile = 32 dpre rows x 32 samples (row = n / co),
// B tile = 32 weight rows x 32 k (k contiguous in memory, transposed through LDS).
// Up to two SOURCES are accumulated separately and added at the end -- the two streams of a dueling network meeting at
// the base output (dX_val + dX_adv, src/dueling.jl:10 backward) -- so the join costs no extra launch.
// =====================================================================================================================
struct GDxSrc { const float* W; const float* dpre; };
struct GDxArgs { GDxSrc src[2]; int nsrc; float* out; const float* ysrc; int ldy, act_src; };
constexpr int X_SB = 34;     // B tile row stride (32 k + 2 pad): fragment reads along the row, conflict-free like W_ST (stores: lds_st4)

// ---- small batches: ONE output tile (32 features x 32 samples) per workgroup, its contraction cut into up to four UNITS -- (source, plan chunk)
// pairs: the two streams of a dueling join x dense chunks of n (plan.dx_kc) or conv chunks of RAW taps -- and every unit contracted by ONE wave
// through a wave-PRIVATE pipeline: own register stages (two K tiles of global loads in flight), own LDS tiles, all four accumulator tiles
// (32 MFMAs per 32-deep K tile), no workgroup barrier inside the loop (LDS operations of one wave execute in order).  r02 ktrace: the dX
// workgroups were the critical path of every backward launch -- a lone wave per SIMD walking load -> LDS -> barrier -> LDS -> 8 MFMAs through
// 16-18 K tiles (8.7-15 us) -- while the chip idled; here the serial chain of a workgroup is its longest unit (4-8 K tiles at config 2).
// The unit sums meet in LDS and are added in canonical order: chunks ascending per source, then source 0 + source 1.
constexpr int U_AW = 32 * 32;            // A tile [32 k][32 samples]: row stride 32, the two 16-sample halves of ODD rows swapped, so the fragment read
                                         // of lanes (i, kq) hits bank 16*((mt ^ kq) & 1) + i: conflict-free without padding
constexpr int U_BW = 32 * X_SB;          // B tile [32 features][34]
constexpr int U_WAVE = U_AW + U_BW;      // floats per wave (8448 B)
constexpr int U_MAX = 4;                 // units per workgroup = waves
#ifndef DQN_U_FT
#define DQN_U_FT 1
#endif
constexpr int U_FT = DQN_U_FT;           // 16-feature tiles per workgroup (1: 16 x 32 output tiles; 2: 32 x 32 -- measured, r03: see dx_units_body)
static size_t dx_units_lds_bytes() { return (size_t)(4 * U_WAVE + 64 + 8) * 4; }
// FT = 16-feature tiles per workgroup: 2 (32 x 32 output tiles) or 1 (16 x 32: twice the workgroups -- a CU draws ~10 B/clk from HBM whatever it has
// in flight (MI355X guide), so a weight-streaming dX confined to 98 CUs (the FC join with 32-feature tiles) could not exceed ~2.3 TB/s)
template <int FT>
__device__ __forceinline__ void dx_units_body(const LayerDev& L, const GDxArgs& A, int B, int S, int kc, int bid, int nblocks, int by, unsigned long long* ktr = nullptr /* debug: words 4..6 of this workgroup's ktrace record */) {
    constexpr int FW = 16 * FT;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    float* As = lds + wave * U_WAVE; float* Bs = As + U_AW;
    int* taps = (int*)(lds + 4 * U_WAVE);             // conv: valid taps of this input position ((tap << 16) | pos), raw-tap ascending
    int* cst = taps + 64;                             // conv: cst[j] = first entry of non-empty chunk j, cst[n] = number of taps, cst[7] = n
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int b0 = by * 32;
    const int w = xcd_remap(bid, nblocks);
    const int nfeat = dense ? L.K : L.cin, khw = L.kh * L.kw;
    int f0, ip = 0, nch;
    if (dense) { const int ftiles = (L.K + FW - 1) / FW; int q, r; fdiv_qr(w, fdiv_of(ftiles), q, r); f0 = r * FW; nch = S; }
    else {
        const int ctiles = L.cin / FW; int r; fdiv_qr(w, fdiv_of(ctiles), ip, r); f0 = r * FW;
        if (tid < 64) {
            // lane (ky*kw + kx) tests its own tap; ballots compact the valid ones in raw-tap order and mark the first valid tap of every chunk
            int iy, ix; fdiv_qr(ip, fdiv_of(L.iw), iy, ix); bool ok = false; int val = 0;      // (reciprocal decodes: fdiv_*, common.h)
            if (tid < khw) {
                int ky, kx; fdiv_qr(tid, fdiv_of(L.kw), ky, kx);
                const int ty = iy - ky, tx = ix - kx;
                if (ty >= 0 && tx >= 0) {
                    int oy, ry, ox, rx; fdiv_qr(ty, fdiv_of(L.sh), oy, ry); fdiv_qr(tx, fdiv_of(L.sw), ox, rx);
                    if (ry == 0 && rx == 0 && oy < L.oh && ox < L.ow) { ok = true; val = (tid << 16) | (oy * L.ow + ox); }
                }
            }
            const int tcr = DQN_CONV_TAP_CHUNK(L); const FDiv ftcr = fdiv_of(tcr);
            const unsigned long long m = __ballot(ok), below = (1ull << tid) - 1ull, mb = m & below;
            const bool first = ok && (mb == 0 || fdiv_q(63 - __clzll(mb), ftcr) != fdiv_q(tid, ftcr));
            const unsigned long long fm = __ballot(first);
            if (ok) taps[__popcll(mb)] = val;
            if (first) cst[__popcll(fm & below)] = __popcll(mb);
            if (tid == 0) { const int n = __popcll(fm); cst[n] = __popcll(m); cst[7] = n; }
        }
        __syncthreads();
        nch = cst[7];
    }
    const int NU = A.nsrc * nch;
    const bool active = wave < NU;
    const int ft = wave % FT, mt = wave / FT;         // the accumulator tile this wave FINISHES (epilogue; waves >= 2 * FT have none)
    const bool fin = wave < 2 * FT;
    // epilogue operand requested up front: the producer's activation for the fused act' multiply
    const int fl = f0 + 16 * ft + l15;
    const size_t feat = dense ? (size_t)min(fl, nfeat - 1) : (size_t)min(fl, nfeat - 1) * L.ih * L.iw + ip;
    // (requested AFTER the first operand tiles: the register allocator parks this long-lived value in an AGPR, which needs the loaded data -- issued
    // first, that copy put a full cold round trip in front of the tile loads: 4-5 us from entry to the first global_load, ktrace r03)
    f32x4 y_e = {0.f, 0.f, 0.f, 0.f};
    const float* y_ptr = (A.ysrc && fin) ? A.ysrc + feat * A.ldy + b0 + 16 * mt + 4 * kq : nullptr;
    f32x4 acc[2 * FT];                                 // [sample tile * FT + feature tile] of this wave's unit
#pragma unroll
    for (int j = 0; j < 2 * FT; j++) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (active) {
        const int wv = __builtin_amdgcn_readfirstlane(wave);      // the unit index in a SCALAR register: everything derived from it (tile counts, cursor, bases) is scalar arithmetic and uniform branches
        const int si = fdiv_q(wv, fdiv_of(nch)), cj = wv - si * nch;
        const GDxSrc& sr = A.src[si];
        const int cot = L.N / 32;
        int nt, n0 = 0, t0 = 0;
        if (dense) { n0 = cj * kc; nt = (min(L.N, n0 + kc) - n0) / 32; }
        else { t0 = cst[cj]; nt = (cst[cj + 1] - t0) * cot; }
        // staging slices of this lane: rows (lane >> 3) + 8p, float4 (lane & 7) of both tiles
        const int row0 = lane >> 3, f4 = lane & 7;
        // r04: operand address = SCALAR base of the tile (scalar ALU; the unit and its tile cursor are wave-uniform) + a 32-bit per-lane byte offset fixed for the whole
        // unit: no VALU address arithmetic per load (every VALU instruction is paid in fp32 MFMA time on gfx950, and a wave is alone on its SIMD here)
        unsigned aoff[4], boff[2 * FT];
#pragma unroll
        for (int p = 0; p < 4; p++) aoff[p] = 4u * ((unsigned)(row0 + 8 * p) * (unsigned)(dense ? B : L.npos * B) + (unsigned)(b0 + 4 * f4));
#pragma unroll
        for (int p = 0; p < 2 * FT; p++) { const unsigned frow = (unsigned)min(f0 + row0 + 8 * p, nfeat - 1); boff[p] = 4u * (dense ? frow * (unsigned)L.N + 4u * f4 : frow * (unsigned)(khw * L.N) + 4u * f4); }
        struct Stage { f32x4 a[4], b[2 * FT]; };
        auto gld = [](unsigned off, const float* base) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
        // tile cursor: gload is called for tiles 0, 1, 2, ... in order (clamped at the last): tile = (tap g_ti of the chunk, channel tile g_ci); the tap of the NEXT
        // tile is read from the table one call ahead
        int g_t = 0, g_ti = 0, g_ci = 0;
        int g_tp = dense ? 0 : taps[t0];
        auto gload = [&](int, Stage& r) {
            size_t sa, sb;
            if (dense) { const unsigned nb = (unsigned)(n0 + 32 * g_t); sa = (size_t)nb * (unsigned)B; sb = nb; }
            else {
                const int tp = __builtin_amdgcn_readfirstlane(g_tp); const unsigned tap = (unsigned)tp >> 16, pos = (unsigned)tp & 0xffffu, cob = (unsigned)g_ci * 32u;
                sa = ((size_t)cob * (unsigned)L.npos + pos) * (unsigned)B; sb = (size_t)tap * (unsigned)L.N + cob;
            }
            const float* pa = uniform_ptr(sr.dpre + sa); const float* pb = uniform_ptr(sr.W + sb);      // (the unit is the wave: uniform, which the compiler cannot see)
#pragma unroll
            for (int p = 0; p < 4; p++) r.a[p] = gld(aoff[p], pa);
#pragma unroll
            for (int p = 0; p < 2 * FT; p++) r.b[p] = gld(boff[p], pb);
            if (g_t + 1 < nt) {
                g_t++; g_ci++;
                if (g_ci == cot) { g_ci = 0; g_ti++; }
                if (!dense) g_tp = taps[t0 + g_ti];
            }
        };
        const int aswz = ((row0 & 1) << 2) ^ f4;     // float4 column of this lane's A rows (all of one parity: row0 + 8p)
        auto lstore = [&](const Stage& r) {
#pragma unroll
            for (int p = 0; p < 4; p++) *reinterpret_cast<f32x4*>(As + (row0 + 8 * p) * 32 + 4 * aswz) = r.a[p];
#pragma unroll
            for (int p = 0; p < 2 * FT; p++) lds_st4(Bs + (row0 + 8 * p) * X_SB + 4 * f4, r.b[p]);
        };
        constexpr int LPS = 4 + 2 * FT;                // loads per stage
#define STAGE_WAIT8(r) do { if constexpr (FT == 2) asm volatile("s_waitcnt vmcnt(8)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]), "+v"(r.b[2]), "+v"(r.b[3 % (2 * FT)]) :: "memory"); \
                            else asm volatile("s_waitcnt vmcnt(6)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b[0]), "+v"(r.b[1]) :: "memory"); } while (0)
        static_assert(LPS == 8 || LPS == 6, "STAGE_WAIT8 counts");
        // fragments of HALF a tile (4 MFMA steps x 2 sample tiles / 2 feature tiles = 16 registers), two sets: the reads of one half ride under
        // the 16 MFMAs of the other, and the next tile's LDS round trip (wait, stores, reads of its first half) under the second half's MFMAs
        struct Frag { float a[4][2], b[4][FT]; };
        auto fread = [&](Frag& f, int h) {
            const float* Ab = As + (16 * h + kq) * 32 + l15;
            const float* Bb = Bs + l15 * X_SB + kq + 16 * h;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                f.a[st][0] = Ab[st * 128 + ((kq & 1) << 4)]; f.a[st][1] = Ab[st * 128 + (((kq & 1) ^ 1) << 4)];
#pragma unroll
                for (int q = 0; q < FT; q++) f.b[st][q] = Bb[16 * q * X_SB + 4 * st];
            }
        };
        auto mma = [&](const Frag& f) {
#pragma unroll
            for (int st = 0; st < 4; st++)
#pragma unroll
                for (int m = 0; m < 2; m++)
#pragma unroll
                    for (int q = 0; q < FT; q++) acc[m * FT + q] = MFMA(f.a[st][m], f.b[st][q], acc[m * FT + q]);
        };
        Stage r0, r1; Frag fa, fb;
        gload(0, r0); gload(1, r1);
        if (y_ptr) y_e = *reinterpret_cast<const f32x4*>(y_ptr);
        STAGE_WAIT8(r0); lstore(r0); gload(2, r0); fread(fa, 0);
        if (ktr) ktr[4] = __builtin_amdgcn_s_memtime();
        for (int t = 0; t < nt; t += 2) {
            // tile t is in LDS, its first half in fa; r1 = tile t + 1 and r0 = tile t + 2 in flight
            fread(fb, 1); mma(fa);
            STAGE_WAIT8(r1); lstore(r1); gload(t + 3, r1);      // LDS operations of a wave execute in order: the stores follow the reads above
            if (t + 1 < nt) fread(fa, 0);
            mma(fb);
            if (t + 1 < nt) {
                fread(fb, 1); mma(fa);
                STAGE_WAIT8(r0); lstore(r0); gload(t + 4, r0);
                if (t + 2 < nt) fread(fa, 0);
                mma(fb);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef STAGE_WAIT8
        if (ktr) ktr[5] = __builtin_amdgcn_s_memtime();
    }
    else if (y_ptr) y_e = *reinterpret_cast<const f32x4*>(y_ptr);
    __syncthreads();                                   // every wave is done with its staging tiles: the unit sums alias them
    f32x4* slot = reinterpret_cast<f32x4*>(lds);       // [unit][accumulator tile][lane]
    if (active) {
#pragma unroll
        for (int j = 0; j < 2 * FT; j++) slot[(wave * 2 * FT + j) * 64 + lane] = acc[j];
    }
    __syncthreads();
    if (ktr) ktr[6] = __builtin_amdgcn_s_memtime();
    if (!fin || fl >= nfeat) return;
    // wave w finishes accumulator tile w = (sample tile mt, feature tile ft): chunk sums ascending per source, then the two sources
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int si = 0; si < A.nsrc; si++) {
        f32x4 tot = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < nch; c++) {
            const f32x4 x = slot[((si * nch + c) * 2 * FT + wave) * 64 + lane];
            if (c == 0) tot = x; else { tot.x = tot.x + x.x; tot.y = tot.y + x.y; tot.z = tot.z + x.z; tot.w = tot.w + x.w; }
        }
        if (si == 0) v = tot; else { v.x = v.x + tot.x; v.y = v.y + tot.y; v.z = v.z + tot.z; v.w = v.w + tot.w; }
    }
    if (A.ysrc) { v.x = dact_f(v.x, y_e.x, A.act_src); v.y = dact_f(v.y, y_e.y, A.act_src); v.z = dact_f(v.z, y_e.z, A.act_src); v.w = dact_f(v.w, y_e.w, A.act_src); }
    const size_t featx = dense ? (size_t)fl : (size_t)fl * L.ih * L.iw + ip;
    st_out4(reinterpret_cast<f32x4*>(A.out + featx * B + b0 + 16 * mt + 4 * kq), v, L.opt & DQN_LOPT_ST_WT);
}
// Large batches (B % 128 == 0): 32 features x 128 SAMPLES per workgroup.  Wave w owns samples 32w..32w+31 and both 16-feature tiles: four
// accumulator tiles per source, 32 MFMAs per K tile and wave between two barriers instead of 8, and the conv prologue (tap list) is paid once
// per 128 samples.  The per-element chain (taps ascending, output channels ascending, sources added at the end) is dx_lds_body's.
constexpr int X_SAW = 144;   // A tile row stride (128 samples + 16 pad): fragment reads hit banks 16*kq + i
__device__ __forceinline__ void dx_lds_body_wide(const LayerDev& L, const GDxArgs& A, int B, int S, int kc, int bid, int nblocks, int by) {
    extern __shared__ float lds[];
    float* As = lds;                                  // [2][32][X_SAW]
    float* Bs = lds + 2 * 32 * X_SAW;                 // [2][32][X_SB]
    int* taps = (int*)(Bs + 2 * 32 * X_SB);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int b0 = by * 128;
    int w = xcd_remap(bid, nblocks);
    int f0, s = 0, ip = 0;
    const int nfeat = dense ? L.K : L.cin;
    if (dense) { const int ftiles = (L.K + 31) / 32; int r; fdiv_qr(w, fdiv_of(ftiles), s, r); f0 = r * 32; }
    else { const int ctiles = L.cin / 32; int r; fdiv_qr(w, fdiv_of(ctiles), ip, r); f0 = r * 32; }
    int nkt;
    const int khw = L.kh * L.kw;
    if (dense) { const int n0 = s * kc, n1 = min(L.N, n0 + kc); nkt = (n1 - n0) / 32; }
    else {
        if (tid < 64) {
            int iy, ix; fdiv_qr(ip, fdiv_of(L.iw), iy, ix); bool ok = false; int val = 0;
            if (tid < L.kh * L.kw) {
                int ky, kx; fdiv_qr(tid, fdiv_of(L.kw), ky, kx);
                const int ty = iy - ky, tx = ix - kx;
                if (ty >= 0 && tx >= 0) {
                    int oy, ry, ox, rx; fdiv_qr(ty, fdiv_of(L.sh), oy, ry); fdiv_qr(tx, fdiv_of(L.sw), ox, rx);
                    if (ry == 0 && rx == 0 && oy < L.oh && ox < L.ow) { ok = true; val = (tid << 16) | (oy * L.ow + ox); }
                }
            }
            const unsigned long long m = __ballot(ok);
            if (ok) taps[__popcll(m & ((1ull << tid) - 1ull))] = val;
            if (tid == 0) taps[255] = __popcll(m);
        }
        __syncthreads();
        nkt = taps[255] * (L.N / 32);
    }
    const int total = nkt * A.nsrc;
    // staging: A = 32 rows x 32 float4 (4 per thread: rows (tid >> 5) + 8p, float4 tid & 31); B = row tid >> 3, float4 tid & 7
    const int arow = tid >> 5, af4 = tid & 31, brow = tid >> 3, bf4 = tid & 7;
    const int frow = min(f0 + brow, nfeat - 1);
    struct Stage { f32x4 a[4]; f32x4 b; };
    // r04: operand address = SCALAR base of the tile (source pointer + tile offset, scalar ALU) + a 32-bit per-thread byte offset fixed for the whole workgroup -- the
    // loads cost no VALU instruction (they used to: five 64-bit multiply-adds per stage, 80 v_mul_lo / v_mad_u64 per four K tiles, and on gfx950 every VALU
    // instruction is paid in fp32 MFMA time, tools/micro/mfma_mix.cpp).  The activations / weights of one layer are < 4 GB.
    unsigned a_off[4];
#pragma unroll
    for (int q = 0; q < 4; q++) a_off[q] = 4u * ((unsigned)(arow + 8 * q) * (unsigned)(dense ? B : L.npos * B) + (unsigned)(b0 + 4 * af4));
    const unsigned b_off = 4u * ((unsigned)frow * (unsigned)(dense ? L.N : khw * L.N) + (unsigned)(4 * bf4));
    auto gld_s = [](unsigned off, const float* base) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(off), "s"(base) : "memory"); return v; };
    // cursor of the staging loads: gload is called for tiles 0, 1, 2, ... in order (past the last tile it stays there); the tap of the NEXT tile is read from the
    // table one call ahead, so no LDS round trip or division sits in front of a load issue
    const int cot = dense ? 1 : L.N / 32;
    int g_kt = 0, g_si = 0, g_k = 0, g_ci = 0, g_ti = 0;
    int g_tp = dense ? 0 : taps[0];
    auto gload = [&](Stage& r) {
        const GDxSrc& sr = A.src[g_si];
        size_t sa, sb;
        if (dense) { const int nb = s * kc + g_k * 32; sa = (size_t)nb * B; sb = (size_t)nb; }
        else {
            const int tp = __builtin_amdgcn_readfirstlane(g_tp); const int tap = tp >> 16, pos = tp & 0xffff, cob = g_ci * 32;
            sa = ((size_t)cob * L.npos + pos) * B; sb = (size_t)tap * L.N + cob;
        }
        const float* pa = sr.dpre + sa; const float* pb = sr.W + sb;
#pragma unroll
        for (int q = 0; q < 4; q++) r.a[q] = gld_s(a_off[q], pa);
        r.b = gld_s(b_off, pb);
        if (g_kt + 1 < total) {
            g_kt++; g_k++; g_ci++;
            if (g_ci == cot) { g_ci = 0; g_ti++; }
            if (g_k == nkt) { g_k = 0; g_ci = 0; g_ti = 0; g_si = 1; }
            if (!dense) g_tp = taps[g_ti];
        }
    };
    auto lstore = [&](int buf, const Stage& r) {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(As + (buf * 32 + arow + 8 * q) * X_SAW + 4 * af4) = r.a[q];
        lds_st4(Bs + (buf * 32 + brow) * X_SB + 4 * bf4, r.b);
    };
#define STAGE_WAITW(N, r) asm volatile("s_waitcnt vmcnt(%5)" : "+v"(r.a[0]), "+v"(r.a[1]), "+v"(r.a[2]), "+v"(r.a[3]), "+v"(r.b) : "n"(N) : "memory")
    f32x4 acc[2][2][2];                               // [source][feature tile][sample tile]
#pragma unroll
    for (int i = 0; i < 8; i++) (&acc[0][0][0])[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // SI = the source's accumulator set, a COMPILE-TIME choice (r04: as a run-time select per MFMA it compiled to a branch and two accumulator moves per MFMA)
    auto compute = [&](int buf, auto SI) {
        constexpr int si = decltype(SI)::value;
        const float* Ab = As + buf * 32 * X_SAW + 32 * wave + l15;
        const float* Bb = Bs + (buf * 32 + l15) * X_SB + kq;
        float af[2][8], bf[2][8];
#pragma unroll
        for (int st = 0; st < 8; st++) {
            af[0][st] = Ab[(4 * st + kq) * X_SAW]; af[1][st] = Ab[(4 * st + kq) * X_SAW + 16];
            bf[0][st] = Bb[4 * st]; bf[1][st] = Bb[16 * X_SB + 4 * st];
        }
#pragma unroll
        for (int st = 0; st < 8; st++)
#pragma unroll
            for (int ft = 0; ft < 2; ft++)
#pragma unroll
                for (int ml = 0; ml < 2; ml++) acc[si][ft][ml] = MFMA(af[ml][st], bf[ft][st], acc[si][ft][ml]);
    };
    if (total > 0) {
        Stage r0, r1;
        gload(r0); STAGE_WAITW(0, r0); lstore(0, r0); __syncthreads();
        gload(r0);
        // two K tiles per round; the rounds of source 0, the round that straddles the sources when nkt is odd (or holds the last tile alone), the rounds of source 1
        auto round = [&](int kt, auto S0, auto S1) {
            gload(r1);
            compute(0, S0);
            STAGE_WAITW(5, r0); lstore(1, r0);
            __syncthreads();
            gload(r0);
            if (kt + 1 < total) compute(1, S1);
            STAGE_WAITW(5, r1); lstore(0, r1);
            __syncthreads();
        };
        std::integral_constant<int, 0> s0; std::integral_constant<int, 1> s1;
        int kt = 0;
        for (; kt + 1 < nkt; kt += 2) round(kt, s0, s0);
        if (kt < nkt) { round(kt, s0, s1); kt += 2; }
        for (; kt < total; kt += 2) round(kt, s1, s1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#undef STAGE_WAITW
    const size_t per_s = (size_t)L.in_feat * B;
#pragma unroll
    for (int ft = 0; ft < 2; ft++) {
        const int fl = f0 + 16 * ft + l15;
        if (fl >= nfeat) continue;
        const size_t feat = dense ? (size_t)fl : (size_t)fl * L.ih * L.iw + ip;
#pragma unroll
        for (int ml = 0; ml < 2; ml++) {
            const int bcol = b0 + 32 * wave + 16 * ml + 4 * kq;
            f32x4 v = acc[0][ft][ml];
            if (A.nsrc > 1) { const f32x4 o = acc[1][ft][ml]; v.x = v.x + o.x; v.y = v.y + o.y; v.z = v.z + o.z; v.w = v.w + o.w; }
            if (S == 1 && A.ysrc) {
                const f32x4 y = *reinterpret_cast<const f32x4*>(A.ysrc + feat * A.ldy + bcol);
                dact_v4(v, y, A.act_src);
            }
            st_out4(reinterpret_cast<f32x4*>(A.out + (size_t)s * per_s + feat * B + bcol), v, L.opt & DQN_LOPT_ST_WT);
        }
    }
}
// most non-empty tap chunks any input position of a conv layer can have: per stride-parity class (ky = iy mod sh, kx = ix mod sw) the distinct
// chunk ids among its taps (edge positions see subsets)
static int conv_max_chunks(const LayerDev& L) {
    const int tc = DQN_CONV_TAP_CHUNK(L); int best = 0;
    for (int py = 0; py < L.sh; py++) for (int px = 0; px < L.sw; px++) {
        int last = -1, n = 0;
        for (int ky = py; ky < L.kh; ky += L.sh) for (int kx = px; kx < L.kw; kx += L.sw) { const int c = (ky * L.kw + kx) / tc; if (c != last) { n++; last = c; } }
        if (n > best) best = n;
    }
    return best;
}
// dX body of a launch: 2 = 32 features x 128 samples per workgroup (large batches; dense plan chunks go through slabs, conv taps unchunked),
// 3 = 32 x 32 tiles with one wave per (source, chunk) unit (dx_units_body; chunks combined in the workgroup), -1 = not covered by the LDS kernels
static int dx_mode(const LayerDev& L, int nsrc, int B, int ldy) {
    const bool no_wide = (L.opt & DQN_LOPT_NO_DX_WIDE) != 0;
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int S = dense ? dqn_nchunks(L.N, L.dx_kc) : 1, kc = dqn_chunk_len(L.N, L.dx_kc);
    if (B % 32 || ldy % 4 || L.N % 32 || (S > 1 && kc % 32) || L.w_off % 4) return -1;
    if (!dense && (L.cin % (16 * U_FT) || L.kh * L.kw > 64 || L.npos > 65535)) return -1;
    const int nch = dense ? S : conv_max_chunks(L);
    if (B % 128 == 0 && !no_wide && (dense || (nch <= 1 && L.cin % 32 == 0)) && (nsrc == 1 || S == 1)) return 2;
    if (nsrc * nch <= U_MAX) return 3;
    return -1;
}
static int dx_cols(int mode) { return mode == 2 ? 128 : 32; }
static size_t dx_lds_bytes(int mode) {
    if (mode == 2) return (size_t)(2 * 32 * X_SAW + 2 * 32 * X_SB) * 4 + 256 * 4;
    return dx_units_lds_bytes();
}

// WIDE: the 128-sample dX body (large batches) -- a kernel of its own: compiled beside the small-batch body it set the register budget of both
// (154 + 36 VGPRs: two waves per SIMD) and halved the workgroups in flight of every B = 32 backward launch
template <bool WIDE>
__global__ __launch_bounds__(256) void k_dx_lds(LayerDev L, GDxArgs A, int B, int S, int kc, int gx, GemmTail tail) {
    karg_warm<sizeof(LayerDev) + sizeof(GDxArgs) + 32 + sizeof(GemmTail)>();
    GEMM_TAIL_PROLOGUE(tail, bid, main_blocks)
    int bq, br; fdiv_qr(bid, fdiv_of(gx), bq, br);
    if constexpr (WIDE) dx_lds_body_wide(L, A, B, S, kc, br, gx, bq);
    else dx_units_body<U_FT>(L, A, B, S, kc, br, gx, bq);
}
// dW and dX of one layer are independent given dpre: ONE launch runs both (blocks [0, dw_blocks) do dW, the rest dX), which
// saves a dispatch and lets the latency-bound dX workgroups share the machine with the dW ones.
template <int NT, bool WIDE>
__global__ __launch_bounds__(256, WIDE ? 1 : 4)      // small batches: <= 128 registers, i.e. FOUR workgroups per CU (r06: at 135 the 800 dX workgroups of conv2's pair ran on 768 slots -- a second round for 32 of them)
void k_dwdx_lds(LayerDev Lw, GDwProbs pr, int nprob, int ldx, int B, int Sw, int kcw, int dw_blocks, int dw_nsplit,
                                                  LayerDev Lx, GDxArgs A, int Sx, int kcx, int dx_gx, GemmTail tail) {
    // the few, long-latency dX workgroups are dispatched FIRST so that they run for the whole kernel while the many short dW
    // workgroups fill the remaining CUs (dispatch order is blockIdx order); a tail of small VALU tasks comes last
    // dispatch order: [priority block][dX workgroups][tail: VALU tasks, Adam job][dW workgroups]: the long-latency dX chains start first, the
    // bandwidth-bound tail streams beside them, the many short dW workgroups fill the slots as they free up
    karg_warm<2 * sizeof(LayerDev) + sizeof(GDwProbs) + sizeof(GDxArgs) + sizeof(GemmTail) + 68>();
    KTRACE_BEGIN()      // record: {grid, block, entry, role (0 prio, 1 dX, 2 tail, 3 dW), -, -, -, exit}  (tools/ktrace_bwd.py)
    const int pre_ = (tail.has_adam && tail.adam.prio.n > 0) ? 1 : 0;
    const int probe = KTRACE_PROBE();
    if (pre_ && blockIdx.x == 0) { extern __shared__ float lds[]; if (!(probe & 2)) prio_block_run(tail.adam.prio, tail.adam.state, reinterpret_cast<long long*>(lds), tail.lds_bytes, KTRACE_REC()); KTRACE_SET(3, 0); KTRACE(7); KTRACE_END(); return; }
    const int bid = (int)blockIdx.x - pre_, ntail = (int)gemm_tail_blocks(tail) - pre_;
    const int dx_blocks = (int)gridDim.x - pre_ - ntail - dw_blocks;
    if (bid < dx_blocks) {
        if (probe & 8) {}      // timing probe: dX workgroups return at once
        else if constexpr (WIDE) { int bq, br; fdiv_qr(bid, fdiv_of(dx_gx), bq, br); dx_lds_body_wide(Lx, A, B, Sx, kcx, br, dx_gx, bq); }
        else { int bq, br; fdiv_qr(bid, fdiv_of(dx_gx), bq, br); dx_units_body<U_FT>(Lx, A, B, Sx, kcx, br, dx_gx, bq, KTRACE_REC()); }
        KTRACE_SET(3, 1);
    }
    else if (bid < dx_blocks + ntail) { gemm_tail_run(tail, (unsigned)(bid - dx_blocks)); KTRACE_SET(3, 2); }
    else { if (!(probe & 1)) dw_section<NT, false>(Lw, pr, nprob, ldx, B, Sw, kcw, bid - dx_blocks - ntail, dw_blocks, dw_nsplit, DwStride{Lw.npos * B, 0, 0}, probe); KTRACE_SET(3, 3); }
    KTRACE(7); KTRACE_END();
}
bool gemm_dx_eligible(const LayerDev& L, int B, int ldy, int nsrc) { return dx_mode(L, nsrc, B, ldy) >= 0; }
// true: the plan chunks of this dX are contracted and combined INSIDE the launch (no partial slabs, no reduce launch)
bool gemm_dx_internal_chunks(const LayerDev& L, int B, int ldy, int nsrc) { return dx_mode(L, nsrc, B, ldy) == 3; }
// nsrc == 2: the two dueling streams (identical geometry), out = dact(dX_src0 + dX_src1)
void launch_gemm_dx(hipStream_t st, const LayerDev& L, int nsrc, const float* const* W, const float* const* dpre, int B, float* out,
                    const float* ysrc, int ldy, int act_src, GemmTail tail) {
    const bool dense = L.kind == DQN_LAYER_DENSE;
    const int S = dense ? dqn_nchunks(L.N, L.dx_kc) : 1, kc = dqn_chunk_len(L.N, L.dx_kc);
    GDxArgs a; a.nsrc = nsrc; a.out = out; a.ysrc = ysrc; a.ldy = ldy; a.act_src = act_src;
    for (int i = 0; i < 2; i++) { const int j = i < nsrc ? i : 0; a.src[i].W = W[j]; a.src[i].dpre = dpre[j]; }
    const int pj = dx_mode(L, nsrc, B, ldy);
    const int fw = pj == 3 ? 16 * U_FT : 32;
    const int gx = dense ? ((L.K + fw - 1) / fw) * (pj == 2 ? S : 1) : (L.cin / fw) * L.ih * L.iw;
    tail.lds_bytes = (unsigned)dx_lds_bytes(pj); tail.probe = HOST_PROBE();
    if (pj == 2) hipLaunchKernelGGL((k_dx_lds<true>), dim3(gx * (B / dx_cols(pj)) + gemm_tail_blocks(tail)), dim3(256), dx_lds_bytes(pj), st, L, a, B, S, kc, gx, tail);
    else hipLaunchKernelGGL((k_dx_lds<false>), dim3(gx * (B / dx_cols(pj)) + gemm_tail_blocks(tail)), dim3(256), dx_lds_bytes(pj), st, L, a, B, S, kc, gx, tail);
}

void launch_gemm_dwdx(hipStream_t st, const LayerDev& Lw, int nprob, const float* const* X, int ldx, const float* const* dpre_w, int B, float* const* out_w,
                      const LayerDev& Lx, int nsrc, const float* const* W, const float* const* dpre_x, float* out_x, const float* ysrc, int ldy, int act_src, GemmTail tail) {
    const int KK = Lw.npos * B, Sw = dqn_nchunks(KK, Lw.dw_kc), kcw = dqn_chunk_len(KK, Lw.dw_kc);
    GDwProbs pr;
    for (int i = 0; i < 2; i++) { const int j = i < nprob ? i : 0; pr.p[i].X = X[j]; pr.p[i].dpre = dpre_w[j]; pr.p[i].out = out_w[j]; }
    const int NT = Lw.N % 64 == 0 ? 4 : (Lw.N % 32 == 0 ? 2 : 1);
    const int dw_units = ((Lw.K + 63) / 64) * (Lw.N / (16 * NT)) * Sw * nprob, dw_nsplit = dw_split_units(Lw, NT, dw_units, B), dw_blocks = dw_units + dw_nsplit;
    const bool dense = Lx.kind == DQN_LAYER_DENSE;
    const int Sx = dense ? dqn_nchunks(Lx.N, Lx.dx_kc) : 1, kcx = dqn_chunk_len(Lx.N, Lx.dx_kc);
    GDxArgs a; a.nsrc = nsrc; a.out = out_x; a.ysrc = ysrc; a.ldy = ldy; a.act_src = act_src;
    for (int i = 0; i < 2; i++) { const int j = i < nsrc ? i : 0; a.src[i].W = W[j]; a.src[i].dpre = dpre_x[j]; }
    const int pj = dx_mode(Lx, nsrc, B, ldy);
    const int fw = pj == 3 ? 16 * U_FT : 32;
    const int gx = dense ? ((Lx.K + fw - 1) / fw) * (pj == 2 ? Sx : 1) : (Lx.cin / fw) * Lx.ih * Lx.iw;
    const size_t lds_w = (size_t)(2 * 64 * W_ST + 2 * 16 * NT * W_ST) * 4, lds_x = dx_lds_bytes(pj);
    size_t lds = lds_w > lds_x ? lds_w : lds_x;
    // the dense pair at B >= 512 runs better with TWO workgroups per CU than with the three its 46.6 KB allow (measured r04, config 5, same box, alternating: 81.7 -> 76.8 us;
    // the conv launches lose 8-11 % the same way and keep three): the request is raised past a third of the CU's LDS
    // (r06, with the dW section's half units in place: three per CU measure the same -- 74.5 vs 74.5-75.3 us, profiles/r06_l_fc3wg_ab.txt -- the request stays)
    if (pj == 2 && dense && B >= 512 && lds < (size_t)(160 * 1024 / 3 + 1024)) lds = (size_t)(160 * 1024 / 3 + 1024);
    tail.lds_bytes = (unsigned)lds; tail.probe = HOST_PROBE();
    const int grid = dw_blocks + gx * (B / dx_cols(pj)) + (int)gemm_tail_blocks(tail);
#define DWDX_LAUNCH(NTv, Wv) hipLaunchKernelGGL((k_dwdx_lds<NTv, Wv>), dim3(grid), dim3(256), lds, st, Lw, pr, nprob, ldx, B, Sw, kcw, dw_blocks, dw_nsplit, Lx, a, Sx, kcx, gx, tail)
    if (pj == 2) { if (NT == 4) DWDX_LAUNCH(4, true); else if (NT == 2) DWDX_LAUNCH(2, true); else DWDX_LAUNCH(1, true); }
    else { if (NT == 4) DWDX_LAUNCH(4, false); else if (NT == 2) DWDX_LAUNCH(2, false); else DWDX_LAUNCH(1, false); }
#undef DWDX_LAUNCH
}
