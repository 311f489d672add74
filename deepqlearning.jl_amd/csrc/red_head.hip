// red_head.hip -- small batches: the split-K reduce of the last hidden (dense) layer AND the head level of batch_train! in ONE launch that fills the chip
// (r05; replaces k_reduce_multi + k_head_td where it applies: 5.1 + 12.9 us on 384 + 32 workgroups at config 2).
//   reference lines: Q = val .+ adv .- mean(adv)            src/dueling.jl:10
//                    best = argmax(Qonline(sp)[:, b]), y = r + (1 - done) * gamma * Qtarget(sp)[best], td = Q(s)[a] - y, loss = sum(huber(w .* td)) / B
//                                                             src/solver.jl:209-224, src/helpers.jl:14-19
// Work decomposition.  The head's input is the hidden layer's output h[k][column], K = 512 rows at config 2, produced as S split-K slabs by the forward launch.  A head
// output is a sum over plan chunks of 32 k's, each chunk ONE k-ascending fma chain from +0 -- so a chunk of 32 hidden rows is the unit that can
// move to another workgroup without changing a bit.  Workgroup (g, stream, c):
//   g       a group of FOUR batch columns b = 4g .. 4g+3 (their s and sp columns of the online net, their sp columns of the target net: 12 columns)
//   stream  advantage / plain Q head (0) or value head (1): each has its own hidden layer
//   c       plan chunk: hidden rows 32c .. 32c+31
// = (B/4) x 2 x (K/32) = 256 workgroups at config 2 -- one per CU, each pulling 10.5 KB of slabs as 16-byte pieces (the 32 workgroups of k_head_td pulled 86 KB each
// across XCDs, which is why folding the reduce into THAT kernel lost: DESIGN.md 6.4 / 6.10).  It
//   A. sums its slab pieces in ascending slab order, + bias, activation (== k_reduce_multi, mode 0) -> 32 x 12 hidden activations; the online s columns go to the
//      activation array the backward pass reads (head dW tail tasks), WRITE-THROUGH (sc1);
//   B. contracts them with its 32 rows of the head weights: the chunk sums of every (slot, column, output) of its stream -> `partials`, write-through;
//   C. drains its stores, takes a ticket on its group's counter (MI355X guide, Guideline 16 R1 / "splitk-seam": sc1 payload -> every storing wave vmcnt(0) -> barrier ->
//      ONE relaxed agent-scope atomic).  31 of the 32 workgroups of a group are done here.
//   D. the LAST arriver of a group reads the group's 960 chunk sums and its 4 columns of both hidden layers with sc1 loads (the producers stored sc1: no acquire fence),
//      adds the chunk sums in ascending order + bias + activation, and runs k_head_td's per-column arithmetic for its four columns: dueling combine, first-max argmax,
//      Bellman target, TD, Huber term, dL/dQ, the heads' dpre -- and the heads' dX (acc = +0; n ascending: fma(dpre[n], W[k][n], acc); act' of the hidden layer) for all K
//      rows of both streams.  It also re-arms the ticket.
// No workgroup ever waits for another one (no spin, no residency assumption): correctness does not depend on dispatch order or placement.
// Every value follows the canonical order of the kernels this replaces, so the step stays bit-identical to them and to the twin.
#include <algorithm>
#include "common.h"

typedef float f32x4r __attribute__((ext_vector_type(4)));

// Pointers that come out of the argument record (device memory) carry no address space: hipcc then emits FLAT loads, which count on lgkmcnt as well as vmcnt -- and the next
// scalar argument fetch (s_load + s_waitcnt lgkmcnt(0)) drains every flat load in flight: the rounds of phase A would serialise.  Everything here is global memory.
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const GLOBAL_AS T* gptr(const T* p) { return (const GLOBAL_AS T*)p; }
template <class T> __device__ __forceinline__ GLOBAL_AS T* gptr(T* p) { return (GLOBAL_AS T*)p; }
// 16-byte write-through store / L1-bypassing load (inline asm: hipcc neither counts nor waits for them -- the waits below do, cdna_hip_programming.md 5.7)
__device__ __forceinline__ void st_sc1_x4(float* p, const f32x4r& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ f32x4r ld_sc1_x4(const float* p) { f32x4r v; asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(gptr(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(gptr(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// activations: TRANS == false instantiations know only identity / relu (selects, no code) -- the Float64 tanh / sigmoid bodies of act_f / dact_f inlined at every use made
// the kernel 5600 instructions of branchy code that each last arriver walks ONCE, cold in the instruction cache (r05 stamps: 3 us for six dX items per lane)
template <bool TRANS> __device__ __forceinline__ float rh_act(float y, int a) { if constexpr (TRANS) return act_f(y, a); else return a == DQN_ACT_RELU ? (y > 0.0f ? y : 0.0f) : y; }
template <bool TRANS> __device__ __forceinline__ float rh_dact(float dy, float y, int a) { if constexpr (TRANS) return dact_f(dy, y, a); else return a == DQN_ACT_RELU ? (y > 0.0f ? dy : 0.0f) : dy; }
// One lane per column: its three Q columns (Q = (val .+ adv) .- mean(adv), src/dueling.jl:10; the parity copies go out from here), then k_td's / k_head_td's TD arithmetic --
// no barrier and no LDS round trip between the two (r05 stamps: Q 0.7 + TD 1.6 us as two barrier-separated phases).  hv: finished head outputs [3][4][NO], bm: [4][4] batch
// scalars, dq: [4][NO] out.  Shared by k_red_head's last arrivers and k_head_cols4.
template <bool TRANS> __device__ __forceinline__ void rh_td_column(const RedHeadArgs& A, const float* hv, const float* bm, float* dq, const int j, const int b) {
    const int B = A.B, nA = A.nA, NO = A.NO, nstream = A.nstream, double_q = A.double_q;
    float q3[3][8], h0r[8], h0v = 0.0f;      // [slot][a]  (nA <= 8); raw head outputs of slot 0 (the heads' act' below)
#pragma unroll
    for (int sl_ = 0; sl_ < 3; sl_++) {
        const float* ar = hv + (sl_ * 4 + j) * NO;
        float araw[8];
#pragma unroll
        for (int a = 0; a < 8; a++) araw[a] = ar[a < nA ? a : nA - 1];      // (clamped, never predicated: a runtime "read or not" per element serialises the LDS round trips)
        if (sl_ == 0) {
#pragma unroll
            for (int a = 0; a < 8; a++) h0r[a] = araw[a];
        }
        if (nstream == 1) {
#pragma unroll
            for (int a = 0; a < 8; a++) q3[sl_][a] = araw[a];
        } else {
            const float vv = ar[NO - 1];
            if (sl_ == 0) h0v = vv;
            float sum = araw[0];
#pragma unroll
            for (int a = 1; a < 8; a++) if (a < nA) sum = sum + araw[a];
            const float mean = sum / (float)nA;
#pragma unroll
            for (int a = 0; a < 8; a++) q3[sl_][a] = (vv + araw[a]) - mean;
        }
    }
#pragma unroll
    for (int a = 0; a < 8; a++) if (a < nA) {
        A.q_on_s[(size_t)b * nA + a] = q3[0][a]; A.q_tg_sp[(size_t)b * nA + a] = q3[2][a];
        A.q_on_sp[(size_t)b * nA + a] = double_q ? q3[1][a] : q3[2][a];
    }
    const float invB = 1.0f / (float)B;
    const int act_i = __float_as_int(bm[4 * j]); const float rew = bm[4 * j + 1], dn = bm[4 * j + 2], w = bm[4 * j + 3];
    A.w_is[b] = w;
    int best = 0; float bq = double_q ? q3[1][0] : q3[2][0];      // argmax over the online net's Q(sp) (double-Q) or the target net's: first max (Julia argmax)
#pragma unroll
    for (int a = 1; a < 8; a++) { const float q = double_q ? q3[1][a] : q3[2][a]; if (a < nA && q > bq) { bq = q; best = a; } }
    float qsp = q3[2][0], qsa = q3[0][0];
#pragma unroll
    for (int a = 1; a < 8; a++) { if (a == best) qsp = q3[2][a]; if (a == act_i) qsa = q3[0][a]; }
    A.best[b] = best;
    const float t1 = 1.0f - dn; const float t2 = t1 * A.gamma; const float t3 = t2 * qsp; const float y = rew + t3;
    A.ytarget[b] = y;
    const float td = qsa - y; A.td[b] = td;
    const float x = w * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
    A.hl[b] = (0.5f * qd) * qd + lin;
    const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
    const float gq = (invB * cl) * w;
    const int act_a = A.st[0].hact;
    if (nstream > 1) {
        const float dv = rh_dact<TRANS>(gq, h0v, A.st[1].hact); dq[j * NO + nA] = dv; A.st[1].dpre[b] = dv;
        const float gm = gq / (float)nA;
#pragma unroll
        for (int a = 0; a < 8; a++) if (a < nA) { const float d = rh_dact<TRANS>((a == act_i ? gq : 0.0f) - gm, h0r[a], act_a); dq[j * NO + a] = d; A.st[0].dpre[(size_t)a * B + b] = d; }
    } else {
#pragma unroll
        for (int a = 0; a < 8; a++) if (a < nA) { const float d = rh_dact<TRANS>(a == act_i ? gq : 0.0f, h0r[a], act_a); dq[j * NO + a] = d; A.st[0].dpre[(size_t)a * B + b] = d; }
    }
}
// dX of the heads for one hidden row i (rows of stream 0 then stream 1) and the group's four columns: acc = +0; n ascending: acc = fma(dpre[n], W[k][n], acc); then act' of
// the hidden layer (its activation y).  dq of the four columns lives in registers; a row of the advantage head's weights is one 16-byte LDS read when it has four outputs
struct RhDx { int K, B, nA, N0s, N1s, pa0, pa1, won1, wpad; float *ds0, *ds1; };      // wpad: floats of padding after every 32 rows of the LDS weight blocks
template <bool TRANS> __device__ __forceinline__ void rh_dx_row(const RhDx& X, const float* Won, const float (&dqr)[4][9], const int i, const f32x4r& y, const int g) {
    const int K = X.K, B = X.B, nA = X.nA, N0s = X.N0s, N1s = X.N1s, pa0 = X.pa0, pa1 = X.pa1, won1 = X.won1; float *ds0 = X.ds0, *ds1 = X.ds1;
    const int st_ = i >= K ? 1 : 0, k = i - st_ * K, Ns = st_ ? N1s : N0s;
    const float* w = Won + (st_ ? won1 : 0) + k * Ns + (k >> 5) * X.wpad;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    if (st_) {      // the value head: one output, dq index nA
        const float wv = w[0];
        const float d0 = dqr[0][8], d1 = dqr[1][8], d2 = dqr[2][8], d3 = dqr[3][8];      // slot 8 = the value head's dq (index nA of the column's NO values), put there by the caller:
                                                                                           // picked out of the register array by a RUN-TIME index it sent the whole array through scratch
        a0 = fmaf(d0, wv, a0); a1 = fmaf(d1, wv, a1); a2 = fmaf(d2, wv, a2); a3 = fmaf(d3, wv, a3);
    } else if (Ns == 4) {
        const f32x4r w4 = *reinterpret_cast<const f32x4r*>(w);
        const float wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int n = 0; n < 4; n++) { a0 = fmaf(dqr[0][n], wv[n], a0); a1 = fmaf(dqr[1][n], wv[n], a1); a2 = fmaf(dqr[2][n], wv[n], a2); a3 = fmaf(dqr[3][n], wv[n], a3); }
    } else {
#pragma unroll
        for (int n = 0; n < 8; n++) if (n < Ns) { const float wv = w[n]; a0 = fmaf(dqr[0][n], wv, a0); a1 = fmaf(dqr[1][n], wv, a1); a2 = fmaf(dqr[2][n], wv, a2); a3 = fmaf(dqr[3][n], wv, a3); }
    }
    const int as = st_ ? pa1 : pa0;
    f32x4r d = y;
    d.x = rh_dact<TRANS>(a0, d.x, as); d.y = rh_dact<TRANS>(a1, d.y, as); d.z = rh_dact<TRANS>(a2, d.z, as); d.w = rh_dact<TRANS>(a3, d.w, as);
    *gptr(reinterpret_cast<f32x4r*>((st_ ? ds1 : ds0) + (size_t)k * B + 4 * g)) = d;
}
template <int SMAX, bool TRANS>      // SMAX: compile-time bound on the slab count (8 or 16): the slab pieces live in registers, loaded UNCONDITIONALLY (clamped) -- a runtime "load or not" per
                         // element makes hipcc branch around every load and wait for each (cdna_hip_programming.md 5, trap (c))
__global__ __launch_bounds__(256) void k_red_head(const RedHeadArgs A, int bump_sample_ctr, int take_pre) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    // the argument record travels BY VALUE in the kernarg segment and every 64-byte line of it is touched up front (one round of independent scalar loads): behind a
    // pointer into device memory the first slab load waited for two dependent scalar round trips (phase A of this kernel: 5.0 us by its stamps, r05)
    {
        typedef const uint32_t __attribute__((address_space(4))) karg_u32;
        karg_u32* kp = (karg_u32*)__builtin_amdgcn_kernarg_segment_ptr(); uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < (int)((sizeof(RedHeadArgs) + 8 + 63) / 64); i++) x ^= kp[16 * i];
        asm volatile("" ::"s"(x));
    }
    const int tid = threadIdx.x;
    const int B = A.B, nA = A.nA, K = A.K, S = A.S, ncon = A.ncon, NC = K >> 5, G = B >> 2, nstream = A.nstream, NO = A.NO, double_q = A.double_q;
    const int g = (int)blockIdx.x % G, r_ = (int)blockIdx.x / G, stream = r_ / NC, c = r_ - stream * NC;
    const RedHeadStream& T = A.st[stream];
    const int N = T.N, o0 = stream == 0 ? 0 : nA;      // this stream's outputs are o0 .. o0 + N - 1 of the NO head outputs (advantage / plain Q first, the value last)
    // ---- LDS carve-up (floats; every offset a multiple of 4)
    const int n0q = (K * A.st[0].N) >> 2, n1q = nstream > 1 ? (K * A.st[1].N) >> 2 : 0;      // float4s of the two online head weight blocks (K % 32 == 0: whole float4s)
    float* act = hs;                                   // [3 slots][32 rows][4 columns]   hidden activations of this chunk (slot 0 = online s, 1 = online sp, 2 = target sp)
    float* Wt = act + 384;                             // [32][N]                          TARGET head weights of this chunk
    float* Won = Wt + 32 * N;                          // stream 0: [K][N0], then stream 1: [K][N1]   ONLINE head weights of both streams (chunk rows for B., all rows for D.)
    const int won1 = 4 * n0q;
    float* hv = Won + 4 * (n0q + n1q);                 // [3][4][NO]   finished head outputs (last arriver)
    float* qs = hv + ((12 * NO + 3) & ~3);             // [3][4][nA]   Q columns
    float* dq = qs + ((12 * nA + 3) & ~3);             // [4][NO]      dL/d(pre-activation) of the heads for the group's four columns
    float* bm = dq + ((4 * NO + 3) & ~3);              // [4][4]       batch scalars: a (bits), r, done, w of the group's columns
    int* flag = reinterpret_cast<int*>(bm + 16);       // [1]          "this workgroup is the group's last arriver"
    f32x4r* ys = reinterpret_cast<f32x4r*>(bm + 20);   // [K * nstream] the group's four columns of both hidden layers (last arriver: act' operand of the heads' dX)
    // TIMING PROBE (DQN_DRQN_STAMPS at create; null in production): 100 MHz s_memrealtime stamps, common to all XCDs.  [0..5] = workgroup 0's phases, [8..15] = group 0's last
    // arriver, [16] = latest exit of any last arriver, [17] = earliest entry of any workgroup (both as atomics)
    unsigned long long* const stamps = A.stamps;
    const unsigned long long t_in = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define RH_STAMP(i) do { if (stamps && blockIdx.x == 0 && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    if (stamps && tid == 0) { atomicMin(stamps + 17, t_in); if (blockIdx.x == 0) stamps[0] = t_in; }
    // ---- A. every load whose address is known at entry goes out in ONE round, into registers: the online head weights of both streams (<= 4 float4 per thread), the chunk's
    // target head weights, the batch scalars, then the slab pieces
    f32x4r wq[4];
    {
        const f32x4r* w0p = reinterpret_cast<const f32x4r*>(A.st[0].W[0]); const f32x4r* w1p = reinterpret_cast<const f32x4r*>(A.st[1].W[0]);      // (one stream: st[1] == st[0])
#pragma unroll
        for (int u = 0; u < 4; u++) {      // ONE load per slot: the source is selected as an address, never as a branch
            int i = tid + 256 * u; if (i >= n0q + n1q) i = n0q + n1q - 1;
            const f32x4r* src = i < n0q ? w0p + i : w1p + (i - n0q);
            wq[u] = *gptr(src);
        }
    }
    const float wt_r = *gptr(T.W[1] + (size_t)32 * c * N + (tid < 32 * N ? tid : 0));
    float bm_r;
    {      // get_batch scalars + IS weight of the group's columns (written by the gather launch): lane (j, field) of threads 128-143; every lane loads (clamped), one instruction
        const int j = (tid >> 2) & 3, fld = tid & 3, b = 4 * g + j;
        const float *s0 = reinterpret_cast<const float*>(A.bm_a), *s1 = A.bm_r, *s2 = A.bm_done, *s3 = A.bm_w;      // (the action index travels as its bit pattern)
        const float* src = fld == 0 ? s0 : (fld == 1 ? s1 : (fld == 2 ? s2 : s3));
        bm_r = *gptr(src + b);
    }
    if (tid == 0) flag[0] = 0;
    // workgroup 0's idle lanes 192-255 also carry the step's bookkeeping IN THE SAME ROUND of loads (as a block of its own in front of the round it cost workgroup 0 -- and
    // with it column group 0's ticket -- a full extra round trip): lanes 192.. publish the pre-gathered batch's indices (take_pre: the previous step's Adam launch gathered
    // this batch; read by the priority block and the parity API), lane 255 ticks the step counters (read by k_adam later in this step) and checks the pre-gather flag
    long long idx_v = 0; unsigned long long step_v = 0, sctr_v = 0; int pv_v = 2;
    const bool bk = blockIdx.x == 0 && tid >= 192;
    if (bk) {
        if (take_pre && tid - 192 < B) idx_v = *gptr(A.idx_pre + (tid - 192));      // (B > 64: the rest of the list follows in a loop below, off the critical path of group 0)
        if (tid == 255) { step_v = *gptr(&A.stt->step); sctr_v = *gptr(&A.stt->sample_ctr); if (take_pre) pv_v = *gptr(&A.stt->pre_valid); }
    }
    f32x4r sl[SMAX]; float pb = 0.0f;
    const int f = tid & 31, slot = tid >> 5;           // threads 0-95: hidden row f of the chunk, slot
    if (tid < 96) {      // ONE branch around the whole round; inside it every load is unconditional (slab index clamped)
        // (both nets' pointers are scalars selected per lane: indexing the record with a per-lane `net` would be a vector load of the pointer)
        const float *pt0 = T.part[0], *pt1 = T.part[1], *pbs0 = T.pbias[0], *pbs1 = T.pbias[1];
        const int net = slot == 2 ? 1 : 0, ncols = net ? B : ncon, col = 4 * g + ((slot == 1 && double_q) ? B : 0);      // (!double_q: ncon == B, slot 1 is discarded below -- it re-reads slot 0's piece instead of running B floats past the row)
        // (piece-major slabs: the 32 rows of this chunk and column quad are 512 contiguous bytes; else one 16-byte piece per 4 * ncols bytes)
        const float* p = (net ? pt1 : pt0) + (A.pm ? ((size_t)(col >> 2) * K + (size_t)(32 * c + f)) * 4 : (size_t)(32 * c + f) * ncols + col);
        const size_t per_s = (size_t)K * ncols;
#pragma unroll
        for (int s = 0; s < SMAX; s++) sl[s] = *gptr(reinterpret_cast<const f32x4r*>(p + (size_t)(s < S ? s : S - 1) * per_s));
        pb = *gptr((net ? pbs1 : pbs0) + 32 * c + f);
    }
    // the weights and scalars (requested first) go into LDS while the slab pieces are still in flight
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = tid + 256 * u; if (i < n0q + n1q) reinterpret_cast<f32x4r*>(Won)[i] = wq[u]; }
    if (tid < 32 * N) Wt[tid] = wt_r;
    if (tid >= 128 && tid < 144) bm[tid - 128] = bm_r;
    if (tid < 96) {      // ascending slab order, + bias, activation (k_reduce_multi mode 0)
        f32x4r tot = sl[0];
#pragma unroll
        for (int s = 1; s < SMAX; s++) if (s < S) { tot.x = tot.x + sl[s].x; tot.y = tot.y + sl[s].y; tot.z = tot.z + sl[s].z; tot.w = tot.w + sl[s].w; }
        if (S > 1) { tot.x = rh_act<TRANS>(tot.x + pb, T.pact); tot.y = rh_act<TRANS>(tot.y + pb, T.pact); tot.z = rh_act<TRANS>(tot.z + pb, T.pact); tot.w = rh_act<TRANS>(tot.w + pb, T.pact); }
        // (S == 1: `part` IS the hidden layer's finished activation -- an unsplit forward, large batches -- and the online s columns are already where the backward pass reads them)
        if (slot == 1 && !double_q) tot = (f32x4r){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4r*>(act + (slot * 32 + f) * 4) = tot;
        if (slot == 0 && S > 1) {
            st_sc1_x4(T.y_on + (size_t)(32 * c + f) * ncon + 4 * g, tot);      // the backward pass (head dW tasks, the hidden layers' act') reads it
            // r06: ... and a second, piece-major copy [group][stream][K][4] for the group's last arriver, whose round of loads then is 1 KB per instruction instead of 64 pieces ncon * 4 bytes apart
            if (A.ypm) st_sc1_x4(A.ypm + (((size_t)g * nstream + stream) * K + (size_t)(32 * c + f)) * 4, tot);
        }
    }
    if (bk) {
        if (take_pre && tid - 192 < B) *gptr(A.idx + (tid - 192)) = idx_v;
        if (take_pre) for (int i = tid - 192 + 64; i < B; i += 64) A.idx[i] = A.idx_pre[i];
        if (tid == 255) { *gptr(&A.stt->step) = step_v + 1; if (bump_sample_ctr) *gptr(&A.stt->sample_ctr) = sctr_v + 1; if (take_pre && pv_v != 2) *gptr(&A.stt->err) = 3; }
    }
    // LDS-only hand-over: s_waitcnt lgkmcnt(0) + s_barrier -- __syncthreads() also drains vmcnt, i.e. would wait here for the write-through store's acknowledgement
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RH_STAMP(1);
    // ---- B. chunk sums of this stream's head outputs: item (slot, column j, output n), one k-ascending chain of 32 from +0 each
    float* Pg = A.partials + (size_t)g * 12 * NO * NC;      // [slot][j][o][chunk]
    if (tid < 12 * N) {
        const int n = tid % N, sj = tid / N, j = sj & 3, sl_ = sj >> 2;
        float acc = 0.0f;
        if (!(sl_ == 1 && !double_q)) {
            const float* w = sl_ == 2 ? Wt + n : Won + (stream ? won1 : 0) + (size_t)32 * c * N + n;
            const float* x = act + sl_ * 128 + j;
#pragma unroll 8
            for (int k = 0; k < 32; k++) acc = fmaf(x[4 * k], w[k * N], acc);
        }
        st_sc1(Pg + ((size_t)(sl_ * 4 + j) * NO + o0 + n) * NC + c, acc);
    }
    // ---- C. publish: every storing wave drains its write-through stores, then ONE relaxed agent-scope ticket
    RH_STAMP(2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    RH_STAMP(3);
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(gptr(A.tickets + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(NC * nstream - 1)) { flag[0] = 1; __hip_atomic_store(gptr(A.tickets + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // everybody has arrived: re-armed for the next launch
    }
    __syncthreads();
    RH_STAMP(4);
    if (!flag[0]) return;
#define RH_STAMP_L(i) do { if (stamps && g == 0 && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    if (stamps && g == 0 && tid == 0) stamps[8] = t_in;
    RH_STAMP_L(9);
    // ---- D. the group's last arriver.  Round trip: the hidden activations of the four columns (thread = rows tid, tid + 256, ... of stream 0 then stream 1) and the chunk
    // sums of every head output (thread = (slot, j, o)), all with L1-bypassing loads (the producers stored write-through)
    const int nout = 12 * NO, KS = K * nstream;
    // (per-stream fields of the record are fetched with scalar loads and selected per lane, as in phase A)
    const float *y0p = A.st[0].y_on, *y1p = A.st[1].y_on; float *ds0 = A.st[0].dsrc, *ds1 = A.st[1].dsrc;
    const int N0s = A.st[0].N, N1s = A.st[1].N, pa0 = A.st[0].pact, pa1 = A.st[1].pact, ha0 = A.st[0].hact, ha1 = A.st[1].hact;
    const float *hb00 = A.st[0].hbias[0], *hb01 = A.st[0].hbias[1], *hb10 = A.st[1].hbias[0], *hb11 = A.st[1].hbias[1];
    // (the dX items are spread over waves 0-2 only, stride 192: wave 3 carries the per-column TD lanes, whose ~25 parity stores would otherwise stand between that wave's
    // dX items and their registers -- r05 stamps: 2.6 us "dX" on the TD wave, all of it waiting for store acknowledgements)
    f32x4r yv[6];
#pragma unroll
    for (int u = 0; u < 6; u++) { int i = (tid < 192 ? tid : 0) + 192 * u; if (i >= KS) i = KS - 1; const int st_ = i >= K ? 1 : 0, k = i - st_ * K;
                                  yv[u] = A.ypm ? ld_sc1_x4(A.ypm + ((size_t)g * KS + i) * 4) : ld_sc1_x4((st_ ? y1p : y0p) + (size_t)k * ncon + 4 * g); }
    {
        const int t2 = tid < nout ? tid : 0;
        const int o = t2 % NO, sj = t2 / NO, sl_ = sj >> 2;
        const int st_ = o >= nA ? 1 : 0, n = o - (st_ ? nA : 0), net = sl_ == 2 ? 1 : 0;
        const float* pp = Pg + (size_t)t2 * NC;
        float pv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) pv[q] = ld_sc1(pp + (q < NC ? q : NC - 1));      // (r06: four 16-byte requests instead of sixteen 4-byte ones measured the same: 1.56 vs 1.64 us for this round of loads)
        const float hb = *gptr((st_ ? (net ? hb11 : hb10) : (net ? hb01 : hb00)) + n);
        float tot = pv[0];
#pragma unroll
        for (int q = 1; q < 16; q++) if (q < NC) tot = tot + pv[q];      // chunk sums added in ascending order
        if (tid < nout) hv[tid] = (sl_ == 1 && !double_q) ? 0.0f : rh_act<TRANS>(tot + hb, st_ ? ha1 : ha0);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(yv[0]), "+v"(yv[1]), "+v"(yv[2]), "+v"(yv[3]), "+v"(yv[4]), "+v"(yv[5])::"memory");
    // r06: the y pieces go through LDS so that the dX loop below can stay ROLLED (everything after the ticket is code only the eight last arrivers ever execute, each on another CU, cold in
    // the instruction cache: the six unrolled, three-way specialised dX items were ~1400 instructions).  What the stamps charged to this phase -- 2.4-2.7 us for ~150 instructions of
    // arithmetic -- was something else, though: rh_dx_row picked the value head's dq out of the register array by a RUN-TIME index, the array went through scratch, and the scratch loads'
    // vmcnt waits made every row wait for the previous row's global store to be acknowledged.  With the index gone (slot 8 of dqr): 1.8 us, and no scratch in the kernel
    if (tid < 192) {
#pragma unroll
        for (int u = 0; u < 6; u++) { const int i = tid + 192 * u; if (i < KS) ys[i] = yv[u]; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    RH_STAMP_L(10); RH_STAMP_L(11);
    if (tid >= 192 && tid < 196) rh_td_column<TRANS>(A, hv, bm, dq, tid - 192, 4 * g + tid - 192);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS-only hand-over of dq (no drain of the parity stores above)
    RH_STAMP_L(12);
    // dX of the heads for the group's four columns: acc = +0; n ascending: acc = fma(dpre[n], W[k][n], acc); then act' of the hidden layer (its activation y).
    // dq of the four columns lives in registers; a row of the advantage head's weights is one 16-byte LDS read when it has four outputs
    const RhDx dxa = {K, B, nA, N0s, N1s, pa0, pa1, won1, 0, ds0, ds1};
    float dqr[4][9];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int o = 0; o < 9; o++) dqr[j][o] = dq[j * NO + (o == 8 ? (nA < NO ? nA : NO - 1) : (o < NO ? o : NO - 1))];      // (clamped: see above; slot 8 = the value head's, see rh_dx_row)
    if (tid < 192) {
#pragma unroll 1
        for (int i = tid; i < KS; i += 192) rh_dx_row<TRANS>(dxa, Won, dqr, i, ys[i], g);
    }
    RH_STAMP_L(13);
    if (stamps && tid == 0) atomicMax(stamps + 16, __builtin_amdgcn_s_memrealtime());
#undef RH_STAMP
#undef RH_STAMP_L
}
// =====================================================================================================================
// k_head_cols4 -- the head level at LARGE batches (the hidden layers' forwards are not split-K: S == 1, `part` is the finished activation).  One workgroup per group of
// four batch columns does everything k_red_head's 2 * K/32 workgroups + last arriver do for that group, with no hand-off at all: it pulls the group's 12 columns of both
// hidden layers (3 * K * nstream values per column group: 48 KB at config 5), both nets' head weights, and runs the chunk chains, the ascending chunk sums, the per-column
// TD arithmetic and the heads' dX -- whose output leaves as 16-byte stores.  Replaces k_head_td there (one workgroup per COLUMN: B x 20 KB of head weights and dX as 4-byte
// stores 2 KB apart: 19.2 us at B = 512).  Same chains, same order: bit-identical.
// The columns come out of the TRANSPOSED copy [column][K] the dense forward's epilogue writes (partT: 4 KB runs per column, coalesced); read as 16-byte pieces of the
// [K][columns] array itself -- one piece per 4 KB at B = 512 -- the same 48 KB took 8.8 us per workgroup (r05 stamps; 4.1 us at B = 128), which is the fallback when no
// transposed copy exists (a forward that did not go through the LDS-tiled GEMM).
// LDS layout: activations [slot][column j][stream][k], k contiguous, 4 floats of padding per 32 k and 12 / 8 between streams / rows: the chain items of a wave (chunk,
// slot, output) read their 16-byte pieces from different bank groups; the head weights likewise (4 floats after every 32 rows).
// =====================================================================================================================
template <bool TRANS>
__global__ __launch_bounds__(256) void k_head_cols4(const RedHeadArgs A, int bump_sample_ctr, int take_pre) {
    extern __shared__ __attribute__((aligned(16))) float hs[];
    {
        typedef const uint32_t __attribute__((address_space(4))) karg_u32;
        karg_u32* kp = (karg_u32*)__builtin_amdgcn_kernarg_segment_ptr(); uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < (int)((sizeof(RedHeadArgs) + 8 + 63) / 64); i++) x ^= kp[16 * i];
        asm volatile("" ::"s"(x));
    }
    const int tid = threadIdx.x;
    const int B = A.B, nA = A.nA, K = A.K, ncon = A.ncon, NC = K >> 5, nstream = A.nstream, NO = A.NO, double_q = A.double_q, KS = K * nstream;
    const int g = (int)blockIdx.x;
    // TIMING PROBE (DQN_DRQN_STAMPS at create; null in production): [0..5] = workgroup 0's phases, [16] = latest exit, [17] = earliest entry of any workgroup
    unsigned long long* const stamps = A.stamps;
#define C4_STAMP(i) do { if (stamps && blockIdx.x == 0 && tid == 0) stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
    if (stamps && tid == 0) { const unsigned long long t_in = __builtin_amdgcn_s_memrealtime(); atomicMin(stamps + 17, t_in); if (blockIdx.x == 0) stamps[0] = t_in; }
    const int N0s = A.st[0].N, N1s = A.st[1].N;
    const int n0q = (K * N0s) >> 2, n1q = nstream > 1 ? (K * N1s) >> 2 : 0;
    constexpr int WPAD = 4;
    const int won1 = K * N0s + NC * WPAD, wsz = won1 + (nstream > 1 ? K * N1s + NC * WPAD : 0);      // padded floats of stream 0's block / of both
    const int KP = NC * 36, SP = KP + 12, RS = (nstream > 1 ? SP + KP : KP) + 8;                     // padded floats per stream, offset of stream 1, row (slot, j) stride
    // ---- LDS carve-up (floats; every offset a multiple of 4)
    float* act = hs;                                   // [3 slots][4 columns] rows of RS floats
    float* Won = act + 12 * RS;                        // online head weights (padded): stream 0 [K][N0], stream 1 [K][N1]
    float* Wtg = Won + wsz;                            // target head weights, same layout
    float* P = Wtg + wsz;                              // [3][4][NO][NC] chunk sums
    float* hv = P + 12 * NO * NC;                      // [3][4][NO]
    float* dq = hv + ((12 * NO + 3) & ~3);             // [4][NO]
    float* bm = dq + ((4 * NO + 3) & ~3);              // [4][4]
    // ---- 1. every load goes out in one round: both nets' head weights (<= 4 + 4 float4 per thread), the batch scalars, the head biases, the 12 columns of the hidden layers
    f32x4r wq[4], wtq[4];
    {
        const f32x4r* w0p = reinterpret_cast<const f32x4r*>(A.st[0].W[0]); const f32x4r* w1p = reinterpret_cast<const f32x4r*>(A.st[1].W[0]);
        const f32x4r* t0p = reinterpret_cast<const f32x4r*>(A.st[0].W[1]); const f32x4r* t1p = reinterpret_cast<const f32x4r*>(A.st[1].W[1]);
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int i = tid + 256 * u; if (i >= n0q + n1q) i = n0q + n1q - 1;
            wq[u] = *gptr(i < n0q ? w0p + i : w1p + (i - n0q));
            wtq[u] = *gptr(i < n0q ? t0p + i : t1p + (i - n0q));
        }
    }
    float bm_r;
    {
        const int j = (tid >> 2) & 3, fld = tid & 3, b = 4 * g + j;
        const float *s0 = reinterpret_cast<const float*>(A.bm_a), *s1 = A.bm_r, *s2 = A.bm_done, *s3 = A.bm_w;
        bm_r = *gptr((fld == 0 ? s0 : (fld == 1 ? s1 : (fld == 2 ? s2 : s3))) + b);
    }
    const int nout = 12 * NO;
    float hb;
    {      // head bias of output item tid = (slot, j, o)
        const float *hb00 = A.st[0].hbias[0], *hb01 = A.st[0].hbias[1], *hb10 = A.st[1].hbias[0], *hb11 = A.st[1].hbias[1];
        const int t2 = tid < nout ? tid : 0, o = t2 % NO, sl_ = (t2 / NO) >> 2, st_ = o >= nA ? 1 : 0, n = o - (st_ ? nA : 0), net = sl_ == 2 ? 1 : 0;
        hb = *gptr((st_ ? (net ? hb11 : hb10) : (net ? hb01 : hb00)) + n);
    }
    long long idx_v = 0; unsigned long long step_v = 0, sctr_v = 0; int pv_v = 2;
    const bool bk = blockIdx.x == 0 && tid >= 192;      // the step's bookkeeping, as in k_red_head
    if (bk) {
        if (take_pre && tid - 192 < B) idx_v = *gptr(A.idx_pre + (tid - 192));
        if (tid == 255) { step_v = *gptr(&A.stt->step); sctr_v = *gptr(&A.stt->sample_ctr); if (take_pre) pv_v = *gptr(&A.stt->pre_valid); }
    }
    f32x4r yv[12];
    const float *q00 = A.st[0].partT[0], *q01 = A.st[0].partT[1], *q10 = A.st[1].partT[0], *q11 = A.st[1].partT[1];
    const bool viaT = q00 != nullptr;      // (uniform)
    const int qpr = KS >> 2;               // float4 per (slot, column) row
    if (viaT) {
        // quad e = tid + 256 u of the 12 rows (slot, j) x KS/4: four consecutive k of ONE column, coalesced along k
        const FDiv fq = fdiv_of(qpr);
#pragma unroll
        for (int u = 0; u < 12; u++) {
            int e = tid + 256 * u; if (e >= 12 * qpr) e = 12 * qpr - 1;
            int cs, i4; fdiv_qr(e, fq, cs, i4);
            const int sl_ = cs >> 2, j = cs & 3, i = 4 * i4, st_ = i >= K ? 1 : 0, k = i - st_ * K;
            const int col = 4 * g + j + ((sl_ == 1 && double_q) ? B : 0);      // (single-Q: the online net did not run on sp; the slot is zeroed in its consumers)
            const float* src = sl_ == 2 ? (st_ ? q11 : q01) : (st_ ? q10 : q00);
            yv[u] = *gptr(reinterpret_cast<const f32x4r*>(src + (size_t)col * K + k));
        }
    } else {
        const float *p00 = A.st[0].part[0], *p01 = A.st[0].part[1], *p10 = A.st[1].part[0], *p11 = A.st[1].part[1];
#pragma unroll
        for (int u = 0; u < 4; u++) {      // hidden row tid + 256 u: the group's four columns as one 16-byte piece per slot
            int i = tid + 256 * u; if (i >= KS) i = KS - 1;
            const int st_ = i >= K ? 1 : 0, k = i - st_ * K;
            const float* on = (st_ ? p10 : p00) + (size_t)k * ncon + 4 * g;
            yv[3 * u + 0] = *gptr(reinterpret_cast<const f32x4r*>(on));
            yv[3 * u + 1] = *gptr(reinterpret_cast<const f32x4r*>(on + (double_q ? B : 0)));
            yv[3 * u + 2] = *gptr(reinterpret_cast<const f32x4r*>((st_ ? p11 : p01) + (size_t)k * B + 4 * g));
        }
    }
    {      // head weights -> LDS, 4 floats of padding after every 32 rows (32 N floats: a multiple of 4, so a float4 never straddles a chunk)
        const FDiv f0 = fdiv_of(8 * N0s), f1 = fdiv_of(8 * (nstream > 1 ? N1s : 1));
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = tid + 256 * u;
            if (i < n0q + n1q) {
                const int st_ = i >= n0q ? 1 : 0, q = i - (st_ ? n0q : 0);
                const int off = (st_ ? won1 : 0) + 4 * q + WPAD * fdiv_q(q, st_ ? f1 : f0);
                *reinterpret_cast<f32x4r*>(Won + off) = wq[u]; *reinterpret_cast<f32x4r*>(Wtg + off) = wtq[u];
            }
        }
    }
    if (tid >= 128 && tid < 144) bm[tid - 128] = bm_r;
    if (viaT) {
        const FDiv fq = fdiv_of(qpr);
#pragma unroll
        for (int u = 0; u < 12; u++) {
            const int e = tid + 256 * u;
            if (e < 12 * qpr) {
                int cs, i4; fdiv_qr(e, fq, cs, i4);
                const int i = 4 * i4, st_ = i >= K ? 1 : 0, k = i - st_ * K;
                *reinterpret_cast<f32x4r*>(act + cs * RS + st_ * SP + 36 * (k >> 5) + (k & 31)) = yv[u];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = tid + 256 * u;
            if (i < KS) {
                const int st_ = i >= K ? 1 : 0, k = i - st_ * K;
                float* d = act + st_ * SP + 36 * (k >> 5) + (k & 31);
#pragma unroll
                for (int sl_ = 0; sl_ < 3; sl_++) { const f32x4r v = yv[3 * u + sl_]; float* ds_ = d + 4 * sl_ * RS; ds_[0] = v.x; ds_[RS] = v.y; ds_[2 * RS] = v.z; ds_[3 * RS] = v.w; }
            }
        }
    }
    if (bk) {
        if (take_pre && tid - 192 < B) *gptr(A.idx + (tid - 192)) = idx_v;
        if (take_pre) for (int i = tid - 192 + 64; i < B; i += 64) A.idx[i] = A.idx_pre[i];
        if (tid == 255) { *gptr(&A.stt->step) = step_v + 1; if (bump_sample_ctr) *gptr(&A.stt->sample_ctr) = sctr_v + 1; if (take_pre && pv_v != 2) *gptr(&A.stt->err) = 3; }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    C4_STAMP(1);
    // ---- 2. chunk sums: item (chunk c, slot, output o) carries the FOUR columns of its group -- one k-ascending chain of 32 from +0 per column (k_red_head phase B)
    for (int it = tid; it < 3 * NO * NC; it += 256) {
        const int o = it % NO, r_ = it / NO, sl_ = r_ % 3, c = r_ / 3;
        const int st_ = o >= nA ? 1 : 0, n = o - (st_ ? nA : 0), N = st_ ? N1s : N0s;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        if (!(sl_ == 1 && !double_q)) {
            const float* w = (sl_ == 2 ? Wtg : Won) + (st_ ? won1 : 0) + c * (32 * N + WPAD) + n;
            const float* x = act + 4 * sl_ * RS + st_ * SP + 36 * c;
#pragma unroll 2
            for (int kk = 0; kk < 8; kk++) {
                const f32x4r x0 = *reinterpret_cast<const f32x4r*>(x + 4 * kk), x1 = *reinterpret_cast<const f32x4r*>(x + RS + 4 * kk);
                const f32x4r x2 = *reinterpret_cast<const f32x4r*>(x + 2 * RS + 4 * kk), x3 = *reinterpret_cast<const f32x4r*>(x + 3 * RS + 4 * kk);
                const float w0 = w[(4 * kk) * N], w1 = w[(4 * kk + 1) * N], w2 = w[(4 * kk + 2) * N], w3 = w[(4 * kk + 3) * N];
                a0 = fmaf(x0.x, w0, a0); a0 = fmaf(x0.y, w1, a0); a0 = fmaf(x0.z, w2, a0); a0 = fmaf(x0.w, w3, a0);
                a1 = fmaf(x1.x, w0, a1); a1 = fmaf(x1.y, w1, a1); a1 = fmaf(x1.z, w2, a1); a1 = fmaf(x1.w, w3, a1);
                a2 = fmaf(x2.x, w0, a2); a2 = fmaf(x2.y, w1, a2); a2 = fmaf(x2.z, w2, a2); a2 = fmaf(x2.w, w3, a2);
                a3 = fmaf(x3.x, w0, a3); a3 = fmaf(x3.y, w1, a3); a3 = fmaf(x3.z, w2, a3); a3 = fmaf(x3.w, w3, a3);
            }
        }
        float* pp = P + ((size_t)(sl_ * 4) * NO + o) * NC + c;      // [slot][j][o][chunk]
        pp[0] = a0; pp[(size_t)NO * NC] = a1; pp[(size_t)2 * NO * NC] = a2; pp[(size_t)3 * NO * NC] = a3;
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    C4_STAMP(2);
    // ---- 3. finished head outputs: chunk sums added in ascending order, + bias, activation
    if (tid < nout) {
        const int o = tid % NO, sl_ = (tid / NO) >> 2, st_ = o >= nA ? 1 : 0;
        const float* pp = P + (size_t)tid * NC;
        float tot = pp[0];
        for (int q = 1; q < NC; q++) tot = tot + pp[q];
        hv[tid] = (sl_ == 1 && !double_q) ? 0.0f : rh_act<TRANS>(tot + hb, st_ ? A.st[1].hact : A.st[0].hact);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    C4_STAMP(3);
    // ---- 4. per-column TD arithmetic (lanes 192-195), then the heads' dX for every hidden row of both streams
    if (tid >= 192 && tid < 196) rh_td_column<TRANS>(A, hv, bm, dq, tid - 192, 4 * g + tid - 192);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    C4_STAMP(4);
    const RhDx dxa = {K, B, nA, N0s, N1s, A.st[0].pact, A.st[1].pact, won1, WPAD, A.st[0].dsrc, A.st[1].dsrc};
    float dqr[4][9];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int o = 0; o < 9; o++) dqr[j][o] = dq[j * NO + (o == 8 ? (nA < NO ? nA : NO - 1) : (o < NO ? o : NO - 1))];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int i = tid + 256 * u;
        if (i < KS) {
            const int st_ = i >= K ? 1 : 0, k = i - st_ * K;
            const float* ys = act + st_ * SP + 36 * (k >> 5) + (k & 31);      // the online s columns (slot 0): act' of the hidden layer
            const f32x4r y = (f32x4r){ys[0], ys[RS], ys[2 * RS], ys[3 * RS]};
            rh_dx_row<TRANS>(dxa, Won, dqr, i, y, g);
        }
    }
    C4_STAMP(5);
    if (stamps && tid == 0) atomicMax(stamps + 16, __builtin_amdgcn_s_memrealtime());
#undef C4_STAMP
}
static size_t head_cols4_lds_bytes(const RedHeadArgs& a) {
    const size_t NC = a.K / 32, KP = NC * 36, RS = (a.nstream > 1 ? 2 * KP + 12 : KP) + 8;
    const size_t wsz = (size_t)a.K * a.st[0].N + NC * 4 + (a.nstream > 1 ? (size_t)a.K * a.st[1].N + NC * 4 : 0);
    size_t f = 12 * RS + 2 * wsz + 12 * (size_t)a.NO * NC + ((12 * (size_t)a.NO + 3) & ~(size_t)3) + ((4 * (size_t)a.NO + 3) & ~(size_t)3) + 16;
    return f * sizeof(float);
}
size_t red_head_lds_bytes(const RedHeadArgs& a) {
    const size_t nmax = (size_t)std::max(a.st[0].N, a.nstream > 1 ? a.st[1].N : 0);
    size_t f = 384 + 32 * nmax + (size_t)a.K * a.st[0].N + (a.nstream > 1 ? (size_t)a.K * a.st[1].N : 0);
    f += ((12 * (size_t)a.NO + 3) & ~(size_t)3) + ((12 * (size_t)a.nA + 3) & ~(size_t)3) + ((4 * (size_t)a.NO + 3) & ~(size_t)3) + 16 + 4;
    f += 4 * (size_t)a.K * a.nstream;      // ys: the last arriver's y pieces
    return f * sizeof(float);
}
// shapes this launch covers (the caller has already checked: fused head level, dense split-K producers of both nets, distinct producers per stream): chunks of 32 hidden rows,
// groups of 4 columns; the hidden rows of both streams within the last arriver's 4 register slots per thread, the online head weights within 4 float4 per thread, <= 16
// chunks (one round of loads per head output), <= 16 slabs
bool red_head_ok(int B, int K, int S, int nA, int nstream, int N0, int N1) {
    const int NO = N0 + (nstream > 1 ? N1 : 0);
    return B % 4 == 0 && B >= 4 && B <= 1024 && K % 32 == 0 && K <= 512 && K * nstream <= 1024 && K * NO <= 4096 && S >= 1 && S <= 16 && nA >= 1 && nA <= 8 && N0 == nA && (nstream == 1 || N1 == 1) &&
           12 * std::max(N0, N1) <= 256 && 32 * std::max(N0, N1) <= 256 && 12 * NO <= 256;
}
void launch_red_head(hipStream_t st, const RedHeadArgs& a, const RedHeadArgs* a_dev, int bump_sample_ctr, int take_pre) {
    const size_t lds = red_head_lds_bytes(a);
    const unsigned grid = (unsigned)((a.B / 4) * (a.K / 32) * a.nstream);
    (void)a_dev;
    auto tr = [](int x) { return x == DQN_ACT_TANH || x == DQN_ACT_SIGMOID; };
    const bool trans = tr(a.st[0].pact) || tr(a.st[0].hact) || (a.nstream > 1 && (tr(a.st[1].pact) || tr(a.st[1].hact)));
    if (a.S == 1) {      // unsplit producers (large batches): one workgroup per column group, no hand-off
        const size_t l4 = head_cols4_lds_bytes(a);
        if (l4 > 64 * 1024) (void)hipFuncSetAttribute(trans ? (const void*)k_head_cols4<true> : (const void*)k_head_cols4<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
        if (trans) hipLaunchKernelGGL((k_head_cols4<true>), dim3(a.B / 4), dim3(256), l4, st, a, bump_sample_ctr, take_pre);
        else hipLaunchKernelGGL((k_head_cols4<false>), dim3(a.B / 4), dim3(256), l4, st, a, bump_sample_ctr, take_pre);
        return;
    }
    else if (a.S <= 8) { if (trans) hipLaunchKernelGGL((k_red_head<8, true>), dim3(grid), dim3(256), lds, st, a, bump_sample_ctr, take_pre); else hipLaunchKernelGGL((k_red_head<8, false>), dim3(grid), dim3(256), lds, st, a, bump_sample_ctr, take_pre); }
    else { if (trans) hipLaunchKernelGGL((k_red_head<16, true>), dim3(grid), dim3(256), lds, st, a, bump_sample_ctr, take_pre); else hipLaunchKernelGGL((k_red_head<16, false>), dim3(grid), dim3(256), lds, st, a, bump_sample_ctr, take_pre); }
}
