// adam_body.h -- globalnorm (helpers.jl:38-46) + Flux Adam (solver.jl:66,228) + update_priorities! (solver.jl:231-233) as ONE job description
// (AdamJob) run either by k_adam (nn_valu.hip) or by TAIL workgroups of the LDS-tiled backward launches (nn_gemm.hip): parameters whose gradient
// is already final are updated under the next, latency-bound backward launch instead of after the last one.
//   blocks of a job:  [prio block, if prio.n > 0] [segs.blocks slab-reduce blocks] [sblocks streaming blocks]
//   slab-reduce block: one element per thread: gradient = ascending sum of S split-K slabs (canonical chunk order), then Adam on it
//   streaming block:   16-B accesses over the job's element ranges (multiples of 4; arrays 16-B aligned), skipping slab ranges inside them
#pragma once
#include "common.h"

#ifndef DQN_ADAM_ST
#define DQN_ADAM_ST 0      /* experiment (r05): bit 0 = m, v stored write-through (sc1), bit 1 = p too */
#endif
__device__ __forceinline__ void adam_st4(float4* p, const float4& v, int wt) {
    typedef float adam_f4 __attribute__((ext_vector_type(4)));
    if (wt) { const adam_f4 x = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory"); }
    else *p = v;
}
// one element: Flux 0.14 Optimise.Adam with Float64 scalars (f64mode) or plain fp32; returns |g| for the max-abs norm
__device__ __forceinline__ float adam_upd(float gi, float& mi, float& vi, float& pi, int f64mode, float lr, double b1, double b2, double eps, double bp1, double bp2, float gscale) {
    if (gscale != 1.0f) gi = gi * gscale;
    float mn, vn, dl;
    if (f64mode) {
        const double c1 = 1.0 - bp1, c2 = 1.0 - bp2;
        const double gd = (double)gi;
        const double t1 = b1 * (double)mi; const double t2 = (1.0 - b1) * gd; mn = (float)(t1 + t2);
        const double u1 = b2 * (double)vi; const double u2 = (1.0 - b2) * gd; const double u3 = u2 * gd; vn = (float)(u1 + u3);
        const double mh = (double)mn / c1; const double vh = (double)vn / c2; const double den = sqrt(vh) + eps; const double q1 = mh / den;
        dl = (float)(q1 * (double)lr);
    } else {
        const float fb1 = (float)b1, fb2 = (float)b2;
        const float t1 = fb1 * mi; const float t2 = (1.0f - fb1) * gi; mn = t1 + t2;
        const float u1 = fb2 * vi; const float u2 = (1.0f - fb2) * gi; const float u3 = u2 * gi; vn = u1 + u3;
        const float mh = mn / (1.0f - (float)bp1); const float vh = vn / (1.0f - (float)bp2); const float den = sqrtf(vh) + (float)eps; const float q1 = mh / den;
        dl = q1 * lr;
    }
    mi = mn; vi = vn; pi = pi - dl;
    return fabsf(gi);
}
// workgroup `bid` (256 threads) of job J.  sidx: 1024 long longs of LDS for the priority block (unused when the job has none or it runs
// elsewhere); wmax: 4 floats of LDS.
__device__ __forceinline__ void adam_job_run(const AdamJob& J, int bid, long long* sidx, float* wmax, bool prio_elsewhere = false, unsigned sidx_bytes = 0, float* fold_buf = nullptr) {
    if (J.prio.n > 0 && !prio_elsewhere) {
        // update_priorities!(replay, indices, td): one DEDICATED workgroup walks the sum-tree while the others stream -- its latency-bound
        // levels ride inside a longer launch instead of costing one of their own; the tree is next read by the following step's sampler
        if (bid == 0) { prio_block_run(J.prio, J.state, sidx, sidx_bytes); return; }
        bid--;
    }
    // beta powers are double-buffered by step parity: this step reads slot (step & 1); the job with `tick` writes slot ((step+1) & 1)
    // (Flux: bp .= bp .* beta AFTER the update), so no block of any job can observe a half-updated value.
    const int slot = (int)(J.state->step & 1ull);
    const double bp1 = J.state->bp[slot][0], bp2 = J.state->bp[slot][1];
    const bool rblock = bid < (int)J.segs.blocks;
    if (J.tick && bid == (int)J.segs.blocks && threadIdx.x == 0) { J.state->bp[slot ^ 1][0] = bp1 * J.b1; J.state->bp[slot ^ 1][1] = bp2 * J.b2; }
    if (J.tick && J.fold_hl && fold_buf && bid == (int)J.segs.blocks) {      // recurrent fused step: loss fold in the twin's order (t outer, b inner, / B per step, / T at the end)
        // rows of B terms come in through LDS (all loads of a round in flight at once); one thread per row adds its B terms in order, thread 0 then adds the
        // rows' lsum / B in row order -- the association of oracle/dqn_ref.c, with the T rows advancing side by side
        const int B = J.fold_B, T = J.fold_T; float loss = 0.0f;
        if (B <= 128) {
            const int rpc = 256 / B;                                   // rows per round
            for (int t0 = 0; t0 < T; t0 += rpc) {
                const int nr = T - t0 < rpc ? T - t0 : rpc;
                if ((int)threadIdx.x < nr * B) fold_buf[threadIdx.x] = J.fold_hl[t0 * B + threadIdx.x];
                __syncthreads();
                float lsum = 0.0f;
                if ((int)threadIdx.x < nr) {
                    const float* r = fold_buf + threadIdx.x * B;
                    for (int b = 0; b < B; b += 8) {
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) v[q] = b + q < B ? r[b + q] : 0.0f;
#pragma unroll
                        for (int q = 0; q < 8; q++) if (b + q < B) lsum = lsum + v[q];
                    }
                    lsum = lsum / (float)B;
                }
                __syncthreads();
                if ((int)threadIdx.x < nr) fold_buf[threadIdx.x] = lsum;
                __syncthreads();
                if (threadIdx.x == 0) for (int r = 0; r < nr; r++) loss = loss + fold_buf[r];
                __syncthreads();
            }
        } else if (threadIdx.x == 0) {
            for (int t = 0; t < T; t++) { float lsum = 0.0f; for (int b = 0; b < B; b++) lsum = lsum + J.fold_hl[t * B + b]; loss = loss + lsum / (float)B; }
        }
        if (threadIdx.x == 0) J.state->loss = loss / (float)T;
    }
    float gmax = 0.0f;
    if (rblock) {
        size_t e = (size_t)bid * blockDim.x + threadIdx.x;      // index into the concatenation of the segments
        for (int q = 0; q < J.segs.n; q++) {
            const size_t len = J.segs.end[q] - J.segs.beg[q];
            if (e < len) {
                const float tot = slab_sum(J.segs.part[q] + e, J.segs.stride[q] ? (size_t)J.segs.stride[q] : len, J.segs.S[q]);
                const size_t i = J.segs.beg[q] + e;
                J.g_out[i] = tot;                                     // the materialised gradient (dqn_get_grads, parity tests)
                gmax = fmaxf(gmax, adam_upd(tot, J.m[i], J.v[i], J.p[i], J.f64mode, J.lr, J.b1, J.b2, J.eps, bp1, bp2, J.gscale));
                break;
            }
            e -= len;
        }
    } else {
        const int sb = bid - (int)J.segs.blocks, nsb = (int)J.sblocks;
        for (int r = 0; r < J.nr; r++) {
            const size_t i0 = J.beg[r] / 4, i1 = J.end[r] / 4;
            for (size_t i = i0 + (size_t)sb * blockDim.x + threadIdx.x; i < i1; i += (size_t)nsb * blockDim.x) {
                bool skip = false;
                for (int q = 0; q < J.segs.n; q++) skip = skip || (4 * i >= J.segs.beg[q] && 4 * i < J.segs.end[q]);
                if (skip) continue;
                const float4 g4 = reinterpret_cast<const float4*>(J.g)[i]; float4 m4 = reinterpret_cast<float4*>(J.m)[i], v4 = reinterpret_cast<float4*>(J.v)[i], p4 = reinterpret_cast<float4*>(J.p)[i];
                gmax = fmaxf(gmax, adam_upd(g4.x, m4.x, v4.x, p4.x, J.f64mode, J.lr, J.b1, J.b2, J.eps, bp1, bp2, J.gscale));
                gmax = fmaxf(gmax, adam_upd(g4.y, m4.y, v4.y, p4.y, J.f64mode, J.lr, J.b1, J.b2, J.eps, bp1, bp2, J.gscale));
                gmax = fmaxf(gmax, adam_upd(g4.z, m4.z, v4.z, p4.z, J.f64mode, J.lr, J.b1, J.b2, J.eps, bp1, bp2, J.gscale));
                gmax = fmaxf(gmax, adam_upd(g4.w, m4.w, v4.w, p4.w, J.f64mode, J.lr, J.b1, J.b2, J.eps, bp1, bp2, J.gscale));
                // m and v are next read a whole step later: small-batch engines store them write-through (J.wt; r05 same-box A/B, profiles/history/r05_m_store_ab.txt: +0.5 %; p too: -0.2 %)
                adam_st4(reinterpret_cast<float4*>(J.m) + i, m4, J.wt | (DQN_ADAM_ST & 1)); adam_st4(reinterpret_cast<float4*>(J.v) + i, v4, J.wt | (DQN_ADAM_ST & 1)); adam_st4(reinterpret_cast<float4*>(J.p) + i, p4, DQN_ADAM_ST & 2);
            }
        }
    }
    // wave max (64 lanes) then one value per block; max is order-independent, so this is exact
    for (int off = 32; off > 0; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = gmax;
    __syncthreads();
    if (threadIdx.x == 0) J.gmax_part[J.slot0 + bid] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));   // folded on demand by k_update_priorities
}
