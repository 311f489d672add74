// gather_body.h -- get_batch (src/prioritized_experience_replay.jl:89-104) of the train path as a DEVICE body, run by k_gather_fb (replay.hip) and
// by the pre-gather workgroups of the Adam launch (nn_valu.hip, k_adam_pg): inside dqn_train_steps(n) step i's last launch already gathers
// step i+1's batch, so that step i+1 starts with its first convolution instead of a gather launch.
#pragma once
#include "common.h"

typedef float gb_f32x4 __attribute__((ext_vector_type(4)));

// batch scalars + IS weights of the B sampled transitions (k_batch_meta's arithmetic), by the first 64 lanes of ONE workgroup at the END of the
// gather launch: its two dependent round trips and the double-precision pow overlap the other workgroups' row traffic
__device__ __forceinline__ void gather_batch_meta(const BatchMeta& M, const long long* rows, int c0, int B, long long cap2, const float* __restrict__ tree,
                                                  const StepState* __restrict__ state) {
    if (!M.a_out || threadIdx.x >= 64) return;
    const int c = c0 + threadIdx.x;
    if (c >= B) return;
    const long long j = rows[threadIdx.x];
    M.a_out[c] = M.a[j]; M.r_out[c] = M.r[j]; M.done_out[c] = (float)M.done[j];
    const float p = tree[cap2 + j] / tree[1];                   // p = prio ./ sum(prio[1:n]), :101
    const float x = (float)state->size * p;                     // n .* p
    M.w_out[c] = (float)pow((double)x, -(double)M.beta);        // .^ (-beta), :102
}
// hp.sample_distinct on the fused sample + gather path (B <= 64): the workgroup reproduces the WHOLE list -- the pre-drawn, already deduped one when it is valid, else the B
// stratified draws followed by sample()'s dedupe (sample_distinct_block: deterministic, so every workgroup of the launch arrives at the same list) -- into `dl`
__device__ __forceinline__ void gather_distinct_list(long long* dl, int B, long long cap2, const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state,
                                                     const long long* __restrict__ idx_pre) {
    __shared__ long long gd_taken[64]; __shared__ float gd_tp[64]; __shared__ int gd_any;
    const bool pre = idx_pre && state->pre_valid;
    const long long size = state->size; const unsigned long long ctr = state->sample_ctr;
    if ((int)threadIdx.x < B) dl[threadIdx.x] = pre ? idx_pre[threadIdx.x] : tree_descend(tree, cap2, size, seed, ctr, (int)threadIdx.x, tree[1] / (float)B);
    __syncthreads();
    if (!pre) sample_distinct_block(tree, cap2, size, seed, ctr, B, dl, gd_taken, gd_tp, &gd_any);
}
// (bx, by): the 64-feature x 64-column tile; tile: 64 x 65 floats of LDS; rows: 64 long longs of LDS
__device__ __forceinline__ void gather_fb_body(const void* __restrict__ s_rows, const void* __restrict__ sp_rows, int u8, int E, int B,
                                               long long* __restrict__ idx, float* __restrict__ x0, int do_sample, long long cap2,
                                               const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state, const BatchMeta& meta,
                                               const long long* __restrict__ idx_pre, int bx, int by, float (*tile)[65], long long* rows) {
    const int f0 = bx * 64, c0 = by * 64, lane = threadIdx.x & 63, w = threadIdx.x >> 6, ld = 2 * B;
    __shared__ long long gd_list[64];
    const bool dist = do_sample && meta.distinct && B <= 64;
    if (dist) gather_distinct_list(gd_list, B, cap2, tree, seed, state, idx_pre);
    if (threadIdx.x < 64) {
        const int c = c0 + threadIdx.x;
        long long r = 0;
        if (c < ld) {
            const int i = c < B ? c : c - B;
            if (dist) { r = gd_list[i]; if (bx == 0 && c < B) idx[i] = r; }
            else if (do_sample) {
                // the indices of this sample() were drawn in the tail of the previous step's priority block unless something changed the tree since
                r = (idx_pre && state->pre_valid) ? idx_pre[i] : tree_descend(tree, cap2, state->size, seed, state->sample_ctr, i, tree[1] / (float)B);
                if (bx == 0 && c < B) idx[i] = r;
            } else r = idx[i];
        }
        rows[threadIdx.x] = r;
    }
    __syncthreads();
    if (!u8 && (E & 3) == 0) {
        // f32 rows: 4 independent 16-B loads per thread are issued before any is consumed (HBM latency overlapped); a 64-feature
        // row segment is 16 lanes x 16 B = one 256-B burst of a sampled transition
        gb_f32x4 v[4];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int q = threadIdx.x + 256 * p, cl = q >> 4, c = c0 + cl, f = f0 + 4 * (q & 15);
            v[p] = (gb_f32x4){0.f, 0.f, 0.f, 0.f};
            if (c < ld && f < E) v[p] = *reinterpret_cast<const gb_f32x4*>((const float*)(c < B ? s_rows : sp_rows) + rows[cl] * E + f);
        }
        // LDS transpose without bank conflicts (r04): ds_write_b32 / ds_read_b32 are serviced per 32-lane group on banks (dword address) mod 32; with the 65-word pitch the 16 float4
        // columns of a row (and the 16 column quads of the read-back) repeat every 8 lanes, so lanes 8 apart rotate which of their four words goes out in a given instruction
        // (stores: by column half + row parity; reads: by 2, the two feature rows of a group being 1 bank apart) -- every instruction then touches 32 distinct banks (PMC r03: 112 896 conflict cycles per launch)
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const int q = threadIdx.x + 256 * p, cl = q >> 4, m = q & 15, fl = 4 * m, rot = (m >> 3) + (cl & 1);      // the four (row parity, column half) classes of a 32-lane group land on the four word offsets
            const float vv[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
            for (int i = 0; i < 4; i++) { const int ii = (i + rot) & 3; tile[cl][fl + ii] = ii == 0 ? vv[0] : ii == 1 ? vv[1] : ii == 2 ? vv[2] : vv[3]; }
        }
    } else {
#pragma unroll 4
        for (int p = 0; p < 16; p++) {
            const int cl = p * 4 + w, c = c0 + cl, f = f0 + lane;
            float v = 0.0f;
            if (c < ld && f < E) {
                const long long row = rows[cl];
                const void* base = c < B ? s_rows : sp_rows;
                if (u8) v = (float)((const unsigned char*)base)[row * E + f] / 255.0f;  // test/test_env.jl:59
                else v = ((const float*)base)[row * E + f];
            }
            tile[cl][lane] = v;
        }
    }
    __syncthreads();
    // 16 lanes x float4 = one 256-B row segment of the arena (64 consecutive columns of one feature); a wave writes 4 feature rows per instruction
    const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int fl = p * 16 + r16, f = f0 + fl, c = c0 + 4 * l16;
        if (f >= E) continue;
        if (c + 3 < ld) {
            float o4[4]; const int rot = (l16 >> 3) << 1;
#pragma unroll
            for (int u = 0; u < 4; u++) { const int uu = (u + rot) & 3; const float t = tile[4 * l16 + uu][fl]; if (uu == 0) o4[0] = t; else if (uu == 1) o4[1] = t; else if (uu == 2) o4[2] = t; else o4[3] = t; }
#if defined(DQN_ADAM_ST) && (DQN_ADAM_ST & 4)
            { const gb_f32x4 ov = {o4[0], o4[1], o4[2], o4[3]}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(x0 + (size_t)f * ld + c), "v"(ov) : "memory"); }
#else
            *reinterpret_cast<gb_f32x4*>(x0 + (size_t)f * ld + c) = (gb_f32x4){o4[0], o4[1], o4[2], o4[3]};
#endif
        }
        else for (int u = 0; u < 4; u++) if (c + u < ld) x0[(size_t)f * ld + c + u] = tile[4 * l16 + u][fl];
    }
    if (bx == 0) gather_batch_meta(meta, rows, c0, B, cap2, tree, state);
}
// u8 rows into a BYTE arena X0b[f][2B] (the first layer converts byte / 255 inside its tile loads): the gather moves 1 byte per element each way
// instead of writing 4 -- at config 5 (B = 512) 57.8 MB per launch instead of 144.5 MB.  256 features x 128 columns per workgroup: every sampled
// row is read in 256-byte segments (16 lanes x 16 B; 8 such loads in flight per lane), the arena is written in 128-byte segments; the tile stays
// packed in LDS (4 features per word, 33 KB).
// (bx, by): the 256-feature x 128-column tile; tile32: 128 x 65 words of LDS ([column][64 words of 4 features], row pitch 65); rows: 128 long longs
__device__ __forceinline__ void gather_u8b_body(const unsigned char* __restrict__ s_rows, const unsigned char* __restrict__ sp_rows, int E, int B,
                                                long long* __restrict__ idx, unsigned char* __restrict__ x0b, int do_sample, long long cap2,
                                                const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state, const BatchMeta& meta,
                                                const long long* __restrict__ idx_pre, int bx, int by, uint32_t* tile32, long long* rows) {
    const int f0 = bx * 256, c0 = by * 128, ld = 2 * B;
    __shared__ long long gd_list[64];
    const bool dist = do_sample && meta.distinct && B <= 64;
    if (dist) gather_distinct_list(gd_list, B, cap2, tree, seed, state, idx_pre);
    if (threadIdx.x < 128) {
        const int c = c0 + threadIdx.x;
        long long r = 0;
        if (c < ld) {
            const int i = c < B ? c : c - B;
            if (dist) { r = gd_list[i]; if (bx == 0 && c < B) idx[i] = r; }
            else if (do_sample) {
                r = (idx_pre && state->pre_valid) ? idx_pre[i] : tree_descend(tree, cap2, state->size, seed, state->sample_ctr, i, tree[1] / (float)B);
                if (bx == 0 && c < B) idx[i] = r;
            } else r = idx[i];
        }
        rows[threadIdx.x] = r;
    }
    __syncthreads();
    // 128 columns x 16 uint4 (256 bytes) = 2048 uint4 per tile, 8 per lane; rows of E bytes are 16-byte aligned when E % 16 == 0, else 4-byte loads
    const bool a16 = (E & 15) == 0;
    uint4 v[8];
#pragma unroll
    for (int p = 0; p < 8; p++) {
        const int q = threadIdx.x + 256 * p, cl = q >> 4, c = c0 + cl, f = f0 + 16 * (q & 15);
        v[p] = make_uint4(0, 0, 0, 0);
        if (c < ld && f < E) {
            const unsigned char* src = (c < B ? s_rows : sp_rows) + rows[cl] * E + f;
            if (a16 && f + 16 <= E) v[p] = *reinterpret_cast<const uint4*>(src);
            else { uint32_t w[4] = {0, 0, 0, 0}; for (int u = 0; u < 4; u++) if (f + 4 * u < E) w[u] = *reinterpret_cast<const uint32_t*>(src + 4 * u); v[p] = make_uint4(w[0], w[1], w[2], w[3]); }
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) {      // rotated word order per 8-lane block: conflict-free like gather_fb_body's stores (r04; PMC r03: 67 % of this kernel's LDS cycles were conflicts)
        const int q = threadIdx.x + 256 * p, m = q & 15, rot = (m >> 3) + ((q >> 4) & 1); uint32_t* d = tile32 + (q >> 4) * 65 + 4 * m;
        const uint32_t vv[4] = {v[p].x, v[p].y, v[p].z, v[p].w};
#pragma unroll
        for (int i = 0; i < 4; i++) { const int ii = (i + rot) & 3; d[ii] = ii == 0 ? vv[0] : ii == 1 ? vv[1] : ii == 2 ? vv[2] : vv[3]; }
    }
    __syncthreads();
    // 32 lanes x 4 columns = one 128-byte segment of a feature row of the arena; a wave writes 2 feature rows per instruction.
    // r06: a lane reads the FOUR words (4 features each) of its four columns once and transposes the 4 x 4 bytes in registers (v_perm_b32: two stages of four) into the four
    // output words of features 4 wg .. 4 wg + 3 -- 32 LDS reads and 64 byte permutes per thread instead of 128 reads and ~400 shift / mask / or instructions (each word was
    // re-read by the four iterations that took one byte of it).  The reads are rotated per 8-lane block as before (lanes 8 apart sit on one bank); the rotation is undone
    // by the second stage's per-lane selectors.
    const int c4 = threadIdx.x & 31, r8 = threadIdx.x >> 5, rot = c4 >> 3, c = c0 + 4 * c4;
    // word r_i = column (i + rot) & 3.  Stage 1: t0 = {r0.b0, r1.b0, r0.b1, r1.b1}, t1 = {r0.b2, r1.b2, r0.b3, r1.b3}, u0 / u1 likewise from r2, r3.
    // Stage 2, perm(u, t, sel): byte b_even of r_i sits at selector index {0, 1, 4, 5}[i], b_odd at {2, 3, 6, 7}[i]; output byte u (column u) takes r_{(u - rot) & 3}
    uint32_t selE = 0, selO = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = (u - rot) & 3, pe = (i & 1) + ((i >> 1) << 2); selE |= (uint32_t)pe << (8 * u); selO |= (uint32_t)(pe + 2) << (8 * u); }
    const uint32_t* tc = tile32 + (4 * c4) * 65;
    const int o0 = ((0 + rot) & 3) * 65, o1 = ((1 + rot) & 3) * 65, o2 = ((2 + rot) & 3) * 65, o3 = ((3 + rot) & 3) * 65;
#pragma unroll 4
    for (int p = 0; p < 8; p++) {
        const int wg = p * 8 + r8, f = f0 + 4 * wg;
        if (f >= E) continue;
        const uint32_t r0 = tc[o0 + wg], r1 = tc[o1 + wg], r2 = tc[o2 + wg], r3 = tc[o3 + wg];
        const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
        const uint32_t u0 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), u1 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
        const uint32_t o[4] = {__builtin_amdgcn_perm(u0, t0, selE), __builtin_amdgcn_perm(u0, t0, selO), __builtin_amdgcn_perm(u1, t1, selE), __builtin_amdgcn_perm(u1, t1, selO)};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (f + j >= E) break;
            if (c + 3 < ld) *reinterpret_cast<uint32_t*>(x0b + (size_t)(f + j) * ld + c) = o[j];
            else for (int u = 0; u < 4; u++) if (c + u < ld) x0b[(size_t)(f + j) * ld + c + u] = (unsigned char)(o[j] >> (8 * u));
        }
    }
    if (bx == 0) { gather_batch_meta(meta, rows, c0, B, cap2, tree, state); gather_batch_meta(meta, rows + 64, c0 + 64, B, cap2, tree, state); }
}
