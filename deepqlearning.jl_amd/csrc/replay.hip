// replay.hip -- PrioritizedReplayBuffer on the device (gfx950).
//   add_exp!            src/prioritized_experience_replay.jl:65-74   -> k_replay_commit (rows are copied straight into the ring by the host API)
//   sample              src/prioritized_experience_replay.jl:82-87   -> k_sample (stratified sum-tree descent, Philox4x32-10)
//   get_batch           src/prioritized_experience_replay.jl:89-104  -> k_gather_fb (train path, batch-innermost), k_gather_rows + k_batch_meta (parity seam)
//   update_priorities!  src/prioritized_experience_replay.jl:76-80   -> k_update_priorities
// The sum-tree is canonical: every internal node is exactly f32(left + right) of its current children, so the root
// (the IS-weight denominator, :101) does not depend on update history and the CPU twin reproduces it bit for bit.
#include "common.h"
#include "gather_body.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ gather (HBM-bound)
// 64 features x 64 columns per workgroup through a padded LDS tile: reads are 256-B row segments of the sampled
// transitions (coalesced along the feature axis), writes are 256-B lines of the batch-innermost arena X0[f][2B]
// (columns 0..B-1 = s, B..2B-1 = sp).  Algorithmic bytes: 2*B*E*sizeof(obs) read + 2*B*E*4 written.
// With do_sample, every workgroup first repeats the (cheap, deterministic) stratified sum-tree descent for the 2B columns
// it needs -- B descents of log2(cap) dependent L2 hits -- instead of waiting for a separate single-workgroup sample
// launch (~4.6 us floor); workgroup (0,0) publishes the indices for k_td.  The Philox counter is bumped by k_td.
__global__ __launch_bounds__(256) void k_gather_fb(const void* __restrict__ s_rows, const void* __restrict__ sp_rows, int u8, int E, int B,
                                                   long long* __restrict__ idx, float* __restrict__ x0, int do_sample, long long cap2,
                                                   const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state, BatchMeta meta,
                                                   const long long* __restrict__ idx_pre) {
    __shared__ float tile[64][65];
    __shared__ long long rows[64];
    gather_fb_body(s_rows, sp_rows, u8, E, B, idx, x0, do_sample, cap2, tree, seed, state, meta, idx_pre, (int)blockIdx.x, (int)blockIdx.y, tile, rows);
}
// u8 rows (config 5: 1e6 transitions, 28 224 B each): 256 features x 64 columns per workgroup so that every sampled row is
// read in 256-B segments (one uchar4 per lane, 16 independent loads in flight per thread); 4x fewer workgroups than the f32
// tiling also means 4x fewer repeats of the descent when it is fused.
__global__ __launch_bounds__(256) void k_gather_fb_u8(const unsigned char* __restrict__ s_rows, const unsigned char* __restrict__ sp_rows, int E, int B,
                                                      long long* __restrict__ idx, float* __restrict__ x0, int do_sample, long long cap2,
                                                      const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state, BatchMeta meta,
                                                      const long long* __restrict__ idx_pre) {
    // the tile stays PACKED in LDS (64 columns x 64 words of 4 features = 16.6 KB instead of 66 KB of floats): several workgroups per CU keep the
    // random 256-B row reads in flight; bytes are unpacked and converted (256-entry table: one IEEE division per value of b, not per element)
    // on the way out
    __shared__ uint32_t tile32[64 * 65];
    __shared__ long long rows[64];
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;      // test/test_env.jl:59
    const int f0 = blockIdx.x * 256, c0 = blockIdx.y * 64, ld = 2 * B;
    if (threadIdx.x < 64) {
        const int c = c0 + threadIdx.x;
        long long r = 0;
        if (c < ld) {
            const int i = c < B ? c : c - B;
            if (do_sample) {
                // the indices of this sample() were drawn in the tail of the previous step's priority block unless something changed the tree since
                r = (idx_pre && state->pre_valid) ? idx_pre[i] : tree_descend(tree, cap2, state->size, seed, state->sample_ctr, i, tree[1] / (float)B);
                if (blockIdx.x == 0 && c < B) idx[i] = r;
            } else r = idx[i];
        }
        rows[threadIdx.x] = r;
    }
    __syncthreads();
    uint32_t v[16];
#pragma unroll
    for (int p = 0; p < 16; p++) {
        const int q = threadIdx.x + 256 * p, cl = q >> 6, c = c0 + cl, f = f0 + 4 * (q & 63);
        v[p] = 0;
        if (c < ld && f < E) v[p] = *reinterpret_cast<const uint32_t*>((c < B ? s_rows : sp_rows) + rows[cl] * E + f);      // E % 4 == 0
    }
#pragma unroll
    for (int p = 0; p < 16; p++) { const int q = threadIdx.x + 256 * p; tile32[(q >> 6) * 65 + (q & 63)] = v[p]; }     // consecutive lanes, consecutive words
    __syncthreads();
    // 16 lanes x float4 = one 256-B row segment of the arena (64 consecutive columns of one feature); a wave writes 4 feature rows per instruction
    const int l16 = threadIdx.x & 15, r16 = threadIdx.x >> 4;
#pragma unroll 4
    for (int p = 0; p < 16; p++) {
        const int fl = p * 16 + r16, f = f0 + fl, c = c0 + 4 * l16, sh = 8 * (fl & 3);
        if (f >= E) continue;
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = lut[(tile32[(4 * l16 + u) * 65 + (fl >> 2)] >> sh) & 0xffu];
        if (c + 3 < ld) *reinterpret_cast<f32x4*>(x0 + (size_t)f * ld + c) = (f32x4){o[0], o[1], o[2], o[3]};
        else for (int u = 0; u < 4; u++) if (c + u < ld) x0[(size_t)f * ld + c + u] = o[u];
    }
    if (blockIdx.x == 0) gather_batch_meta(meta, rows, c0, B, cap2, tree, state);
}
// u8 rows into a BYTE arena X0b[f][2B] (the first layer converts byte / 255 inside its tile loads): the gather moves 1 byte per element each way
// instead of writing 4 -- at config 5 (B = 512) 57.8 MB per launch instead of 144.5 MB.  256 features x 128 columns per workgroup: every sampled
// row is read in 256-byte segments (16 lanes x 16 B; 8 such loads in flight per lane), the arena is written in 128-byte segments; the tile stays
// packed in LDS (4 features per word, 33 KB).
__global__ __launch_bounds__(256) void k_gather_fb_u8b(const unsigned char* __restrict__ s_rows, const unsigned char* __restrict__ sp_rows, int E, int B,
                                                       long long* __restrict__ idx, unsigned char* __restrict__ x0b, int do_sample, long long cap2,
                                                       const float* __restrict__ tree, unsigned long long seed, const StepState* __restrict__ state, BatchMeta meta,
                                                       const long long* __restrict__ idx_pre) {
    __shared__ uint32_t tile32[128 * 65];
    __shared__ long long rows[128];
    gather_u8b_body(s_rows, sp_rows, E, B, idx, x0b, do_sample, cap2, tree, seed, state, meta, idx_pre, (int)blockIdx.x, (int)blockIdx.y, tile32, rows);
}
void launch_gather_fb(hipStream_t st, const void* s_rows, const void* sp_rows, int obs_u8, int E, int B, long long* idx, float* x0, int do_sample,
                      long long cap2, const float* tree, unsigned long long seed, const StepState* state, const BatchMeta& meta, const long long* idx_pre, int arena_u8) {
    if (arena_u8) {      // (engine_program.hip guarantees obs_u8, E % 4 == 0 and B % 2 == 0 here)
        dim3 grid((E + 255) / 256, (2 * B + 127) / 128);
        hipLaunchKernelGGL(k_gather_fb_u8b, grid, dim3(256), 0, st, (const unsigned char*)s_rows, (const unsigned char*)sp_rows, E, B, idx, (unsigned char*)x0,
                           do_sample, cap2, tree, seed, state, meta, idx_pre);
        return;
    }
    if (obs_u8 && (E & 3) == 0) {
        dim3 grid((E + 255) / 256, (2 * B + 63) / 64);
        hipLaunchKernelGGL(k_gather_fb_u8, grid, dim3(256), 0, st, (const unsigned char*)s_rows, (const unsigned char*)sp_rows, E, B, idx, x0,
                           do_sample, cap2, tree, seed, state, meta, idx_pre);
        return;
    }
    dim3 grid((E + 63) / 64, (2 * B + 63) / 64);
    hipLaunchKernelGGL(k_gather_fb, grid, dim3(256), 0, st, s_rows, sp_rows, obs_u8, E, B, idx, x0, do_sample, cap2, tree, seed, state, meta, idx_pre);
}

__global__ void k_gather_rows(const void* __restrict__ rows, int u8, int E, const long long* __restrict__ idx, float* __restrict__ out) {
    const long long row = idx[blockIdx.y];
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < E; f += gridDim.x * blockDim.x) {
        float v = u8 ? (float)((const unsigned char*)rows)[row * E + f] / 255.0f : ((const float*)rows)[row * E + f];
        out[(size_t)blockIdx.y * E + f] = v;
    }
}
void launch_gather_rows(hipStream_t st, const void* rows, int obs_u8, int E, int n, const long long* idx, float* out) {
    dim3 grid((E + 255) / 256 < 64 ? (E + 255) / 256 : 64, n);
    hipLaunchKernelGGL(k_gather_rows, grid, dim3(256), 0, st, rows, obs_u8, E, idx, out);
}

// obs[n][E] (host layout) -> x[E][n] (batch-innermost) for the policy path (src/policy.jl:38-64)
__global__ void k_transpose_obs(const float* __restrict__ obs, int E, int n, float* __restrict__ x) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)E * n) return;
    const int b = (int)(t % n); const size_t f = t / n;
    x[t] = obs[(size_t)b * E + f];
}
void launch_transpose_obs(hipStream_t st, const float* obs, int E, int n, float* x) {
    const size_t tot = (size_t)E * n;
    hipLaunchKernelGGL(k_transpose_obs, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, obs, E, n, x);
}

// ------------------------------------------------------------------ add_exp! (metadata + priorities + tree)
__global__ __launch_bounds__(1024) void k_replay_commit(int n, long long start, long long cap, long long cap2, const int* __restrict__ a_in,
                                                        const float* __restrict__ r_in, const unsigned char* __restrict__ done_in,
                                                        const float* __restrict__ td_in, float eps, float alpha, int* a, float* r,
                                                        unsigned char* done, float* tree, StepState* state) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const long long slot = (start + i) % cap;
        a[slot] = a_in[i]; r[slot] = r_in[i]; done[slot] = done_in[i] ? 1 : 0;
        const float td = td_in ? td_in[i] : fabsf(r_in[i]);   // default td_err = abs(expe.r), :65
        if (!(td + eps > 0.0f)) state->err = 1;                // @assert td_err + eps > 0, :66
        tree[cap2 + slot] = prio_f(td, eps, alpha);
    }
    __syncthreads();
    // the written leaves form <= 2 contiguous ranges (ring wrap); rebuild their ancestors level by level
    long long lo[2], hi[2]; int nr = 1;
    if (n >= cap) { lo[0] = 0; hi[0] = cap - 1; }
    else {
        const long long s0 = start % cap, e0 = (start + n - 1) % cap;
        if (e0 >= s0) { lo[0] = s0; hi[0] = e0; } else { lo[0] = s0; hi[0] = cap - 1; lo[1] = 0; hi[1] = e0; nr = 2; }
    }
    for (long long width = cap2; width > 1; width >>= 1) {          // width = number of nodes on the child level
        for (int q = 0; q < nr; q++) {
            const long long a0 = (width + lo[q]) >> 1, a1 = (width + hi[q]) >> 1;
            for (long long node = a0 + threadIdx.x; node <= a1; node += blockDim.x) tree[node] = tree[2 * node] + tree[2 * node + 1];
            lo[q] >>= 1; hi[q] >>= 1;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { long long s = state->size + n; state->size = s > cap ? cap : s; state->pre_valid = 0; }
}
void launch_replay_commit(hipStream_t st, int n, long long start, long long cap, long long cap2, const int* a_in, const float* r_in,
                          const unsigned char* done_in, const float* td_in, float eps, float alpha, int* a, float* r, unsigned char* done,
                          float* tree, StepState* state) {
    hipLaunchKernelGGL(k_replay_commit, dim3(1), dim3(1024), 0, st, n, start, cap, cap2, a_in, r_in, done_in, td_in, eps, alpha, a, r, done, tree, state);
}

// ------------------------------------------------------------------ sample
__global__ __launch_bounds__(1024) void k_sample(int B, long long cap2, const float* __restrict__ tree, unsigned long long seed,
                                                 long long* __restrict__ idx, StepState* state, int bump, int distinct) {
    __shared__ long long taken[1024]; __shared__ float tp[1024];
    const unsigned long long ctr = state->sample_ctr;
    const long long size = state->size;
    const float total = tree[1], seg = total / (float)B;
    for (int i = threadIdx.x; i < B; i += blockDim.x) idx[i] = tree_descend(tree, cap2, size, seed, ctr, i, seg);
    if (distinct && B <= 1024) {      // hp.sample_distinct: redraw later duplicates on the residual priorities (sample_distinct_block / sample_distinct_fix, common.h)
        __shared__ int any_dup;
        __syncthreads();
        sample_distinct_block(tree, cap2, size, seed, ctr, B, idx, taken, tp, &any_dup);
    }
    __syncthreads();
    if (threadIdx.x == 0 && bump) { state->sample_ctr = ctr + 1; state->pre_valid = 0; }      // pre-drawn indices belonged to the counter just consumed
}
void launch_sample(hipStream_t st, int B, long long cap2, const float* tree, unsigned long long seed, long long* idx, StepState* state, int bump, int distinct) {
    int bs = ((B + 63) / 64) * 64; if (bs > 1024) bs = 1024;
    hipLaunchKernelGGL(k_sample, dim3(1), dim3(bs), 0, st, B, cap2, tree, seed, idx, state, bump, distinct);
}

// ------------------------------------------------------------------ get_batch scalars + IS weights (parity seam)
__global__ void k_batch_meta(int B, long long cap2, const long long* __restrict__ idx, const int* __restrict__ a, const float* __restrict__ r,
                             const unsigned char* __restrict__ done, const float* __restrict__ tree, float beta, const StepState* state,
                             int* a_out, float* r_out, float* done_out, float* w_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const long long j = idx[i];
    a_out[i] = a[j]; r_out[i] = r[j]; done_out[i] = (float)done[j];
    const float p = tree[cap2 + j] / tree[1];                   // p = prio ./ sum(prio[1:n]), :101
    const float x = (float)state->size * p;                     // n .* p
    w_out[i] = (float)pow((double)x, -(double)beta);            // .^ (-beta), :102
}
void launch_batch_meta(hipStream_t st, int B, long long cap2, const long long* idx, const int* a, const float* r, const unsigned char* done,
                       const float* tree, float beta, const StepState* state, int* a_out, float* r_out, float* done_out, float* w_out) {
    hipLaunchKernelGGL(k_batch_meta, dim3((B + 255) / 256), dim3(256), 0, st, B, cap2, idx, a, r, done, tree, beta, state, a_out, r_out, done_out, w_out);
}

// ------------------------------------------------------------------ update_priorities! (host-called seam) / grad-norm fold (tick_adam != 0, on demand)
__global__ __launch_bounds__(1024) void k_update_priorities(int n, long long cap2, const long long* __restrict__ idx, const float* __restrict__ td,
                                                            float eps, float alpha, float* tree, StepState* state, int tick_adam, double beta1,
                                                            double beta2, const float* __restrict__ gmax_part, int n_gmax,
                                                            long long* __restrict__ idx_pre, unsigned long long seed, int pre_B) {
    __shared__ __attribute__((aligned(16))) long long sidx[1024 + 4096];      // n <= 1024 indices + the dense top of the sum-tree (8192 floats: the level of 4096 nodes and everything above)
    __shared__ float smax[16];
    if (tick_adam) {   // globalnorm (helpers.jl:38-46): fold the Adam kernel's per-block max-abs (max is order-independent => exact)
        float g = 0.0f;
        for (int j = threadIdx.x; j < n_gmax; j += blockDim.x) g = fmaxf(g, gmax_part[j]);
        for (int off = 32; off > 0; off >>= 1) g = fmaxf(g, __shfl_xor(g, off));
        if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = g;
        __syncthreads();
        if (threadIdx.x == 0) { for (int w = 1; w < (int)(blockDim.x >> 6); w++) g = fmaxf(g, smax[w]); state->gnorm_bits = __float_as_uint(g); }
    }
    if (n > 0) { if (threadIdx.x == 0) state->pre_valid = 0; prio_update_block(n, cap2, idx, td, eps, alpha, tree, state, sidx, (unsigned)sizeof sidx); }
    if (n > 0 && idx_pre) {      // the tree is final and the next sample()'s Philox counter is known: draw its indices now (see prio_block_run)
        __syncthreads();
        const unsigned long long ctr = state->sample_ctr; const long long size = state->size;
        const float seg = tree[1] / (float)pre_B;
        for (int i = threadIdx.x; i < pre_B; i += blockDim.x) idx_pre[i] = tree_descend(tree, cap2, size, seed, ctr, i, seg);
        __syncthreads();
        if (threadIdx.x == 0) state->pre_valid = 1;
    }
}
void launch_update_priorities(hipStream_t st, int n, long long cap2, const long long* idx, const float* td, float eps, float alpha,
                              float* tree, StepState* state, int tick_adam, double beta1, double beta2, const float* gmax_part, int n_gmax,
                              long long* idx_pre, unsigned long long seed, int pre_B) {
    int bs = ((n + 63) / 64) * 64; if (bs < 64) bs = 64;
    if (tick_adam && bs < 1024) bs = 1024;      // the fold: one load per lane for the ~2100 live slots of config 2
    hipLaunchKernelGGL(k_update_priorities, dim3(1), dim3(bs), 0, st, n, cap2, idx, td, eps, alpha, tree, state, tick_adam, beta1, beta2, gmax_part, n_gmax, idx_pre, seed, pre_B);
}

// ------------------------------------------------------------------ (loss, grad_norm) -> host mailbox: the last launch of a dqn_train_step that returns scalars
// globalnorm (helpers.jl:38-46) is the max over the Adam jobs' per-block maxima (exact: max is order-independent); the record goes straight into mapped
// pinned host memory, payload first, then -- after a system-scope fence -- its sequence number
__global__ __launch_bounds__(1024) void k_publish_scalars(StepState* state, const float* __restrict__ gmax_part, int n_gmax, unsigned long long* pub_ctr, StepMail* mail) {
    __shared__ float smax[16];
    float g = 0.0f;
    for (int j = threadIdx.x; j < n_gmax; j += blockDim.x) g = fmaxf(g, gmax_part[j]);
    for (int off = 32; off > 0; off >>= 1) g = fmaxf(g, __shfl_xor(g, off));
    if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = g;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) g = fmaxf(g, smax[w]);
        state->gnorm_bits = __float_as_uint(g);
        const unsigned long long seq = *pub_ctr + 1; *pub_ctr = seq;
        StepMail* m = mail + (seq & (DQN_MAIL_SLOTS - 1));
        // a device-side assertion failure is CONSUMED here (delivered in exactly one record): left sticky, every step already enqueued behind the failing one would
        // publish it again and the host would report it once per outstanding ticket; the host sweeps every arrived record in order (mail_sweep, engine.hip)
        m->loss = state->loss; m->gnorm = g; m->err = atomicExch(&state->err, 0); m->step = state->step;
        __threadfence_system();
        __hip_atomic_store(&m->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
void launch_publish_scalars(hipStream_t st, StepState* state, const float* gmax_part, int n_gmax, unsigned long long* pub_ctr, StepMail* mail) {
    hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(1024), 0, st, state, gmax_part, n_gmax, pub_ctr, mail);
}

// ------------------------------------------------------------------ checkpoint import: rebuild every internal node of the sum-tree from the leaves
__global__ void k_tree_level(float* tree, long long width) {      // width = number of nodes on the PARENT level
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) { const long long node = width + i; tree[node] = tree[2 * node] + tree[2 * node + 1]; }
}
void launch_tree_rebuild(hipStream_t st, float* tree, long long cap2) {
    for (long long width = cap2 >> 1; width >= 1; width >>= 1)
        hipLaunchKernelGGL(k_tree_level, dim3((unsigned)((width + 255) / 256)), dim3(256), 0, st, tree, width);
}
