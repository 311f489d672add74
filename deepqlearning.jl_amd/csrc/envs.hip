// envs.hip -- vectorised environments on the device: the env loop of dqn_train! (src/solver.jl:82-132) for n
// lock-stepped copies without any host round trip.  TestMDP restates test/test_env.jl:10-87, SimpleGridWorld restates the
// POMDPModels defaults (third-party; recalled).  All randomness is Philox4x32-10 with counter (vector step, env, purpose) so
// that the CPU twin (oracle/dqn_ref.c) reproduces trajectories bit for bit.
//
// One vector step is: [policy forward on pol_x -> greedy]  k_env_step  k_env_observe2   -- every argument is a fixed device
// pointer (the step counter, the eps schedule and the ring cursor live in RolloutDev), so the whole step replays as a hipGraph.
#include "common.h"

#define ENV_HEAD_LDS 8192      // floats of head outputs staged in LDS by k_env_step (32 KB)

__device__ __forceinline__ uint32_t env_rand(unsigned long long seed, unsigned long long t, int env, uint32_t purpose) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)env, purpose};
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
    return c[0];
}
__device__ __forceinline__ float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

// element f of the observation of an env in state (sw = the 4 TestMDP state bytes packed little-endian | px, py).  No local
// arrays: a dynamically indexed one would live in scratch.
__device__ __forceinline__ float obs_elem(const EnvDev& V, uint32_t sw, int px_, int py_, int f, unsigned char* raw) {
    if (V.kind == DQN_ENV_TESTMDP) {
        const int hw = V.H * V.W, c = f / hw, px = f - c * hw;            // obs[.., c] = observations[s[end - c]]  (test/test_env.jl:56-58)
        const int sel = (int)((sw >> (8 * (3 - c))) & 0xffu) - 1;
        const unsigned char b = V.images[sel * hw + px];
        *raw = b; return (float)b / 255.0f;
    }
    *raw = 0; return (float)(f == 0 ? px_ : py_);                          // Float32[x, y]
}
__device__ __forceinline__ void reset_state(const EnvDev& V, int i, unsigned long long t, uint32_t* sw, int* tm_t, int* px, int* py) {
    if (V.kind == DQN_ENV_TESTMDP) { *sw = 0x01010101u; *tm_t = 1; }                                                  // initialstate, :46-52
    else { *px = 1 + (int)(env_rand(V.seed, t, i, 5u) % (uint32_t)V.size_x); *py = 1 + (int)(env_rand(V.seed, t, i, 6u) % (uint32_t)V.size_y); }
}
__device__ __forceinline__ void load_state(const EnvDev& V, int i, uint32_t* sw, int* px, int* py) {
    *sw = 0x01010101u; *px = *py = 0;
    if (V.kind == DQN_ENV_TESTMDP) *sw = *(const uint32_t*)(V.tm_s + i * 4);
    else { *px = V.gw_pos[i * 2]; *py = V.gw_pos[i * 2 + 1]; }
}
__device__ __forceinline__ void store_reset(const EnvDev& V, int i, unsigned long long t) {
    uint32_t sw; int tm = 1, px, py; reset_state(V, i, t, &sw, &tm, &px, &py);
    if (V.kind == DQN_ENV_TESTMDP) { *(uint32_t*)(V.tm_s + i * 4) = sw; V.tm_t[i] = tm; } else { V.gw_pos[i * 2] = px; V.gw_pos[i * 2 + 1] = py; }
}

// observation of every env into (a) the staged rows[i][E] in the replay storage dtype and (b) batch-innermost x[E][n] (policy input)
__global__ void k_env_observe(EnvDev V, void* __restrict__ rows, int rows_u8, float* __restrict__ x) {
    const int n = V.n, E = V.E;
    const size_t tot = (size_t)n * E;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < tot; q += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(q / E), f = (int)(q % E);
        uint32_t sw; int px, py; load_state(V, i, &sw, &px, &py);
        unsigned char vb; const float v = obs_elem(V, sw, px, py, f, &vb);
        if (rows) { if (rows_u8) ((unsigned char*)rows)[q] = vb; else ((float*)rows)[q] = v; }
        if (x) x[(size_t)f * n + i] = v;
    }
}
void launch_env_observe(hipStream_t st, const EnvDev& V, void* rows, int rows_u8, float* x) {
    const size_t tot = (size_t)V.n * V.E; unsigned blocks = (unsigned)((tot + 255) / 256); if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_env_observe, dim3(blocks), dim3(256), 0, st, V, rows, rows_u8, x);
}

// add_exp!'s tree part for n new leaves at ring positions start .. start + n - 1 (already stored): their sum-tree ancestors, replay size, pre_valid.  One workgroup;
// lvl[0][i] = leaf i's new priority (non-wrapping range), published by a barrier before the call.  Runs at the end of k_env_step, or -- with k_act_head doing the
// per-copy work -- as workgroup 0 of the observe launch.
__device__ __forceinline__ void env_tree_rebuild(const ReplayMeta& R, const int n, const long long start, float (*lvl)[1024], float (*rim)[2]) {
    const long long s0 = start % R.cap, e0 = (start + n - 1) % R.cap;
    const bool wrap = n >= R.cap || e0 < s0;
    if (!wrap) {
        // the n new leaves are one contiguous range: rebuild their ancestors out of LDS.  Only the two children at the rim of
        // each level's dirty range come from the (unchanged) tree; all of those are fetched up front in one round of loads.
        int nlev = 0; for (long long w = R.cap2; w > 1; w >>= 1) nlev++;
        if ((int)threadIdx.x < 2 * nlev) {
            const int d = threadIdx.x >> 1, right = threadIdx.x & 1;
            const long long lo = s0 >> d, hi = e0 >> d, width = R.cap2 >> d;
            float v = 0.0f;
            if (!right && (lo & 1)) v = R.tree[width + lo - 1];
            if (right && !(hi & 1)) v = R.tree[width + hi + 1];
            rim[d][right] = v;
        }
        __syncthreads();
        long long lo = s0, hi = e0, width = R.cap2; int cur = 0, d = 0;
        for (; d < nlev && hi - lo >= 2; d++) {
            const long long plo = lo >> 1, phi = hi >> 1;
            for (long long pp = plo + threadIdx.x; pp <= phi; pp += blockDim.x) {
                const long long c0 = 2 * pp, c1 = 2 * pp + 1;
                const float l = c0 < lo ? rim[d][0] : lvl[cur][c0 - lo];
                const float r = c1 > hi ? rim[d][1] : lvl[cur][c1 - lo];
                const float sum = l + r;
                lvl[cur ^ 1][pp - plo] = sum; R.tree[(width >> 1) + pp] = sum;
            }
            __syncthreads();
            lo = plo; hi = phi; width >>= 1; cur ^= 1;
        }
        if (threadIdx.x == 0) {
            // the dirty range is down to <= 2 nodes: the rest of the path to the root is a serial chain, walked without barriers
            float v0 = lvl[cur][0], v1 = hi > lo ? lvl[cur][1] : 0.0f;
            for (; d < nlev; d++) {
                const long long plo = lo >> 1, phi = hi >> 1;
                float n0, n1 = 0.0f;
                if (hi == lo) n0 = (lo & 1) ? rim[d][0] + v0 : v0 + rim[d][1];
                else if (plo == phi) n0 = v0 + v1;
                else { n0 = rim[d][0] + v0; n1 = v1 + rim[d][1]; }
                R.tree[(width >> 1) + plo] = n0; if (phi > plo) R.tree[(width >> 1) + phi] = n1;
                v0 = n0; v1 = n1; lo = plo; hi = phi; width >>= 1;
            }
        }
    } else {
        __syncthreads();
        long long lo[2], hi[2]; int nr = 1;
        if (n >= R.cap) { lo[0] = 0; hi[0] = R.cap - 1; }
        else { lo[0] = s0; hi[0] = R.cap - 1; lo[1] = 0; hi[1] = e0; nr = 2; }
        for (long long width = R.cap2; width > 1; width >>= 1) {
            for (int q = 0; q < nr; q++) {
                const long long a0 = (width + lo[q]) >> 1, a1 = (width + hi[q]) >> 1;
                for (long long node = a0 + threadIdx.x; node <= a1; node += blockDim.x) R.tree[node] = R.tree[2 * node] + R.tree[2 * node + 1];
                lo[q] >>= 1; hi[q] >>= 1;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) { long long s = R.state->size + n; R.state->size = s > R.cap ? R.cap : s; R.state->pre_valid = 0; }     // the tree changed: pre-drawn indices are stale
}

// ONE workgroup, thread i = env i (n <= 1024):
//   tick the step counter and the ring cursor; apply the reset the previous step left pending (src/solver.jl:99-132);
//   Q column of the env's observation from the head outputs (split-K slabs reduced on the fly; dueling (v + a) - mean(a),
//   src/dueling.jl:13-16) and its first-max argmax = action(policy, obs) (src/policy.jl:38-64);
//   eps-greedy (POMDPTools EpsGreedyPolicy: rand(rng) < eps ? rand(rng, actions) : greedy), act! (:89): transition, reward, terminal;
//   add_exp!(replay, exp, abs(exp.r)) (:91-94): metadata + leaf priority, then the sum-tree ancestors of the written leaves.
__global__ __launch_bounds__(1024) void k_env_step(EnvDev V, RolloutDev* rs, ActHeads Hd, ReplayMeta R) {
    const int n = V.n;
    const unsigned long long t_prev = (unsigned long long)rs->t, t = t_prev + 1;
    const long long start = (rs->widx + n) % R.cap;
    float eps = rs->eps_start - (float)t * ((rs->eps_start - rs->eps_stop) / rs->eps_steps);     // LinearDecaySchedule, fp32
    if (!(rs->eps_steps > 0.0f) || eps < rs->eps_stop) eps = rs->eps_stop;
    // head outputs of the acting forward -> LDS in one round of independent loads (split-K slabs included); the per-env
    // reduction below then adds them in canonical (ascending-slab) order out of LDS
    __shared__ float hl[ENV_HEAD_LDS];
    __shared__ float hbias[DQN_MAX_ACTIONS + 1];
    __shared__ float lvl[2][1024]; __shared__ float rim[48][2];      // sum-tree rebuild: dirty values of the current/next level, rim children per level
    for (int k = threadIdx.x; k <= V.nA; k += blockDim.x) hbias[k] = k < V.nA ? (Hd.adv.S > 1 ? Hd.adv.bias[k] : 0.0f) : (Hd.dueling && Hd.val.S > 1 ? Hd.val.bias[0] : 0.0f);
    const int Sa = Hd.adv.S < 1 ? 1 : Hd.adv.S, Sv = Hd.dueling ? (Hd.val.S < 1 ? 1 : Hd.val.S) : 0;
    const int per_col = Sa * V.nA + Sv;                     // floats per env: [a][slab] then [slab] of the value head
    const bool in_lds = (long long)per_col * n <= ENV_HEAD_LDS;
    if (in_lds) {
        // 4 independent loads in flight per thread (a runtime-trip-count loop would serialise the round trips)
        const int tot = per_col * n;
        for (int q0 = threadIdx.x; q0 < tot; q0 += 4 * blockDim.x) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int q = q0 + u * blockDim.x; v[u] = 0.0f;
                if (q < tot) {
                    const int i = q % n, k = q / n;             // consecutive lanes = consecutive columns: coalesced
                    if (k < Sa * V.nA) { const int a = k / Sa, sl = k - a * Sa; v[u] = Hd.adv.p[(size_t)sl * Hd.adv.per_s + (size_t)a * Hd.adv.ld + i]; }
                    else { const int sl = k - Sa * V.nA; v[u] = Hd.val.p[(size_t)sl * Hd.val.per_s + i]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int q = q0 + u * blockDim.x; if (q < tot) hl[q] = v[u]; }
        }
    }
    __syncthreads();
    if (in_lds) {
        // one thread per (head value, env): canonical ascending-slab sum (+ bias, activation) out of LDS, 8 reads in flight;
        // the result replaces the value's slab-0 slot, which no other thread reads
        for (int q = threadIdx.x; q < (V.nA + (Hd.dueling ? 1 : 0)) * n; q += blockDim.x) {
            const int k = q / n, i = q - k * n;
            const bool isval = k == V.nA; const int S = isval ? Sv : Sa; const int base = (isval ? Sa * V.nA : k * Sa) * n + i;
            float tot = hl[base]; int sl = 1;
            for (; sl + 8 <= S; sl += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = hl[base + (sl + u) * n];
#pragma unroll
                for (int u = 0; u < 8; u++) tot = tot + v[u];
            }
            for (; sl < S; sl++) tot = tot + hl[base + sl * n];
            const HeadSrc& h = isval ? Hd.val : Hd.adv;
            hl[base] = h.S > 1 ? act_f(tot + hbias[k], h.act) : tot;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { rs->t = (long long)t; rs->widx = start; }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (V.eval_mode && V.pending[i]) continue;              // evaluation: one episode per copy, finished copies idle
        if (V.pending[i]) {
            V.fin_eps[i] += 1; V.fin_reward[i] += (double)V.ep_reward[i]; V.ep_reward[i] = 0.0f; V.ep_step[i] = 0; V.pending[i] = 0;
            store_reset(V, i, t_prev);
        }
        if (V.kind == DQN_ENV_TESTMDP) *(uint32_t*)(V.tm_prev + i * 4) = *(const uint32_t*)(V.tm_s + i * 4);       // s of this transition = observation of the pre-step state
        else { V.gw_prev[i * 2] = V.gw_pos[i * 2]; V.gw_prev[i * 2 + 1] = V.gw_pos[i * 2 + 1]; }
        int a = 0;
        {   // Q column without per-lane arrays (they would live in scratch): advantages are re-read for the second pass
            auto adv_k = [&](int k) -> float {
                return in_lds ? hl[(k * Sa) * n + i] : head_val(Hd.adv, k, i);
            };
            float v = 0.0f, mean = 0.0f;
            if (Hd.dueling) {
                v = in_lds ? hl[(Sa * V.nA) * n + i] : head_val(Hd.val, 0, i);
                float sum = adv_k(0);
                for (int k = 1; k < V.nA; k++) sum = sum + adv_k(k);
                mean = sum / (float)V.nA;
            }
            float qbest = 0.0f;
            for (int k = 0; k < V.nA; k++) {
                const float ak = adv_k(k), qk = Hd.dueling ? (v + ak) - mean : ak;
                Hd.q_out[(size_t)i * V.nA + k] = qk;
                if (k == 0 || qk > qbest) { qbest = qk; a = k; }
            }
            Hd.amax[i] = a;
        }
        if (u01(env_rand(V.seed, t, i, 1u)) < eps) a = (int)(env_rand(V.seed, t, i, 2u) % (uint32_t)V.nA);
        float r; unsigned char done;
        if (V.kind == DQN_ENV_TESTMDP) {
            signed char* s = V.tm_s + i * 4;
            const bool was_second = s[3] == 2;                                // was_in_second(s), :62-64
            const signed char s0 = s[1], s1 = s[2], s2 = s[3];               // circshift(s, -1)
            const signed char last = a < 3 ? (signed char)(a + 1) : s2;       // a < 4 ? a : s_new[end-1]  (1-based), :69-74
            s[0] = s0; s[1] = s1; s[2] = s2; s[3] = last;
            r = (last == 1 ? -0.1f : (last == 2 ? 0.0f : 0.1f));
            if (was_second) r = r * -10.0f;                                   // :77-83
            V.tm_t[i] += 1; done = V.tm_t[i] >= V.max_time;                   // isterminal: t >= max_time, :85-87
        } else {
            int* p = V.gw_pos + i * 2; float rv = 0.0f;
            for (int k = 0; k < V.n_reward; k++) if (p[0] == V.reward_xy[k][0] && p[1] == V.reward_xy[k][1]) rv = V.reward_val[k];
            const bool at_reward = rv != 0.0f;
            const bool intended = u01(env_rand(V.seed, t, i, 3u)) < V.tprob;
            const int other = (int)(env_rand(V.seed, t, i, 4u) % 3u);
            const int eff = intended ? a : (a + 1 + other) % 4;
            const int dx = eff == 2 ? -1 : (eff == 3 ? 1 : 0), dy = eff == 0 ? 1 : (eff == 1 ? -1 : 0);
            const int nx = p[0] + dx, ny = p[1] + dy;
            if (!at_reward && nx >= 1 && nx <= V.size_x && ny >= 1 && ny <= V.size_y) { p[0] = nx; p[1] = ny; }
            r = rv; done = at_reward;
        }
        V.actions[i] = a; V.rewards[i] = r; V.dones[i] = done; V.ep_reward[i] += r; V.ep_step[i] += 1;
        if (V.eval_mode) {      // basic_evaluation: r_tot += rew in Float64; while !done && step <= max_episode_length (src/evaluation_policy.jl:27-34)
            V.fin_reward[i] += (double)r; V.pending[i] = (done || V.ep_step[i] > V.max_episode_length) ? 1 : 0; continue;
        }
        V.pending[i] = (done || V.ep_step[i] >= V.max_episode_length) ? 1 : 0;
        const long long slot = (start + i) % R.cap;
        R.a[slot] = a; R.r[slot] = r; R.done[slot] = done ? 1 : 0;
        const float td = V.prioritized ? fabsf(r) : 0.0f;                     // add_exp!(replay, exp, abs(exp.r)) / 0f0, src/solver.jl:91-94
        if (!(td + R.eps > 0.0f)) R.state->err = 1;
        const float pr = prio_f(td, R.eps, R.alpha);
        R.tree[R.cap2 + slot] = pr; lvl[0][i] = pr;
    }
    if (V.eval_mode) return;                                    // no add_exp! (uniform branch: eval_mode is a kernel argument)
    env_tree_rebuild(R, n, start, lvl, rim);
}
void launch_env_step(hipStream_t st, const EnvDev& V, RolloutDev* rs, const ActHeads& Hd, const ReplayMeta& R) {
    hipLaunchKernelGGL(k_env_step, dim3(1), dim3(1024), 0, st, V, rs, Hd, R);
}

// after k_env_step: the transition's rows go straight into the ring -- s = observation of the saved pre-step state, sp = observe(env)
// (src/solver.jl:90) -- and the NEXT step's observation (of the reset state if the episode just ended) becomes the policy input
// x[E][n].  Nothing is staged: every element is regenerated from the few bytes of env state (TestMDP images stay L2-resident).
// Workgroups [0, rows_blocks) walk the rows (blockIdx -> env, f fastest: coalesced rows), the rest walk x (i fastest).
template <int VEC>
__device__ __forceinline__ void obs_vec(const EnvDev& V, uint32_t sw, int px_, int py_, int f, float* of, unsigned char* ob) {
    if (V.kind == DQN_ENV_TESTMDP) {
        const int hw = V.H * V.W; int c = f / hw, px = f - c * hw;        // obs[.., c] = observations[s[end - c]]  (test/test_env.jl:56-58)
#pragma unroll
        for (int u = 0; u < VEC; u++) {
            const int sel = (int)((sw >> (8 * (3 - c))) & 0xffu) - 1;
            const unsigned char b = V.images[sel * hw + px];
            ob[u] = b; of[u] = (float)b / 255.0f;
            if (++px == hw) { px = 0; c++; }
        }
    } else {
#pragma unroll
        for (int u = 0; u < VEC; u++) { ob[u] = 0; of[u] = (float)((f + u) == 0 ? px_ : py_); }
    }
}
template <typename RowT, int VEC>
__global__ __launch_bounds__(256) void k_env_observe2(EnvDev V, const RolloutDev* __restrict__ rs0, RowT* __restrict__ s_rows, RowT* __restrict__ sp_rows,
                                                      long long cap, float* __restrict__ x, unsigned bx, unsigned rows_blocks, int grouped, int tree_wg, ReplayMeta R) {
    typedef RowT RowV __attribute__((ext_vector_type(VEC)));
    typedef float FloatV __attribute__((ext_vector_type(VEC)));
    const unsigned n = V.n, E = V.E;
    if (tree_wg && blockIdx.x == 0) {
        // (k_act_head did the per-copy part of add_exp!) workgroup 0, dispatched first: the new leaves' ancestors, beside the row writers instead of in front of them
        __shared__ float lvl[2][1024]; __shared__ float rim[48][2];
        const long long start = rs0->widx;                          // ticked by k_act_head: first ring position of this step's n experiences
        for (unsigned i = threadIdx.x; i < n; i += blockDim.x) { long long slot = start + i; if (slot >= cap) slot -= cap; lvl[0][i] = R.tree[R.cap2 + slot]; }
        env_tree_rebuild(R, (int)n, start, lvl, rim);
        return;
    }
    const unsigned blk = blockIdx.x - (unsigned)tree_wg;
    if (blk < rows_blocks) {
        const unsigned i = blk / bx, fv = (blk - i * bx) * blockDim.x + threadIdx.x;
        if (fv >= E / VEC) return;
        const unsigned f = fv * VEC;
        const RolloutDev* rs = rs0 + (grouped ? (i >> 2) : 0u);
        long long slot = rs->widx + i; if (slot >= cap) slot -= cap;
        uint32_t sw0 = 0x01010101u, sw1; int p0x = 0, p0y = 0, p1x, p1y;
        load_state(V, (int)i, &sw1, &p1x, &p1y);
        if (V.kind == DQN_ENV_TESTMDP) sw0 = *(const uint32_t*)(V.tm_prev + i * 4); else { p0x = V.gw_prev[i * 2]; p0y = V.gw_prev[i * 2 + 1]; }
        float of[VEC]; unsigned char ob[VEC]; RowT r[VEC];
        const size_t dst = (size_t)slot * E + f;
        obs_vec<VEC>(V, sw0, p0x, p0y, (int)f, of, ob);
#pragma unroll
        for (int u = 0; u < VEC; u++) r[u] = sizeof(RowT) == 1 ? (RowT)ob[u] : (RowT)of[u];
        __builtin_nontemporal_store(*(const RowV*)r, (RowV*)(s_rows + dst));      // ring rows are next read by a gather, steps later: streamed past the L2s (a kernel boundary otherwise writes back what the launch left dirty)
        obs_vec<VEC>(V, sw1, p1x, p1y, (int)f, of, ob);
#pragma unroll
        for (int u = 0; u < VEC; u++) r[u] = sizeof(RowT) == 1 ? (RowT)ob[u] : (RowT)of[u];
        __builtin_nontemporal_store(*(const RowV*)r, (RowV*)(sp_rows + dst));
    } else {
        const unsigned nv = n / VEC, qv = (blk - rows_blocks) * blockDim.x + threadIdx.x;
        if (qv >= nv * E) return;
        const unsigned f = qv / nv, i0 = (qv - f * nv) * VEC;             // VEC | n: the VEC elements share the feature
        const RolloutDev* rs = rs0 + (grouped ? (i0 >> 2) : 0u);          // (grouped: VEC == 4, the four copies are one group; every group's record holds the same t)
        const unsigned long long t = (unsigned long long)rs->t;
        float nx[VEC];
#pragma unroll
        for (int u = 0; u < VEC; u++) {
            uint32_t sw; int px, py, tm = 0; load_state(V, (int)(i0 + u), &sw, &px, &py);
            if (V.pending[i0 + u] && !V.eval_mode) reset_state(V, (int)(i0 + u), t, &sw, &tm, &px, &py);
            unsigned char b; nx[u] = obs_elem(V, sw, px, py, (int)f, &b);
        }
        { typedef float obs_f4 __attribute__((ext_vector_type(4)));      // the policy input is read by the very next launch, mostly from other XCDs: written through (sc1) as it is produced
          if constexpr (VEC == 4) { const obs_f4 v = {nx[0], nx[1], nx[2], nx[3]}; asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(x + (size_t)f * n + i0), "v"(v) : "memory"); }
          else *(FloatV*)(x + (size_t)f * n + i0) = *(const FloatV*)nx; }
    }
}
void launch_env_observe2(hipStream_t st, const EnvDev& V, const RolloutDev* rs, int rows_u8, void* s_rows, void* sp_rows, long long cap, float* x, int grouped, const ReplayMeta* tree) {
    const bool v4 = V.E % 4 == 0 && V.n % 4 == 0; const unsigned vec = v4 ? 4 : 1;
    const unsigned bx = (V.E / vec + 255) / 256, rows_blocks = V.eval_mode ? 0 : bx * V.n, x_blocks = (unsigned)(((size_t)V.n / vec * V.E + 255) / 256);
    const int tw = (tree && !V.eval_mode) ? 1 : 0; ReplayMeta R; memset(&R, 0, sizeof R); if (tw) R = *tree;
    if (grouped && !v4) grouped = 0;      // (callers only group when n % 4 == 0; the four-byte path reads record 0)
    const dim3 g(rows_blocks + x_blocks + tw), b(256);
    if (rows_u8) { if (v4) hipLaunchKernelGGL((k_env_observe2<unsigned char, 4>), g, b, 0, st, V, rs, (unsigned char*)s_rows, (unsigned char*)sp_rows, cap, x, bx, rows_blocks, grouped, tw, R);
                   else hipLaunchKernelGGL((k_env_observe2<unsigned char, 1>), g, b, 0, st, V, rs, (unsigned char*)s_rows, (unsigned char*)sp_rows, cap, x, bx, rows_blocks, grouped, tw, R); }
    else { if (v4) hipLaunchKernelGGL((k_env_observe2<float, 4>), g, b, 0, st, V, rs, (float*)s_rows, (float*)sp_rows, cap, x, bx, rows_blocks, grouped, tw, R);
           else hipLaunchKernelGGL((k_env_observe2<float, 1>), g, b, 0, st, V, rs, (float*)s_rows, (float*)sp_rows, cap, x, bx, rows_blocks, grouped, tw, R); }
}

// apply pending resets (end of a rollout call) or reset everything (dqn_envs_reset)
__global__ void k_env_reset_pending(EnvDev V, const RolloutDev* rs, int force_all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V.n) return;
    if (!force_all && !V.pending[i]) return;
    if (!force_all) { V.fin_eps[i] += 1; V.fin_reward[i] += (double)V.ep_reward[i]; } else V.dones[i] = 0;     // dones[] keeps the flags of the last act! (inspection)
    V.ep_reward[i] = 0.0f; V.ep_step[i] = 0; V.pending[i] = 0;
    store_reset(V, i, force_all ? 0ull : (unsigned long long)rs->t);
}
void launch_env_reset_pending(hipStream_t st, const EnvDev& V, const RolloutDev* rs, int force_all) {
    hipLaunchKernelGGL(k_env_reset_pending, dim3((V.n + 255) / 256), dim3(256), 0, st, V, rs, force_all);
}
