// act_head.hip -- the tail of the device env loop's ACTING step in one chip-filling launch (r06): the split-K reduce of the last hidden
// (dense) layer, the head level, Q = (val .+ adv) .- mean(adv) (src/dueling.jl:10), action(policy, obs) = first-max argmax (src/policy.jl:38-64), eps-greedy, act!
// and add_exp!'s per-experience part (src/solver.jl:89-94) -- what took k_reduce_multi + k_valu_multi + most of k_env_step: three launches, the last of them ONE
// workgroup walking five dependent round trips (profiles/history/r05_n_acting_step.txt: 4.7 + 4.8 + 11.4 us for 32 copies).
//
// Decomposition = k_red_head's (red_head.hip), one slot instead of three: a head output is a sum over plan chunks of 32 hidden rows, each chunk ONE k-ascending fma
// chain from +0, so a chunk is the unit that can move to another workgroup without changing a bit.  Workgroup (g, stream, c) -- ONE WAVE:
//   g       four env copies i = 4g .. 4g+3 (columns of the policy forward)
//   stream  advantage / plain Q head (0) or value head (1)
//   c       plan chunk: hidden rows 32c .. 32c+31
//   A. lanes 0-31: the chunk's slab pieces (16 bytes per row and slab) in one round of loads, summed in ascending slab order, + bias, activation (== k_reduce_multi
//      mode 0); lanes 32-63: the chunk's 32 rows of the head weights; lanes 0-3 also request their copy's env state -- the group's last arriver will want it
//   B. lanes (j, n): the chunk sum of output n of copy j -> `partials`, write-through
//   C. vmcnt(0), ONE relaxed agent-scope ticket on the group's counter (MI355X guide, Guideline 16 R1; the hand-off k_red_head uses)
//   D. the group's LAST arriver adds the chunk sums in ascending order (+ bias, activation), and lanes 0-3 run k_env_step's per-copy body for their copy: the pending
//      reset, the Q column + argmax, the eps-greedy draw, the transition, the replay metadata and the leaf priority.  It ticks the group's copy of the rollout
//      counters: RolloutDev is an array with one record per group, so that nobody reads a counter another workgroup of the same launch writes.
// (r06, measured and dropped: ONE workgroup of 16 waves per group -- the chunk sums handed over through LDS, no write-through partials, no ticket: three fabric round trips
//  fewer, but eight workgroups pulling 114 KB of slabs each: 10.6 vs 9.3 us, profiles/r06_ae_act_step_new.txt.)
// What is left of k_env_step -- the sum-tree ancestors of the n new leaves, replay size, pre_valid -- runs as workgroup 0 of the observe launch (envs.hip), beside
// the row writers instead of in front of them.  No workgroup waits for another one.  Same arithmetic, same order as the launches this replaces: bit-identical
// trajectories (tests/test_envs_gpu.py, both schedules: DQN_NO_ACT_HEAD=1 keeps the four-launch tail).
#include <algorithm>
#include "common.h"

typedef float f32x4h __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const GLOBAL_AS T* gp(const T* p) { return (const GLOBAL_AS T*)p; }
template <class T> __device__ __forceinline__ GLOBAL_AS T* gp(T* p) { return (GLOBAL_AS T*)p; }
// write-through store / L1-bypassing load of the hand-off payload (tracked by the compiler: __hip_atomic_* at agent scope)
__device__ __forceinline__ void st_wt(float* p, float v) { __hip_atomic_store(gp(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_wt(const float* p) { return __hip_atomic_load(gp(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ uint32_t ah_rand(unsigned long long seed, unsigned long long t, int env, uint32_t purpose) {      // == env_rand (envs.hip)
    uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)env, purpose};
    philox4x32_10((uint32_t)seed, (uint32_t)(seed >> 32), c);
    return c[0];
}
__device__ __forceinline__ float ah_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }

template <int SMAX>
__global__ __launch_bounds__(64) void k_act_head(const ActHeadArgs A) {
    __shared__ __attribute__((aligned(16))) float act[32 * 4];      // [32 rows][4 copies]  hidden activations of this chunk
    __shared__ float Wl[32 * 8];                                     // [32][N]              head weights of this chunk
    __shared__ float hv[4 * 9];                                      // [4][NO]              finished head outputs (last arriver)
    {      // the argument record's cache lines are requested at once (red_head.hip: a dependent chain of scalar loads otherwise)
        typedef const uint32_t __attribute__((address_space(4))) karg_u32;
        karg_u32* kp = (karg_u32*)__builtin_amdgcn_kernarg_segment_ptr(); uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < (int)((sizeof(ActHeadArgs) + 63) / 64); i++) x ^= kp[16 * i];
        asm volatile("" ::"s"(x));
    }
    const EnvDev& V = A.V; const ReplayMeta& R = A.R;
    const int lane = threadIdx.x;
    const int n = A.n, nA = A.nA, K = A.K, S = A.S, NC = K >> 5, G = n >> 2, nstream = A.nstream, NO = A.NO;
    const int g = (int)blockIdx.x % G, r_ = (int)blockIdx.x / G, stream = r_ / NC, c = r_ - stream * NC;
    // (per-stream fields: scalars selected by the uniform `stream`, never a vector load of a pointer)
    const float* part = stream ? A.st[1].part : A.st[0].part; const float* pbias = stream ? A.st[1].pbias : A.st[0].pbias; const float* Wg = stream ? A.st[1].W : A.st[0].W;
    const int N = stream ? A.st[1].N : A.st[0].N, pact = stream ? A.st[1].pact : A.st[0].pact, o0 = stream ? nA : 0;
    RolloutDev* const rs = A.rs + g;
    // ---- A. ONE round of loads, the same instructions on every lane (no divergent branch around a load: the compiler serialises the two sides' loads on register reuse --
    // first build: the head weights' round trip stood in front of half the slab loads, the slab round trip in front of the env state's; lanes 32-63 repeat lanes 0-31's slab
    // addresses, which costs nothing).  What only the group's last arriver uses is requested here as well -- the group's rollout record, the heads' biases, the four copies' env
    // state -- through a lane-dependent (opaque zero) address, so that the values stay in vector registers until they are used instead of being waited for on the spot
    int vz = 0; asm volatile("" : "+v"(vz));
    const RolloutDev* rsv = rs + vz;
    const long long rs_t = rsv->t, rs_widx = rsv->widx; const float rs_e0 = rsv->eps_start, rs_e1 = rsv->eps_stop, rs_es = rsv->eps_steps;
    float hb_v = 0.0f; int ha_v = 0;
    {
        const int t2 = lane < 4 * NO ? lane : 0, o = t2 % NO, st_ = o >= nA ? 1 : 0, nn = o - (st_ ? nA : 0);
        const float* hb0 = A.st[0].hbias; const float* hb1 = A.st[1].hbias;
        hb_v = *gp((st_ ? hb1 : hb0) + nn); ha_v = st_ ? A.st[1].hact : A.st[0].hact;
    }
    f32x4h sl[SMAX]; float pb, wv[4];
    const int i_env = 4 * g + (lane & 3);
    const int f = lane & 31, row = 32 * c + f, wtot = 32 * N;
    {
        const float* p = part + (A.pm ? ((size_t)g * K + row) * 4 : (size_t)row * n + 4 * g);
        const size_t per_s = (size_t)K * n;
#pragma unroll
        for (int s = 0; s < SMAX; s++) sl[s] = *gp(reinterpret_cast<const f32x4h*>(p + (size_t)(s < S ? s : S - 1) * per_s));
        pb = *gp(pbias + row);
#pragma unroll
        for (int u = 0; u < 4; u++) { int i = lane + 64 * u; if (i >= wtot) i = wtot - 1; wv[u] = *gp(Wg + (size_t)32 * c * N + i); }      // the chunk's 32 x N head weights (N <= 8)
    }
    // the copy's env state: two words whose addresses are selected, not branched on (TestMDP: the four state bytes, the time; SimpleGridWorld: x, y)
    const bool tmdp = V.kind == DQN_ENV_TESTMDP;
    const int* w0p = tmdp ? reinterpret_cast<const int*>(V.tm_s) + i_env : V.gw_pos + 2 * i_env;
    const int* w1p = tmdp ? V.tm_t + i_env : V.gw_pos + 2 * i_env + 1;
    const int w0_v = *gp(w0p), w1_v = *gp(w1p);
    unsigned char pend_v = *gp(V.pending + i_env); int eps_step_v = *gp(V.ep_step + i_env); float ep_rew_v = *gp(V.ep_reward + i_env);
    if (lane < 32) {
        f32x4h tot = sl[0];
#pragma unroll
        for (int s = 1; s < SMAX; s++) if (s < S) { tot.x = tot.x + sl[s].x; tot.y = tot.y + sl[s].y; tot.z = tot.z + sl[s].z; tot.w = tot.w + sl[s].w; }
        if (S > 1) { tot.x = act_f(tot.x + pb, pact); tot.y = act_f(tot.y + pb, pact); tot.z = act_f(tot.z + pb, pact); tot.w = act_f(tot.w + pb, pact); }      // (S == 1: `part` IS the finished activation)
        *reinterpret_cast<f32x4h*>(act + lane * 4) = tot;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = lane + 64 * u; if (i < wtot) Wl[i] = wv[u]; }
    __syncthreads();
    // ---- B. chunk sums: item (copy j, output nn), one k-ascending chain of 32 from +0
    float* Pg = A.partials + (size_t)g * 4 * NO * NC;      // [j][o][chunk]
    if (lane < 4 * N) {
        const int nn = lane % N, j = lane / N;
        float acc = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 32; k++) acc = fmaf(act[4 * k + j], Wl[k * N + nn], acc);
        st_wt(Pg + ((size_t)j * NO + o0 + nn) * NC + c, acc);
    }
    // ---- C. publish: the wave drains its write-through stores, then ONE relaxed agent-scope ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int last = 0;
    if (lane == 0) {
        const unsigned t = __hip_atomic_fetch_add(gp(A.tickets + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)(NC * nstream - 1)) { last = 1; __hip_atomic_store(gp(A.tickets + g), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // re-armed for the next launch
    }
    last = __builtin_amdgcn_readfirstlane(last);
    if (!last) return;
    // ---- D. the group's last arriver
    {
        const int t2 = lane < 4 * NO ? lane : 0;
        const float* pp = Pg + (size_t)t2 * NC;
        float pv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) pv[q] = ld_wt(pp + (q < NC ? q : NC - 1));
        float tot = pv[0];
#pragma unroll
        for (int q = 1; q < 16; q++) if (q < NC) tot = tot + pv[q];      // chunk sums added in ascending order
        if (lane < 4 * NO) hv[lane] = act_f(tot + hb_v, ha_v);
    }
    __syncthreads();
    if (lane >= 4) return;
    // ---- k_env_step's per-copy body (envs.hip), one lane per copy
    const int i = i_env;
    uint32_t sw_v = tmdp ? (uint32_t)w0_v : 0x01010101u; int tm_v = tmdp ? w1_v : 0, px_v = tmdp ? 0 : w0_v, py_v = tmdp ? 0 : w1_v;
    const unsigned long long t_prev = (unsigned long long)rs_t, t = t_prev + 1;
    const long long start = (rs_widx + n) % R.cap;
    float eps = rs_e0 - (float)t * ((rs_e0 - rs_e1) / rs_es);     // LinearDecaySchedule, fp32
    if (!(rs_es > 0.0f) || eps < rs_e1) eps = rs_e1;
    // this group's record.  Every wave of the group requested it at entry (phase A) and only this one -- the last arriver -- uses what it got; the others are past their ticket
    // or will find it taken, and drop the value: no workgroup ever USES a word another workgroup of the same launch writes (the next launch reads the ticked record behind a
    // kernel boundary)
    if (lane == 0) { rs->t = (long long)t; rs->widx = start; }
    if (V.eval_mode && pend_v) return;                              // evaluation: one episode per copy, finished copies idle
    if (pend_v) {
        V.fin_eps[i] += 1; V.fin_reward[i] += (double)ep_rew_v; ep_rew_v = 0.0f; eps_step_v = 0;
        if (V.kind == DQN_ENV_TESTMDP) { sw_v = 0x01010101u; tm_v = 1; }                                                  // initialstate, test/test_env.jl:46-52
        else { px_v = 1 + (int)(ah_rand(V.seed, t_prev, i, 5u) % (uint32_t)V.size_x); py_v = 1 + (int)(ah_rand(V.seed, t_prev, i, 6u) % (uint32_t)V.size_y); }
    }
    if (V.kind == DQN_ENV_TESTMDP) *reinterpret_cast<uint32_t*>(V.tm_prev + i * 4) = sw_v;       // s of this transition = observation of the pre-step state
    else { V.gw_prev[i * 2] = px_v; V.gw_prev[i * 2 + 1] = py_v; }
    int a = 0;
    {
        const float* h = hv + lane * NO;
        float v = 0.0f, mean = 0.0f;
        if (nstream > 1) {
            v = h[nA];
            float sum = h[0];
            for (int k = 1; k < nA; k++) sum = sum + h[k];
            mean = sum / (float)nA;
        }
        float qbest = 0.0f;
        for (int k = 0; k < nA; k++) {
            const float ak = h[k], qk = nstream > 1 ? (v + ak) - mean : ak;
            A.q_out[(size_t)i * nA + k] = qk;
            if (k == 0 || qk > qbest) { qbest = qk; a = k; }
        }
        A.amax[i] = a;
    }
    if (ah_u01(ah_rand(V.seed, t, i, 1u)) < eps) a = (int)(ah_rand(V.seed, t, i, 2u) % (uint32_t)nA);
    float r; unsigned char done;
    if (V.kind == DQN_ENV_TESTMDP) {
        const signed char s1 = (signed char)((sw_v >> 8) & 0xffu), s2 = (signed char)((sw_v >> 16) & 0xffu), s3 = (signed char)(sw_v >> 24);      // s[1], s[2], s[3] (little-endian)
        const bool was_second = s3 == 2;                                      // was_in_second(s), :62-64
        const signed char lastc = a < 3 ? (signed char)(a + 1) : s3;          // circshift(s, -1); a < 4 ? a : s_new[end-1]  (1-based), :69-74
        sw_v = (uint32_t)(unsigned char)s1 | ((uint32_t)(unsigned char)s2 << 8) | ((uint32_t)(unsigned char)s3 << 16) | ((uint32_t)(unsigned char)lastc << 24);
        r = (lastc == 1 ? -0.1f : (lastc == 2 ? 0.0f : 0.1f));
        if (was_second) r = r * -10.0f;                                       // :77-83
        tm_v += 1; done = tm_v >= V.max_time;                                 // isterminal: t >= max_time, :85-87
        *reinterpret_cast<uint32_t*>(V.tm_s + i * 4) = sw_v; V.tm_t[i] = tm_v;
    } else {
        float rv = 0.0f;
        for (int k = 0; k < V.n_reward; k++) if (px_v == V.reward_xy[k][0] && py_v == V.reward_xy[k][1]) rv = V.reward_val[k];
        const bool at_reward = rv != 0.0f;
        const bool intended = ah_u01(ah_rand(V.seed, t, i, 3u)) < V.tprob;
        const int other = (int)(ah_rand(V.seed, t, i, 4u) % 3u);
        const int eff = intended ? a : (a + 1 + other) % 4;
        const int dx = eff == 2 ? -1 : (eff == 3 ? 1 : 0), dy = eff == 0 ? 1 : (eff == 1 ? -1 : 0);
        const int nx = px_v + dx, ny = py_v + dy;
        if (!at_reward && nx >= 1 && nx <= V.size_x && ny >= 1 && ny <= V.size_y) { px_v = nx; py_v = ny; }
        V.gw_pos[i * 2] = px_v; V.gw_pos[i * 2 + 1] = py_v;
        r = rv; done = at_reward;
    }
    ep_rew_v += r; eps_step_v += 1;
    V.actions[i] = a; V.rewards[i] = r; V.dones[i] = done; V.ep_reward[i] = ep_rew_v; V.ep_step[i] = eps_step_v;
    if (V.eval_mode) {      // basic_evaluation: r_tot += rew in Float64; while !done && step <= max_episode_length (src/evaluation_policy.jl:27-34)
        V.fin_reward[i] += (double)r; V.pending[i] = (done || eps_step_v > V.max_episode_length) ? 1 : 0; return;
    }
    V.pending[i] = (done || eps_step_v >= V.max_episode_length) ? 1 : 0;
    const long long slot = (start + i) % R.cap;
    R.a[slot] = a; R.r[slot] = r; R.done[slot] = done ? 1 : 0;
    const float td = V.prioritized ? fabsf(r) : 0.0f;                         // add_exp!(replay, exp, abs(exp.r)) / 0f0, src/solver.jl:91-94
    if (!(td + R.eps > 0.0f)) R.state->err = 1;
    R.tree[R.cap2 + slot] = prio_f(td, R.eps, R.alpha);                       // the ancestors: workgroup 0 of the observe launch
}

// shapes this launch covers: groups of four copies, chunks of 32 hidden rows, <= 16 chunks and <= 16 slabs, head outputs of a copy group within one wave
bool act_head_ok(int n, int K, int S, int nA, int nstream, int N0, int N1) {
    const int NO = N0 + (nstream > 1 ? N1 : 0);
    return n % 4 == 0 && n >= 4 && n <= 1024 && K % 32 == 0 && K >= 32 && K <= 512 && S >= 1 && S <= 16 && nA >= 1 && nA <= 8 && N0 == nA && (nstream == 1 || N1 == 1) && 4 * NO <= 64;
}
void launch_act_head(hipStream_t st, const ActHeadArgs& a) {
    const unsigned grid = (unsigned)((a.n / 4) * (a.K / 32) * a.nstream);
    if (a.S <= 8) hipLaunchKernelGGL((k_act_head<8>), dim3(grid), dim3(64), 0, st, a);
    else hipLaunchKernelGGL((k_act_head<16>), dim3(grid), dim3(64), 0, st, a);
}
