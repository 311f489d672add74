// engine_drqn.hip -- C ABI of the recurrent path: EpisodeReplayBuffer (src/episode_replay.jl:3-95) and the DRQN train step
// (src/solver.jl:239-287); the launch program itself is built by engine_program.hip.
#include "engine.h"

// ---------------------------------------------------------------- DRQN: EpisodeReplayBuffer + recurrent batch_train!
extern "C" int dqn_episode_commit(dqn_engine_t* e) { if (!e) return fail("null engine handle");          // add_episode! (src/episode_replay.jl:54-60)
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    const int len = (int)e->ep_cur_len;
    e->ep_len_host[(size_t)e->ep_widx] = len;
    HIPCHK(hipMemcpyAsync(e->ep_len + e->ep_widx, &e->ep_len_host[(size_t)e->ep_widx], 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ep_widx = (e->ep_widx + 1) % e->ep_cap; if (e->ep_size < e->ep_cap) e->ep_size++;
    e->ep_cur_len = 0; return 0;
}
extern "C" int dqn_episode_add(dqn_engine_t* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done, int n) { if (!e) return fail("null engine handle");
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    const size_t row = (size_t)e->E * 4;
    for (int i = 0; i < n; i++) {                              // add_exp! (:46-52): push; the episode is stored when done
        if (a[i] < 0 || a[i] >= e->nA) return fail("action index %d out of range 0..%d", a[i], e->nA - 1);
        if (e->ep_cur_len < e->T) {                            // only the first trace_length transitions can ever be sampled (:82-92)
            const size_t slot = (size_t)e->ep_widx * e->T + (size_t)e->ep_cur_len;
            HIPCHK(hipMemcpyAsync((char*)e->ep_s + slot * row, (const char*)s + (size_t)i * row, row, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync((char*)e->ep_sp + slot * row, (const char*)sp + (size_t)i * row, row, hipMemcpyHostToDevice, e->stream));
            const unsigned char d8 = done[i] ? 1 : 0;
            HIPCHK(hipMemcpyAsync(e->ep_a + slot, a + i, 4, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipMemcpyAsync(e->ep_r + slot, r + i, 4, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->ep_done + slot, &d8, 1, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
        }
        e->ep_cur_len++;
        if (done[i] && dqn_episode_commit(e)) return -1;
    }
    return 0;
}
// ---------------------------------------------------------------- checkpoint / resume of the episode replay (SURVEY 8f-3 for config 4)
// episodes first .. first+n-1 in slot order: rows [n][T][E] (only the first trace_length transitions of an episode are ever stored, see
// dqn_episode_add), a / r / done [n][T], len [n] (the episode's TRUE length, which the start draw uses).  An episode still being collected
// (dqn_episode_add without its terminal transition) is not part of a checkpoint: commit or finish it first.
extern "C" int dqn_episode_export(dqn_engine_t* e, int64_t first, int64_t n, float* s, float* sp, int32_t* a, float* r, uint8_t* done, int32_t* len) { if (!e) return fail("null engine handle");
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    if (first < 0 || n < 0 || first + n > e->ep_size) return fail("BoundsError: episodes %lld..%lld outside 0..%lld", (long long)first, (long long)(first + n - 1), (long long)e->ep_size - 1);
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t T = (size_t)e->T, row = (size_t)e->E * 4, off = (size_t)first * T, cnt = (size_t)n * T;
    if (s) HIPCHK(hipMemcpy(s, (const char*)e->ep_s + off * row, cnt * row, hipMemcpyDeviceToHost));
    if (sp) HIPCHK(hipMemcpy(sp, (const char*)e->ep_sp + off * row, cnt * row, hipMemcpyDeviceToHost));
    if (a) HIPCHK(hipMemcpy(a, e->ep_a + off, cnt * 4, hipMemcpyDeviceToHost));
    if (r) HIPCHK(hipMemcpy(r, e->ep_r + off, cnt * 4, hipMemcpyDeviceToHost));
    if (done) HIPCHK(hipMemcpy(done, e->ep_done + off, cnt, hipMemcpyDeviceToHost));
    if (len) for (int64_t i = 0; i < n; i++) len[i] = e->ep_len_host[(size_t)(first + i)];
    return 0;
}
// replaces the whole episode replay: n <= capacity committed episodes go to slots 0..n-1; the ring cursor becomes n mod capacity (override it,
// and the host sampler's draw counter, with dqn_set_counters: widx, sample_ctr)
extern "C" int dqn_episode_import(dqn_engine_t* e, int64_t n, const float* s, const float* sp, const int32_t* a, const float* r, const uint8_t* done, const int32_t* len) { if (!e) return fail("null engine handle");
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    if (n < 0 || n > e->ep_cap) return fail("import of %lld episodes into an episode replay of capacity %lld", (long long)n, (long long)e->ep_cap);
    const size_t T = (size_t)e->T, row = (size_t)e->E * 4, cnt = (size_t)n * T;
    for (int64_t i = 0; i < n; i++) {
        if (len[i] < 1) return fail("episode %lld: length %d < 1", (long long)i, len[i]);
        const int m = len[i] < e->T ? len[i] : e->T;
        for (int t = 0; t < m; t++) if (a[(size_t)i * T + t] < 0 || a[(size_t)i * T + t] >= e->nA) return fail("action index %d out of range 0..%d", a[(size_t)i * T + t], e->nA - 1);
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->ep_s, s, cnt * row, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(e->ep_sp, sp, cnt * row, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->ep_a, a, cnt * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(e->ep_r, r, cnt * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->ep_done, done, cnt, hipMemcpyHostToDevice));
    for (int64_t i = 0; i < n; i++) e->ep_len_host[(size_t)i] = len[i];
    HIPCHK(hipMemcpy(e->ep_len, e->ep_len_host.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    e->ep_size = n; e->ep_widx = n % e->ep_cap; e->ep_cur_len = 0; e->ep_perm.clear();
    return 0;
}
extern "C" int dqn_episode_count(dqn_engine_t* e, int64_t* cur, int64_t* cap) { if (!e) return fail("null engine handle"); NEED_REC(e); if (cur) *cur = e->ep_size; if (cap) *cap = e->ep_cap; return 0; }
static int drqn_check(dqn_engine* e, const int64_t* ep_idx, const int32_t* ep_start) {
    if (e->ep_size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    for (int b = 0; b < e->B; b++) {
        if (ep_idx[b] < 0 || ep_idx[b] >= e->ep_size) return fail("BoundsError: episode index %lld outside 0..%lld", (long long)ep_idx[b], (long long)e->ep_size - 1);
        const int len = e->ep_len_host[(size_t)ep_idx[b]];
        if (len > 0 && (ep_start[b] < 0 || ep_start[b] >= len)) return fail("episode start %d outside 0..%d", ep_start[b], len - 1);
    }
    return 0;
}
static int drqn_upload_draws(dqn_engine* e, const int64_t* ep_idx, const int32_t* ep_start) {
    HIPCHK(hipMemcpyAsync(e->ep_idx, ep_idx, (size_t)e->B * 8, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipMemcpyAsync(e->ep_start, ep_start, (size_t)e->B * 4, hipMemcpyHostToDevice, e->stream));
    return 0;
}
extern "C" int dqn_episode_get_batch(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* s, int32_t* a, float* r, float* sp, float* done, int32_t* mask) { if (!e) return fail("null engine handle");
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    if (drqn_check(e, ep_idx, ep_start) || drqn_upload_draws(e, ep_idx, ep_start)) return -1;
    EpGatherArgs g; g.ep_s = e->ep_s; g.ep_sp = e->ep_sp; g.ep_a = e->ep_a; g.ep_r = e->ep_r; g.ep_done = e->ep_done; g.ep_len = e->ep_len; g.ep_idx = e->ep_idx; g.ep_start = e->ep_start;
    g.E = e->E; g.B = e->B; g.T = e->T; g.x0 = e->x0; g.a_out = e->r_a; g.r_out = e->r_r; g.done_out = e->r_done; g.mask_out = e->r_mask;
    launch_gather_episodes(e->stream, g);
    const int TB = e->Bc, E = e->E;
    std::vector<float> x((size_t)E * 2 * TB), rr(TB), dd(TB), mm(TB); std::vector<int> aa(TB);
    HIPCHK(hipMemcpyAsync(x.data(), e->x0, x.size() * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipMemcpyAsync(aa.data(), e->r_a, TB * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(rr.data(), e->r_r, TB * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipMemcpyAsync(dd.data(), e->r_done, TB * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(mm.data(), e->r_mask, TB * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
    for (int k = 0; k < TB; k++) {      // device arena is [feature][column]; the seam returns [T][B][obs]
        if (s) for (int f = 0; f < E; f++) s[(size_t)k * E + f] = x[(size_t)f * 2 * TB + k];
        if (sp) for (int f = 0; f < E; f++) sp[(size_t)k * E + f] = x[(size_t)f * 2 * TB + TB + k];
        if (a) a[k] = aa[k]; if (r) r[k] = rr[k]; if (done) done[k] = dd[k]; if (mask) mask[k] = (int32_t)mm[k];
    }
    return 0;
}
// sample(rng, 1:n, B, replace=false); ep_start = rand(rng, 1:length(ep))  (src/episode_replay.jl:75,81) -- host-side SplitMix draws
static int drqn_host_draws(dqn_engine* e, std::vector<int64_t>& di, std::vector<int32_t>& ds) {
    if (e->ep_size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    auto next = [&]() { uint64_t z = (e->drqn_draws += 0x9E3779B97F4A7C15ull) ^ e->hp.seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); };
    // partial Fisher-Yates on a persistent identity permutation: B swaps, read the prefix, undo the swaps (O(B) per step, same
    // draws as shuffling a fresh 0..n-1 vector)
    std::vector<int64_t>& perm = e->ep_perm;
    if ((long long)perm.size() != e->ep_size) { perm.resize((size_t)e->ep_size); for (size_t i = 0; i < perm.size(); i++) perm[i] = (int64_t)i; }
    std::vector<size_t> js((size_t)e->B);
    for (int b = 0; b < e->B; b++) { js[b] = b + (size_t)(next() % (perm.size() - b)); std::swap(perm[b], perm[js[b]]); }
    di.assign(perm.begin(), perm.begin() + e->B); ds.resize(e->B);
    for (int b = e->B - 1; b >= 0; b--) std::swap(perm[b], perm[js[b]]);
    for (int b = 0; b < e->B; b++) { const int len = e->ep_len_host[(size_t)di[b]]; ds[b] = len > 0 ? (int32_t)(next() % (uint64_t)len) : 0; }
    return 0;
}
// fused recurrent step: the draws of a step go into ITS slot of the mapped host buffer (the slot is a launch parameter of the step's kernel, fixed per graph node): no H2D
// copy launch, no device-side counter, and the host never waits for the GPU except for the previous launch of the same graph instance (two instances alternate)
static void drqn_ring_put(dqn_engine* e, int slot, const int64_t* ep_idx, const int32_t* ep_start) {
    const size_t o = (size_t)slot * e->B;
    memcpy(e->draw_idx_h + o, ep_idx, (size_t)e->B * 8);
    for (int b = 0; b < e->B; b++) {      // what the kernel needs of (length, start): the number of rows the prefix copy delivers (src/episode_replay.jl:82-92)
        const int len = e->ep_len_host[(size_t)ep_idx[b]]; int np = (len < e->T ? len : e->T) - ep_start[b]; e->draw_start_h[o + b] = np < 0 ? 0 : np;
    }
}
extern "C" int dqn_train_step_drqn(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* loss, float* grad_norm) { if (!e) return fail("null engine handle");
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    std::vector<int64_t> di; std::vector<int32_t> ds;
    if (!ep_idx) { if (drqn_host_draws(e, di, ds)) return -1; ep_idx = di.data(); ep_start = ds.data(); }
    if (drqn_check(e, ep_idx, ep_start)) return -1;
    if (build_program(e)) return -1;
    const int par = e->drqn_fused ? e->drqn_one_par : 0, evi = 2 + par;
    if (e->drqn_fused) {
        if (e->draw_ev_used[evi]) HIPCHK(hipEventSynchronize(e->draw_ev[evi]));      // the previous launch of THIS instance has read its slot
        drqn_ring_put(e, 16 + par, ep_idx, ep_start); __atomic_thread_fence(__ATOMIC_RELEASE);
        e->drqn_slot_next = 16 + par;
    } else if (drqn_upload_draws(e, ep_idx, ep_start)) return -1;
    const bool xch = e->world > 1 || (e->comm && e->force_comm);      // replicas (or DQN_FORCE_ALLREDUCE at world 1, tests): all-reduce of the materialised gradient between backward and Adam
    if (e->hp.use_graph && !e->profiling && !xch) {
        if (!e->g_drqn[par] && capture(e, false, PH_ALL, &e->g_drqn[par])) return -1;
        HIPCHK(hipGraphLaunch(e->g_drqn[par], e->stream));
    } else { enqueue_step(e, false, PH_PRE); if (xch && exchange_grads(e)) return -1; enqueue_step(e, false, PH_POST); if (e->launch_failed) { e->launch_failed = false; return fail("the recurrent step could not be enqueued (dynamic LDS refused)"); } }
    if (e->drqn_fused) { HIPCHK(hipEventRecord(e->draw_ev[evi], e->stream)); e->draw_ev_used[evi] = true; e->drqn_one_par ^= 1; }
    if (loss || grad_norm) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
// n sampled recurrent steps back to back (dqn_train_steps on a recurrent engine).  Fused step: runs of DRQN_GROUP steps replay as ONE graph whose k-th step reads slot
// (instance * 8 + k) -- the draws of the whole run are written first, then one hipGraphLaunch.
int drqn_train_steps(dqn_engine* e, int n, float* loss, float* grad_norm) {
    enum { DRQN_GROUP = 8 };
    if (build_program(e)) return -1;
    const bool grouped = e->drqn_fused && e->hp.use_graph && !e->profiling && e->world == 1 && !(e->comm && e->force_comm);
    std::vector<int64_t> di; std::vector<int32_t> ds;
    for (int i = 0; i < n;) {
        if (grouped && n - i >= DRQN_GROUP) {
            const int par = e->drqn_grp_par;
            if (e->draw_ev_used[par]) HIPCHK(hipEventSynchronize(e->draw_ev[par]));
            for (int k = 0; k < DRQN_GROUP; k++) { if (drqn_host_draws(e, di, ds) || drqn_check(e, di.data(), ds.data())) return -1; drqn_ring_put(e, par * DRQN_GROUP + k, di.data(), ds.data()); }
            __atomic_thread_fence(__ATOMIC_RELEASE);
            if (!e->g_drqn_k[par]) { e->drqn_slot_next = par * DRQN_GROUP; if (capture(e, false, PH_ALL, &e->g_drqn_k[par], DRQN_GROUP)) return -1; }
            HIPCHK(hipGraphLaunch(e->g_drqn_k[par], e->stream));
            HIPCHK(hipEventRecord(e->draw_ev[par], e->stream)); e->draw_ev_used[par] = true; e->drqn_grp_par ^= 1;
            i += DRQN_GROUP; continue;
        }
        if (dqn_train_step_drqn(e, nullptr, nullptr, nullptr, nullptr)) return -1;
        i++;
    }
    if (loss || grad_norm) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
extern "C" int dqn_reset_state(dqn_engine_t* e) { if (!e) return fail("null engine handle");             // resetstate!(policy) (src/policy.jl:32-34)
    HIPCHK(hipSetDevice(e->device));
    if (!e->hp.recurrence) return 0;
    return policy_state(e, e->pol_state_n > 0 ? e->pol_state_n : 1, true);
}
extern "C" int dqn_get_hidden(dqn_engine_t* e, float* hc, size_t n) { if (!e) return fail("null engine handle");   // hiddenstates(m) (src/helpers.jl:61-63): per LSTM layer h then c, [out][streams]
    HIPCHK(hipSetDevice(e->device)); size_t off = 0;
    if (e->hp.recurrence && e->pol_state_n == 0 && policy_state(e, 1, true)) return -1;      // no forward yet: one stream at state0
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_state_n; if (off + 2 * m > n) return fail("get_hidden: buffer too small");
        HIPCHK(hipMemcpyAsync(hc + off, e->pol_h[i][e->pol_flip], m * 4, hipMemcpyDeviceToHost, e->stream)); off += m;
        HIPCHK(hipMemcpyAsync(hc + off, e->pol_c[i][e->pol_flip], m * 4, hipMemcpyDeviceToHost, e->stream)); off += m;
    }
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
extern "C" int dqn_set_hidden(dqn_engine_t* e, const float* hc, size_t n) { if (!e) return fail("null engine handle");   // sethiddenstates!(m, hs) (src/helpers.jl:71-79)
    HIPCHK(hipSetDevice(e->device)); size_t off = 0;
    if (e->hp.recurrence && e->pol_state_n == 0 && policy_state(e, 1, true)) return -1;
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_state_n; if (off + 2 * m > n) return fail("set_hidden: buffer too small");
        HIPCHK(hipMemcpyAsync(e->pol_h[i][e->pol_flip], hc + off, m * 4, hipMemcpyHostToDevice, e->stream)); off += m;
        HIPCHK(hipMemcpyAsync(e->pol_c[i][e->pol_flip], hc + off, m * 4, hipMemcpyHostToDevice, e->stream)); off += m;
    }
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}

// TIMING PROBE (tools/drqn_phases.py; not part of the public header): s_memrealtime stamps (100 MHz) of workgroup 0 at the phase boundaries of the fused recurrent
// step, recorded when the engine was created under DQN_DRQN_STAMPS=1
extern "C" __attribute__((visibility("default"))) int dqn_debug_drqn_stamps(dqn_engine_t* e, uint64_t* out, size_t n) { if (!e) return fail("null engine handle");
    if (!e->drqn_stamps) return fail("no stamps: create the engine under DQN_DRQN_STAMPS=1 and run a fused recurrent step first");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(out, e->drqn_stamps, std::min<size_t>(n, 32) * 8, hipMemcpyDeviceToHost)); return 0;
}
