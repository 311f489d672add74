// engine_envs.hip -- C ABI of the device-resident environments: dqn_envs_create / dqn_rollout / dqn_evaluate /
// dqn_envs_peek; kernels in envs.hip.
#include "engine.h"

// ---------------------------------------------------------------- vectorised environments on the device
static void free_env_arrays(EnvDev& V) {      // the per-copy arrays of an evaluation env set (images and spec are shared with the training set)
    hipFree(V.tm_s); hipFree(V.tm_prev); hipFree(V.tm_t); hipFree(V.gw_pos); hipFree(V.gw_prev);
    hipFree(V.actions); hipFree(V.rewards); hipFree(V.dones); hipFree(V.pending); hipFree(V.ep_reward); hipFree(V.ep_step); hipFree(V.fin_eps); hipFree(V.fin_reward);
    memset(&V, 0, sizeof V);
}
void free_envs(dqn_engine* e) {
    EnvDev& V = e->env;
    hipFree(e->env_images); hipFree(V.tm_s); hipFree(V.tm_prev); hipFree(V.tm_t); hipFree(V.gw_pos); hipFree(V.gw_prev); hipFree(e->roll);
    hipFree(V.actions); hipFree(V.rewards); hipFree(V.dones); hipFree(V.pending); hipFree(V.ep_reward); hipFree(V.ep_step); hipFree(V.fin_eps); hipFree(V.fin_reward);
    e->env_images = nullptr; e->roll = nullptr; memset(&V, 0, sizeof V); e->has_envs = false;
    free_env_arrays(e->eval_env); hipFree(e->eval_roll); e->eval_roll = nullptr; e->eval_n = 0;
    drop_act(e, e->act); drop_act(e, e->evalp);
}
extern "C" int dqn_envs_create(dqn_engine_t* e, const dqn_env_spec* sp) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("device environments drive the feed-forward path (recurrence = false)");
    if (sp->n_envs < 1 || sp->n_envs > std::min<long long>(1024, e->cap)) return fail("n_envs must be in 1..min(1024, replay capacity)");
    if (sp->max_episode_length < 1) return fail("max_episode_length must be >= 1");
    HIPCHK(hipStreamSynchronize(e->stream)); free_envs(e);
    EnvDev& V = e->env; const int n = sp->n_envs;
    V.kind = sp->kind; V.n = n; V.E = e->E; V.nA = e->nA; V.max_episode_length = sp->max_episode_length; V.seed = sp->seed; V.prioritized = e->hp.prioritized_replay ? 1 : 0;
    const bool u8 = e->hp.obs_dtype == DQN_OBS_U8;
    if (sp->kind == DQN_ENV_TESTMDP) {
        if (!sp->images) return fail("TestMDP needs its three images");
        if (sp->o_stack < 1 || sp->o_stack > 4 || sp->o_stack != e->hp.obs_c) return fail("TestMDP: o_stack (%d) must equal obs_c (%d) and be <= 4", sp->o_stack, e->hp.obs_c);
        if (e->nA != 4) return fail("TestMDP has 4 actions, the network has %d outputs", e->nA);
        V.H = e->hp.obs_h; V.W = e->hp.obs_w; V.max_time = sp->max_time;
        const size_t ib = (size_t)3 * V.H * V.W;
        DM(e->env_images, ib); HIPCHK(hipMemcpy(e->env_images, sp->images, ib, hipMemcpyHostToDevice)); V.images = e->env_images;
        DM(V.tm_s, (size_t)n * 4); DM(V.tm_prev, (size_t)n * 4); DM(V.tm_t, n);
    } else if (sp->kind == DQN_ENV_GRIDWORLD) {
        if (u8) return fail("SimpleGridWorld observations are Float32[x, y]: use obs_dtype f32");
        if (e->E != 2 || e->nA != 4) return fail("SimpleGridWorld: observation has 2 elements and there are 4 actions (network: %d in, %d out)", e->E, e->nA);
        if (sp->n_reward_cells < 0 || sp->n_reward_cells > 8) return fail("at most 8 reward cells");
        V.size_x = sp->size_x; V.size_y = sp->size_y; V.tprob = sp->tprob; V.n_reward = sp->n_reward_cells;
        for (int k = 0; k < V.n_reward; k++) { V.reward_xy[k][0] = sp->reward_xy[k][0]; V.reward_xy[k][1] = sp->reward_xy[k][1]; V.reward_val[k] = sp->reward_val[k]; }
        DM(V.gw_pos, (size_t)n * 2); DM(V.gw_prev, (size_t)n * 2);
    } else return fail("unknown environment kind %d", sp->kind);
    DM(V.actions, n); DM(V.rewards, n); DM(V.dones, n); DM(V.pending, n); DM(V.ep_reward, n); DM(V.ep_step, n); DM(V.fin_eps, n); DM(V.fin_reward, n); DM(e->roll, DQN_ROLL_RECORDS);
    HIPCHK(hipMemsetAsync(V.fin_eps, 0, (size_t)n * 8, e->stream)); HIPCHK(hipMemsetAsync(V.fin_reward, 0, (size_t)n * 8, e->stream));
    HIPCHK(hipMemsetAsync(V.actions, 0, (size_t)n * 4, e->stream)); HIPCHK(hipMemsetAsync(V.rewards, 0, (size_t)n * 4, e->stream));
    HIPCHK(hipMemsetAsync(e->roll, 0, sizeof(RolloutDev) * DQN_ROLL_RECORDS, e->stream));
    e->has_envs = true;
    return dqn_envs_reset(e);
}
extern "C" int dqn_envs_reset(dqn_engine_t* e) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create");
    launch_env_reset_pending(e->stream, e->env, e->roll, 1);
    return 0;
}
// the acting program: online net forward on the n columns of pol_x (batch-innermost), then Q columns + first-max argmax
// (action(policy, obs), src/policy.jl:38-64) -- the same tiled kernels and the same plan as the train step, compiled once per n
static int build_act_program(dqn_engine* e, dqn_engine::ActProg& ap, const EnvDev& V, RolloutDev* rs) {
    const int n = V.n;
    if (ap.n == n) return 0;
    if (policy_ws(e, std::max(n, std::max(e->env.n, e->eval_n)))) return -1;      // one workspace serves both env sets (no realloc when they alternate)
    drop_act(e, ap);
    e->prog_names.reserve(512);
    e->sink = &ap.steps; e->alloc_sink = &ap.allocs;
    const bool mf = e->hp.use_mfma != 0;
    std::vector<std::vector<int>> levels; std::vector<int> val, adv;
    for (int i = 0; i < e->nl; i++) { if (e->L[i].stream == DQN_STREAM_BASE) levels.push_back({i}); else if (e->L[i].stream == DQN_STREAM_VAL) val.push_back(i); else adv.push_back(i); }
    for (size_t j = 0; j < std::max(val.size(), adv.size()); j++) { std::vector<int> lv; if (j < val.size()) lv.push_back(val[j]); if (j < adv.size()) lv.push_back(adv[j]); levels.push_back(lv); }
    const float* P = e->p_on;
    HeadSrc head[DQN_MAX_LAYERS];
    // the fused tail (act_head.hip): reduce of the heads' producers + heads + Q / argmax + eps-greedy + act! + add_exp!'s per-experience part in ONE launch, where the shapes allow
    // (the conditions of the train step's fused reduce + head launch, engine_program.hip); else k_reduce_multi + the heads' forward + k_env_step
    const int lq = e->hp.dueling ? e->last_adv : e->last_base, lvh = e->hp.dueling ? e->last_val : -1;
    bool use_ah = !e->opt.no_act_head && levels.size() >= 2; int ah_pa = -1, ah_pv = -1, ah_S = 0; bool ah_pm = false; const float* ah_part[2] = {nullptr, nullptr};
    if (use_ah) {
        const LayerDev& La = e->L[lq]; ah_pa = La.src; ah_pv = lvh >= 0 ? e->L[lvh].src : -1;
        bool ok = La.kind == DQN_LAYER_DENSE && ah_pa >= 0 && (lvh < 0 || (e->L[lvh].kind == DQN_LAYER_DENSE && ah_pv >= 0 && ah_pv != ah_pa));
        auto in_lv = [&](const std::vector<int>& v, int l) { for (int x : v) if (x == l) return true; return false; };
        const auto& hl = levels.back(); const auto& pl = levels[levels.size() - 2];
        if (ok) ok = (int)hl.size() == (lvh >= 0 ? 2 : 1) && in_lv(hl, lq) && (lvh < 0 || in_lv(hl, lvh)) && (int)pl.size() == (lvh >= 0 ? 2 : 1) && in_lv(pl, ah_pa) && (lvh < 0 || in_lv(pl, ah_pv));
        if (ok) {
            const LayerDev& Pa = e->L[ah_pa]; ah_S = dqn_nchunks(Pa.K, Pa.fwd_kc);
            ok = Pa.kind == DQN_LAYER_DENSE && Pa.N == La.K && dqn_chunk_len(La.K, La.fwd_kc) == 32 && dqn_nchunks(La.K, La.fwd_kc) * 32 == La.K;
            if (ok && lvh >= 0) { const LayerDev& Pv = e->L[ah_pv]; const LayerDev& Lv = e->L[lvh];
                ok = Pv.kind == DQN_LAYER_DENSE && Pv.N == Pa.N && dqn_nchunks(Pv.K, Pv.fwd_kc) == ah_S && dqn_chunk_len(Lv.K, Lv.fwd_kc) == 32 && Lv.K == La.K; }
            if (ok) ok = act_head_ok(n, La.K, ah_S, e->nA, lvh >= 0 ? 2 : 1, La.N, lvh >= 0 ? e->L[lvh].N : 0);
        }
        use_ah = ok;
    }
    for (size_t li = 0; li < levels.size(); li++) {
        const auto& lv = levels[li]; const bool last = li + 1 == levels.size();
        if (use_ah && last) break;                          // the head level runs inside k_act_head
        const bool ah_prod = use_ah && li + 2 == levels.size();      // this level = the heads' producers: their slabs stay unreduced
        struct Prob { int l; const float* X; float *Y, *part; int S; };
        std::vector<Prob> pr;
        for (int l : lv) { const LayerDev& L = e->L[l]; Prob q; q.l = l; q.X = L.src < 0 ? e->pol_x : e->pol_act[L.src]; q.Y = e->pol_act[l]; q.S = dqn_nchunks(L.K, L.fwd_kc);
                           q.part = q.S > 1 ? palloc(e, (size_t)q.S * L.out_feat * n) : nullptr; pr.push_back(q); }
        bool geo = true; for (int l : lv) geo = geo && same_geo(e->L[lv[0]], e->L[l]);
        std::vector<bool> done(pr.size(), false);
        auto emit_gemm = [&](const std::vector<int>& ids, const char* name) {
            const LayerDev L = e->L[pr[ids[0]].l]; const int np = (int)ids.size();
            struct A { const float *W[4], *bias[4], *X[4]; int ldx[4], col0[4], ncols[4]; float* out[4]; } a;
            for (int i = 0; i < np; i++) { const Prob& q = pr[ids[i]]; const LayerDev& Lq = e->L[q.l]; a.W[i] = P + Lq.w_off; a.bias[i] = P + Lq.b_off; a.X[i] = q.X; a.ldx[i] = n; a.col0[i] = 0; a.ncols[i] = n; a.out[i] = q.S > 1 ? q.part : q.Y; }
            // slabs only k_act_head reads are written piece-major (GFwdProb::pm), when ONE launch produces them all
            const int pm = (ah_prod && ah_S > 1 && np == (int)pr.size() && !e->opt.no_rh_pm) ? 1 : 0; if (pm) ah_pm = true;
            ap.steps.push_back({name, [=](dqn_engine* en) { launch_gemm_fwd(en->stream, L, np, a.W, a.bias, a.X, a.ldx, a.col0, a.ncols, a.out, nullptr, pm); }});
            for (int id : ids) done[id] = true;
        };
        if (mf && pr.size() <= 4) {
            int ldx[4], c0[4], nc[4]; std::vector<int> all;
            for (size_t i = 0; i < pr.size(); i++) { all.push_back((int)i); ldx[i] = n; c0[i] = 0; nc[i] = n; }
            if (geo && gemm_fwd_eligible(e->L[lv[0]], (int)pr.size(), ldx, c0, nc)) emit_gemm(all, pname(e, "act_fwd", e->L[lv[0]].kind, lv[0]));
            else for (size_t i = 0; i < pr.size(); i++) if (gemm_fwd_eligible(e->L[pr[i].l], 1, ldx, c0, nc)) emit_gemm({(int)i}, pname(e, "act_fwd", e->L[pr[i].l].kind, pr[i].l));
        }
        std::vector<VTask> pend;
        for (size_t i = 0; i < pr.size(); i++) {
            if (done[i]) continue;
            const Prob q = pr[i]; const LayerDev L = e->L[q.l];
            if (mf && mfma_fwd_ok(L, n)) ap.steps.push_back({pname(e, "act_fwd", L.kind, q.l), [=](dqn_engine* en) { launch_mfma_fwd(en->stream, L, P, q.X, n, 0, n, q.Y, q.part, false); }});
            else { VTask t; memset(&t, 0, sizeof t); t.kind = 0; t.L = L; t.P = P; t.X = q.X; t.ldx = n; t.col0 = 0; t.ncols = n; t.S = q.S; t.kc = dqn_chunk_len(L.K, L.fwd_kc); t.out = q.S > 1 ? q.part : q.Y; add_valu(e, pend, t); }
        }
        flush_valu(e, pend, pname(e, "act_fwd_valu", e->L[lv[0]].kind, lv[0]));
        std::vector<RSeg> segs;
        for (const Prob& q : pr) {
            const LayerDev& L = e->L[q.l];
            HeadSrc h; h.p = q.Y; h.ld = n; h.S = 1; h.per_s = 0; h.bias = P + L.b_off; h.act = L.act;
            if (ah_prod) { ah_part[q.l == ah_pa ? 0 : 1] = q.S > 1 ? q.part : q.Y; head[q.l] = h; continue; }      // reduced inside k_act_head (S == 1: the finished activation)
            if (q.S > 1) {
                if (last) { h.p = q.part; h.S = q.S; h.per_s = (unsigned long long)L.out_feat * n; }      // reduced on the fly by k_env_step
                else { RSeg r; memset(&r, 0, sizeof r); r.part = q.part; r.S = q.S; r.elems = (unsigned long long)L.out_feat * n; r.mode = 0; r.bias = P + L.b_off; r.per_n = L.npos * n; r.act = L.act; r.out = q.Y; segs.push_back(r); }
            }
            head[q.l] = h;
        }
        emit_reduce(e, segs, pname(e, "act_reduce", e->L[lv[0]].kind, lv[0]));
    }
    ReplayMeta R; R.cap = e->cap; R.cap2 = e->cap2; R.a = e->ra; R.r = e->rr; R.done = e->rdone; R.tree = e->tree; R.state = e->state; R.eps = e->hp.prio_eps; R.alpha = e->hp.prio_alpha;
    const bool u8 = e->hp.obs_dtype == DQN_OBS_U8;
    void *srows = e->s_rows, *sprows = e->sp_rows; float* px = e->pol_x; const long long cap = e->cap; const EnvDev Vc = V;
    if (use_ah) {
        ActHeadArgs h; memset(&h, 0, sizeof h);
        const LayerDev& La = e->L[lq];
        h.n = n; h.nA = e->nA; h.K = La.K; h.S = ah_S; h.nstream = lvh >= 0 ? 2 : 1; h.NO = e->nA + (lvh >= 0 ? 1 : 0); h.pm = ah_pm ? 1 : 0;
        for (int st = 0; st < 2; st++) {
            const int hl_ = (st == 1 && lvh >= 0) ? lvh : lq, pl_ = (st == 1 && lvh >= 0) ? ah_pv : ah_pa; const LayerDev& H = e->L[hl_]; const LayerDev& Pl = e->L[pl_]; ActHeadStream& T = h.st[st];
            T.part = ah_part[(st == 1 && lvh >= 0) ? 1 : 0]; T.pbias = P + Pl.b_off; T.pact = Pl.act; T.W = P + H.w_off; T.hbias = P + H.b_off; T.N = H.N; T.hact = H.act;
        }
        const int Gc = n / 4, NC = La.K / 32;
        h.partials = palloc(e, (size_t)Gc * 4 * h.NO * NC); h.tickets = (unsigned*)palloc(e, (size_t)Gc);
        hipMemsetAsync(h.tickets, 0, (size_t)Gc * 4, e->stream);
        h.q_out = e->pol_q; h.amax = e->pol_a; h.rs = rs; h.V = V; h.R = R;
        e->sink = nullptr; e->alloc_sink = nullptr;
        ap.steps.push_back({"act_head_step", [=](dqn_engine* en) { launch_act_head(en->stream, h); }});
        ap.steps.push_back({"env_observe_tree", [=](dqn_engine* en) { launch_env_observe2(en->stream, Vc, rs, u8, srows, sprows, cap, px, 1, &R); }});
        ap.n = n; ap.fused_tail = true; return 0;
    }
    e->sink = nullptr; e->alloc_sink = nullptr; ap.fused_tail = false;
    ActHeads Hd; memset(&Hd, 0, sizeof Hd); Hd.adv = head[lq]; if (e->hp.dueling) Hd.val = head[e->last_val]; Hd.dueling = e->hp.dueling; Hd.q_out = e->pol_q; Hd.amax = e->pol_a;
    // act!, add_exp!, observe, episode bookkeeping
    ap.steps.push_back({"env_step_commit", [=](dqn_engine* en) { launch_env_step(en->stream, Vc, rs, Hd, R); }});
    ap.steps.push_back({"env_observe", [=](dqn_engine* en) { launch_env_observe2(en->stream, Vc, rs, u8, srows, sprows, cap, px); }});
    ap.n = n; return 0;
}
static int act_graph(dqn_engine* e, dqn_engine::ActProg& ap) {
    if (ap.graph) return 0;
    hipGraph_t g;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (auto& s : ap.steps) s.fn(e);
    HIPCHK(hipStreamEndCapture(e->stream, &g));
    HIPCHK(hipGraphInstantiate(&ap.graph, g, nullptr, nullptr, 0)); HIPCHK(hipGraphDestroy(g)); return 0;
}
// F acting steps (+ one plain sampled train step) as one graph
static int cycle_graph(dqn_engine* e, dqn_engine::ActProg& ap, int F, bool with_train) {
    if (ap.cycle && ap.cycle_F == F && ap.cycle_train == with_train) return 0;
    if (ap.cycle) { hipGraphExecDestroy(ap.cycle); ap.cycle = nullptr; }
    hipGraph_t g;
    (void)hipGetLastError();
    e->step_take_pre = e->step_pregather = false;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (int f = 0; f < F; f++) for (auto& s : ap.steps) s.fn(e);
    if (with_train) enqueue_step(e, true, PH_ALL);
    const hipError_t lerr = hipGetLastError();
    HIPCHK(hipStreamEndCapture(e->stream, &g));
    if (lerr != hipSuccess) { hipGraphDestroy(g); return fail("HIP error %s while capturing the rollout cycle", hipGetErrorString(lerr)); }
    HIPCHK(hipGraphInstantiate(&ap.cycle, g, nullptr, nullptr, 0)); HIPCHK(hipGraphDestroy(g));
    ap.cycle_F = F; ap.cycle_train = with_train; return 0;
}
// one vector step of the reference's cadence (src/solver.jl:136-140: a train step every train_freq ENV steps) as ONE graph: the acting step, then its `due` train steps
// back to back with the pipelined gather of dqn_train_steps (step i's Adam launch gathers step i + 1's batch) -- one graph launch instead of the acting graph + the
// first / grouped / last graphs of a dqn_train_steps(due) call
static int envc_graph(dqn_engine* e, dqn_engine::ActProg& ap, int due) {
    if (ap.envc && ap.envc_due == due) return 0;
    if (ap.envc) { hipGraphExecDestroy(ap.envc); ap.envc = nullptr; }
    hipGraph_t g;
    (void)hipGetLastError();
    const bool pg = e->pg_ok;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    for (auto& s : ap.steps) s.fn(e);
    for (int i = 0; i < due; i++) { e->step_take_pre = pg && i > 0; e->step_pregather = pg && i + 1 < due; enqueue_step(e, true, PH_ALL); }
    e->step_take_pre = e->step_pregather = false;
    hipError_t lerr = hipGetLastError();
    if (e->launch_failed) { e->launch_failed = false; if (lerr == hipSuccess) lerr = hipErrorInvalidValue; }
    HIPCHK(hipStreamEndCapture(e->stream, &g));
    if (lerr != hipSuccess) { hipGraphDestroy(g); return fail("HIP error %s while capturing the env-cadence cycle", hipGetErrorString(lerr)); }
    HIPCHK(hipGraphInstantiate(&ap.envc, g, nullptr, nullptr, 0)); HIPCHK(hipGraphDestroy(g));
    if (!e->opt.no_graph_upload) (void)hipGraphUpload(ap.envc, e->stream);
    (void)hipGetLastError();
    ap.envc_due = due; return 0;
}
extern "C" int dqn_rollout(dqn_engine_t* e, int n_steps, const dqn_rollout_cfg* cfg, dqn_rollout_stats* out) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create");
    if (cfg->t0 < 1) return fail("t0 counts from 1 (src/solver.jl:82)");
    EnvDev& V = e->env; const int n = V.n;
    if (build_act_program(e, e->act, V, e->roll)) return -1;
    if (cfg->train_freq > 0 && build_program(e)) return -1;       // may reallocate split-K workspaces: before any capture
    RolloutDev h; h.t = cfg->t0 - 1; h.widx = ((e->widx - n) % e->cap + e->cap) % e->cap; h.eps_start = cfg->eps_start; h.eps_stop = cfg->eps_stop; h.eps_steps = cfg->eps_steps; h.pad = 0;
    { std::vector<RolloutDev> hs(DQN_ROLL_RECORDS, h);      // one record per group of four copies (k_act_head ticks its group's), record 0 = the four-launch tail's
      HIPCHK(hipMemcpyAsync(e->roll, hs.data(), sizeof(RolloutDev) * DQN_ROLL_RECORDS, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }   // hs lives in this scope
    launch_env_observe(e->stream, V, nullptr, 0, e->pol_x);
    const bool graph = e->hp.use_graph && !e->profiling;
    if (graph && act_graph(e, e->act)) return -1;
    long long trained = 0;
    // whole cycles -- train_freq acting steps ending on a train step (or 4 acting steps when nothing trains) -- replay as ONE graph where the
    // schedule allows it: single device, the train step due exactly at the cycle's last step, the replay already holding a batch, no target sync
    // before the cycle's last step
    const bool single = e->world <= 1 && !(e->comm && e->force_comm);
    const int F = cfg->train_freq > 0 ? cfg->train_freq : 4;
    const bool envc = cfg->cadence_env_steps != 0;      // train_freq / target_update_freq count ENV steps (src/solver.jl:136-145): n / train_freq train steps per vector step
    const bool cyc = graph && single && F >= 2 && F <= 16 && !e->opt.no_rollout_cycle && !envc;
    for (int k = 0; k < n_steps; k++) {
        const long long t = cfg->t0 + k;
        if (envc) {
            // the whole vector step as one graph where every vector step owes the same number of train steps and the replay holds a batch once this step's experiences are in
            if (graph && single && !e->opt.no_rollout_cycle && !e->tiny && cfg->train_freq > 0 && n % cfg->train_freq == 0 && n / cfg->train_freq <= 64 && std::min(e->cap, e->size + n) >= e->B) {
                const int due_c = n / cfg->train_freq;
                e->step_publish = false;
                if (envc_graph(e, e->act, due_c)) return -1;
                HIPCHK(hipGraphLaunch(e->act.envc, e->stream));
                e->widx = (e->widx + n) % e->cap; e->size = std::min(e->cap, e->size + n); trained += due_c;
                if (cfg->target_update_freq > 0 && (t * n) / cfg->target_update_freq != ((t - 1) * n) / cfg->target_update_freq) { if (dqn_sync_target(e)) return -1; }
                continue;
            }
            if (graph) HIPCHK(hipGraphLaunch(e->act.graph, e->stream));
            else for (auto& s : e->act.steps) { prof_begin(e, s.name); s.fn(e); prof_end(e); }
            e->widx = (e->widx + n) % e->cap; e->size = std::min(e->cap, e->size + n);
            const long long due = cfg->train_freq > 0 ? (t * n) / cfg->train_freq - ((t - 1) * n) / cfg->train_freq : 0;
            if (due > 0 && e->size >= e->B) { if (dqn_train_steps(e, (int)due, nullptr, nullptr)) return -1; trained += due; }      // back to back: the pipelined gather applies
            if (cfg->target_update_freq > 0 && (t * n) / cfg->target_update_freq != ((t - 1) * n) / cfg->target_update_freq) { if (dqn_sync_target(e)) return -1; }
            continue;
        }
        if (cyc && k + F <= n_steps) {
            const long long tl = t + F - 1;      // the cycle's last step
            bool ok = cfg->train_freq > 0 ? (tl % cfg->train_freq == 0 && std::min(e->cap, e->size + (long long)F * n) >= e->B) : true;
            if (cfg->target_update_freq > 0) for (long long u = t; u < tl; u++) ok = ok && (u % cfg->target_update_freq != 0);
            if (ok) {
                if (cycle_graph(e, e->act, F, cfg->train_freq > 0)) return -1;
                HIPCHK(hipGraphLaunch(e->act.cycle, e->stream));
                for (int f = 0; f < F; f++) { e->widx = (e->widx + n) % e->cap; e->size = std::min(e->cap, e->size + n); }
                if (cfg->train_freq > 0) trained++;
                if (cfg->target_update_freq > 0 && tl % cfg->target_update_freq == 0) { if (dqn_sync_target(e)) return -1; }
                k += F - 1; continue;
            }
        }
        if (graph) HIPCHK(hipGraphLaunch(e->act.graph, e->stream));
        else for (auto& s : e->act.steps) { prof_begin(e, s.name); s.fn(e); prof_end(e); }
        e->widx = (e->widx + n) % e->cap; e->size = std::min(e->cap, e->size + n);
        if (cfg->train_freq > 0 && t % cfg->train_freq == 0 && e->size >= e->B) { if (run_step(e, true)) return -1; trained++; }     // :134-139
        if (cfg->target_update_freq > 0 && t % cfg->target_update_freq == 0) { if (dqn_sync_target(e)) return -1; }                // :142-145
    }
    launch_env_reset_pending(e->stream, V, e->roll, 0);      // episode bookkeeping of the last step (src/solver.jl:99-132)
    if (out) {
        std::vector<long long> fe(n); std::vector<double> fr(n);
        HIPCHK(hipMemcpyAsync(fe.data(), V.fin_eps, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipMemcpyAsync(fr.data(), V.fin_reward, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
        out->last_loss = out->last_grad_norm = 0.0f;
        if (trained) { if (fetch_scalars(e, &out->last_loss, &out->last_grad_norm)) return -1; } else HIPCHK(hipStreamSynchronize(e->stream));
        out->episodes = 0; out->reward_sum = 0.0; out->train_steps = trained;
        for (int i = 0; i < n; i++) { out->episodes += fe[i]; out->reward_sum += fr[i]; }
    }
    return 0;
}
// basic_evaluation (src/evaluation_policy.jl:17-42) on the device: n_eval copies of the training MDP run one greedy episode each
// (while !done && step <= max_episode_length), rewards summed in Float64 like the reference's r_tot; returns the averages.
extern "C" int dqn_evaluate(dqn_engine_t* e, int n_eval, int max_episode_length, uint64_t seed, double* avg_reward, double* avg_steps) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create (the evaluation copies share its MDP)");
    if (n_eval < 1 || n_eval > 1024) return fail("n_eval must be in 1..1024");
    if (max_episode_length < 1) return fail("max_episode_length must be >= 1");
    EnvDev& W = e->eval_env;
    if (e->eval_n != n_eval) {
        HIPCHK(hipStreamSynchronize(e->stream)); drop_act(e, e->evalp); free_env_arrays(W); hipFree(e->eval_roll); e->eval_roll = nullptr; e->eval_n = 0;
        W = e->env; W.n = n_eval; W.eval_mode = 1;
        W.tm_s = W.tm_prev = nullptr; W.tm_t = nullptr; W.gw_pos = W.gw_prev = nullptr; W.actions = nullptr; W.rewards = nullptr; W.dones = W.pending = nullptr;
        W.ep_reward = nullptr; W.ep_step = nullptr; W.fin_eps = nullptr; W.fin_reward = nullptr;
        if (W.kind == DQN_ENV_TESTMDP) { DM(W.tm_s, (size_t)n_eval * 4); DM(W.tm_prev, (size_t)n_eval * 4); DM(W.tm_t, n_eval); }
        else { DM(W.gw_pos, (size_t)n_eval * 2); DM(W.gw_prev, (size_t)n_eval * 2); }
        DM(W.actions, n_eval); DM(W.rewards, n_eval); DM(W.dones, n_eval); DM(W.pending, n_eval); DM(W.ep_reward, n_eval); DM(W.ep_step, n_eval); DM(W.fin_eps, n_eval); DM(W.fin_reward, n_eval);
        DM(e->eval_roll, DQN_ROLL_RECORDS);
        e->eval_n = n_eval;
    }
    if (W.seed != seed || W.max_episode_length != max_episode_length) { W.seed = seed; W.max_episode_length = max_episode_length; drop_act(e, e->evalp); }   // baked into the program
    if (build_act_program(e, e->evalp, W, e->eval_roll)) return -1;
    RolloutDev h; memset(&h, 0, sizeof h);                                   // t = 0; eps schedule (0, 0, 1): always greedy
    h.eps_steps = 1.0f;
    { std::vector<RolloutDev> hs(DQN_ROLL_RECORDS, h);
      HIPCHK(hipMemcpyAsync(e->eval_roll, hs.data(), sizeof(RolloutDev) * DQN_ROLL_RECORDS, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
    HIPCHK(hipMemsetAsync(W.fin_reward, 0, (size_t)n_eval * 8, e->stream));
    launch_env_reset_pending(e->stream, W, e->eval_roll, 1);                   // reset!(env), resetstate!(policy)
    launch_env_observe(e->stream, W, nullptr, 0, e->pol_x);
    const bool graph = e->hp.use_graph && !e->profiling;
    if (graph && act_graph(e, e->evalp)) return -1;
    std::vector<unsigned char> pend(n_eval);
    for (int k = 0; k <= max_episode_length; k++) {
        if (graph) HIPCHK(hipGraphLaunch(e->evalp.graph, e->stream)); else for (auto& s : e->evalp.steps) s.fn(e);
        if ((k & 7) == 7) {     // every 8 vector steps: stop early once every episode is over
            HIPCHK(hipMemcpyAsync(pend.data(), W.pending, n_eval, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
            bool alive = false; for (int i = 0; i < n_eval; i++) alive = alive || !pend[i];
            if (!alive) break;
        }
    }
    std::vector<double> fr(n_eval); std::vector<int> st(n_eval);
    HIPCHK(hipMemcpyAsync(fr.data(), W.fin_reward, (size_t)n_eval * 8, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemcpyAsync(st.data(), W.ep_step, (size_t)n_eval * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
    double r = 0.0, s = 0.0;
    for (int i = 0; i < n_eval; i++) { r += fr[i]; s += (double)st[i]; }      // avg_r += r_tot; avg_steps += step, episode order
    if (avg_reward) *avg_reward = r / n_eval;
    if (avg_steps) *avg_steps = s / n_eval;
    return 0;
}
extern "C" int dqn_envs_info(dqn_engine_t* e, int* n_envs, int* fused_tail) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create");
    if (build_act_program(e, e->act, e->env, e->roll)) return -1;
    if (n_envs) *n_envs = e->env.n;
    if (fused_tail) *fused_tail = e->act.fused_tail ? 1 : 0;
    return 0;
}
extern "C" int dqn_envs_peek(dqn_engine_t* e, float* obs, int32_t* actions, float* rewards, uint8_t* dones) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create");
    EnvDev& V = e->env; const int n = V.n;
    if (obs) {
        if (policy_ws(e, n)) return -1;
        launch_env_observe(e->stream, V, nullptr, 0, e->pol_x);
        std::vector<float> x((size_t)e->E * n); HIPCHK(hipMemcpyAsync(x.data(), e->pol_x, x.size() * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
        for (int i = 0; i < n; i++) for (int f = 0; f < e->E; f++) obs[(size_t)i * e->E + f] = x[(size_t)f * n + i];
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    if (actions) HIPCHK(hipMemcpy(actions, V.actions, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (rewards) HIPCHK(hipMemcpy(rewards, V.rewards, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (dones) HIPCHK(hipMemcpy(dones, V.dones, (size_t)n, hipMemcpyDeviceToHost));
    return 0;
}

