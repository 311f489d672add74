// engine_program.hip -- the static launch program of the train step (batch_train!, src/solver.jl:191-236 and :239-287): every pointer,
// shape and plan is fixed at engine creation, so the step is compiled ONCE into a list of launches and replayed (hipGraph).
#include "engine.h"

// ---------------------------------------------------------------- static launch program
// Every pointer, shape and plan is fixed at engine creation, so the train step is compiled ONCE into a list of launches
// (closures) and merely replayed (and captured into a hipGraph).  Small independent kernels are batched: one k_valu_multi
// launch per network level, one k_reduce_multi per level, head reductions folded into k_td.
float* palloc(dqn_engine* e, size_t n) { float* d = nullptr; hipMalloc((void**)&d, n * 4); (e->alloc_sink ? *e->alloc_sink : e->prog_allocs).push_back(d); return d; }
bool same_geo(const LayerDev& a, const LayerDev& b) {
    return a.kind == b.kind && a.act == b.act && a.K == b.K && a.N == b.N && a.npos == b.npos && a.cin == b.cin && a.kh == b.kh && a.kw == b.kw &&
           a.sh == b.sh && a.sw == b.sw && a.ih == b.ih && a.iw == b.iw && a.fwd_kc == b.fwd_kc && a.src == b.src;
}
void add_valu(dqn_engine* e, std::vector<VTask>& pend, const VTask& t) { pend.push_back(t); }
// prio: the step's priority block (update_priorities! + the next step's index draw) rides as workgroup 0 of this launch
void flush_valu(dqn_engine* e, std::vector<VTask>& pend, const char* name, const PrioArgs* prio) {
    if (pend.empty()) return;
    unsigned blocks = 0;
    for (auto& t : pend) { t.first_block = blocks; blocks += valu_task_blocks(t); }
    VTask* dev = upload(e, pend); const int n = (int)pend.size();
    if (prio) {
        const PrioArgs pa = *prio; StepState* stt = e->state;
        char nm[80]; snprintf(nm, sizeof nm, "%s+prio", name); e->prog_names.push_back(nm); const char* name2 = e->prog_names.back().c_str();
        (e->sink ? *e->sink : e->prog).push_back({name2, [=](dqn_engine* en) { launch_valu_multi(en->stream, dev, n, blocks, &pa, stt); }});
    }
    else (e->sink ? *e->sink : e->prog).push_back({name, [=](dqn_engine* en) { launch_valu_multi(en->stream, dev, n, blocks); }});
    pend.clear();
}
void emit_reduce(dqn_engine* e, std::vector<RSeg>& segs, const char* name) {
    if (segs.empty()) return;
    unsigned blocks = 0;
    for (auto& r : segs) { r.first_block = blocks; blocks += (unsigned)((r.elems + 255) / 256); }
    RSeg* dev = upload(e, segs); const int n = (int)segs.size();
    (e->sink ? *e->sink : e->prog).push_back({name, [=](dqn_engine* en) { launch_reduce_multi(en->stream, dev, n, blocks); }});
    segs.clear();
}
const char* pname(dqn_engine* e, const char* op, int kind, int i) {
    char b[32]; snprintf(b, sizeof b, "%s_%s%d", op, kind == DQN_LAYER_CONV ? "conv" : "dense", i); e->prog_names.push_back(b); return e->prog_names.back().c_str();
}
int build_program(dqn_engine* e) {
    if (e->prog_built) return 0;
    HIPCHK(hipSetDevice(e->device));
    e->pg_ok = false; e->adam_step = -1; e->gmax_used = 0;
    HIPCHK(hipMemsetAsync(e->gmax_part, 0, (size_t)gmax_slots(e->Pint) * 4, e->stream));      // per-block maxima of an earlier program shape must not survive
    e->prog_names.reserve(512);
    const int B = e->Bc /* columns of one sequence set: batch_size, or T*batch_size for DRQN */, ncon = e->ncon, ld0 = 2 * B, Bb = e->B, T = e->T;
    const bool mf = e->hp.use_mfma != 0, rec = e->hp.recurrence != 0;
    // forward views: an LSTM layer's batched part is its bias-free input projection Gx = Wi*x over ALL columns (a dense layer
    // K = n_in, N = 4H writing gx_*); the recurrence then runs as T small launches.
    LayerDev LV[DQN_MAX_LAYERS]; float *fwd_on[DQN_MAX_LAYERS], *fwd_tg[DQN_MAX_LAYERS];
    for (int i = 0; i < e->nl; i++) {
        LV[i] = e->L[i]; fwd_on[i] = e->act_on[i]; fwd_tg[i] = e->act_tg[i];
        if (e->L[i].kind == DQN_LAYER_LSTM) { LV[i].kind = DQN_LAYER_DENSE; LV[i].out_feat = LV[i].N; LV[i].b_off = LV[i].z_off; LV[i].act = DQN_ACT_IDENTITY; fwd_on[i] = e->gx_on[i]; fwd_tg[i] = e->gx_tg[i]; }
    }
    std::vector<std::vector<int>> levels; std::vector<int> val, adv;
    for (int i = 0; i < e->nl; i++) { if (e->L[i].stream == DQN_STREAM_BASE) levels.push_back({i}); else if (e->L[i].stream == DQN_STREAM_VAL) val.push_back(i); else adv.push_back(i); }
    for (size_t j = 0; j < std::max(val.size(), adv.size()); j++) { std::vector<int> lv; if (j < val.size()) lv.push_back(val[j]); if (j < adv.size()) lv.push_back(adv[j]); levels.push_back(lv); }
    // ---------------- recurrent networks with column-group dW chunks (plan dw_kc = -cg): the step is ONE column-parallel launch + the Adam launch (drqn_cols.hip; BASELINE config 4)
    {
        int cgm = 0; bool all_same = true;
        for (int i = 0; i < e->nl; i++) { if (e->L[i].dw_kc < 0) cgm = -e->L[i].dw_kc; all_same = all_same && e->L[i].dw_kc == e->L[0].dw_kc; }
        if (cgm) {
            if (!rec || !all_same) return fail("plan: column-group dW chunks (dw_kc < 0) need recurrence = true and the same dw_kc on every layer");
            const int nset = e->hp.double_q ? 3 : 2; const LayerDev& L0 = e->L[0];
            bool ok = drqn_fused_cg(e->L, e->nl, e->E, Bb, T, e->nA, e->hp.dueling, e->hp.double_q, 1) > 0 && Bb % cgm == 0 && nset * 4 * L0.H * cgm <= 1024 && !e->comm && !e->sim_world && e->world <= 1;
            if (ok) { ok = dqn_nchunks(L0.K, L0.fwd_kc) == 1; for (int i = 1; i < e->nl; i++) ok = ok && dqn_nchunks(e->L[i].N, e->L[i].dx_kc) == 1; }
            if (!ok) return fail("plan: column-group dW chunks (dw_kc = %d) need a network the fused recurrent step covers -- Chain(flattenbatch, LSTM, Dense) with or without the dueling split, "
                                 "H a multiple of 8 up to 64, unsplit input projection and head dX, ONE device without a communicator -- use dw_kc >= 0 (plan = NULL picks a plan that fits; with a communicator dqn_comm_init re-derives it)", -cgm);
            DrqnColsArgs a; memset(&a, 0, sizeof a);
            a.B = Bb; a.T = T; a.H = L0.H; a.E = e->E; a.nA = e->nA; a.dueling = e->hp.dueling; a.double_q = e->hp.double_q; a.cg = cgm; a.nset = nset; a.gamma = e->hp.gamma; a.Pint = (unsigned)e->Pint;
            a.wi_off = (unsigned)L0.w_off; a.b_off = (unsigned)L0.b_off; a.wh_off = (unsigned)L0.wh_off; a.h0_off = (unsigned)L0.h0_off; a.c0_off = (unsigned)L0.c0_off;
            const int ha = e->hp.dueling ? e->last_adv : e->last_base, hv = e->hp.dueling ? e->last_val : -1;
            for (int hd = 0; hd < 2; hd++) { const int l = hd == 0 ? ha : hv; if (l < 0) continue; const LayerDev& L = e->L[l];
                a.hw_off[hd] = (unsigned)L.w_off; a.hb_off[hd] = (unsigned)L.b_off; a.hN[hd] = L.N; a.hact[hd] = L.act; a.h_S[hd] = dqn_nchunks(L.K, L.fwd_kc); a.h_kc[hd] = dqn_chunk_len(L.K, L.fwd_kc); }
            a.p_on = e->p_on; a.p_tg = e->p_tg; a.ep_s = e->ep_s; a.ep_sp = e->ep_sp; a.ep_a = e->ep_a; a.ep_r = e->ep_r; a.ep_done = e->ep_done; a.ep_len = e->ep_len;
            if (!e->draw_idx_h) {      // the draw ring: mapped, coherent pinned host memory (survives program rebuilds; freed with the engine)
                HIPCHK(hipHostMalloc((void**)&e->draw_idx_h, sizeof(long long) * DQN_DRAW_SLOTS * Bb, hipHostMallocMapped | hipHostMallocCoherent));
                HIPCHK(hipHostMalloc((void**)&e->draw_start_h, sizeof(int) * DQN_DRAW_SLOTS * Bb, hipHostMallocMapped | hipHostMallocCoherent));
                memset(e->draw_idx_h, 0, sizeof(long long) * DQN_DRAW_SLOTS * Bb); memset(e->draw_start_h, 0, sizeof(int) * DQN_DRAW_SLOTS * Bb);
                HIPCHK(hipHostGetDevicePointer((void**)&e->draw_idx_d, e->draw_idx_h, 0)); HIPCHK(hipHostGetDevicePointer((void**)&e->draw_start_d, e->draw_start_h, 0));
                for (int k = 0; k < 4; k++) { HIPCHK(hipEventCreateWithFlags(&e->draw_ev[k], hipEventDisableTiming)); e->draw_ev_used[k] = false; }
            }
            a.ring_idx = e->draw_idx_d; a.ring_np = e->draw_start_d;
            if (e->opt.drqn_probe & 4) {      // TIMING PROBE: the draws come from DEVICE memory (episode 0, one row, for every column -- wrong numbers, right schedule): what the PCIe read of the host ring costs
                long long* di = (long long*)palloc(e, (size_t)DQN_DRAW_SLOTS * Bb * 2); int* dn = (int*)palloc(e, (size_t)DQN_DRAW_SLOTS * Bb);
                std::vector<long long> zi((size_t)DQN_DRAW_SLOTS * Bb, 0); std::vector<int> zn((size_t)DQN_DRAW_SLOTS * Bb, 1);
                HIPCHK(hipMemcpy(di, zi.data(), zi.size() * 8, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(dn, zn.data(), zn.size() * 4, hipMemcpyHostToDevice));
                a.ring_idx = di; a.ring_np = dn;
            }
            const int G = Bb / cgm;
            a.slabs = palloc(e, (size_t)G * e->Pint); a.hl = palloc(e, (size_t)B); a.td = e->td; a.st = e->state;
            a.probe = e->opt.drqn_probe | (mf ? 0 : 2);      // hp.use_mfma = 0: the VALU form of the input projection (the same chains, the same bits)
            if (e->opt.drqn_stamps) { a.stamps = (unsigned long long*)palloc(e, 64); e->drqn_stamps = a.stamps; }
            e->prog.push_back({"drqn_cols", [=](dqn_engine* en) { if (launch_drqn_cols(en->stream, a, en->drqn_slot_next++)) en->launch_failed = true; }});      // the slot is baked into the captured node
            e->prog_post_begin = e->prog.size();
            AdamJob J; memset(&J, 0, sizeof J);
            J.p = e->p_on; J.m = e->m; J.v = e->v; J.g = e->grad; J.g_out = e->grad; J.state = e->state; J.gmax_part = e->gmax_part;
            J.f64mode = e->hp.adam_f64_scalars; J.lr = e->hp.learning_rate; J.b1 = e->hp.adam_beta1; J.b2 = e->hp.adam_beta2; J.eps = e->hp.adam_eps; J.gscale = 1.0f;
            J.segs.n = 1; J.segs.beg[0] = 0; J.segs.end[0] = e->Pint; J.segs.part[0] = a.slabs; J.segs.S[0] = G; J.segs.stride[0] = e->Pint; J.segs.blocks = (unsigned)((e->Pint + 255) / 256);
            J.nr = 0; J.sblocks = 1; J.tick = 1; J.slot0 = 0; J.fold_hl = a.hl; J.fold_T = T; J.fold_B = Bb;
            e->adam_step = (long)e->prog.size();
            e->prog.push_back({"adam", [=](dqn_engine* en) { launch_adam(en->stream, J); }});
            e->gmax_used = (int)(J.segs.blocks + J.sblocks);
            e->tiny = false; e->drqn_fused = true; e->arena_u8 = false; e->prio_forked = false; e->prio_in_bwd = false; e->dp_gather = false; e->dp_overlap = false; e->prog_pre1_end = 0;
            e->final_reduce_step = -1; memset(&e->adam_segs, 0, sizeof e->adam_segs);
            e->prog_built = true;
            return 0;
        }
        e->drqn_fused = false;
    }
    // ---------------- networks that fit in LDS: the WHOLE step is one single-workgroup launch (tiny_step.hip; BASELINE config 1)
    e->tiny = false;
    if (!rec && !e->comm && !e->sim_world && e->world <= 1 && e->hp.prioritized_replay && Bb <= 64 && e->nl <= TINY_MAX_LAYERS &&
        (int)levels.size() <= TINY_MAX_LAYERS && e->Pint <= 16384 && (size_t)e->Pint * B <= 262144 /* ~5 MACs per parameter and column on ONE CU: <= ~9 us of arithmetic */ && !e->no_tiny) {
        bool ok = true; size_t fl = 0;
        TinyArgs a; memset(&a, 0, sizeof a);
        for (int i = 0; i < e->nl; i++) { const LayerDev& L = e->L[i]; ok = ok && L.kind == DQN_LAYER_DENSE && dqn_nchunks(L.N, L.dx_kc) == 1 && L.src < i; a.L[i] = L; }
        auto take = [&](size_t n) { const int o = (int)fl; fl += (n + 3) / 4 * 4; return o; };
        a.pon_off = take(e->Pint); a.ptg_off = take(e->Pint); a.g_off = take(e->Pint); a.x0_off = take((size_t)e->E * ld0); a.misc_off = take((size_t)B * 3 * e->nA);
        for (int i = 0; i < e->nl; i++) { a.on_off[i] = take((size_t)e->L[i].N * ncon); a.tg_off[i] = take((size_t)e->L[i].N * B); a.d_off[i] = take((size_t)e->L[i].N * B); }
        if (fl * 4 < 7808 + 64 * 4 + 1024) fl = (7808 + 64 * 4 + 1024) / 4;      // room for the priority block's path state (it reuses the whole region)
        ok = ok && fl * 4 <= 144 * 1024;      // one workgroup may hold up to 160 KB of LDS on gfx950 (beyond 64 KB the launcher raises the function's dynamic-LDS limit)
        if (ok) {
            a.nl = e->nl; a.nlev = (int)levels.size(); a.B = B; a.nA = e->nA; a.E = e->E; a.ncon = ncon; a.dueling = e->hp.dueling; a.double_q = e->hp.double_q; a.obs_u8 = e->hp.obs_dtype == DQN_OBS_U8 ? 1 : 0; a.distinct = e->hp.sample_distinct ? 1 : 0;
            a.last_base = e->last_base; a.last_val = e->last_val; a.last_adv = e->last_adv;
            a.gamma = e->hp.gamma; a.beta = e->hp.prio_beta; a.prio_eps = e->hp.prio_eps; a.prio_alpha = e->hp.prio_alpha; a.cap2 = e->cap2; a.seed = e->hp.seed; a.P = e->Pint;
            for (size_t li = 0; li < levels.size(); li++) { a.lev_n[li] = (int)levels[li].size(); a.lev_l[li][0] = levels[li][0]; a.lev_l[li][1] = levels[li].size() > 1 ? levels[li][1] : levels[li][0]; }
            a.lds_bytes = (unsigned)(fl * 4);
            a.p_tg = e->p_tg; a.p_on = e->p_on; a.m = e->m; a.v = e->v; a.grad = e->grad; a.state = e->state; a.gmax_part = e->gmax_part; a.tree = e->tree;
            a.idx = e->idx; a.idx_pre = e->idx_pre; a.s_rows = e->s_rows; a.sp_rows = e->sp_rows; a.ra = e->ra; a.rr = e->rr; a.rdone = e->rdone;
            a.x0 = e->x0; a.w_is = e->w_is; a.td = e->td; a.q_on_s = e->q_on_s; a.q_on_sp = e->q_on_sp; a.q_tg_sp = e->q_tg_sp; a.ytarget = e->ytarget; a.best = e->best;
            a.f64mode = e->hp.adam_f64_scalars; a.lr = e->hp.learning_rate; a.b1 = e->hp.adam_beta1; a.b2 = e->hp.adam_beta2; a.adam_eps = e->hp.adam_eps;
            const TinyArgs* a_dev = upload(e, std::vector<TinyArgs>(1, a)); const unsigned lds = a.lds_bytes;
            e->prog.push_back({"tiny_step", [=](dqn_engine* en) { if (launch_tiny_step(en->stream, a_dev, lds, en->step_sampled ? 1 : 0, en->opt.tiny_stop)) en->launch_failed = true; }});
            e->tiny = true; e->arena_u8 = false; e->prio_forked = false; e->prio_in_bwd = false; e->dp_gather = false; e->dp_overlap = false; e->prog_pre1_end = 0;
            e->prog_post_begin = e->prog.size(); e->final_reduce_step = -1; memset(&e->adam_segs, 0, sizeof e->adam_segs); e->gmax_used = 1;
            for (int i = 0; i < e->nl; i++) e->L[i].xu8 = 0;
            e->prog_built = true;
            return 0;
        }
    }
    // ---------------- u8 replay: keep the observation arena in BYTES when its only consumers are the LDS-tiled forward and dW launches of ONE
    // first layer (they convert byte / 255 inside their tile loads): the gather writes 1 byte per element instead of 4 and the first layer reads
    // a quarter of the bytes.  Everything else (VALU / direct-MFMA fallbacks, heads fed by the observation, the operand all-gather) needs floats.
    e->arena_u8 = false;
    for (int i = 0; i < e->nl; i++) { e->L[i].xu8 = 0; LV[i].xu8 = 0; }
    if (e->hp.obs_dtype == DQN_OBS_U8 && !rec && mf && B % 4 == 0 && e->E % 4 == 0 && levels.size() > 1 && !e->opt.no_u8_arena) {
        int n_src = 0; for (int i = 0; i < e->nl; i++) if (e->L[i].src < 0) n_src++;
        const int l0 = levels[0][0];
        int ldx2[2] = {ld0, ld0}, c02[2] = {0, B}, nc2[2] = {ncon, B};
        if (n_src == 1 && levels[0].size() == 1 && e->L[l0].src < 0 && (e->L[l0].kind == DQN_LAYER_CONV || (!e->comm && !e->sim_world)) &&
            gemm_fwd_eligible(LV[l0], 2, ldx2, c02, nc2) && gemm_dw_eligible(e->L[l0], B, ld0)) { e->arena_u8 = true; e->L[l0].xu8 = 1; LV[l0].xu8 = 1; }
    }
    // ---------------- small batches: the head level (forwards of both nets), the TD kernel and the head layers' dX run as ONE launch with a
    // workgroup per batch column (k_head_td); the heads' dW/db and the loss fold ride as tail tasks of the next backward launch
    int hv_l = -1, ha_l = -1; bool fuse_heads = false;
    // (r03: at ANY batch -- at B = 512 the four launches it replaces, head forwards / slab reduce / single-workgroup k_td / head dX, took 40 us)
    if (!rec && e->B <= e->opt.head_fuse_maxb && !e->opt.no_head_fuse) {
        const auto& lv = levels.back();
        if (e->hp.dueling && lv.size() == 2 && lv[0] == e->last_val && lv[1] == e->last_adv) { hv_l = lv[0]; ha_l = lv[1]; }
        else if (!e->hp.dueling && lv.size() == 1 && lv[0] == e->last_base) ha_l = lv[0];
        if (ha_l >= 0) {
            fuse_heads = true; size_t lds = (size_t)(1 + e->nA) * 4;
            for (int l : lv) {
                const LayerDev& L = e->L[l];
                fuse_heads = fuse_heads && L.kind == DQN_LAYER_DENSE && dqn_nchunks(L.N, L.dx_kc) == 1 && (L.src < 0 || e->L[L.src].kind != DQN_LAYER_LSTM);
                lds += ((size_t)3 * L.K + (size_t)3 * L.N * dqn_nchunks(L.K, L.fwd_kc) + (size_t)3 * L.N) * 4;
            }
            fuse_heads = fuse_heads && lds <= 60 * 1024;
        }
    }
    float* hl_buf = fuse_heads ? palloc(e, (size_t)B) : nullptr;      // per-column Huber terms (folded into the loss by a tail task)
    float* actT[DQN_MAX_LAYERS][2] = {};                             // transposed copies [column][feature] of the head layers' inputs (written by the split-K reduce)
    bool wantT[DQN_MAX_LAYERS] = {};
    // ---------------- r05: when the head layers sit on dense hidden layers whose forward ran split-K, the split-K reduce AND the head level are ONE chip-filling launch
    // (red_head.hip: workgroup = 4 batch columns x stream x plan chunk of 32 hidden rows, the last arriver of a column group does TD + the heads' dX) instead of
    // k_reduce_multi (384 workgroups) + k_head_td (B workgroups)
    bool fuse_rh = false, rh_pm = false; int rh_pa = -1, rh_pv = -1, rh_S = 0; const float* rh_part[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [stream][net]
    const float* rh_partT[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    if (fuse_heads && levels.size() >= 2 && !e->opt.no_red_head && !e->opt.probe_no_tg && !e->opt.head_dbg) {
        const LayerDev& La = e->L[ha_l]; rh_pa = La.src; rh_pv = hv_l >= 0 ? e->L[hv_l].src : -1;
        bool ok = rh_pa >= 0 && (hv_l < 0 || (rh_pv >= 0 && rh_pv != rh_pa));
        const auto& pl = levels[levels.size() - 2];
        auto in_pl = [&](int l) { for (int x : pl) if (x == l) return true; return false; };
        if (ok) ok = in_pl(rh_pa) && (hv_l < 0 || in_pl(rh_pv)) && (int)pl.size() == (hv_l >= 0 ? 2 : 1);
        if (ok) {
            const LayerDev& Pa = e->L[rh_pa]; const int S = dqn_nchunks(Pa.K, Pa.fwd_kc); rh_S = S;
            // S > 1: k_red_head (split-K producers, small batches).  S == 1 -- finished activations of an unsplit forward, i.e. large batches: k_head_cols4, one workgroup per
            // column group (k_red_head's (group, stream, chunk) decomposition is SLOWER there: 27.4 vs 21.3 us for k_head_td at B = 512, profiles/history/r05_k_cfg5_red_head_ab.txt --
            // 4096 workgroups each staging both streams' head weights)
            ok = Pa.kind == DQN_LAYER_DENSE && (S > 1 || e->opt.no_head_cols4 != 1) && dqn_chunk_len(La.K, La.fwd_kc) == 32 && dqn_nchunks(La.K, La.fwd_kc) * 32 == La.K;
            if (ok && hv_l >= 0) { const LayerDev& Pv = e->L[rh_pv]; const LayerDev& Lv = e->L[hv_l];
                ok = Pv.kind == DQN_LAYER_DENSE && Pv.N == Pa.N && dqn_nchunks(Pv.K, Pv.fwd_kc) == S && dqn_chunk_len(Lv.K, Lv.fwd_kc) == 32 && Lv.K == La.K; }
            if (ok) ok = red_head_ok(B, La.K, S, e->nA, hv_l >= 0 ? 2 : 1, La.N, hv_l >= 0 ? e->L[hv_l].N : 0);
        }
        fuse_rh = ok;
    }
    // k_head_td and k_head_cols4 (unsplit producers) read their input columns out of transposed copies
    if (fuse_heads && (!fuse_rh || rh_S == 1)) for (int l : levels.back()) if (e->L[l].src >= 0) wantT[e->L[l].src] = true;
    HeadSrc head[DQN_MAX_LAYERS][2];   // per (layer, net): where k_td finds the layer's output
    // ---------------- forward: online net on [s ; sp] (src/solver.jl:210,220), target net on sp (:211)
    for (size_t li = 0; li < levels.size(); li++) {
        // the head layers' split-K slabs are reduced inside the single-workgroup TD kernel only while that is cheaper than a reduce
        // launch (small batches); at B = 512 the 7680 head values x 16 slabs belong on many workgroups
        const auto& lv = levels[li]; const bool last = li + 1 == levels.size() && !rec && e->B <= 64;
        if (fuse_heads && li + 1 == levels.size()) continue;      // computed inside k_head_td
        struct Prob { int l, net; const float *P, *X; int ldx, col0, ncols; float *Y, *part; int S; };
        std::vector<Prob> pr;
        const bool probe_no_tg = e->opt.probe_no_tg != 0;      // TIMING PROBE (wrong numbers, right schedule): the forward launches without the target network's problems
        for (int l : lv) for (int net = 0; net < 2; net++) {
            if (net == 1 && probe_no_tg) { if (wantT[l]) actT[l][1] = palloc(e, (size_t)LV[l].out_feat * B); continue; }
            const LayerDev& L = LV[l]; Prob q; q.l = l; q.net = net; q.P = net ? e->p_tg : e->p_on;
            float** act = net ? e->act_tg : e->act_on;
            q.X = L.src < 0 ? e->x0 : act[L.src]; q.ldx = L.src < 0 ? ld0 : (net ? B : ncon); q.col0 = (L.src < 0 && net) ? B : 0; q.ncols = net ? B : ncon;
            q.Y = net ? fwd_tg[l] : fwd_on[l]; q.S = dqn_nchunks(L.K, L.fwd_kc); q.part = q.S > 1 ? palloc(e, (size_t)q.S * L.out_feat * q.ncols) : nullptr;
            pr.push_back(q);
        }
        bool geo = true; for (int l : lv) geo = geo && same_geo(LV[lv[0]], LV[l]);
        std::vector<bool> done(pr.size(), false);
        auto emit_gemm = [&](const std::vector<int>& ids, const char* name) {
            const LayerDev L = LV[pr[ids[0]].l]; const int n = (int)ids.size();
            struct A { const float *W[4], *bias[4], *X[4]; int ldx[4], col0[4], ncols[4]; float* out[4]; float* outT[4]; } a;
            bool any_t = false;
            for (int i = 0; i < n; i++) {
                const Prob& q = pr[ids[i]]; const LayerDev& Lq = LV[q.l]; a.W[i] = q.P + Lq.w_off; a.bias[i] = q.P + Lq.b_off; a.X[i] = q.X; a.ldx[i] = q.ldx; a.col0[i] = q.col0; a.ncols[i] = q.ncols; a.out[i] = q.S > 1 ? q.part : q.Y;
                // an UNSPLIT dense layer that feeds k_head_td writes the transposed copy of its output itself (r04: without it the head kernel read its columns one 64-byte
                // sector per element at B = 512 -- 11 us of its 25); split-K layers get theirs from the reduce launch below
                a.outT[i] = nullptr;
                if (wantT[q.l] && q.S == 1 && Lq.kind == DQN_LAYER_DENSE) { a.outT[i] = actT[q.l][q.net] = palloc(e, (size_t)Lq.out_feat * q.ncols); any_t = true; }
            }
            if (any_t) for (int i = 0; i < n; i++) if (!a.outT[i]) { any_t = false; for (int j = 0; j < n; j++) { if (a.outT[j]) actT[pr[ids[j]].l][pr[ids[j]].net] = nullptr; a.outT[j] = nullptr; } break; }      // all problems of the launch or none
            // split-K slabs that only k_red_head reads are written piece-major (GFwdProb::pm): every problem of the launch belongs to the head's producer layers
            bool pm_all = fuse_rh && rh_S > 1 && !e->opt.no_rh_pm; for (int i = 0; i < n; i++) pm_all = pm_all && (pr[ids[i]].l == rh_pa || pr[ids[i]].l == rh_pv) && pr[ids[i]].S > 1;
            if (pm_all) rh_pm = true;
            const int pm = pm_all ? 1 : 0;
            e->prog.push_back({name, [=](dqn_engine* en) { launch_gemm_fwd(en->stream, L, n, a.W, a.bias, a.X, a.ldx, a.col0, a.ncols, a.out, any_t ? a.outT : nullptr, pm); }});
            for (int id : ids) done[id] = true;
        };
        if (mf) {
            std::vector<int> all; int ldx[4], c0[4], nc[4];
            for (size_t i = 0; i < pr.size() && i < 4; i++) { all.push_back((int)i); ldx[i] = pr[i].ldx; c0[i] = pr[i].col0; nc[i] = pr[i].ncols; }
            if (geo && pr.size() <= 4 && gemm_fwd_eligible(LV[lv[0]], (int)pr.size(), ldx, c0, nc)) emit_gemm(all, pname(e, "fwd", e->L[lv[0]].kind, lv[0]));
            else for (size_t i = 0; i + 1 < pr.size(); i += 2) {
                int l2[2] = {pr[i].ldx, pr[i + 1].ldx}, c2[2] = {pr[i].col0, pr[i + 1].col0}, n2[2] = {pr[i].ncols, pr[i + 1].ncols};
                if (gemm_fwd_eligible(LV[pr[i].l], 2, l2, c2, n2)) emit_gemm({(int)i, (int)i + 1}, pname(e, "fwd", e->L[pr[i].l].kind, pr[i].l));
            }
        }
        std::vector<VTask> pend;
        for (size_t i = 0; i < pr.size(); i++) {
            if (done[i]) continue;
            const Prob q = pr[i]; const LayerDev L = LV[q.l];
            if (mf && mfma_fwd_ok(L, q.ncols)) {
                e->prog.push_back({pname(e, q.net ? "fwd_tg" : "fwd_on", L.kind, q.l), [=](dqn_engine* en) { launch_mfma_fwd(en->stream, L, q.P, q.X, q.ldx, q.col0, q.ncols, q.Y, q.part, false); }});
            } else {
                VTask t; memset(&t, 0, sizeof t); t.kind = 0; t.L = L; t.P = q.P; t.X = q.X; t.ldx = q.ldx; t.col0 = q.col0; t.ncols = q.ncols; t.S = q.S; t.kc = dqn_chunk_len(L.K, L.fwd_kc);
                t.out = q.S > 1 ? q.part : q.Y; add_valu(e, pend, t);
            }
        }
        flush_valu(e, pend, pname(e, "fwd_valu", e->L[lv[0]].kind, lv[0]));
        std::vector<RSeg> segs;
        for (const Prob& q : pr) {
            const LayerDev& L = LV[q.l];
            HeadSrc h; h.p = q.Y; h.ld = q.ncols; h.S = 1; h.per_s = 0; h.bias = q.P + L.b_off; h.act = L.act;
            if (fuse_rh && (q.l == rh_pa || q.l == rh_pv)) { rh_part[q.l == rh_pa ? 0 : 1][q.net] = q.S > 1 ? q.part : q.Y;      // reduced inside k_red_head (S == 1: the finished activation)
                                                             rh_partT[q.l == rh_pa ? 0 : 1][q.net] = q.S > 1 ? nullptr : actT[q.l][q.net]; }
            else if (q.S > 1) {
                if (last) { h.p = q.part; h.S = q.S; h.per_s = (unsigned long long)L.out_feat * q.ncols; }   // reduced on the fly by k_td
                else { RSeg r; memset(&r, 0, sizeof r); r.part = q.part; r.S = q.S; r.elems = (unsigned long long)L.out_feat * q.ncols; r.mode = 0; r.bias = q.P + L.b_off; r.per_n = L.npos * q.ncols; r.act = L.act; r.out = q.Y;
                       if (wantT[q.l]) { r.outT = actT[q.l][q.net] = palloc(e, (size_t)L.out_feat * q.ncols); r.ncolsT = q.ncols; }
                       segs.push_back(r); }
            }
            head[q.l][q.net] = h;
        }
        emit_reduce(e, segs, pname(e, "fwd_reduce", e->L[lv[0]].kind, lv[0]));
        if (e->L[lv[0]].kind == DQN_LAYER_LSTM) {
            // the recurrence: T launches, each advancing the online s-sequence, the online sp-sequence (double-Q) and the target
            // sp-sequence by one step from the reset state (Flux.reset!, src/solver.jl:249-250,271)
            const int l = lv[0]; const LayerDev L = e->L[l]; const int H = L.H;
            if (lstm_seq_fits(H, Bb, T)) {        // small LSTM: the whole recurrence of the three sequence sets in ONE launch
                LstmSeqArgs a; memset(&a, 0, sizeof a); a.H = H; a.B = Bb; a.T = T; int ns = 0;
                auto seq = [&](const float* P, const float* gx, float* hout, float* cst, int ld, int c0, bool keep) {
                    LstmSeqF& q = a.s[ns++]; q.Gx = gx; q.Hout = hout; q.Cst = cst; q.ld = ld; q.c0 = c0; q.Wh = P + L.wh_off; q.bias = P + L.b_off; q.h0 = P + L.h0_off; q.c0v = P + L.c0_off;
                    if (keep) { q.gates = e->gates[l]; q.tc = e->tcb[l]; q.hprev_out = e->hprev_buf[l]; q.cprev_out = e->cprev_buf[l]; q.keep_ld = B; q.keep_c0 = 0; }
                };
                seq(e->p_on, e->gx_on[l], e->act_on[l], e->cst_on[l], ncon, 0, true);
                if (e->hp.double_q) seq(e->p_on, e->gx_on[l], e->act_on[l], e->cst_on[l], ncon, B, false);
                seq(e->p_tg, e->gx_tg[l], e->act_tg[l], e->cst_tg[l], B, 0, false);
                a.nseq = ns;
                e->prog.push_back({pname(e, "lstm_seq", L.kind, l), [=](dqn_engine* en) { launch_lstm_seq(en->stream, a); }});
            } else
            for (int t = 0; t < T; t++) {
                LstmStepArgs a; memset(&a, 0, sizeof a); a.H = H; a.B = Bb; int ns = 0;
                auto seq = [&](const float* P, const float* gx, float* hout, float* cst, int ld, int c0, bool keep) {
                    LstmSeq& q = a.s[ns++]; q.Gx = gx; q.Hout = hout; q.Cst = cst; q.ld = ld; q.c0 = c0; q.Wh = P + L.wh_off; q.bias = P + L.b_off;
                    if (t == 0) { q.hprev = P + L.h0_off; q.hp_ld = 1; q.hp_bs = 0; q.cprev = P + L.c0_off; q.cp_ld = 1; q.cp_bs = 0; }
                    else { q.hprev = hout + c0 + (t - 1) * Bb; q.hp_ld = ld; q.hp_bs = 1; q.cprev = cst + c0 + (t - 1) * Bb; q.cp_ld = ld; q.cp_bs = 1; }
                    if (keep) { q.gates = e->gates[l]; q.tc = e->tcb[l]; q.hprev_out = e->hprev_buf[l]; q.cprev_out = e->cprev_buf[l]; q.keep_ld = B; q.keep_c0 = 0; }
                };
                seq(e->p_on, e->gx_on[l], e->act_on[l], e->cst_on[l], ncon, 0, true);
                if (e->hp.double_q) seq(e->p_on, e->gx_on[l], e->act_on[l], e->cst_on[l], ncon, B, false);
                seq(e->p_tg, e->gx_tg[l], e->act_tg[l], e->cst_tg[l], B, 0, false);
                a.nseq = ns;
                e->prog.push_back({pname(e, "lstm_step", L.kind, l), [=](dqn_engine* en) { launch_lstm_step_t(en->stream, a, t); }});
            }
        }
    }
    // ---------------- dueling reduce + argmax + Bellman target + TD + Huber + dL/dQ + priority update
    {
        TdArgs t; memset(&t, 0, sizeof t);
        t.B = B; t.nA = e->nA; t.ncon = ncon; t.dueling = e->hp.dueling; t.double_q = e->hp.double_q; t.prioritized = e->hp.prioritized_replay;
        t.gamma = e->hp.gamma; t.prio_beta = e->hp.prio_beta; t.prio_eps = e->hp.prio_eps; t.prio_alpha = e->hp.prio_alpha; t.cap2 = e->cap2;
        t.idx = e->idx; t.a = e->ra; t.r = e->rr; t.done = e->rdone; t.tree = e->tree;
        const int lq = e->hp.dueling ? e->last_adv : e->last_base;
        t.on_adv = head[lq][0]; t.tg_adv = head[lq][1]; t.d_adv = e->dact[lq];
        if (e->hp.dueling) { t.on_val = head[e->last_val][0]; t.tg_val = head[e->last_val][1]; t.d_val = e->dact[e->last_val]; }
        t.idx_mut = e->idx; t.idx_pre = e->idx_pre; t.w_is = e->w_is; t.td = e->td; t.q_on_s = e->q_on_s; t.q_on_sp = e->q_on_sp; t.q_tg_sp = e->q_tg_sp; t.ytarget = e->ytarget; t.best = e->best; t.st = e->state;
        if (fuse_rh) {
            RedHeadArgs h; memset(&h, 0, sizeof h);
            const LayerDev& La = e->L[ha_l];
            h.B = B; h.nA = e->nA; h.K = La.K; h.S = dqn_nchunks(e->L[rh_pa].K, e->L[rh_pa].fwd_kc); h.ncon = ncon; h.nstream = hv_l >= 0 ? 2 : 1; h.NO = e->nA + (hv_l >= 0 ? 1 : 0); h.double_q = e->hp.double_q;
            h.gamma = e->hp.gamma; h.pm = rh_pm ? 1 : 0;
            for (int st = 0; st < h.nstream; st++) {
                const int hl_ = st == 0 ? ha_l : hv_l, pl_ = st == 0 ? rh_pa : rh_pv; const LayerDev& H = e->L[hl_]; const LayerDev& P = e->L[pl_]; RedHeadStream& T = h.st[st];
                T.part[0] = rh_part[st][0]; T.part[1] = rh_part[st][1]; T.partT[0] = rh_partT[st][0]; T.partT[1] = rh_partT[st][1]; T.pbias[0] = e->p_on + P.b_off; T.pbias[1] = e->p_tg + P.b_off; T.pact = P.act;
                T.W[0] = e->p_on + H.w_off; T.W[1] = e->p_tg + H.w_off; T.hbias[0] = e->p_on + H.b_off; T.hbias[1] = e->p_tg + H.b_off; T.N = H.N; T.hact = H.act;
                T.y_on = e->act_on[pl_]; T.dpre = e->dact[hl_]; T.dsrc = e->dact[pl_];
            }
            if (h.nstream == 1) h.st[1] = h.st[0];      // (never read: keeps every pointer of the record valid)
            { bool allT = true; for (int st = 0; st < h.nstream; st++) allT = allT && h.st[st].partT[0] && h.st[st].partT[1];      // transposed copies: all four or none
              if (!allT || e->opt.no_head_cols4 == 2) for (int st = 0; st < 2; st++) h.st[st].partT[0] = h.st[st].partT[1] = nullptr; }
            h.bm_a = e->gb_a2; h.bm_r = e->gb_r2; h.bm_done = e->gb_done2; h.bm_w = e->gb_w2;
            h.w_is = e->w_is; h.td = e->td; h.q_on_s = e->q_on_s; h.q_on_sp = e->q_on_sp; h.q_tg_sp = e->q_tg_sp; h.ytarget = e->ytarget; h.best = e->best; h.hl = hl_buf; h.stt = e->state;
            h.idx = e->idx; h.idx_pre = e->idx_pre;
            const int Gc = B / 4, NC = h.K / 32;
            h.partials = palloc(e, (size_t)Gc * 12 * h.NO * NC); h.tickets = (unsigned*)palloc(e, (size_t)Gc);
            h.ypm = (h.S > 1 && !e->opt.no_rh_pm) ? palloc(e, (size_t)Gc * h.nstream * h.K * 4) : nullptr;
            HIPCHK(hipMemset(h.tickets, 0, (size_t)Gc * 4));      // armed once; every launch's last arrivers re-arm their groups
            if (e->opt.drqn_stamps) { h.stamps = (unsigned long long*)palloc(e, 64); HIPCHK(hipMemset(h.stamps, 0, 256)); e->drqn_stamps = h.stamps; }
            const RedHeadArgs* h_dev = upload(e, std::vector<RedHeadArgs>(1, h));
            e->prog.push_back({h.S == 1 ? "head_cols4" : "red_head", [=](dqn_engine* en) { launch_red_head(en->stream, h, h_dev, en->step_sampled ? 1 : 0, en->step_take_pre ? 1 : 0); }});
        }
        else if (fuse_heads) {
            HeadTdArgs h; memset(&h, 0, sizeof h);
            h.B = B; h.nA = e->nA; h.dueling = e->hp.dueling; h.double_q = e->hp.double_q; h.gamma = e->hp.gamma; h.prio_beta = e->hp.prio_beta; h.cap2 = e->cap2;
            h.bm_a = e->gb_a2; h.bm_r = e->gb_r2; h.bm_done = e->gb_done2; h.bm_w = e->gb_w2;
            h.w_is = e->w_is; h.td = e->td; h.q_on_s = e->q_on_s; h.q_on_sp = e->q_on_sp; h.q_tg_sp = e->q_tg_sp; h.ytarget = e->ytarget; h.best = e->best; h.st = e->state;
            h.hl = hl_buf; h.idx = e->idx; h.idx_pre = e->idx_pre;
            auto fill = [&](HeadLayer& H, int l) {
                const LayerDev& L = e->L[l];
                H.K = L.K; H.N = L.N; H.S = dqn_nchunks(L.K, L.fwd_kc); H.kc = dqn_chunk_len(L.K, L.fwd_kc); H.act = L.act;
                H.W[0] = e->p_on + L.w_off; H.bias[0] = e->p_on + L.b_off; H.W[1] = e->p_tg + L.w_off; H.bias[1] = e->p_tg + L.b_off;
                if (L.src < 0) { H.X[0] = H.X[1] = e->x0; H.ldx[0] = H.ldx[1] = ld0; H.c0[0] = 0; H.c0[1] = B; }
                else { H.X[0] = e->act_on[L.src]; H.ldx[0] = ncon; H.c0[0] = 0; H.X[1] = e->act_tg[L.src]; H.ldx[1] = B; H.c0[1] = 0; H.XT[0] = actT[L.src][0]; H.XT[1] = actT[L.src][1]; }
                H.dpre = e->dact[l];
                if (L.src >= 0) { H.dsrc = e->dact[L.src]; H.ysrc = e->act_on[L.src]; H.ldy = ncon; H.act_src = e->L[L.src].act; }
            };
            fill(h.adv, ha_l);
            if (hv_l >= 0) { fill(h.val, hv_l); h.join = (e->L[hv_l].src == e->L[ha_l].src && e->L[ha_l].src >= 0) ? 1 : 0; }
            {   // stage the head weights in LDS when they fit beside the input columns
                bool ok = true; size_t wb = 0;
                for (int l : levels.back()) { const LayerDev& L = e->L[l]; ok = ok && ((size_t)L.K * L.N) % 4 == 0 && L.w_off % 4 == 0; wb += (size_t)2 * L.K * L.N * 4; }
                h.stage_w = 0;
                if (ok) { h.stage_w = 1; if (head_td_lds_bytes(h) > 60 * 1024) h.stage_w = 0; }
                (void)wb;
            }
            h.dbg = e->opt.head_dbg;
            const HeadTdArgs* h_dev = upload(e, std::vector<HeadTdArgs>(1, h));
            e->prog.push_back({"head_td", [=](dqn_engine* en) { launch_head_td(en->stream, h, h_dev, en->step_sampled ? 1 : 0, en->step_take_pre ? 1 : 0); }});
        }
        else if (!rec) e->prog.push_back({"td_huber", [=](dqn_engine* en) { TdArgs a = t; a.bump_sample_ctr = en->step_sampled ? 1 : 0; a.take_pre = en->step_take_pre ? 1 : 0; launch_td(en->stream, a); }});
        else {
            TdDrqnArgs d; memset(&d, 0, sizeof d); d.B = Bb; d.T = T; d.nA = e->nA; d.ncon = ncon; d.dueling = e->hp.dueling; d.double_q = e->hp.double_q; d.gamma = e->hp.gamma;
            d.on_val = t.on_val; d.on_adv = t.on_adv; d.tg_val = t.tg_val; d.tg_adv = t.tg_adv; d.d_val = t.d_val; d.d_adv = t.d_adv;
            d.a = e->r_a; d.r = e->r_r; d.done = e->r_done; d.mask = e->r_mask; d.td = e->td; d.st = e->state;
            e->prog.push_back({"td_huber_drqn", [=](dqn_engine* en) { launch_td_drqn(en->stream, d); }});
        }
    }
    {
        // update_priorities! (src/solver.jl:231-233) needs only idx and td.  For small batches it rides as a dedicated block of the Adam launch; at
        // B > 64 its B leaf paths x log2(cap) levels (~70 us at B = 512, 1e6 leaves) would be that launch's tail, so it runs on a side stream
        // concurrently with the whole backward pass and is joined before the optimizer (a graph fork/join costs ~15 us -- only worth it here).
        // ... unless a backward launch can carry it as a workgroup of its own (r03: the fork + join nodes themselves cost 12 + 10 us of the main
        // stream at config 5, and the block -- split in two, update then draws -- is shorter than the 57-87 us launches it rides in)
        bool big_in_bwd = false;
        if (!rec && e->hp.prioritized_replay && Bb > 64 && Bb <= 1024 && mf && !e->comm && !e->sim_world && !e->opt.prio_fork) {
            int carriers = 0;
            for (const auto& lvq : levels) for (int l2 : lvq) { const LayerDev& L2 = e->L[l2]; if (L2.kind != DQN_LAYER_LSTM && gemm_dw_eligible(L2, B, L2.src < 0 ? ld0 : ncon)) { carriers++; break; } }
            big_in_bwd = carriers >= 2;
        }
        e->prio_in_bwd = big_in_bwd;
        if (!rec && e->hp.prioritized_replay && Bb > 64 && !big_in_bwd) {
            e->prio_forked = true;
            e->prog.push_back({"prio_fork", [](dqn_engine* en) {
                hipEventRecord(en->ev_fork, en->stream); hipStreamWaitEvent(en->stream2, en->ev_fork, 0);
                launch_update_priorities(en->stream2, en->B, en->cap2, en->idx, en->td, en->hp.prio_eps, en->hp.prio_alpha, en->tree, en->state, 0, 1.0, 1.0, nullptr, 0,
                                         en->hp.sample_distinct ? nullptr : en->idx_pre, en->hp.seed, en->B);      // + the next step's index draw (stratified mode)
                hipEventRecord(en->ev_join, en->stream2); }});
        } else e->prio_forked = false;
    }
    // ---------------- data-parallel replicas: which layers' dW is computed AFTER the exchange from gathered operands (dp.hip)
    const int W = e->sim_world ? e->sim_world : e->world;
    const bool dp_on = (e->comm || e->sim_world) && !rec && !e->opt.dp_allreduce;
    bool dp_layer[DQN_MAX_LAYERS] = {}; int n_dp = 0;
    if (dp_on) for (int i = 0; i < e->nl; i++) {
        const LayerDev& L = e->L[i];
        // a wide dense layer whose operands (X: K x B, dpre: N x B per rank) are smaller than its gradient, one chain over all W*B samples
        if (L.kind == DQN_LAYER_DENSE && mf && (size_t)(L.K + 1) * L.N > (size_t)4 * (L.K + L.N) * B && L.dw_kc == 0 && B % 32 == 0 && gemm_dw_eligible(L, W * B, B) && n_dp < 8) { dp_layer[i] = true; n_dp++; }
    }
    // ---------------- backward of the online net on the s columns (Zygote through src/solver.jl:219-225)
    std::vector<RSeg> final_segs;   // dW split-K slabs: nothing reads the gradient before Adam, so ONE reduce launch at the end
    // ---------------- Adam overlap (single GPU).  A layer's gradient is final once the launch of its level has run (fused heads: once the launch
    // carrying their dW tail tasks has run), and from then on nothing reads its parameters (its dX ran at its own level).  So the Adam update of
    // every such layer -- stream (unsplit dW) or slab reduce + update (split-K dW) -- rides as TAIL workgroups of the NEXT backward launch: the
    // bandwidth-bound work overlaps the latency-bound conv backward launches; update_priorities! (needs only idx, td) rides on the first of them.
    // The final k_adam is left with the first level's layers and the beta-power tick.
    bool segs_ok = true;
    for (int i = 0; i < e->nl; i++) { const LayerDev& L = e->L[i]; if (L.kind != DQN_LAYER_LSTM && dqn_nchunks(L.npos * B, L.dw_kc) > 1) segs_ok = segs_ok && L.w_off % 4 == 0 && ((size_t)(L.K + 1) * L.N) % 4 == 0; }
    // (removed in r06: DQN_ADAM_MODE=1, the Adam update of layers whose gradient is already final carried as tail workgroups of the backward launches -- measured slower in
    //  rounds 2, 3, 4 and, as "the stream in the free CU slots", 5: 159.3 vs 156.9 us/step in r02, +3.4 us per carrying launch for -2.75 us of Adam in r05; docs/history/r02.md, r05.md)
    bool prio_placed = false, prio_draw_pending = false;
    auto base_job = [&]() {
        AdamJob J; memset(&J, 0, sizeof J);
        J.p = e->p_on; J.m = e->m; J.v = e->v; J.g = e->grad; J.g_out = e->grad; J.state = e->state; J.gmax_part = e->gmax_part;
        J.f64mode = e->hp.adam_f64_scalars; J.lr = e->hp.learning_rate; J.b1 = e->hp.adam_beta1; J.b2 = e->hp.adam_beta2; J.eps = e->hp.adam_eps; J.gscale = 1.0f;
        J.wt = (e->nl > 0 && (e->L[0].opt & DQN_LOPT_ST_WT)) ? 1 : 0;
        return J;
    };
    auto prio_args = [&]() { PrioArgs pa; memset(&pa, 0, sizeof pa); pa.n = B; pa.cap2 = e->cap2; pa.idx = e->idx; pa.td = e->td; pa.eps = e->hp.prio_eps; pa.alpha = e->hp.prio_alpha; pa.tree = e->tree;
                              // hp.sample_distinct: pre-drawn (and deduped by the priority block) for the small batches whose fused sample + gather launch understands the list
                              // r06: ... and for the large batches whose priority workgroup rides a backward launch (prio_in_bwd): the list it leaves in idx_pre is final -- the
                              // Adam launch's pre-gather reads it as it reads a stratified one
                              const bool dist_ok = !e->hp.sample_distinct || (Bb <= 64 && !(e->hp.obs_dtype == DQN_OBS_U8 && !e->arena_u8 && (e->E & 3) == 0)) || (Bb > 64 && e->prio_in_bwd);
                              if ((Bb <= 64 || e->prio_in_bwd) && dist_ok) { pa.idx_pre = e->idx_pre; pa.seed = e->hp.seed; pa.B = Bb; pa.distinct = e->hp.sample_distinct ? 1 : 0; }      // the fused sample+gather launch (B <= 64) consumes them
                              return pa; };
    const bool prio_in_adam = e->hp.prioritized_replay && !rec && (Bb <= 64 || e->prio_in_bwd);      // larger batches: a backward launch's workgroup, or the side stream (prio_fork)
    // pre-gather (common.h PreGather): needs the priority block (which also draws the next indices) OUT of the Adam launch -- it rides as
    // workgroup 0 of the first LDS-tiled backward launch instead
    // large batches: the priority update runs on the side stream (prio_fork) and draws the next indices there; k_td takes the pre-drawn batch
    const bool pg_want = e->hp.prioritized_replay && !rec && (prio_in_adam ? (fuse_heads || e->prio_in_bwd) : e->prio_forked) && !e->sim_world &&
                         (e->hp.obs_dtype != DQN_OBS_U8 || e->arena_u8) && !e->opt.no_pregather && (!e->hp.sample_distinct || Bb <= 64 || e->prio_in_bwd);      // u8 rows: only onto the byte arena; distinct mode at B > 64 without a carrying backward launch: sample launch + gather launch every step
    bool joined = false;
    std::vector<VTask> tail_pend;    // small tasks waiting for a launch to ride on (fused heads: their dW/db and the loss fold)
    auto make_tail = [&](std::vector<VTask>& v) {
        GemmTail t = gemm_no_tail();
        if (v.empty()) return t;
        unsigned blocks = 0;
        for (auto& q : v) { q.first_block = blocks; blocks += valu_task_blocks(q); }
        t.tasks = upload(e, v); t.n = (int)v.size(); t.blocks = blocks; v.clear();
        return t;
    };
    // dp_overlap: with fused heads the operands of the flagged wide dense layers (X from the forward pass, dpre from k_head_td) are final HERE, before any
    // backward launch: pack and exchange them now, on the exchange stream, while the conv backward runs (SURVEY 8e "overlapped with backward")
    // Measured at world 1, where there is nothing to hide, the fork + join of the exchange stream cost 26 us per step (one-graph replica step 155.8 -> 181.9 us): it pays
    // once the first all-gather takes longer than that.  r06: decided HERE from the world size and the bytes instead of by an environment knob -- the wide operands of one
    // rank over W - 1 ring hops of one 153 GB/s xGMI link + 15 us of collective latency (the worst shape RCCL can pick on point-to-point links; unmeasured: no multi-GPU
    // box) against the 26 us; DQN_DP_OVERLAP = 1 / 0 still forces it (tests, A/B on real hardware)
    bool want_overlap = e->opt.dp_overlap > 0;
    if (e->opt.dp_overlap < 0 && dp_on) {
        double bytes_a = 0.0; for (int i = 0; i < e->nl; i++) if (dp_layer[i]) bytes_a += 4.0 * (double)(e->L[i].K + e->L[i].N) * (double)B;
        want_overlap = W >= 2 && !e->sim_world && (double)(W - 1) * bytes_a / 153e9 * 1e6 + 15.0 > 26.0;
    }
    bool overlap_cand = dp_on && n_dp > 0 && fuse_heads && levels.size() >= 2 && want_overlap;
    if (overlap_cand) for (int i = 0; i < e->nl; i++) if (dp_layer[i]) { bool in_lvl = false; for (int l : levels[levels.size() - 2]) in_lvl = in_lvl || l == i; overlap_cand = overlap_cand && in_lvl; }
    e->dp_overlap = false; e->prog_pre1_end = 0;
    if (overlap_cand) { e->prog.push_back({"dp_pack_wide", [](dqn_engine* en) { if (en->dp_overlap) launch_dp_pack(en->stream, en->dp_pk_a); }}); e->prog_pre1_end = e->prog.size(); }
    for (int li = (int)levels.size() - 1; li >= 0; li--) {
        const auto& lv = levels[li];
        std::vector<VTask> pend;
        if (fuse_heads && li + 1 == (int)levels.size()) {
            // the head layers' dX already ran inside k_head_td; their dW/db (one B-long chain per weight) and the loss fold are tail tasks
            for (int l : lv) {
                const LayerDev L = e->L[l];
                const int S = dqn_nchunks(B, L.dw_kc);      // large batches: plan chunks of the B-long chains, slabs summed with the other dW slabs
                float* part = S > 1 ? palloc(e, (size_t)S * (L.K + 1) * L.N) : nullptr;
                VTask t; memset(&t, 0, sizeof t); t.kind = 1; t.L = L; t.X = L.src < 0 ? e->x0 : e->act_on[L.src]; t.ldx = L.src < 0 ? ld0 : ncon; t.dpre = e->dact[l]; t.B = B; t.S = S; t.kc = dqn_chunk_len(B, L.dw_kc);
                t.out = S > 1 ? part : e->grad + L.w_off; tail_pend.push_back(t);
                if (S > 1) { RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)(L.K + 1) * L.N; r.mode = 2; r.out = e->grad + L.w_off; final_segs.push_back(r); }
            }
            VTask f; memset(&f, 0, sizeof f); f.kind = 3; f.dpre = hl_buf; f.B = B; f.out = &e->state->loss; tail_pend.push_back(f);
            continue;
        }
        bool dw_done_sibling = false;   // the level's two sibling layers got their dW from one fused launch
        struct DwL { bool on = false; LayerDev L; int nprob = 0; const float* X[2]; int ldx = 0; const float* d[2]; float* o[2]; const char* name = ""; } dwl;
        struct DxL { bool on = false; LayerDev L; int nsrc = 0; const float* W[2]; const float* d[2]; float* out = nullptr; const float* ys = nullptr; int act_src = 0; const char* name = ""; } dxl;
        auto flush_dw = [&]() { if (!dwl.on) return; const DwL a = dwl; e->prog.push_back({a.name, [=](dqn_engine* en) { launch_gemm_dw(en->stream, a.L, a.nprob, a.X, a.ldx, a.d, B, a.o); }}); dwl.on = false; };
        auto flush_dx = [&]() { if (!dxl.on) return; const DxL a = dxl; e->prog.push_back({a.name, [=](dqn_engine* en) { launch_gemm_dx(en->stream, a.L, a.nsrc, a.W, a.d, B, a.out, a.ys, ncon, a.act_src); }}); dxl.on = false; };
        for (int k = (int)lv.size() - 1; k >= 0; k--) {
            const int l = lv[k]; const LayerDev L = e->L[l];
            const float* X = L.src < 0 ? e->x0 : e->act_on[L.src]; const int ldx = L.src < 0 ? ld0 : ncon;
            float* dpre = e->dact[l];
            if (L.kind == DQN_LAYER_LSTM) {
                // BPTT over the s-sequence: T single-workgroup steps produce dG (gate pre-activation gradients) for all columns,
                // then Wi|b, Wh and the input gradient are ordinary dense contractions over the T*B columns.
                float* grad = e->grad;
                if (lstm_seq_fits(L.H, Bb, T)) {
                    LstmBwdArgs a; a.t = 0; a.T = T; a.H = L.H; a.B = Bb; a.TB = B; a.gates = e->gates[l]; a.tc = e->tcb[l]; a.cprev = e->cprev_buf[l]; a.Wh = e->p_on + L.wh_off;
                    a.dH = dpre; a.dG = e->dG[l]; a.dhn = e->dhn[l]; a.dcn = e->dcn[l]; a.g_h0 = grad + L.h0_off; a.g_c0 = grad + L.c0_off;
                    e->prog.push_back({pname(e, "lstm_bwd_seq", L.kind, l), [=](dqn_engine* en) { launch_lstm_bwd_seq(en->stream, a); }});
                } else
                for (int t = T - 1; t >= 0; t--) {
                    LstmBwdArgs a; a.t = t; a.T = T; a.H = L.H; a.B = Bb; a.TB = B; a.gates = e->gates[l]; a.tc = e->tcb[l]; a.cprev = e->cprev_buf[l]; a.Wh = e->p_on + L.wh_off;
                    a.dH = dpre; a.dG = e->dG[l]; a.dhn = e->dhn[l]; a.dcn = e->dcn[l]; a.g_h0 = grad + L.h0_off; a.g_c0 = grad + L.c0_off;
                    e->prog.push_back({pname(e, "lstm_bwd", L.kind, l), [=](dqn_engine* en) { launch_lstm_bwd_step(en->stream, a); }});
                }
                LayerDev Vi = L; Vi.kind = DQN_LAYER_DENSE; Vi.out_feat = L.N; Vi.act = DQN_ACT_IDENTITY;                    // Wi | b  : (K+1) x 4H
                LayerDev Vh = Vi; Vh.K = L.H; Vh.in_feat = L.H; Vh.w_off = L.wh_off; Vh.b_off = L.wh_off + (size_t)L.H * L.N;  // Wh | junk
                const float* dG = e->dG[l];
                auto emit_dw1 = [&](const LayerDev V, const float* Xv, int ldv, const char* nm) {
                    const int S = dqn_nchunks(B, V.dw_kc);
                    float* part = S > 1 ? palloc(e, (size_t)S * (V.K + 1) * V.N) : nullptr; float* dst = S > 1 ? part : grad + V.w_off;
                    // small recurrent layers (config 4: (25+1) x 128 and (32+1) x 128 weights, 256 columns): the two dW contractions as VALU tasks of ONE launch
                    // (15 us) beat two LDS-tiled MFMA launches of 13 us each (r03: 113.4 -> 102.7 us/step); the MFMA tiles win once the sample chains get long
                    const bool small_dw = B <= 256 && (size_t)(V.K + 1) * V.N <= 16384 && !e->opt.lstm_dw_mfma;
                    if (mf && !small_dw && gemm_dw_eligible(V, B, ldv)) { struct A { const float* X[1]; const float* d[1]; float* o[1]; } a; a.X[0] = Xv; a.d[0] = dG; a.o[0] = dst;
                        e->prog.push_back({nm, [=](dqn_engine* en) { launch_gemm_dw(en->stream, V, 1, a.X, ldv, a.d, B, a.o); }}); }
                    else if (mf && !small_dw && mfma_dw_ok(V, B)) e->prog.push_back({nm, [=](dqn_engine* en) { launch_mfma_dw(en->stream, V, Xv, ldv, dG, B, grad, part, false); }});
                    else { VTask t; memset(&t, 0, sizeof t); t.kind = 1; t.L = V; t.X = Xv; t.ldx = ldv; t.dpre = dG; t.B = B; t.S = S; t.kc = dqn_chunk_len(B, V.dw_kc); t.out = dst; add_valu(e, pend, t); }
                    if (S > 1) { RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)(V.K + 1) * V.N; r.mode = 2; r.out = grad + V.w_off; final_segs.push_back(r); }
                };
                emit_dw1(Vh, e->hprev_buf[l], B, pname(e, "dw_wh", L.kind, l));      // first: its junk bias row is then overwritten by nothing that matters
                emit_dw1(Vi, X, ldx, pname(e, "dw_wi", L.kind, l));
                if (L.src >= 0) {
                    const int src = L.src; const int act_src = e->L[src].act; float* out = e->dact[src]; const float* ysrc = e->act_on[src]; const float* P = e->p_on;
                    const int S = (mf && gemm_dx_internal_chunks(Vi, B, ncon)) ? 1 : dqn_nchunks(Vi.N, Vi.dx_kc);      // internal: the launch combines its plan chunks itself
                    float* part = S > 1 ? palloc(e, (size_t)S * Vi.in_feat * B) : nullptr;
                    if (mf && gemm_dx_eligible(Vi, B, ncon)) { struct A1 { const float* W[1]; const float* d[1]; } a; a.W[0] = P + Vi.w_off; a.d[0] = dG; float* dst = S > 1 ? part : out; const float* ys = S > 1 ? nullptr : ysrc;
                        e->prog.push_back({pname(e, "dx", L.kind, l), [=](dqn_engine* en) { launch_gemm_dx(en->stream, Vi, 1, a.W, a.d, B, dst, ys, ncon, act_src); }}); }
                    else if (mf && mfma_dx_ok(Vi, B, ncon)) e->prog.push_back({pname(e, "dx", L.kind, l), [=](dqn_engine* en) { launch_mfma_dx(en->stream, Vi, P, dG, B, out, part, nullptr, ysrc, ncon, act_src, false); }});
                    else { VTask t; memset(&t, 0, sizeof t); t.kind = 2; t.L = Vi; t.P = P; t.dpre = dG; t.B = B; t.S = S; t.kc = dqn_chunk_len(Vi.N, Vi.dx_kc); t.out = S > 1 ? part : out; t.ysrc = ysrc; t.ldy = ncon; t.act_src = act_src; add_valu(e, pend, t); }
                    if (S > 1) { flush_valu(e, pend, pname(e, "bwd_valu", L.kind, l)); std::vector<RSeg> one; RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)Vi.in_feat * B; r.mode = 1; r.act = act_src; r.ysrc = ysrc; r.B = B; r.ldy = ncon; r.out = out; one.push_back(r); emit_reduce(e, one, pname(e, "dx_reduce", L.kind, l)); }
                }
                continue;
            }
            if (!dp_layer[l]) {   // dW / db  (layers flagged for the gather exchange get theirs after the collective)
                const int S = dqn_nchunks(L.npos * B, L.dw_kc);
                float* part = S > 1 ? palloc(e, (size_t)S * (L.K + 1) * L.N) : nullptr;
                float* grad = e->grad;
                float* dst = S > 1 ? part : grad + L.w_off;
                if (mf && gemm_dw_eligible(L, B, ldx)) {
                    // sibling layers of this level with identical geometry and the same input share ONE launch
                    if (k == (int)lv.size() - 1 && lv.size() == 2 && same_geo(e->L[lv[0]], e->L[lv[1]]) && e->L[lv[0]].dw_kc == e->L[lv[1]].dw_kc) {
                        const LayerDev L0 = e->L[lv[0]]; const int S0 = S;
                        float* part0 = S0 > 1 ? palloc(e, (size_t)S0 * (L0.K + 1) * L0.N) : nullptr;
                        flush_dw(); dwl.on = true; dwl.L = L; dwl.nprob = 2; dwl.ldx = ldx; dwl.name = pname(e, "dw2", L.kind, l);
                        dwl.X[0] = X; dwl.d[0] = dpre; dwl.o[0] = dst; dwl.X[1] = X; dwl.d[1] = e->dact[lv[0]]; dwl.o[1] = S0 > 1 ? part0 : grad + L0.w_off;
                        if (S0 > 1) { RSeg r; memset(&r, 0, sizeof r); r.part = part0; r.S = S0; r.elems = (unsigned long long)(L0.K + 1) * L0.N; r.mode = 2; r.out = grad + L0.w_off; final_segs.push_back(r); }
                        dw_done_sibling = true;
                    } else if (!(dw_done_sibling && k == 0 && lv.size() == 2)) {
                        flush_dw(); dwl.on = true; dwl.L = L; dwl.nprob = 1; dwl.ldx = ldx; dwl.name = pname(e, "dw", L.kind, l);
                        dwl.X[0] = dwl.X[1] = X; dwl.d[0] = dwl.d[1] = dpre; dwl.o[0] = dwl.o[1] = dst;
                    }
                }
                else if (mf && mfma_dw_ok(L, B)) e->prog.push_back({pname(e, "dw", L.kind, l), [=](dqn_engine* en) { launch_mfma_dw(en->stream, L, X, ldx, dpre, B, grad, part, false); }});
                else { VTask t; memset(&t, 0, sizeof t); t.kind = 1; t.L = L; t.X = X; t.ldx = ldx; t.dpre = dpre; t.B = B; t.S = S; t.kc = dqn_chunk_len(L.npos * B, L.dw_kc); t.out = dst; add_valu(e, pend, t); }
                if (S > 1 && !(dw_done_sibling && k == 0 && lv.size() == 2)) { RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)(L.K + 1) * L.N; r.mode = 2; r.out = grad + L.w_off; final_segs.push_back(r); }
            }
            if (L.src < 0) continue;
            // dX, then act' of the producing layer; the two streams of a dueling net meet at the base output (dX_val + dX_adv)
            const int src = L.src; const bool is_join = e->hp.dueling && src == e->last_base && L.stream != DQN_STREAM_BASE;
            const bool dense = L.kind == DQN_LAYER_DENSE; const int S_plan = dense ? dqn_nchunks(L.N, L.dx_kc) : 1;
            const float* P = e->p_on; const int act_src = e->L[src].act;
            const bool join_lds = is_join && lv.size() == 2 && mf && same_geo(e->L[lv[0]], e->L[lv[1]]) && e->L[lv[0]].dx_kc == e->L[lv[1]].dx_kc &&
                                  gemm_dx_eligible(L, B, ncon, 2) && (S_plan == 1 || gemm_dx_internal_chunks(L, B, ncon, 2));
            if (join_lds) {
                // both streams in ONE launch: the kernel accumulates dX_val and dX_adv separately and adds them (val first)
                if (k == (int)lv.size() - 1) {
                    const LayerDev Lv = e->L[lv[0]], La = e->L[lv[1]];
                    flush_dx(); dxl.on = true; dxl.L = Lv; dxl.nsrc = 2; dxl.W[0] = P + Lv.w_off; dxl.d[0] = e->dact[lv[0]]; dxl.W[1] = P + La.w_off; dxl.d[1] = e->dact[lv[1]];
                    dxl.out = e->dact[src]; dxl.ys = e->act_on[src]; dxl.act_src = act_src; dxl.name = pname(e, "dx_join", L.kind, l);
                }
                continue;
            }
            float* out = e->dact[src]; const float *addend = nullptr, *ysrc = e->act_on[src];
            if (is_join && !joined) { out = e->join_tmp; ysrc = nullptr; joined = true; }
            else if (is_join) { addend = e->join_tmp; flush_valu(e, pend, pname(e, "bwd_valu", L.kind, l)); }   // depends on the first stream's dX
            const int S = (mf && !addend && gemm_dx_internal_chunks(L, B, ncon)) ? 1 : S_plan;      // internal: the launch combines its plan chunks itself (no slabs)
            float* part = S > 1 ? palloc(e, (size_t)S * L.in_feat * B) : nullptr;
            if (mf && !addend && gemm_dx_eligible(L, B, ncon)) {
                flush_dx(); dxl.on = true; dxl.L = L; dxl.nsrc = 1; dxl.W[0] = dxl.W[1] = P + L.w_off; dxl.d[0] = dxl.d[1] = dpre;
                dxl.out = S > 1 ? part : out; dxl.ys = S > 1 ? nullptr : ysrc; dxl.act_src = act_src; dxl.name = pname(e, "dx", L.kind, l);
                if (S > 1) flush_dx();   // its partial slabs are reduced right below
            }
            else if (mf && L.N >= 16 && mfma_dx_ok(L, B, ncon)) e->prog.push_back({pname(e, "dx", L.kind, l), [=](dqn_engine* en) { launch_mfma_dx(en->stream, L, P, dpre, B, out, part, addend, ysrc, ncon, act_src, false); }});
            else { VTask t; memset(&t, 0, sizeof t); t.kind = 2; t.L = L; t.P = P; t.dpre = dpre; t.B = B; t.S = S; t.kc = dqn_chunk_len(L.N, L.dx_kc); t.out = S > 1 ? part : out; t.addend = addend; t.ysrc = ysrc; t.ldy = ncon; t.act_src = act_src; add_valu(e, pend, t); }
            if (S > 1) {
                flush_valu(e, pend, pname(e, "bwd_valu", L.kind, l));
                std::vector<RSeg> one; RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)L.in_feat * B; r.mode = 1; r.act = act_src; r.addend = addend; r.ysrc = ysrc; r.B = B; r.ldy = ncon; r.out = out; one.push_back(r);
                emit_reduce(e, one, pname(e, "dx_reduce", L.kind, l));
            }
        }
        GemmTail tail = gemm_no_tail();
        if (!tail_pend.empty()) {
            if (dwl.on || dxl.on) tail = make_tail(tail_pend);                       // rides in the last workgroups of this level's LDS-tiled launch
            else { for (auto& t : tail_pend) pend.push_back(t); tail_pend.clear(); } // or joins this level's VALU task table
        }
        if (pg_want && prio_in_adam && (dwl.on || dxl.on) && prio_draw_pending) {      // second half of a split priority block: the next sample()'s draws
            tail.adam = base_job(); tail.adam.prio = prio_args(); tail.adam.prio.phase = 2; tail.has_adam = 1; prio_draw_pending = false; prio_placed = true;
        }
        else if (pg_want && prio_in_adam && (dwl.on || dxl.on) && !prio_placed && !prio_draw_pending) {
            // the block rides as workgroup 0 of this launch.  When a LATER backward launch can carry a workgroup too, the block is SPLIT: update_priorities!
            // here, the draws there -- each half shorter than the dX chains it hides under (r03 ktrace: the whole block lived 11 us, longer than any)
            bool later = false;
            for (int lj = li - 1; lj >= 0 && !later; lj--) for (int l2 : levels[lj]) {
                const LayerDev& L2 = e->L[l2]; const int ldx2 = L2.src < 0 ? ld0 : ncon;
                if (mf && L2.kind != DQN_LAYER_LSTM && !dp_layer[l2] && gemm_dw_eligible(L2, B, ldx2)) later = true;
            }
            tail.adam = base_job(); tail.adam.prio = prio_args(); tail.has_adam = 1;
            if (later) { tail.adam.prio.phase = 1; prio_draw_pending = true; } else prio_placed = true;
        }
        if (pg_want && prio_in_adam && !prio_placed && !pend.empty() && !tail.has_adam && !e->prio_in_bwd /* large batches: only the long LDS-tiled launches can hide the block */) { PrioArgs pa = prio_args(); if (prio_draw_pending) { pa.phase = 2; prio_draw_pending = false; } flush_valu(e, pend, pname(e, "bwd_valu", e->L[lv[0]].kind, lv[0]), &pa); prio_placed = true; }
        flush_valu(e, pend, pname(e, "bwd_valu", e->L[lv[0]].kind, lv[0]));
        const char* tsuf = (tail.has_adam && adam_job_blocks(tail.adam) == 1 && tail.adam.prio.n > 0) ? "+prio" : "+adam_tail";      // a job that is only the priority block
        auto tailed = [&](const char* base) { if (!tail.has_adam) return base; char nm[80]; snprintf(nm, sizeof nm, "%s%s", base, tsuf); e->prog_names.push_back(nm); return e->prog_names.back().c_str(); };
        if (dwl.on && dxl.on) {      // dW and dX of this level in ONE launch
            const DwL a = dwl; const DxL x = dxl; dwl.on = dxl.on = false;
            char nm[80]; snprintf(nm, sizeof nm, "%s+%s%s", a.name, x.name, tail.has_adam ? tsuf : ""); e->prog_names.push_back(nm); const char* name = e->prog_names.back().c_str();
            e->prog.push_back({name, [=](dqn_engine* en) { launch_gemm_dwdx(en->stream, a.L, a.nprob, a.X, a.ldx, a.d, B, a.o, x.L, x.nsrc, x.W, x.d, x.out, x.ys, ncon, x.act_src, tail); }});
        }
        else if (dwl.on) { const DwL a = dwl; dwl.on = false; e->prog.push_back({tailed(a.name), [=](dqn_engine* en) { launch_gemm_dw(en->stream, a.L, a.nprob, a.X, a.ldx, a.d, B, a.o, 0, 0, 0, tail); }}); }
        else if (dxl.on) { const DxL a = dxl; dxl.on = false; e->prog.push_back({tailed(a.name), [=](dqn_engine* en) { launch_gemm_dx(en->stream, a.L, a.nsrc, a.W, a.d, B, a.out, a.ys, ncon, a.act_src, tail); }}); }
        flush_dw(); flush_dx();
    }
    if (!tail_pend.empty()) { std::vector<VTask> own(tail_pend); tail_pend.clear(); flush_valu(e, own, "head_dw"); }      // single-level network: nothing to ride on
    if (e->prio_forked) e->prog.push_back({"prio_join", [](dqn_engine* en) { hipStreamWaitEvent(en->stream, en->ev_join, 0); }});
    memset(&e->adam_segs, 0, sizeof e->adam_segs);
    {
        bool ok = !final_segs.empty() && final_segs.size() <= 8; unsigned long long tot = 0;
        for (auto& r : final_segs) { const unsigned long long beg = (unsigned long long)(r.out - e->grad); ok = ok && beg % 4 == 0 && r.elems % 4 == 0 && r.S2 == 0; }
        if (ok) {
            for (auto& r : final_segs) { const int q = e->adam_segs.n++; e->adam_segs.beg[q] = (unsigned long long)(r.out - e->grad); e->adam_segs.end[q] = e->adam_segs.beg[q] + r.elems; e->adam_segs.part[q] = r.part; e->adam_segs.S[q] = r.S; tot += r.elems; }
            e->adam_segs.blocks = (unsigned)((tot + 255) / 256);
        }
        if (!final_segs.empty()) e->final_reduce_step = (long)e->prog.size();
    }
    const std::vector<RSeg> slab_segs(final_segs);      // emit_reduce consumes the list
    emit_reduce(e, final_segs, "dw_reduce_all");
    e->dp_gather = false;
    DpSumArgs dsum; memset(&dsum, 0, sizeof dsum);
    struct DpDw { LayerDev L; const float* X; unsigned long long x_off, d_off; };
    std::vector<DpDw> dpdw;
    if (n_dp > 0) {
        // per-rank block: [X of each distinct producer: K x B][dpre of each flagged layer: N x B][every other gradient range]
        DpPackArgs pk; memset(&pk, 0, sizeof pk); unsigned long long off = 0;
        std::vector<std::pair<const float*, unsigned long long>> xs;
        for (int l = 0; l < e->nl; l++) if (dp_layer[l]) {
            const LayerDev& L = e->L[l]; const float* X = L.src < 0 ? e->x0 : e->act_on[L.src]; const int ldx = L.src < 0 ? ld0 : ncon;
            unsigned long long xo = ~0ull; for (auto& q : xs) if (q.first == X) xo = q.second;
            if (xo == ~0ull) { xo = off; xs.push_back({X, xo}); DpRegion r; memset(&r, 0, sizeof r); r.S = 1; r.src = X; r.dst = off; r.n = (unsigned long long)L.K * B; r.B = B; r.ld = ldx; pk.r[pk.n++] = r; off += r.n; }
            DpRegion d; memset(&d, 0, sizeof d); d.S = 1; d.src = e->dact[l]; d.dst = off; d.n = (unsigned long long)L.N * B; d.B = 1; d.ld = 1; pk.r[pk.n++] = d;
            dpdw.push_back({L, X, xo, off}); off += d.n;
        }
        // the complement of the flagged layers' (K+1) x N blocks inside the internal gradient vector
        std::vector<std::pair<unsigned long long, unsigned long long>> holes;
        for (int l = 0; l < e->nl; l++) if (dp_layer[l]) holes.push_back({e->L[l].w_off, e->L[l].w_off + (unsigned long long)(e->L[l].K + 1) * e->L[l].N});
        std::sort(holes.begin(), holes.end());
        unsigned long long cur = 0;
        // sub-ranges whose gradient still sits in split-K slabs (conv dW) are reduced by the pack kernel itself
        std::vector<RSeg> slabs(slab_segs); std::sort(slabs.begin(), slabs.end(), [](const RSeg& x, const RSeg& y) { return x.out < y.out; });
        bool pack_folds = !slabs.empty();
        for (auto& r : slabs) pack_folds = pack_folds && r.S2 == 0 && r.mode == 2;
        auto small = [&](unsigned long long a, unsigned long long b) {
            if (b <= a) return;
            DpRange q; q.src = off; q.dst = a; q.n = b - a; dsum.r[dsum.n++] = q;
            unsigned long long cur2 = a;
            auto plain = [&](unsigned long long x, unsigned long long y) { if (y <= x) return; DpRegion r; memset(&r, 0, sizeof r); r.src = e->grad + x; r.dst = off + (x - a); r.n = y - x; r.B = 1; r.ld = 1; r.S = 1; pk.r[pk.n++] = r; };
            if (pack_folds) for (auto& sg : slabs) {
                const unsigned long long sb = (unsigned long long)(sg.out - e->grad), se = sb + sg.elems;
                if (sb < a || se > b) continue;
                plain(cur2, sb);
                DpRegion r; memset(&r, 0, sizeof r); r.src = sg.part; r.dst = off + (sb - a); r.n = sg.elems; r.B = 1; r.ld = 1; r.S = sg.S; r.per_s = sg.elems; pk.r[pk.n++] = r;
                cur2 = se;
            }
            plain(cur2, b);
            off += b - a;
        };
        for (auto& h : holes) { small(cur, h.first); cur = h.second; }
        small(cur, e->Pint);
        off = (off + 3) / 4 * 4;
        if (e->dp_count != off) { hipFree(e->dp_send); hipFree(e->dp_recv); e->dp_send = e->dp_recv = nullptr; HIPCHK(hipMalloc((void**)&e->dp_send, off * 4)); HIPCHK(hipMalloc((void**)&e->dp_recv, off * 4 * W)); e->dp_count = off; }
        pk.send = e->dp_send; dsum.recv = e->dp_recv; dsum.stride = off; dsum.world = W; dsum.grad = e->grad;
        unsigned long long off_a = 0;      // end of the wide layers' operands inside the block
        for (auto& q : dpdw) off_a = std::max(off_a, q.d_off + (unsigned long long)q.L.N * B);
        if (overlap_cand && off_a % 4 == 0 && off_a < off) {
            e->dp_overlap = true; e->dp_count_a = off_a;
            hipFree(e->dp_recv_b); e->dp_recv_b = nullptr; HIPCHK(hipMalloc((void**)&e->dp_recv_b, (off - off_a) * 4 * W));
            DpPackArgs pa; memset(&pa, 0, sizeof pa); DpPackArgs pb; memset(&pb, 0, sizeof pb); pa.send = pb.send = e->dp_send;
            for (int i = 0; i < pk.n; i++) { if (pk.r[i].dst < off_a) pa.r[pa.n++] = pk.r[i]; else pb.r[pb.n++] = pk.r[i]; }
            e->dp_pk_a = pa; pk = pb;
            dsum.recv = e->dp_recv_b; dsum.stride = off - off_a; for (int i = 0; i < dsum.n; i++) dsum.r[i].src -= off_a;
        }
        e->prog.push_back({"dp_pack", [=](dqn_engine* en) { launch_dp_pack(en->stream, pk); }});
        e->dp_gather = true; e->dp_pack_folds = pack_folds;
        // the ascending sum over ranks of the small gradient ranges rides inside k_adam (its slab-reduce blocks) when the ranges allow it
        memset(&e->dp_adam_segs, 0, sizeof e->dp_adam_segs);
        bool af = dsum.n > 0 && dsum.n <= 8; unsigned long long tot = 0;
        for (int i = 0; i < dsum.n; i++) af = af && dsum.r[i].dst % 4 == 0 && dsum.r[i].n % 4 == 0;
        if (af) for (int i = 0; i < dsum.n; i++) { AdamSegs& A = e->dp_adam_segs; const int q = A.n++; A.beg[q] = dsum.r[i].dst; A.end[q] = dsum.r[i].dst + dsum.r[i].n; A.part[q] = dsum.recv + dsum.r[i].src; A.S[q] = W; A.stride[q] = dsum.stride; tot += dsum.r[i].n; }
        e->dp_adam_segs.blocks = (unsigned)((tot + 255) / 256);
        e->dp_adam_folds = af;
    }
    e->prog_post_begin = e->prog.size();
    if (e->dp_gather) {
        // big dense dW over the W*B gathered samples (rank-major = the concatenated batch); siblings with the same X and geometry share a launch
        const float* recv = e->dp_recv; const int cnt = (int)(e->dp_overlap ? e->dp_count_a : e->dp_count); float* grad = e->grad;      // rank stride of the buffer that holds X | dpre
        std::vector<bool> used(dpdw.size(), false);
        for (size_t i = 0; i < dpdw.size(); i++) {
            if (used[i]) continue; used[i] = true;
            size_t j = i + 1; for (; j < dpdw.size(); j++) if (!used[j] && dpdw[j].x_off == dpdw[i].x_off && same_geo(dpdw[i].L, dpdw[j].L)) break;
            const bool pair = j < dpdw.size(); if (pair) used[j] = true;
            const DpDw a = dpdw[i], b = pair ? dpdw[j] : dpdw[i]; const int np = pair ? 2 : 1;
            e->prog.push_back({pname(e, "dp_dw", a.L.kind, (int)i), [=](dqn_engine* en) {
                const float* X[2] = {recv + a.x_off, recv + b.x_off}; const float* d[2] = {recv + a.d_off, recv + b.d_off}; float* o[2] = {grad + a.L.w_off, grad + b.L.w_off};
                launch_gemm_dw(en->stream, a.L, np, X, B, d, W * B, o, /*ldd*/ B, /*tiles per rank*/ B / 32, /*rank stride*/ cnt); }});
        }
        if (!e->dp_adam_folds) e->prog.push_back({"dp_sum_ranks", [=](dqn_engine* en) { launch_dp_unpack_sum(en->stream, dsum); }});
    }
    {
        AdamJob J = base_job();
        J.nr = 1; J.beg[0] = 0; J.end[0] = e->Pint; J.sblocks = (unsigned)adam_blocks(e->Pint); J.tick = 1; J.slot0 = 0;
        if (e->hp.prioritized_replay && !rec && !e->prio_forked && !prio_placed) { J.prio = prio_args(); if (prio_draw_pending) { J.prio.phase = 2; prio_draw_pending = false; } }
        memset(&e->pg, 0, sizeof e->pg); e->pg_ok = pg_want && (prio_placed || e->prio_forked);
        if (e->pg_ok) {
            PreGather& G = e->pg; G.on = 1; G.s_rows = e->s_rows; G.sp_rows = e->sp_rows; G.E = e->E; G.B = B; G.idx_pre = e->idx_pre; G.x0 = e->x0; G.cap2 = e->cap2; G.tree = e->tree; G.seed = e->hp.seed;
            G.meta.distinct = e->hp.sample_distinct ? 1 : 0;
            G.meta.a = e->ra; G.meta.r = e->rr; G.meta.done = e->rdone; G.meta.beta = e->hp.prio_beta; G.meta.a_out = e->gb_a2; G.meta.r_out = e->gb_r2; G.meta.done_out = e->gb_done2; G.meta.w_out = e->gb_w2;
            G.u8b = e->arena_u8 ? 1 : 0;
            if (G.u8b) { G.gx = (e->E + 255) / 256; G.gy = (2 * B + 127) / 128; } else { G.gx = (e->E + 63) / 64; G.gy = (2 * B + 63) / 64; }
        }
        e->adam_step = (long)e->prog.size();
        J.gscale = e->world > 1 ? 1.0f / (float)e->world : 1.0f;
        const bool fold = e->adam_segs.n > 0 && !e->comm && !e->sim_world;     // with a communicator the gradient must be materialised before the all-reduce
        if (e->dp_gather && e->dp_adam_folds) J.segs = e->dp_adam_segs; else if (fold) J.segs = e->adam_segs;
        e->prog.push_back({"adam", [=](dqn_engine* en) { launch_adam(en->stream, J, en->step_pregather ? &en->pg : nullptr); }});
        e->gmax_used = (int)(J.segs.blocks + J.sblocks);
    }
    e->prog_built = true;
    return 0;
}
