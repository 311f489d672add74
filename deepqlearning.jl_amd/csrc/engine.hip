// engine.hip -- host side of libdqn_mi355x.so: one engine per GPU owns replay storage, sum-tree, online/target
// parameters, gradients, Adam state, the batch arena and ONE HIP stream; the train step
// (batch_train!, src/solver.jl:191-236) is enqueued as a fixed kernel sequence and replayed from a hipGraph.
// C ABI: include/dqn_mi355x.h.  No CPU fallback exists: every entry point that computes needs the HIP device.



#include "engine.h"

static thread_local char g_err[1024] = "";
int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return -1;
}

// ---------------------------------------------------------------- RCCL (dlopen'ed; only for data-parallel replicas)
struct Id128 { char b[128]; };   // ncclUniqueId (passed by value)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr; int (*CommUserRank)(void*, int*) = nullptr; int (*CommCuDevice)(void*, int*) = nullptr;
};
static Rccl g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) return fail("cannot dlopen librccl: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, Id128, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommCount"); g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommUserRank");
    g_rccl.CommCuDevice = (int (*)(void*, int*))dlsym(g_rccl.lib, "ncclCommCuDevice");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.AllGather) return fail("librccl is missing nccl symbols");
    return 0;
}

void prof_begin(dqn_engine* e, const char* name) {
    if (!e->profiling) return;
    ProfEntry pe; pe.name = name; hipEventCreate(&pe.a); hipEventCreate(&pe.b); hipEventRecord(pe.a, e->stream); e->prof.push_back(pe);
}
void prof_end(dqn_engine* e) { if (e->profiling) hipEventRecord(e->prof.back().b, e->stream); }
#define RUN(e, name, call) do { prof_begin(e, name); call; prof_end(e); } while (0)

extern "C" const char* dqn_last_error(void) { return g_err; }
extern "C" int dqn_version(void) { return 1; }

extern "C" int dqn_hparams_default(dqn_hparams* hp) {
    memset(hp, 0, sizeof *hp);
    hp->batch_size = 32; hp->obs_h = 1; hp->obs_w = 1; hp->obs_dtype = DQN_OBS_F32;
    hp->learning_rate = 1e-4f; hp->adam_beta1 = 0.9; hp->adam_beta2 = 0.999; hp->adam_eps = 1e-8; hp->adam_f64_scalars = 1;
    hp->gamma = 1.0f; hp->double_q = 1; hp->dueling = 1; hp->prioritized_replay = 1; hp->buffer_size = 1000;
    hp->prio_alpha = 0.6f; hp->prio_beta = 0.4f; hp->prio_eps = 1e-3f; hp->seed = 0; hp->use_graph = 1; hp->use_mfma = 1;
    return 0;
}

// geometry of every layer; shared by dqn_plan_default (host only) and dqn_engine_create
static int build_layers(const dqn_layer_desc* d, int n, const dqn_hparams* hp, LayerDev* L, int* lb, int* lv, int* la, size_t* P, size_t* Pint) {
    if (n <= 0 || n > DQN_MAX_LAYERS) return fail("bad layer count %d", n);
    *lb = *lv = *la = -1; size_t off = 0, eoff = 0;
    for (int i = 0; i < n; i++) {
        LayerDev& l = L[i]; memset(&l, 0, sizeof l);
        l.kind = d[i].kind; l.act = d[i].act; l.stream = d[i].stream;
        int prev;
        if (l.stream == DQN_STREAM_BASE) prev = *lb; else if (l.stream == DQN_STREAM_VAL) prev = *lv >= 0 ? *lv : *lb; else prev = *la >= 0 ? *la : *lb;
        l.src = prev;
        int c, h, w;
        if (prev < 0) { c = hp->obs_c; h = hp->obs_h; w = hp->obs_w; }
        else if (L[prev].kind == DQN_LAYER_CONV) { c = L[prev].cout; h = L[prev].oh; w = L[prev].ow; }
        else { c = L[prev].out_feat; h = 1; w = 1; }
        l.in_feat = c * h * w;
        if (l.kind == DQN_LAYER_CONV) {
            l.cin = d[i].cin; l.cout = d[i].cout; l.kh = d[i].kh; l.kw = d[i].kw; l.sh = d[i].sh; l.sw = d[i].sw;
            if (l.cin != c) return fail("layer %d: conv cin %d != incoming channels %d", i, l.cin, c);
            if (l.kh > h || l.kw > w || l.sh < 1 || l.sw < 1) return fail("layer %d: conv kernel/stride does not fit the %dx%d input", i, h, w);
            l.ih = h; l.iw = w; l.oh = (h - l.kh) / l.sh + 1; l.ow = (w - l.kw) / l.sw + 1;
            l.K = l.cin * l.kh * l.kw; l.N = l.cout; l.npos = l.oh * l.ow; l.out_feat = l.cout * l.npos;
        } else if (l.kind == DQN_LAYER_DENSE) {
            if (d[i].n_in != l.in_feat) return fail("layer %d: dense n_in %d != incoming features %d", i, d[i].n_in, l.in_feat);
            l.K = d[i].n_in; l.N = d[i].n_out; l.npos = 1; l.out_feat = l.N; l.ih = l.iw = l.oh = l.ow = 1;
        } else if (l.kind == DQN_LAYER_LSTM) {
            if (!hp->recurrence) return fail("DeepQLearningError: you passed in a recurrent model but recurrence is set to false");   // src/solver.jl:45-47
            if (l.stream != DQN_STREAM_BASE) return fail("LSTM layers are supported in the base chain only");
            if (d[i].n_in != l.in_feat) return fail("layer %d: LSTM n_in %d != incoming features %d", i, d[i].n_in, l.in_feat);
            l.H = d[i].n_out; l.K = d[i].n_in; l.N = 4 * l.H; l.npos = 1; l.out_feat = l.H; l.act = DQN_ACT_IDENTITY; l.ih = l.iw = l.oh = l.ow = 1;
        } else return fail("layer %d: unknown kind %d", i, l.kind);
        if (l.kind == DQN_LAYER_LSTM) {
            const size_t kn = (size_t)l.K * l.N, hn = (size_t)l.H * l.N;
            l.ew_off = eoff; eoff += kn; l.ewh_off = eoff; eoff += hn; l.eb_off = eoff; eoff += l.N; l.eh0_off = eoff; eoff += l.H; l.ec0_off = eoff; eoff += l.H;
            off = (off + 3) / 4 * 4; l.w_off = off; off += kn; l.b_off = off; off += l.N; l.wh_off = off; off += hn; off += l.N /* junk bias row of the Wh dW pass */;
            l.h0_off = off; off += l.H; l.c0_off = off; off += l.H; off = (off + 3) / 4 * 4; l.z_off = off; off += l.N;
        } else {
            l.ew_off = eoff; eoff += (size_t)l.K * l.N; l.eb_off = eoff; eoff += l.N;
            off = (off + 3) / 4 * 4; l.w_off = off; off += (size_t)l.K * l.N; l.b_off = off; off += l.N;
        }
        if (l.stream == DQN_STREAM_BASE) *lb = i; else if (l.stream == DQN_STREAM_VAL) *lv = i; else *la = i;
    }
    *P = eoff; *Pint = (off + 3) / 4 * 4;
    if (hp->dueling) {
        if (*lv < 0 || *la < 0 || L[*lv].out_feat != 1 || L[*la].out_feat != hp->n_actions)
            return fail("DeepQLearningError: the qnetwork provided is incompatible with dueling");   // src/dueling.jl:47
    } else if (*lb < 0 || L[*lb].out_feat != hp->n_actions) return fail("network output size != n_actions");
    if (hp->n_actions > DQN_MAX_ACTIONS) return fail("n_actions > %d unsupported", DQN_MAX_ACTIONS);
    return 0;
}
// The default summation-order plan.  Chosen for gfx950 occupancy: long forward contractions
// are cut into ~512-element chunks, dense dX into 256-element chunks, conv dW into ~256-sample chunks.
// the ONE place the library reads its environment (besides trace builds): at dqn_engine_create, and -- the communicator switches only -- at dqn_comm_init
void read_opts(EngineOpts& o, bool comm_only) {
    auto I = [](const char* k, int dflt) { const char* v = getenv(k); return v ? atoi(v) : dflt; };
    auto F = [](const char* k) { return getenv(k) != nullptr ? 1 : 0; };
    o.force_allreduce = F("DQN_FORCE_ALLREDUCE"); o.dp_allreduce = F("DQN_DP_ALLREDUCE"); o.dp_overlap = I("DQN_DP_OVERLAP", -1); o.dp_no_one_graph = F("DQN_DP_NO_ONE_GRAPH");
    if (comm_only) return;
    // switches that keep a SECOND SCHEDULE of the same arithmetic under parity tests (tests/ name each of them) or that a tool under tools/ drives
    o.no_tiny = F("DQN_NO_TINY"); o.fwd_m32 = I("DQN_FWD_M32", -1); o.no_fwd_wres = F("DQN_NO_FWD_WRES"); o.sim_world = I("DQN_SIM_WORLD", 0); o.no_pregather = F("DQN_NO_PREGATHER");
    o.no_red_head = F("DQN_NO_RED_HEAD"); o.no_head_cols4 = I("DQN_NO_HEAD_COLS4", 0); o.dw_split = I("DQN_DW_SPLIT", 128); o.no_rh_pm = F("DQN_NO_RH_PM"); o.no_act_head = F("DQN_NO_ACT_HEAD"); o.drqn_stamps = F("DQN_DRQN_STAMPS");
    // r06 removed (the constants they set are now the only behaviour): DQN_ADAM_MODE, DQN_PRIO_LEVEL, DQN_PRIO_NOSPLIT, DQN_PRIO_FORK, DQN_HEAD_FUSE_MAXB,
    // DQN_NO_HEAD_FUSE, DQN_NO_U8_ARENA, DQN_LSTM_DW_MFMA, DQN_NO_GRAPH_UPLOAD, DQN_NO_ST_WT, DQN_NO_DX_WIDE, DQN_NO_ROLLOUT_CYCLE, DQN_MID_GROUP, DQN_MID_BIG
    // (docs/history/r06.md lists each with the number that decided it)
#ifdef DQN_KTRACE
    // TIMING PROBES (wrong numbers, right schedule): trace builds only -- the product library does not read them
    o.head_dbg = I("DQN_HEAD_DBG", 0); o.probe_no_tg = F("DQN_PROBE_NO_TG"); o.drqn_probe = I("DQN_DRQN_PROBE", 0); o.tiny_stop = I("DQN_TINY_STOP", 0);
#endif
}
static void default_plan(const LayerDev* L, int n, int B, dqn_layer_plan* out, const dqn_hparams* hp, bool allow_cg = true) {
    bool rec = false; for (int i = 0; i < n; i++) rec = rec || L[i].kind == DQN_LAYER_LSTM;
    for (int i = 0; i < n; i++) {
        out[i].fwd_kc = 0;
        // (small batches only: at B >= 128 the 3 B columns of a step already fill the chip -- 512 workgroups for the 3136 -> 512 layers of config 5 --
        // and the split only bought slab traffic plus a reduce launch: 16 us of the 807 us step, r03_g)
        if (L[i].K > 1024 && B < 128) { const int s = (L[i].K + 511) / 512; int kc = (L[i].K + s - 1) / s; kc = (kc + 3) / 4 * 4; out[i].fwd_kc = kc; }
        // (r06: chunks of 544 -- six slabs instead of seven for the 3136-wide layers -- looked 1 % faster until the probe showed why: fewer slabs dropped the launch below the tile
        //  picker's workgroup threshold, which then chose 16-channel tiles; with those tiles chosen for dense layers outright (fwd_pick_nt, nn_gemm.hip) seven slabs of 448 win:
        //  forward launch 14.9 (448, 32-channel tiles) -> 14.0 (544, 16) -> 13.2 us (448, 16); profiles/r06_zl_plan_probe_fwd.txt, r06_zo_fwd_nt_probe.txt)
        else if (L[i].kind == DQN_LAYER_DENSE && L[i].N < 16 && L[i].K >= 128) out[i].fwd_kc = 32;   // heads: 16+ short chains instead of one long one (at any batch: unsplit, the B = 512 head launch took 25 us instead of 7 + 6)
        out[i].dx_kc = (L[i].kind != DQN_LAYER_CONV && L[i].N > 512) ? 256 : 0;
        // small batches (the 32 x 32 output tiles of k_dx_units, nn_gemm.hip): cut the dX contraction so that an output tile has up to four
        // independent chains, one per wave -- dense: N/4 (N/2 per stream at the dueling join, where two sources meet); conv: RAW taps per chunk
        // such that an interior input position has <= 4 chunks with a valid tap.  Only the rounding order changes.
        const bool join = L[i].stream != DQN_STREAM_BASE && L[i].src >= 0 && L[L[i].src].stream == DQN_STREAM_BASE;
        if (L[i].kind == DQN_LAYER_DENSE && L[i].N <= 512 && B <= 64 && L[i].N >= 128) {
            const int S = join ? 2 : 4; const int kc = ((L[i].N + S - 1) / S + 31) / 32 * 32; if (kc < L[i].N) out[i].dx_kc = kc;
        } else if (L[i].kind == DQN_LAYER_CONV && B <= 64) {
            const int valid = ((L[i].kh + L[i].sh - 1) / L[i].sh) * ((L[i].kw + L[i].sw - 1) / L[i].sw), raw = ((valid + 3) / 4) * L[i].sw;
            // r06: ... but only where the chain of an output element is longer than eight K tiles (valid taps x channels > 256): the 4x4 / stride-2 layer of the Nature net has four
            // valid taps x 64 channels and runs faster as ONE chain per element than as four (its launch 12.6 -> 11.8 us with the dW workgroups of the rule below beside it;
            // 4 / 8 raw taps per chunk: 12.1; profiles/r06_zj_plan_probe_dx.txt); the 3x3 layer (nine taps x 64) keeps its three chunks (unsplit: 15.4 vs 11.7 us)
            if (raw < L[i].kh * L[i].kw && valid * L[i].N > 256) out[i].dx_kc = raw;
        }
        out[i].dw_kc = 0;
        // head layers at large batches, recurrent networks only (their dW is a launch of its own): 64-sample chains on 8x more threads instead of one
        // B-long chain per output.  Feed-forward networks compute the heads' dW as tail tasks of the widest backward launch, where a 512-deep chain
        // hides, and unsplit gradients keep the slab sums foldable into the Adam launch (r03: the separate reduce launch cost 12-19 us at config 5)
        if (L[i].kind == DQN_LAYER_DENSE && L[i].N < 16 && B >= 128 && rec) out[i].dw_kc = 64;
        if (L[i].kind == DQN_LAYER_CONV) {   // positions per chunk so that (K/64 row tiles) x chunks >= ~512 workgroups
            const int tgt = 512;      // (r03: 256 / 128 workgroups measured no faster)
            const int mrows_ = (L[i].K + 63) / 64, st = (tgt + mrows_ - 1) / mrows_; int ppc = L[i].npos / st; if (ppc < 1) ppc = 1;
            // r06: an EVEN number of 32-sample K tiles per chunk where a chunk is 3+ of them -- the dW loop is software-pipelined two tiles per round and an odd count pays a
            // clamped half round -- as long as >= 3/4 of the target workgroups remain (config 2's 8x8 layer: 3 -> 4 positions, 536 -> 400 workgroups: 10.1 -> 9.45 us,
            // profiles/r06_conv_dw_chunk_probe.txt; 5 / 6 / 8 positions: 10.4 / 10.8 / 11.3)
            if (B % 32 == 0 && B < 128) { const int kt = ppc * (B / 32); if (kt >= 3 && (kt & 1) && (B / 32) % 2 == 1 && ((L[i].npos + ppc) / (ppc + 1)) * mrows_ * 4 >= 3 * tgt) ppc += 1; }
            // r06: a conv layer that HAS a dX (every one but the first) computes its dW in the launch that also carries that dX -- chains of lone waves on most of the chip's
            // slots -- and there the target of ~512 dW workgroups is wrong: THREE K tiles per chunk (config 2: 81 -> 27 chunks = 216 workgroups beside the second convolution's 800
            // dX workgroups, 49 -> 17 for the third) shorten both that launch and, with a third of the slabs, the Adam launch: 8374 -> 8555 steps/s
            // (profiles/r06_zh_conv_dw_chunk_probe.txt: 2 / 3 / 4 / 5 / 6 / 9 positions: launch 15.1 / 12.7 / 14.4 / 14.4 / 15.0 / 17.9 us)
            if (B % 32 == 0 && B < 128 && L[i].src >= 0) { const int tpp = B / 32; ppc = (3 + tpp - 1) / tpp; if (ppc > L[i].npos) ppc = L[i].npos; }
            out[i].dw_kc = ppc * B;
            // large batches: a position is 4+ K tiles deep, so chunks are cut in SAMPLES (32-aligned, they may start inside a position) to land
            // on <= 1024 workgroups -- whole positions gave 536 workgroups for the 8x8 conv layer at B = 512, i.e. three on some CUs and two on the rest
            if (B >= 128 && B % 32 == 0) {
                const int tgt2 = 1024;      // measured at config 5: 512 / 768 / 1024 -> conv dW 57 / 52 / 49 us
                const int mrows = (L[i].K + 63) / 64, KK = L[i].npos * B; int ch = tgt2 / mrows; if (ch < 1) ch = 1;
                int kc = ((KK + ch - 1) / ch + 31) / 32 * 32; if (kc < KK) out[i].dw_kc = kc; else out[i].dw_kc = 0;
            }
        }
    }
    // recurrent networks the fused column-parallel step covers (drqn_cols.hip): dW / db chunks = groups of cg batch columns, one workgroup each
    // (allow_cg = false: replicas -- the fused step has no exchange point between its gradient chunks and Adam, so an engine that is given a communicator takes the
    // multi-launch recurrent program, whose gradient is materialised before the all-reduce: dqn_comm_init)
    const int cg = allow_cg ? drqn_fused_cg(L, n, hp->obs_c * hp->obs_h * hp->obs_w, B, hp->trace_length, hp->n_actions, hp->dueling, hp->double_q, hp->recurrence) : 0;
    if (cg) for (int i = 0; i < n; i++) out[i].dw_kc = -cg;
}
extern "C" int dqn_plan_version(void) { return DQN_PLAN_VERSION; }
extern "C" int dqn_plan_default(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, dqn_layer_plan* plan_out) {
    LayerDev L[DQN_MAX_LAYERS]; int lb, lv, la; size_t P, Pi;
    if (build_layers(layers, n_layers, hp, L, &lb, &lv, &la, &P, &Pi)) return -1;
    default_plan(L, n_layers, hp->batch_size, plan_out, hp); return 0;
}


extern "C" int dqn_engine_destroy(dqn_engine_t* e);

static int engine_init(dqn_engine* e, const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, const dqn_layer_plan* plan, int device);
extern "C" int dqn_engine_create(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, const dqn_layer_plan* plan, int device,
                                 dqn_engine_t** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no HIP device: libdqn_mi355x has no CPU fallback (hipGetDeviceCount found %d devices)", ndev);
    if (device < 0 || device >= ndev) return fail("device %d out of range (0..%d)", device, ndev - 1);
    if (hp->batch_size < 1 || hp->batch_size > 1024) return fail("batch_size %d unsupported (1..1024)", hp->batch_size);
    if (hp->buffer_size < hp->batch_size) return fail("AssertionError: r.max_size >= r.batch_size");   // ...replay.jl:84
    dqn_engine* e = new dqn_engine();
    if (engine_init(e, layers, n_layers, hp, plan, device)) {   // every failure path releases what was allocated so far (the message survives: destroy never calls fail)
        dqn_engine_destroy(e); return -1;
    }
    *out = e; return 0;
}
// dynamic LDS of the single-workgroup TD kernel (nn_valu.hip, launch_td) for this shape
static size_t td_lds_bytes(int B, int nA, int ncon) { return (size_t)B * sizeof(float) + (size_t)(1 + nA) * (ncon + B) * sizeof(float); }
static int engine_init(dqn_engine* e, const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, const dqn_layer_plan* plan, int device) {
    e->device = device; e->hp = *hp; e->B = hp->batch_size; e->nA = hp->n_actions; e->E = hp->obs_c * hp->obs_h * hp->obs_w;
    if (build_layers(layers, n_layers, hp, e->L, &e->last_base, &e->last_val, &e->last_adv, &e->P, &e->Pint)) return -1;
    e->nl = n_layers;
    read_opts(e->opt);      // every experiment / test switch, once; nothing below (or later) looks at the environment
    e->no_tiny = e->opt.no_tiny != 0;
    e->mid_group = e->opt.mid_group;      // middle steps of dqn_train_steps per graph launch (1 = one step per graph: no grouped graphs at all)
    e->mid_big = e->opt.mid_group > 1 ? e->opt.mid_big : 0;
    { const int lopt = (e->opt.fwd_m32 > 0 ? DQN_LOPT_FWD_M32 : 0) | (e->opt.fwd_m32 == 0 ? DQN_LOPT_NO_FWD_M32 : 0) | (e->opt.no_dx_wide ? DQN_LOPT_NO_DX_WIDE : 0) | (e->opt.no_fwd_wres ? DQN_LOPT_NO_FWD_WRES : 0) |
                       ((std::min(255, std::max(0, e->opt.dw_split / 16)) & 0xff) << 8) | 
                       ((!e->opt.no_st_wt && (long long)hp->batch_size * (hp->recurrence ? hp->trace_length : 1) <= 64) ? DQN_LOPT_ST_WT : 0); for (int i = 0; i < e->nl; i++) e->L[i].opt = lopt; }
    if (e->opt.sim_world >= 1 && !hp->recurrence) { e->sim_world = e->opt.sim_world; e->world = e->opt.sim_world; }   // tests: one process plays k identical ranks
    dqn_layer_plan defp[DQN_MAX_LAYERS];
    e->plan_defaulted = plan == nullptr;
    if (!plan) { default_plan(e->L, e->nl, e->B, defp, hp); plan = defp; }
    for (int i = 0; i < e->nl; i++) {
        e->L[i].fwd_kc = plan[i].fwd_kc; e->L[i].dx_kc = plan[i].dx_kc; e->L[i].dw_kc = plan[i].dw_kc;
        if (e->L[i].kind == DQN_LAYER_CONV && e->L[i].dw_kc > 0 && e->L[i].dw_kc % e->B && (e->B % 32 || e->L[i].dw_kc % 32)) return fail("plan: conv dw_kc must be a multiple of batch_size (or, for batch sizes divisible by 32, of 32)");
        if (e->L[i].dw_kc < 0 && (!hp->recurrence || e->B % (-e->L[i].dw_kc))) return fail("plan: dw_kc < 0 (column-group chunks of %d batch columns) needs recurrence = true and a group size that divides batch_size", -e->L[i].dw_kc);
    }
    HIPCHK(hipSetDevice(device));
    gemm_set_ktrace(nullptr);      // trace builds: (re)reads DQN_PROBE; a no-op otherwise
    if (!hp->recurrence) {
        // the TD kernel keeps every head output of the step in LDS: (B + (1 + nA) * (ncon + B)) floats.  Shapes that do not fit the
        // workgroup limit of this device are refused HERE with a message instead of failing at launch time (a failed launch would
        // leave the step running backward and Adam on stale gradients).
        hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, device));
        const size_t need = td_lds_bytes(e->B, e->nA, hp->double_q ? 2 * e->B : e->B), have = prop.sharedMemPerBlock;
        if (need > have) return fail("batch_size %d x n_actions %d needs %zu bytes of LDS in the TD kernel, the device offers %zu per workgroup: lower batch_size or n_actions", e->B, e->nA, need, have);
    }
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&e->stream3, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_xa, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_xb, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_xc, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));

    const int B = e->B;
    e->T = hp->recurrence ? hp->trace_length : 1;
    if (hp->recurrence && (e->T < 1 || (long long)e->T * B > 65536)) return fail("trace_length %d unsupported", e->T);
    if (hp->recurrence && hp->obs_dtype == DQN_OBS_U8) return fail("DeepQLearningError: obs_dtype = u8 is not supported with recurrence = true (the episode replay stores Float32 rows, src/episode_replay.jl:3-20)");
    const int Bc = e->Bc = e->T * B;            // columns per sequence set: B, or T*B time-major columns for DRQN
    e->ncon = hp->double_q ? 2 * Bc : Bc;
    DM(e->L_dev, e->nl); HIPCHK(hipMemcpy(e->L_dev, e->L, sizeof(LayerDev) * e->nl, hipMemcpyHostToDevice));
    DM(e->p_on, e->Pint); DM(e->p_tg, e->Pint); DM(e->grad, e->Pint); DM(e->m, e->Pint); DM(e->v, e->Pint); DM(e->io_tmp, e->P);
    HIPCHK(hipMemset(e->p_on, 0, e->Pint * 4)); HIPCHK(hipMemset(e->p_tg, 0, e->Pint * 4)); HIPCHK(hipMemset(e->grad, 0, e->Pint * 4));
    HIPCHK(hipMemset(e->m, 0, e->Pint * 4)); HIPCHK(hipMemset(e->v, 0, e->Pint * 4));
    DM(e->state, 1);
    // scalar mailbox: a coherent, mapped pinned ring the publish launch writes and the host polls (a failure here only disables the fast path)
    DM(e->pub_ctr, 1); HIPCHK(hipMemset(e->pub_ctr, 0, sizeof(unsigned long long)));
    if (hipHostMalloc((void**)&e->mail_host, sizeof(StepMail) * DQN_MAIL_SLOTS, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
        memset(e->mail_host, 0, sizeof(StepMail) * DQN_MAIL_SLOTS);
        if (hipHostGetDevicePointer((void**)&e->mail_dev, e->mail_host, 0) != hipSuccess) { hipHostFree(e->mail_host); e->mail_host = nullptr; e->mail_dev = nullptr; }
    } else { e->mail_host = nullptr; }
    (void)hipGetLastError();
    if (!e->mail_host) { e->mail_host = (StepMail*)calloc(DQN_MAIL_SLOTS, sizeof(StepMail)); e->mail_dev = nullptr; e->mail_plain = true; }      // no device-side publish: dqn_train_step_async degrades to synchronous steps recorded by the host
    StepState s0; memset(&s0, 0, sizeof s0); s0.bp[0][0] = s0.bp[1][0] = hp->adam_beta1; s0.bp[0][1] = s0.bp[1][1] = hp->adam_beta2;
    HIPCHK(hipMemcpy(e->state, &s0, sizeof s0, hipMemcpyHostToDevice));
    e->cap = hp->recurrence ? B : hp->buffer_size; while (e->cap2 < e->cap) e->cap2 <<= 1;   // DRQN keeps episodes instead (below)
    const size_t osz = hp->obs_dtype == DQN_OBS_U8 ? 1 : 4;
    { unsigned char *a = nullptr, *b = nullptr; DM(a, (size_t)e->cap * e->E * osz); DM(b, (size_t)e->cap * e->E * osz); e->s_rows = a; e->sp_rows = b; }
    DM(e->ra, e->cap); DM(e->rr, e->cap); DM(e->rdone, e->cap); DM(e->tree, 2 * (size_t)e->cap2);
    HIPCHK(hipMemset(e->tree, 0, 2 * (size_t)e->cap2 * 4));
    DM(e->st_a, dqn_engine::ADD_CHUNK); DM(e->st_r, dqn_engine::ADD_CHUNK); DM(e->st_done, dqn_engine::ADD_CHUNK); DM(e->st_td, dqn_engine::ADD_CHUNK);
    DM(e->idx, B); HIPCHK(hipMemset(e->idx, 0, B * 8)); DM(e->idx_pre, B); HIPCHK(hipMemset(e->idx_pre, 0, B * 8)); DM(e->x0, (size_t)e->E * 2 * Bc);
    size_t pmax = 1, jmax = 1;
    for (int i = 0; i < e->nl; i++) {
        const LayerDev& l = e->L[i];
        DM(e->act_on[i], (size_t)l.out_feat * e->ncon); DM(e->act_tg[i], (size_t)l.out_feat * Bc); DM(e->dact[i], (size_t)l.out_feat * Bc);
        const size_t sf = dqn_nchunks(l.K, l.fwd_kc); if (sf > 1) pmax = std::max(pmax, sf * (size_t)l.out_feat * e->ncon);
        const size_t sw = dqn_nchunks(l.npos * Bc, l.dw_kc); if (sw > 1) pmax = std::max(pmax, sw * (size_t)(l.K + 1) * l.N);
        const size_t sx = l.kind != DQN_LAYER_CONV ? dqn_nchunks(l.N, l.dx_kc) : 1; if (sx > 1) pmax = std::max(pmax, sx * (size_t)l.in_feat * Bc);
        jmax = std::max(jmax, (size_t)l.in_feat * Bc);
        if (l.kind == DQN_LAYER_LSTM) {
            DM(e->gx_on[i], (size_t)l.N * e->ncon); DM(e->gx_tg[i], (size_t)l.N * Bc); DM(e->cst_on[i], (size_t)l.H * e->ncon); DM(e->cst_tg[i], (size_t)l.H * Bc);
            DM(e->gates[i], (size_t)l.N * Bc); DM(e->tcb[i], (size_t)l.H * Bc); DM(e->hprev_buf[i], (size_t)l.H * Bc); DM(e->cprev_buf[i], (size_t)l.H * Bc);
            DM(e->dG[i], (size_t)l.N * Bc); DM(e->dhn[i], (size_t)l.H * B); DM(e->dcn[i], (size_t)l.H * B);
        }
    }
    if (hp->recurrence) {   // EpisodeReplayBuffer (src/episode_replay.jl:3-40): buffer_size EPISODES, first trace_length transitions of each
        e->ep_cap = hp->buffer_size; e->ep_len_host.assign((size_t)e->ep_cap, 0);
        if (hp->obs_dtype != DQN_OBS_F32) return fail("DRQN episode storage is float32 only");
        DM(e->ep_s, (size_t)e->ep_cap * e->T * e->E); DM(e->ep_sp, (size_t)e->ep_cap * e->T * e->E); DM(e->ep_a, (size_t)e->ep_cap * e->T); DM(e->ep_r, (size_t)e->ep_cap * e->T);
        DM(e->ep_done, (size_t)e->ep_cap * e->T); DM(e->ep_len, e->ep_cap); HIPCHK(hipMemset(e->ep_len, 0, (size_t)e->ep_cap * 4));
        DM(e->ep_idx, B); DM(e->ep_start, B); DM(e->r_a, Bc); DM(e->r_r, Bc); DM(e->r_done, Bc); DM(e->r_mask, Bc);
    }
    e->partials_elems = pmax; DM(e->partials, 2 * pmax);   // second half: the target net's split-K partials (fused on+tg launches)
    DM(e->join_tmp, jmax); DM(e->gmax_part, gmax_slots(e->Pint)); HIPCHK(hipMemset(e->gmax_part, 0, (size_t)gmax_slots(e->Pint) * 4));
    DM(e->w_is, B); DM(e->td, Bc); DM(e->q_on_s, (size_t)B * e->nA); DM(e->q_on_sp, (size_t)B * e->nA); DM(e->q_tg_sp, (size_t)B * e->nA);
    DM(e->ytarget, B); DM(e->best, B);
    DM(e->gb_rows, (size_t)B * e->E); DM(e->gb_r, B); DM(e->gb_done, B); DM(e->gb_w, B); DM(e->gb_a, B); DM(e->gb_idx, B);
    DM(e->gb_r2, B); DM(e->gb_done2, B); DM(e->gb_w2, B); DM(e->gb_a2, B);
    HIPCHK(hipStreamSynchronize(e->stream));
    return 0;
}

void drop_graphs(dqn_engine* e) {
    for (int i = 0; i < 2; i++) {
        if (e->g_full[i]) { hipGraphExecDestroy(e->g_full[i]); e->g_full[i] = nullptr; }
        if (e->g_full_pub[i]) { hipGraphExecDestroy(e->g_full_pub[i]); e->g_full_pub[i] = nullptr; }
        if (e->g_pre[i]) { hipGraphExecDestroy(e->g_pre[i]); e->g_pre[i] = nullptr; }
        for (int j = 0; j < 2; j++) if (e->g_pgv[i][j]) { hipGraphExecDestroy(e->g_pgv[i][j]); e->g_pgv[i][j] = nullptr; }
    }
    if (e->g_pgv_pub) { hipGraphExecDestroy(e->g_pgv_pub); e->g_pgv_pub = nullptr; }
    if (e->g_post) { hipGraphExecDestroy(e->g_post); e->g_post = nullptr; }
    if (e->g_post_pg) { hipGraphExecDestroy(e->g_post_pg); e->g_post_pg = nullptr; }
    if (e->g_mid) { hipGraphExecDestroy(e->g_mid); e->g_mid = nullptr; }
    for (int k = 0; k < 2; k++) { if (e->g_drqn_k[k]) { hipGraphExecDestroy(e->g_drqn_k[k]); e->g_drqn_k[k] = nullptr; } if (e->g_drqn[k]) { hipGraphExecDestroy(e->g_drqn[k]); e->g_drqn[k] = nullptr; } }
    if (e->g_mid_big) { hipGraphExecDestroy(e->g_mid_big); e->g_mid_big = nullptr; } e->mid_big_warm = false;
    if (e->g_pre_tp) { hipGraphExecDestroy(e->g_pre_tp); e->g_pre_tp = nullptr; }
    for (int i = 0; i < 3; i++) if (e->g_pre1[i]) { hipGraphExecDestroy(e->g_pre1[i]); e->g_pre1[i] = nullptr; }
    for (int i = 0; i < 4; i++) if (e->g_dp_one[i]) { hipGraphExecDestroy(e->g_dp_one[i]); e->g_dp_one[i] = nullptr; }
    if (e->g_pre2) { hipGraphExecDestroy(e->g_pre2); e->g_pre2 = nullptr; }
    for (dqn_engine::ActProg* a : {&e->act, &e->evalp}) { if (a->graph) { hipGraphExecDestroy(a->graph); a->graph = nullptr; } if (a->cycle) { hipGraphExecDestroy(a->cycle); a->cycle = nullptr; } if (a->envc) { hipGraphExecDestroy(a->envc); a->envc = nullptr; } }
}
void drop_act(dqn_engine* e, dqn_engine::ActProg& a) {
    if (a.graph) { hipGraphExecDestroy(a.graph); a.graph = nullptr; }
    if (a.cycle) { hipGraphExecDestroy(a.cycle); a.cycle = nullptr; }
    if (a.envc) { hipGraphExecDestroy(a.envc); a.envc = nullptr; }
    if (!a.allocs.empty()) hipStreamSynchronize(e->stream);
    for (void* p : a.allocs) hipFree(p);
    a.allocs.clear(); a.steps.clear(); a.n = 0;
}
static void free_policy_ws(dqn_engine* e) {
    hipFree(e->pol_obs); hipFree(e->pol_x); hipFree(e->pol_q); hipFree(e->pol_a);
    for (int i = 0; i < e->nl; i++) { hipFree(e->pol_act[i]); e->pol_act[i] = nullptr; }
    e->pol_obs = e->pol_x = e->pol_q = nullptr; e->pol_a = nullptr; e->pol_n = 0;
}
extern "C" int dqn_engine_destroy(dqn_engine_t* e) {
    if (!e) return 0;
    hipSetDevice(e->device);
    if (e->stream) hipStreamSynchronize(e->stream);
    drop_graphs(e);
    if (e->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(e->comm);
    if (e->ktrace_buf) { gemm_set_ktrace(nullptr); hipFree(e->ktrace_buf); }
    hipFree(e->L_dev); hipFree(e->p_on); hipFree(e->p_tg); hipFree(e->grad); hipFree(e->m); hipFree(e->v); hipFree(e->io_tmp); hipFree(e->state);
    hipFree(e->s_rows); hipFree(e->sp_rows); hipFree(e->ra); hipFree(e->rr); hipFree(e->rdone); hipFree(e->tree);
    if (e->state_host) hipHostFree(e->state_host);
    if (e->mail_host) { if (e->mail_plain) free(e->mail_host); else hipHostFree(e->mail_host); }
    if (e->draw_idx_h) hipHostFree(e->draw_idx_h);
    if (e->draw_start_h) hipHostFree(e->draw_start_h);
    for (int k = 0; k < 4; k++) if (e->draw_ev[k]) hipEventDestroy(e->draw_ev[k]);
    hipFree(e->pub_ctr);
    hipFree(e->st_a); hipFree(e->st_r); hipFree(e->st_done); hipFree(e->st_td); hipFree(e->idx); hipFree(e->idx_pre); hipFree(e->x0);
    for (int i = 0; i < e->nl; i++) { hipFree(e->act_on[i]); hipFree(e->act_tg[i]); hipFree(e->dact[i]); }
    hipFree(e->join_tmp); hipFree(e->partials); hipFree(e->gmax_part); hipFree(e->w_is); hipFree(e->td); hipFree(e->q_on_s); hipFree(e->q_on_sp); hipFree(e->q_tg_sp);
    hipFree(e->ytarget); hipFree(e->best); hipFree(e->gb_rows); hipFree(e->gb_r); hipFree(e->gb_done); hipFree(e->gb_w); hipFree(e->gb_a); hipFree(e->gb_idx);
    hipFree(e->gb_r2); hipFree(e->gb_done2); hipFree(e->gb_w2); hipFree(e->gb_a2);
    free_policy_ws(e); free_envs(e); hipFree(e->dp_send); hipFree(e->dp_recv);
    for (void* p : e->prog_allocs) hipFree(p);
    hipFree(e->ep_s); hipFree(e->ep_sp); hipFree(e->ep_a); hipFree(e->ep_r); hipFree(e->ep_done); hipFree(e->ep_len); hipFree(e->ep_idx); hipFree(e->ep_start);
    hipFree(e->r_a); hipFree(e->r_r); hipFree(e->r_done); hipFree(e->r_mask);
    for (int i = 0; i < e->nl; i++) { hipFree(e->gx_on[i]); hipFree(e->gx_tg[i]); hipFree(e->cst_on[i]); hipFree(e->cst_tg[i]); hipFree(e->gates[i]); hipFree(e->tcb[i]); hipFree(e->hprev_buf[i]);
        hipFree(e->cprev_buf[i]); hipFree(e->dG[i]); hipFree(e->dhn[i]); hipFree(e->dcn[i]); for (int k = 0; k < 2; k++) { hipFree(e->pol_h[i][k]); hipFree(e->pol_c[i][k]); } hipFree(e->pol_gx[i]); }
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->stream2) hipStreamDestroy(e->stream2);
    if (e->stream3) hipStreamDestroy(e->stream3);
    if (e->ev_xa) hipEventDestroy(e->ev_xa); if (e->ev_xb) hipEventDestroy(e->ev_xb); if (e->ev_xc) hipEventDestroy(e->ev_xc);
    hipFree(e->dp_recv_b); hipFree(e->sim_idx); hipFree(e->sim_td);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);

    delete e; return 0;
}
extern "C" int dqn_engine_get_plan(dqn_engine_t* e, dqn_layer_plan* p) { if (!e) return fail("null engine handle");
    for (int i = 0; i < e->nl; i++) { p[i].fwd_kc = e->L[i].fwd_kc; p[i].dx_kc = e->L[i].dx_kc; p[i].dw_kc = e->L[i].dw_kc; } return 0;
}
extern "C" int dqn_n_params(dqn_engine_t* e, size_t* n) { if (!e) return fail("null engine handle"); *n = e->P; return 0; }

// ---------------------------------------------------------------- parameters
static int put_vec(dqn_engine* e, const float* host, float* dev) {   // external layout -> internal
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->io_tmp, host, e->P * 4, hipMemcpyHostToDevice, e->stream));
    launch_convert_params(e->stream, e->L_dev, e->nl, e->io_tmp, dev, 1, e->P);
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
static int get_vec(dqn_engine* e, const float* dev, float* host) {
    HIPCHK(hipSetDevice(e->device));
    launch_convert_params(e->stream, e->L_dev, e->nl, dev, e->io_tmp, 0, e->P);
    HIPCHK(hipMemcpyAsync(host, e->io_tmp, e->P * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
extern "C" int dqn_batch_arena_elem_bytes(dqn_engine_t* e, int* bytes) { if (!e) return fail("null engine handle");
    if (build_program(e)) return -1;
    *bytes = e->arena_u8 ? 1 : 4; return 0;
}
extern "C" int dqn_set_params(dqn_engine_t* e, int which, const float* flat, size_t n) { if (!e) return fail("null engine handle");
    if (n != e->P) return fail("set_params: got %zu values, the network has %zu parameters", n, e->P);
    return put_vec(e, flat, which == DQN_NET_TARGET ? e->p_tg : e->p_on);
}
extern "C" int dqn_get_params(dqn_engine_t* e, int which, float* flat, size_t n) { if (!e) return fail("null engine handle");
    if (n != e->P) return fail("get_params: size mismatch (%zu vs %zu)", n, e->P);
    return get_vec(e, which == DQN_NET_TARGET ? e->p_tg : e->p_on, flat);
}
extern "C" int dqn_get_grads(dqn_engine_t* e, float* flat, size_t n) { if (!e) return fail("null engine handle");
    if (n != e->P) return fail("get_grads: size mismatch"); return get_vec(e, e->grad, flat);
}
extern "C" int dqn_sync_target(dqn_engine_t* e) { if (!e) return fail("null engine handle");   // Flux.loadparams!(target_q, params(active_q)), src/solver.jl:142-145
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->p_tg, e->p_on, e->Pint * 4, hipMemcpyDeviceToDevice, e->stream)); return 0;
}
extern "C" int dqn_get_adam_state(dqn_engine_t* e, float* m, float* v, double* bp, size_t n) { if (!e) return fail("null engine handle");
    if (n != e->P) return fail("size mismatch");
    if (m && get_vec(e, e->m, m)) return -1;
    if (v && get_vec(e, e->v, v)) return -1;
    if (bp) { StepState s; HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipMemcpy(&s, e->state, sizeof s, hipMemcpyDeviceToHost)); const int sl = (int)((s.step + 1) & 1); bp[0] = s.bp[sl][0]; bp[1] = s.bp[sl][1]; }
    return 0;
}
extern "C" int dqn_set_adam_state(dqn_engine_t* e, const float* m, const float* v, const double* bp, size_t n) { if (!e) return fail("null engine handle");
    if (n != e->P) return fail("size mismatch");
    if (m && put_vec(e, m, e->m)) return -1;
    if (v && put_vec(e, v, e->v)) return -1;
    if (bp) {
        StepState s; HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipMemcpy(&s, e->state, sizeof s, hipMemcpyDeviceToHost));
        s.bp[0][0] = s.bp[1][0] = bp[0]; s.bp[0][1] = s.bp[1][1] = bp[1]; HIPCHK(hipMemcpy(e->state, &s, sizeof s, hipMemcpyHostToDevice));
    }
    return 0;
}

// ---------------------------------------------------------------- replay
extern "C" int dqn_replay_add(dqn_engine_t* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done,
                              const float* td_err, int n) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_episode_add (EpisodeReplayBuffer, src/episode_replay.jl)");
    const size_t row = (size_t)e->E * (e->hp.obs_dtype == DQN_OBS_U8 ? 1 : 4);
    for (int i = 0; i < n; i++) {
        if (a[i] < 0 || a[i] >= e->nA) return fail("action index %d out of range 0..%d", a[i], e->nA - 1);
        const float td = td_err ? td_err[i] : fabsf(r[i]);
        if (!(td + e->hp.prio_eps > 0.0f)) return fail("AssertionError: td_err + r.eps > 0");   // ...replay.jl:66
    }
    const int chunk = (int)std::min<long long>(dqn_engine::ADD_CHUNK, e->cap);   // a chunk never writes one ring slot twice
    for (int o = 0; o < n; o += chunk) {
        const int c = std::min(chunk, n - o);
        // rows go straight into their ring slots (<= 2 contiguous segments)
        const long long first = std::min<long long>(c, e->cap - e->widx);
        const char *sp0 = (const char*)sp + (size_t)o * row, *s0 = (const char*)s + (size_t)o * row;
        HIPCHK(hipMemcpyAsync((char*)e->s_rows + (size_t)e->widx * row, s0, (size_t)first * row, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync((char*)e->sp_rows + (size_t)e->widx * row, sp0, (size_t)first * row, hipMemcpyHostToDevice, e->stream));
        if (first < c) {
            HIPCHK(hipMemcpyAsync(e->s_rows, s0 + (size_t)first * row, (size_t)(c - first) * row, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->sp_rows, sp0 + (size_t)first * row, (size_t)(c - first) * row, hipMemcpyHostToDevice, e->stream));
        }
        HIPCHK(hipMemcpyAsync(e->st_a, a + o, (size_t)c * 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->st_r, r + o, (size_t)c * 4, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->st_done, done + o, (size_t)c, hipMemcpyHostToDevice, e->stream));
        if (td_err) HIPCHK(hipMemcpyAsync(e->st_td, td_err + o, (size_t)c * 4, hipMemcpyHostToDevice, e->stream));
        launch_replay_commit(e->stream, c, e->widx, e->cap, e->cap2, e->st_a, e->st_r, e->st_done, td_err ? e->st_td : nullptr, e->hp.prio_eps,
                             e->hp.prio_alpha, e->ra, e->rr, e->rdone, e->tree, e->state);
        HIPCHK(hipStreamSynchronize(e->stream));   // the staging buffers are reused by the next chunk
        e->widx = (e->widx + c) % e->cap; e->size = std::min(e->cap, e->size + c);
    }
    return 0;
}
extern "C" int dqn_replay_size(dqn_engine_t* e, int64_t* cur, int64_t* cap) { if (!e) return fail("null engine handle"); if (cur) *cur = e->size; if (cap) *cap = e->cap; return 0; }
extern "C" int dqn_replay_get_priorities(dqn_engine_t* e, float* prio, int64_t n) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(prio, e->tree + e->cap2, (size_t)n * 4, hipMemcpyDeviceToHost)); return 0;
}
// ---------------------------------------------------------------- checkpoint / resume
extern "C" int dqn_replay_export(dqn_engine_t* e, int64_t first, int64_t n, void* s, void* sp, int32_t* a, float* r, uint8_t* done, float* prio) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_episode_export");
    if (first < 0 || n < 0 || first + n > e->size) return fail("BoundsError: rows %lld..%lld outside 0..%lld", (long long)first, (long long)(first + n - 1), (long long)e->size - 1);
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t row = (size_t)e->E * (e->hp.obs_dtype == DQN_OBS_U8 ? 1 : 4);
    if (s) HIPCHK(hipMemcpy(s, (const char*)e->s_rows + (size_t)first * row, (size_t)n * row, hipMemcpyDeviceToHost));
    if (sp) HIPCHK(hipMemcpy(sp, (const char*)e->sp_rows + (size_t)first * row, (size_t)n * row, hipMemcpyDeviceToHost));
    if (a) HIPCHK(hipMemcpy(a, e->ra + first, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (r) HIPCHK(hipMemcpy(r, e->rr + first, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (done) HIPCHK(hipMemcpy(done, e->rdone + first, (size_t)n, hipMemcpyDeviceToHost));
    if (prio) HIPCHK(hipMemcpy(prio, e->tree + e->cap2 + first, (size_t)n * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int dqn_replay_import(dqn_engine_t* e, int64_t n, const void* s, const void* sp, const int32_t* a, const float* r, const uint8_t* done, const float* prio) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_episode_import");
    if (n < 0 || n > e->cap) return fail("import of %lld transitions into a replay of capacity %lld", (long long)n, (long long)e->cap);
    for (int64_t i = 0; i < n; i++) {
        if (a[i] < 0 || a[i] >= e->nA) return fail("action index %d out of range 0..%d", a[i], e->nA - 1);
        if (!(prio[i] > 0.0f)) return fail("AssertionError: all(new_priorities .> 0f0)");
    }
    HIPCHK(hipStreamSynchronize(e->stream));
    const size_t row = (size_t)e->E * (e->hp.obs_dtype == DQN_OBS_U8 ? 1 : 4);
    HIPCHK(hipMemcpy(e->s_rows, s, (size_t)n * row, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(e->sp_rows, sp, (size_t)n * row, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->ra, a, (size_t)n * 4, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(e->rr, r, (size_t)n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(e->rdone, done, (size_t)n, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(e->tree, 0, 2 * (size_t)e->cap2 * 4));
    HIPCHK(hipMemcpy(e->tree + e->cap2, prio, (size_t)n * 4, hipMemcpyHostToDevice));
    launch_tree_rebuild(e->stream, e->tree, e->cap2);
    StepState st; HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipMemcpy(&st, e->state, sizeof st, hipMemcpyDeviceToHost));
    st.size = n; st.pre_valid = 0; HIPCHK(hipMemcpy(e->state, &st, sizeof st, hipMemcpyHostToDevice));
    e->size = n; e->widx = n % e->cap;
    return 0;
}
extern "C" int dqn_get_counters(dqn_engine_t* e, dqn_counters* out) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    StepState st; HIPCHK(hipMemcpy(&st, e->state, sizeof st, hipMemcpyDeviceToHost));
    if (e->hp.recurrence) { out->size = e->ep_size; out->widx = e->ep_widx; out->sample_ctr = e->drqn_draws; out->train_steps = st.step; return 0; }      // episode replay: episodes, ring cursor, the host sampler's draw counter
    out->size = e->size; out->widx = e->widx; out->sample_ctr = st.sample_ctr; out->train_steps = st.step; return 0;
}
extern "C" int dqn_set_counters(dqn_engine_t* e, const dqn_counters* in) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    const long long have = e->hp.recurrence ? e->ep_size : e->size, capn = e->hp.recurrence ? e->ep_cap : e->cap;
    if (in->size != have) return fail("counters.size (%lld) differs from the replay's (%lld): import the replay first", (long long)in->size, have);
    if (in->widx < 0 || in->widx >= capn) return fail("counters.widx out of range");
    StepState st; HIPCHK(hipMemcpy(&st, e->state, sizeof st, hipMemcpyDeviceToHost));
    // the Adam beta powers are double-buffered by step parity: after a step with counter S the LIVE pair (the one the next step reads) sits in
    // slot (S + 1) & 1 -- the slot dqn_get_adam_state reports.  Whatever the new parity, make both slots hold the live pair.
    const int live = (int)((st.step + 1ull) & 1ull);
    st.bp[live ^ 1][0] = st.bp[live][0]; st.bp[live ^ 1][1] = st.bp[live][1];
    if (!e->hp.recurrence) st.sample_ctr = in->sample_ctr;
    st.step = in->train_steps; st.pre_valid = 0;
    HIPCHK(hipMemcpy(e->state, &st, sizeof st, hipMemcpyHostToDevice));
    if (e->hp.recurrence) { e->ep_widx = in->widx; e->drqn_draws = in->sample_ctr; } else e->widx = in->widx;
    return 0;
}
static int check_idx(dqn_engine* e, const int64_t* idx, int n) {
    for (int i = 0; i < n; i++) if (idx[i] < 0 || idx[i] >= e->size) return fail("BoundsError: index %lld outside 0..%lld", (long long)idx[i], (long long)e->size - 1);
    return 0;
}
extern "C" int dqn_replay_sample(dqn_engine_t* e, int64_t* idx_out) { if (!e) return fail("null engine handle");
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");   // ...replay.jl:83
    HIPCHK(hipSetDevice(e->device));
    launch_sample(e->stream, e->B, e->cap2, e->tree, e->hp.seed, e->idx, e->state, 1, e->hp.sample_distinct);
    if (idx_out) { HIPCHK(hipMemcpyAsync(idx_out, e->idx, (size_t)e->B * 8, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
    return 0;
}
extern "C" int dqn_replay_get_batch(dqn_engine_t* e, const int64_t* idx, float* s, int32_t* a, float* r, float* sp, float* done, float* w) { if (!e) return fail("null engine handle");
    if (check_idx(e, idx, e->B)) return -1;
    HIPCHK(hipSetDevice(e->device));
    const int B = e->B; const size_t rb = (size_t)B * e->E * 4; const int u8 = e->hp.obs_dtype == DQN_OBS_U8;
    HIPCHK(hipMemcpyAsync(e->gb_idx, idx, (size_t)B * 8, hipMemcpyHostToDevice, e->stream));
    if (s) { launch_gather_rows(e->stream, e->s_rows, u8, e->E, B, e->gb_idx, e->gb_rows); HIPCHK(hipMemcpyAsync(s, e->gb_rows, rb, hipMemcpyDeviceToHost, e->stream)); }
    if (sp) { launch_gather_rows(e->stream, e->sp_rows, u8, e->E, B, e->gb_idx, e->gb_rows); HIPCHK(hipMemcpyAsync(sp, e->gb_rows, rb, hipMemcpyDeviceToHost, e->stream)); }
    launch_batch_meta(e->stream, B, e->cap2, e->gb_idx, e->ra, e->rr, e->rdone, e->tree, e->hp.prio_beta, e->state, e->gb_a, e->gb_r, e->gb_done, e->gb_w);
    if (a) HIPCHK(hipMemcpyAsync(a, e->gb_a, B * 4, hipMemcpyDeviceToHost, e->stream));
    if (r) HIPCHK(hipMemcpyAsync(r, e->gb_r, B * 4, hipMemcpyDeviceToHost, e->stream));
    if (done) HIPCHK(hipMemcpyAsync(done, e->gb_done, B * 4, hipMemcpyDeviceToHost, e->stream));
    if (w) HIPCHK(hipMemcpyAsync(w, e->gb_w, B * 4, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
static int check_state_err(dqn_engine* e) {
    StepState s; HIPCHK(hipMemcpy(&s, e->state, sizeof s, hipMemcpyDeviceToHost));
    if (s.err == 1) return fail("AssertionError: td_err + r.eps > 0");
    if (s.err == 2) return fail("AssertionError: all(new_priorities .> 0f0)");
    return 0;
}
extern "C" int dqn_update_priorities(dqn_engine_t* e, const int64_t* idx, const float* td, int n) { if (!e) return fail("null engine handle");
    if (check_idx(e, idx, n)) return -1;
    // the reference asserts BEFORE it assigns (...replay.jl:77-79), so a bad TD error must leave the priorities untouched; with eps > 0
    // (|td| + eps)^alpha fails to be > 0 only for NaN
    for (int i = 0; i < n; i++) if (td[i] != td[i]) return fail("AssertionError: all(new_priorities .> 0f0)");
    HIPCHK(hipSetDevice(e->device));
    for (int o = 0; o < n; o += e->B) {   // the device buffers hold B entries
        const int c = std::min(e->B, n - o);
        HIPCHK(hipMemcpyAsync(e->gb_idx, idx + o, (size_t)c * 8, hipMemcpyHostToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(e->gb_w, td + o, (size_t)c * 4, hipMemcpyHostToDevice, e->stream));
        launch_update_priorities(e->stream, c, e->cap2, e->gb_idx, e->gb_w, e->hp.prio_eps, e->hp.prio_alpha, e->tree, e->state, 0, 1.0, 1.0, nullptr, 0);
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    return check_state_err(e);
}

// ---------------------------------------------------------------- the train step
void fwd_layer(dqn_engine* e, const LayerDev& l, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, const char* name) {
    prof_begin(e, name);
    if (!(e->hp.use_mfma && launch_mfma_fwd(e->stream, l, P, X, ldx, col0, ncols, Y, e->partials)))
        launch_valu_fwd(e->stream, l, P, X, ldx, col0, ncols, Y, e->partials);
    prof_end(e);
}

void enqueue_step(dqn_engine* e, bool sample, int phase) {
    e->step_sampled = sample;
    if (phase == PH_PRE2) { for (size_t i = e->prog_pre1_end; i < e->prog_post_begin; i++) { if ((long)i == e->final_reduce_step && ((e->adam_segs.n > 0 && !e->comm && !e->sim_world) || (e->dp_gather && e->dp_pack_folds))) continue; RUN(e, e->prog[i].name, e->prog[i].fn(e)); } return; }
    if (phase != PH_POST) {
        if (e->hp.recurrence) {
            EpGatherArgs g; g.ep_s = e->ep_s; g.ep_sp = e->ep_sp; g.ep_a = e->ep_a; g.ep_r = e->ep_r; g.ep_done = e->ep_done; g.ep_len = e->ep_len; g.ep_idx = e->ep_idx; g.ep_start = e->ep_start;
            g.E = e->E; g.B = e->B; g.T = e->T; g.x0 = e->x0; g.a_out = e->r_a; g.r_out = e->r_r; g.done_out = e->r_done; g.mask_out = e->r_mask;
            if (!e->drqn_fused) RUN(e, "gather_episodes", launch_gather_episodes(e->stream, g));      // the fused recurrent step gathers its own columns (drqn_cols.hip)
        } else {
            // the descent is fused into the gather (every workgroup repeats it) while that is cheaper than a launch of its own:
            // small batches.  At B = 512 / 1e6 leaves the repeats cost more than the ~5 us launch, so sample once, then gather.
            // hp.sample_distinct (r05): the fused launch's workgroups dedupe the list themselves (gather_distinct_list) -- except on the u8-rows-to-float-arena kernel,
            // which keeps its own index code: there one workgroup draws AND dedupes (k_sample), then the gather reads the list
            const bool dist_fused_ok = !(e->hp.obs_dtype == DQN_OBS_U8 && !e->arena_u8 && (e->E & 3) == 0);
            const bool fused = sample && e->B <= 64 && (!e->hp.sample_distinct || dist_fused_ok);
            if (e->step_take_pre || e->tiny) {}      // the previous step's Adam launch gathered this batch (PreGather) / the step's one launch gathers itself (tiny_step.hip)
            else {
            if (sample && !fused) RUN(e, "sample", launch_sample(e->stream, e->B, e->cap2, e->tree, e->hp.seed, e->idx, e->state, 0, e->hp.sample_distinct));    // k_td bumps the Philox counter
            BatchMeta bm; bm.a = e->ra; bm.r = e->rr; bm.done = e->rdone; bm.beta = e->hp.prio_beta; bm.a_out = e->gb_a2; bm.r_out = e->gb_r2; bm.done_out = e->gb_done2; bm.w_out = e->gb_w2; bm.distinct = e->hp.sample_distinct ? 1 : 0;
            RUN(e, fused ? "sample_gather" : "gather", launch_gather_fb(e->stream, e->s_rows, e->sp_rows, e->hp.obs_dtype == DQN_OBS_U8, e->E, e->B, e->idx, e->x0,
                                                                        fused ? 1 : 0, e->cap2, e->tree, e->hp.seed, e->state, bm, (e->hp.sample_distinct && !fused) ? nullptr : e->idx_pre, e->arena_u8 ? 1 : 0));
            }
        }
        for (size_t i = 0; i < (phase == PH_PRE1 ? e->prog_pre1_end : e->prog_post_begin); i++) {
            if ((long)i == e->final_reduce_step && ((e->adam_segs.n > 0 && !e->comm && !e->sim_world) || (e->dp_gather && e->dp_pack_folds))) continue;   // folded into k_adam / k_dp_pack   // folded into k_adam
            RUN(e, e->prog[i].name, e->prog[i].fn(e));
        }
    }
    if (phase != PH_PRE && phase != PH_PRE1) for (size_t i = e->prog_post_begin; i < e->prog.size(); i++) RUN(e, (e->step_pregather && (long)i == e->adam_step) ? "adam+gather" : e->prog[i].name, e->prog[i].fn(e));
    if (e->step_publish && phase == PH_ALL)
        RUN(e, "publish", launch_publish_scalars(e->stream, e->state, e->gmax_part, (e->gmax_used > 0 && e->gmax_used <= gmax_slots(e->Pint)) ? e->gmax_used : gmax_slots(e->Pint), e->pub_ctr, e->mail_dev));
}
int exchange_grads(dqn_engine* e);
int exchange_segment(dqn_engine* e, int seg);
int capture(dqn_engine* e, bool sample, int phase, hipGraphExec_t* out, int repeat) {
    hipGraph_t g;
    (void)hipGetLastError();
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    int xrc = 0;
    if (phase == PH_DP_ONE) {
        // the replica step with its collective(s) INSIDE the graph (RCCL supports stream capture): one graph launch and no host-side collective enqueue
        // per step.  dp_overlap: the exchange stream is forked from / joined to the captured stream with events, so all-gather A runs beside PRE2.
        if (e->dp_gather && e->dp_overlap) {
            enqueue_step(e, sample, PH_PRE1);
            hipEventRecord(e->ev_xa, e->stream); hipStreamWaitEvent(e->stream3, e->ev_xa, 0); xrc |= exchange_segment(e, 0);
            enqueue_step(e, sample, PH_PRE2);
            hipEventRecord(e->ev_xb, e->stream); hipStreamWaitEvent(e->stream3, e->ev_xb, 0); xrc |= exchange_segment(e, 1);
            hipEventRecord(e->ev_xc, e->stream3); hipStreamWaitEvent(e->stream, e->ev_xc, 0);
            enqueue_step(e, sample, PH_POST);
        } else { enqueue_step(e, sample, PH_PRE); xrc |= exchange_grads(e); enqueue_step(e, sample, PH_POST); }
    } else
    for (int r = 0; r < repeat; r++) enqueue_step(e, sample, phase);
    hipError_t lerr = hipGetLastError();          // a launch refused during capture never becomes a graph node
    if (e->launch_failed) { e->launch_failed = false; if (lerr == hipSuccess) lerr = hipErrorInvalidValue; }      // a launcher could not get the dynamic LDS it needs on this device
    { const hipError_t ce = hipStreamEndCapture(e->stream, &g); if (ce != hipSuccess || xrc) { (void)hipGetLastError(); if (ce == hipSuccess) hipGraphDestroy(g); return fail("capturing the train step failed (%s%s)", hipGetErrorString(ce), xrc ? "; collective refused" : ""); } }
    if (lerr != hipSuccess) { hipGraphDestroy(g); return fail("HIP error %s while capturing the train step", hipGetErrorString(lerr)); }
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    // the FIRST launch of an executable graph prepares its packets on the device side; done here instead, a short timed call (the driver's 20 steps
    // launch the 4-step middle graph for the first time inside the timed region: its warm-up of 5 steps is too short to reach it) does not pay it
    if (!e->opt.no_graph_upload) (void)hipGraphUpload(*out, e->stream);
    (void)hipGetLastError();
    HIPCHK(hipGraphDestroy(g)); return 0;
}
// dp_overlap: segment 0 = the wide layers' operands [0, dp_count_a) -> dp_recv, segment 1 = the small gradients [dp_count_a, dp_count) -> dp_recv_b, both on stream3
int exchange_segment(dqn_engine* e, int seg) {
    const size_t beg = seg ? e->dp_count_a : 0, cnt = seg ? e->dp_count - e->dp_count_a : e->dp_count_a; float* recv = seg ? e->dp_recv_b : e->dp_recv;
    if (e->sim_world) { for (int r = 0; r < e->sim_world; r++) HIPCHK(hipMemcpyAsync(recv + (size_t)r * cnt, e->dp_send + beg, cnt * 4, hipMemcpyDeviceToDevice, e->stream3)); return 0; }
    const int rc = g_rccl.AllGather(e->dp_send + beg, recv, cnt, /*ncclFloat*/ 7, e->comm, e->stream3);
    if (rc) return fail("ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return 0;
}
int exchange_grads(dqn_engine* e) {
    if (e->dp_gather) {      // every rank's packed block -> all ranks (rank-major), on the engine stream between the two halves of the step
        if (e->sim_world) {
            for (int r = 0; r < e->sim_world; r++) HIPCHK(hipMemcpyAsync(e->dp_recv + (size_t)r * e->dp_count, e->dp_send, e->dp_count * 4, hipMemcpyDeviceToDevice, e->stream));
            return 0;
        }
        const int rc = g_rccl.AllGather(e->dp_send, e->dp_recv, e->dp_count, /*ncclFloat*/ 7, e->comm, e->stream);
        if (rc) return fail("ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
        return 0;
    }
    if (e->sim_world) return fail("DQN_SIM_WORLD needs the gather exchange (no wide dense layer in this network, or DQN_DP_ALLREDUCE is set)");
    const int rc = g_rccl.AllReduce(e->grad, e->grad, e->Pint, /*ncclFloat*/ 7, /*ncclSum*/ 0, e->comm, e->stream);
    if (rc) return fail("ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return 0;
}
// take_pre / pregather: only dqn_train_steps sets them (a sampled step follows / preceded this one and nothing touches the replay in between)
int run_step(dqn_engine* e, bool sample, bool take_pre, bool pregather) {
    if (build_program(e)) return -1;
    const int gi = sample ? 0 : 1;
    e->step_take_pre = e->step_pregather = false;
    if (sample && e->pg_ok && (take_pre || pregather) && e->world <= 1 && !(e->comm && e->force_comm)) {
        e->step_take_pre = take_pre; e->step_pregather = pregather;
        const bool pub = e->step_publish && e->mail_dev != nullptr && take_pre && !pregather; e->step_publish = pub;      // the LAST step of a dqn_train_steps call may publish its scalars
        int rc = 0;
        if (e->hp.use_graph && !e->profiling) {
            hipGraphExec_t& g = pub ? e->g_pgv_pub : e->g_pgv[take_pre ? 1 : 0][pregather ? 1 : 0];
            if (!g && capture(e, true, PH_ALL, &g)) rc = -1; else HIPCHK(hipGraphLaunch(g, e->stream));
        } else { enqueue_step(e, true, PH_ALL); HIPCHK(hipGetLastError()); }
        e->step_take_pre = e->step_pregather = false; e->step_publish = false;
        if (!rc && pub) e->pub_issued++;
        return rc;
    }
    if (e->world > 1 || (e->comm && e->force_comm)) {      // data-parallel replicas (also DQN_SIM_WORLD: world = k without a communicator)
        // scalars on replicas (r05): the publish launch is enqueued EAGERLY behind the step (behind the exchange and the Adam launch that takes grad_norm after it), outside the
        // step's graphs -- one kernel boundary, no fold launch / D2H copy / stream synchronize, so the drop-in seam costs a replica what it costs a single device
        const bool pub_dp = e->step_publish && e->mail_dev != nullptr && !e->profiling; e->step_publish = false;
        auto publish_dp = [&]() { if (!pub_dp) return; launch_publish_scalars(e->stream, e->state, e->gmax_part, (e->gmax_used > 0 && e->gmax_used <= gmax_slots(e->Pint)) ? e->gmax_used : gmax_slots(e->Pint), e->pub_ctr, e->mail_dev); e->pub_issued++; };
        // the pre-gather is rank-local (own replay, own arena), so it works on replicas too: first half without the gather launch, the Adam
        // launch of the second half gathers the next batch.  Every rank takes the same variant (same configuration, same call).
        const bool tp = sample && e->pg_ok && take_pre, pgth = sample && e->pg_ok && pregather;
        e->step_take_pre = tp; e->step_pregather = pgth;
        int rc = 0;
        if (!e->opt.dp_no_one_graph && e->hp.use_graph && !e->profiling && !e->sim_world && e->comm && e->dp_one_state >= 0) {
            hipGraphExec_t& g = e->g_dp_one[(tp ? 2 : 0) + (pgth ? 1 : 0) + 0];
            if (!g && gi == 0) { if (capture(e, sample, PH_DP_ONE, &g)) { e->dp_one_state = -1; g = nullptr; } else e->dp_one_state = 1; }
            if (g && gi == 0) { HIPCHK(hipGraphLaunch(g, e->stream)); e->step_take_pre = e->step_pregather = false; publish_dp(); return 0; }
        }
        if (e->dp_gather && e->dp_overlap) {
            // three segments: [.. head level, pack of the wide operands] | all-gather A on stream3, concurrently: [rest of the backward pass, pack of the small
            // gradients] | all-gather B on stream3 | [wide dW over the gathered samples, sum over ranks, Adam]
            const bool gr = e->hp.use_graph && !e->profiling;
            hipGraphExec_t& g1 = e->g_pre1[tp ? 2 : gi]; hipGraphExec_t& gpost = pgth ? e->g_post_pg : e->g_post;
            if (gr) { if (!g1 && capture(e, sample, PH_PRE1, &g1)) rc = -1; if (!rc && !e->g_pre2 && capture(e, sample, PH_PRE2, &e->g_pre2)) rc = -1; if (!rc && !gpost && capture(e, sample, PH_POST, &gpost)) rc = -1; }
            if (!rc) {
                if (gr) HIPCHK(hipGraphLaunch(g1, e->stream)); else { enqueue_step(e, sample, PH_PRE1); HIPCHK(hipGetLastError()); }
                HIPCHK(hipEventRecord(e->ev_xa, e->stream)); HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_xa, 0));
                if (exchange_segment(e, 0)) rc = -1;
            }
            if (!rc) {
                if (gr) HIPCHK(hipGraphLaunch(e->g_pre2, e->stream)); else { enqueue_step(e, sample, PH_PRE2); HIPCHK(hipGetLastError()); }
                HIPCHK(hipEventRecord(e->ev_xb, e->stream)); HIPCHK(hipStreamWaitEvent(e->stream3, e->ev_xb, 0));
                if (exchange_segment(e, 1)) rc = -1;
                HIPCHK(hipEventRecord(e->ev_xc, e->stream3)); HIPCHK(hipStreamWaitEvent(e->stream, e->ev_xc, 0));
            }
            if (!rc) { if (gr) HIPCHK(hipGraphLaunch(gpost, e->stream)); else { enqueue_step(e, sample, PH_POST); HIPCHK(hipGetLastError()); } }
        }
        else if (e->hp.use_graph && !e->profiling) {
            hipGraphExec_t& gpre = tp ? e->g_pre_tp : e->g_pre[gi];
            hipGraphExec_t& gpost = pgth ? e->g_post_pg : e->g_post;
            if (!gpre && capture(e, sample, PH_PRE, &gpre)) rc = -1;
            if (!rc && !gpost && capture(e, sample, PH_POST, &gpost)) rc = -1;
            if (!rc) {
                HIPCHK(hipGraphLaunch(gpre, e->stream));
                if (exchange_grads(e)) rc = -1;
                else HIPCHK(hipGraphLaunch(gpost, e->stream));
            }
        } else { enqueue_step(e, sample, PH_PRE); HIPCHK(hipGetLastError()); if (exchange_grads(e)) rc = -1; else { enqueue_step(e, sample, PH_POST); HIPCHK(hipGetLastError()); } }
        e->step_take_pre = e->step_pregather = false;
        if (!rc) publish_dp();
        return rc;
    }
    // publish (dqn_train_step with scalar outputs, dqn_train_step_async): the step's last launch also writes (loss, grad_norm) into the host mailbox
    const bool pub = e->step_publish && e->mail_dev != nullptr; e->step_publish = pub;
    int rc = 0;
    if (e->hp.use_graph && !e->profiling) {
        hipGraphExec_t& g = pub ? e->g_full_pub[gi] : e->g_full[gi];
        if (!g && capture(e, sample, PH_ALL, &g)) rc = -1;
        else { const hipError_t le = hipGraphLaunch(g, e->stream); if (le != hipSuccess) rc = fail("HIP error %s launching the train-step graph", hipGetErrorString(le)); }
    } else { enqueue_step(e, sample, PH_ALL); const hipError_t le = hipGetLastError(); if (le != hipSuccess || e->launch_failed) { e->launch_failed = false; rc = fail("HIP error %s enqueuing the train step", hipGetErrorString(le)); } }      // eager launches: a refused launch (bad LDS size, bad grid) is an error, not a silent no-op
    e->step_publish = false;
    if (!rc && pub) e->pub_issued++;
    return rc;
}
// ---- the scalar mailbox, host side
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    asm volatile("yield" ::: "memory");
#elif defined(__powerpc64__)
    asm volatile("or 27,27,27" ::: "memory");
#else
    asm volatile("" ::: "memory");
#endif
}
#define DQN_MAIL_SPIN_US 20000       /* spin on the mapped record this long (a step is 25 us - 1 ms), then stop burning the core and poll the stream */
#define DQN_MAIL_TIMEOUT_S 30        /* a publish that has not arrived after this long while the stream is still busy: a kernel of the step is hung -- say so */
static inline double now_s() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int mail_err_msg(int err, unsigned long long step) {
    // (`step` is the step counter at PUBLISH time: inside dqn_train_steps(n) / a rollout cycle only the last step publishes, so the failure is at or before it)
    if (err == 2) return fail("AssertionError: all(new_priorities .> 0f0) (at or before train step %llu)", step);
    if (err == 3) return fail("internal error: a pre-gathered batch was consumed after the replay changed (StepState::pre_valid != 2; at or before train step %llu)", step);
    return fail("device-side error %d at or before train step %llu", err, step);
}
// wait until publish `ticket` has arrived (its record holds seq == ticket).  Returns 0 = arrived, 1 = not yet (wait == false), -1 = error (message set)
static int mail_arrive(dqn_engine* e, unsigned long long ticket, bool wait) {
    volatile StepMail* m = e->mail_host + (ticket & (DQN_MAIL_SLOTS - 1));
    auto seq_now = [&]() { return __atomic_load_n(&m->seq, __ATOMIC_ACQUIRE); };
    if (seq_now() == ticket) return 0;
    if (!wait) return 1;
    const double t0 = now_s();
    for (unsigned spin = 0;; spin++) {
        if (seq_now() == ticket) return 0;
        cpu_relax();
        if ((spin & 1023) == 1023 && now_s() - t0 > 1e-6 * DQN_MAIL_SPIN_US) break;
    }
    for (;;) {      // the step is long (or stuck): poll the stream instead of the record
        const hipError_t q = hipStreamQuery(e->stream);
        if (seq_now() == ticket) return 0;
        if (q == hipSuccess) { (void)hipGetLastError(); return seq_now() == ticket ? 0 : fail("step scalars: the stream is idle but publish %llu did not arrive (slot holds %llu): the step's publish launch was refused", ticket, (unsigned long long)seq_now()); }
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return fail("step scalars: HIP error %s while waiting for publish %llu", hipGetErrorString(q), ticket); }
        (void)hipGetLastError();
        if (now_s() - t0 > (double)DQN_MAIL_TIMEOUT_S) return fail("step scalars: publish %llu did not arrive within %d s and the stream is still busy -- a kernel of the train step is hung (DQN_MAIL_TIMEOUT_S)", ticket, DQN_MAIL_TIMEOUT_S);
        struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr);
    }
}
// device-side assertion failures travel in the records (k_publish_scalars consumes StepState::err, so each failure sits in exactly ONE record): every arrived record is
// looked at once, in publish order (publishes execute in stream order, so arrival is in order too); the first error found is reported -- by whichever call sweeps first:
// a step-scalars fetch or the next dqn_train_step / dqn_train_step_async -- naming the step that failed
static int mail_sweep(dqn_engine* e) {
    while (e->mail_swept < e->pub_issued) {
        const unsigned long long t = e->mail_swept + 1;
        volatile StepMail* m = e->mail_host + (t & (DQN_MAIL_SLOTS - 1));
        if (__atomic_load_n(&m->seq, __ATOMIC_ACQUIRE) != t) break;      // not yet arrived (or already overwritten: mail_reserve keeps that from happening)
        e->mail_swept = t;
        const int err = m->err; if (err) return mail_err_msg(err, m->step);
    }
    return 0;
}
// before publish number pub_issued + 1 is enqueued: the record it will overwrite (ticket pub_issued + 1 - DQN_MAIL_SLOTS) must have been swept -- the host never runs more
// than DQN_MAIL_SLOTS publishes ahead of the device, so no error record is ever lost unseen.  Also surfaces errors of earlier steps promptly
static int mail_reserve(dqn_engine* e) {
    if (!e->mail_dev) return 0;
    if (e->pub_issued + 1 > DQN_MAIL_SLOTS) { const unsigned long long need = e->pub_issued + 1 - DQN_MAIL_SLOTS; if (e->mail_swept < need && mail_arrive(e, need, true)) return -1; }
    return mail_sweep(e);
}
// the record of publish `ticket`
static int wait_mail(dqn_engine* e, unsigned long long ticket, bool wait, float* loss, float* gn, unsigned long long* published) {
    if (ticket == 0 || ticket > e->pub_issued) return fail("step scalars: ticket %llu was never issued (newest %llu)", ticket, e->pub_issued);
    if (ticket + DQN_MAIL_SLOTS <= e->pub_issued) return fail("step scalars: ticket %llu is older than the %d newest publishes (newest %llu)", ticket, DQN_MAIL_SLOTS, e->pub_issued);
    const int a = mail_arrive(e, ticket, wait);
    if (a < 0) return -1;
    if (published) *published = a == 0 ? ticket : 0;
    if (mail_sweep(e)) return -1;      // an assertion failure of this or an earlier step (reported once)
    if (a) return 0;
    volatile StepMail* m = e->mail_host + (ticket & (DQN_MAIL_SLOTS - 1));
    if (loss) *loss = m->loss;
    if (gn) *gn = m->gnorm;
    return 0;
}
static bool mailbox_ok(dqn_engine* e) { return e->mail_dev && !e->sim_world && !e->profiling && !e->hp.recurrence; }      // replicas included (r05: published eagerly behind the step, run_step)
int fetch_scalars(dqn_engine* e, float* loss, float* gn) {
    // globalnorm (helpers.jl:38-46): fold the Adam kernel's per-block maxima only when the host asks for the scalar
    if (gn) launch_update_priorities(e->stream, 0, e->cap2, e->idx, e->td, e->hp.prio_eps, e->hp.prio_alpha, e->tree, e->state, 1, 1.0, 1.0, e->gmax_part, (e->gmax_used > 0 && e->gmax_used <= gmax_slots(e->Pint)) ? e->gmax_used : gmax_slots(e->Pint));
    // pinned landing buffer: a device -> pageable copy is staged and costs ~100 us per call (seen as a fixed cost of every dqn_train_steps)
    if (!e->state_host) HIPCHK(hipHostMalloc((void**)&e->state_host, sizeof(StepState), hipHostMallocDefault));
    HIPCHK(hipMemcpyAsync(e->state_host, e->state, sizeof(StepState), hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
    const StepState s = *e->state_host;
    if (s.err) { HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&e->state->err, 0, 1, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }      // reported once, not on every later call
    if (s.err == 2) return fail("AssertionError: all(new_priorities .> 0f0)");
    if (s.err == 3) return fail("internal error: a pre-gathered batch was consumed after the replay changed (StepState::pre_valid != 2)");
    if (s.err) return mail_err_msg(s.err, (unsigned long long)s.step);
    if (loss) *loss = s.loss;
    if (gn) { float g; memcpy(&g, &s.gnorm_bits, 4); *gn = g; }
    return 0;
}
extern "C" int dqn_train_step(dqn_engine_t* e, const int64_t* idx, float* loss, float* grad_norm, float* td_out) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_train_step_drqn (src/solver.jl:239-287)");
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (idx) { if (check_idx(e, idx, e->B)) return -1; HIPCHK(hipMemcpyAsync(e->idx, idx, (size_t)e->B * 8, hipMemcpyHostToDevice, e->stream)); }
    // scalars only (what batch_train! returns, src/solver.jl:235): the step's last launch publishes them to the host mailbox -- no fold launch, no D2H copy,
    // no stream synchronize; the host spins on the record
    const bool mail = (loss || grad_norm) && !td_out && mailbox_ok(e);
    if (mail && mail_reserve(e)) return -1;
    e->step_publish = mail;
    if (run_step(e, idx == nullptr)) return -1;
    if (mail) return wait_mail(e, e->pub_issued, true, loss, grad_norm, nullptr);
    if (td_out) HIPCHK(hipMemcpyAsync(td_out, e->td, (size_t)e->B * 4, hipMemcpyDeviceToHost, e->stream));
    if (loss || grad_norm || td_out) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
// the same step WITHOUT waiting: returns as soon as the step is enqueued; *ticket names the (loss, grad_norm) record its last launch will publish.
// The reference only looks at batch_train!'s return values every log_freq env steps (src/solver.jl:154-167): the shim's loop fetches them there.
extern "C" int dqn_train_step_async(dqn_engine_t* e, const int64_t* idx, uint64_t* ticket) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_train_step_drqn (src/solver.jl:239-287)");
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (!e->mail_host) return fail("dqn_train_step_async: no mailbox on this engine");
    if (idx) { if (check_idx(e, idx, e->B)) return -1; HIPCHK(hipMemcpyAsync(e->idx, idx, (size_t)e->B * 8, hipMemcpyHostToDevice, e->stream)); }
    if (!mailbox_ok(e)) {
        // no device-side publish here (replicas, a profiled step, or the mapped-host allocation failed at create): the call DEGRADES to a synchronous step whose scalars are
        // recorded under the ticket by the host, so that a caller written against the async seam -- the shim's dqn_train! -- runs unchanged
        if (mail_sweep(e)) return -1;
        float l = 0.0f, g = 0.0f;
        if (run_step(e, idx == nullptr)) return -1;
        if (fetch_scalars(e, &l, &g)) return -1;
        if (mail_sweep(e)) return -1;      // (the stream is idle: every device-published record has arrived -- their error fields are looked at before the host's own record is numbered past them)
        const unsigned long long t = ++e->pub_issued; e->mail_swept = t;
        StepMail* m = e->mail_host + (t & (DQN_MAIL_SLOTS - 1));
        m->loss = l; m->gnorm = g; m->err = 0; m->step = 0; __atomic_store_n(&m->seq, t, __ATOMIC_RELEASE);
        if (e->pub_ctr) HIPCHK(hipMemcpy(e->pub_ctr, &t, sizeof t, hipMemcpyHostToDevice));      // the device-side publish counter follows (a later mailbox publish continues the sequence)
        if (ticket) *ticket = t;
        return 0;
    }
    if (mail_reserve(e)) return -1;
    e->step_publish = true;
    if (run_step(e, idx == nullptr)) return -1;
    if (ticket) *ticket = e->pub_issued;
    return 0;
}
// (loss, grad_norm) of the step that returned `ticket` (one of the DQN_MAIL_SLOTS newest).  wait != 0: blocks until the record has arrived.  wait == 0: returns at
// once; *published = ticket if the record was there (outputs written), 0 if the step has not finished yet (outputs untouched).
extern "C" int dqn_step_scalars(dqn_engine_t* e, uint64_t ticket, int wait, float* loss, float* grad_norm, uint64_t* published) { if (!e) return fail("null engine handle");
    if (!e->mail_host) return fail("dqn_step_scalars: no mailbox on this engine");
    unsigned long long pub = 0; const int rc = wait_mail(e, ticket, wait != 0, loss, grad_norm, &pub);
    if (published) *published = pub;
    return rc;
}
// TEST HOOK (DQN_SIM_WORLD = k): one data-parallel step in which this process plays k ranks with k DISTINCT batches.  Each simulated rank runs
// the first half of the step (gather .. backward .. dp_pack) on its own index list and its packed block lands in ITS slot of the gathered
// buffer -- exactly what ncclAllGather delivers -- then the second half (wide dW over the k*B gathered samples, sum over ranks, Adam) runs once.
// Equivalent single-device step: the concatenated batch of k*B samples, which is what tests/test_dp_gpu.py compares with.
// Priority updates of ranks 0..k-2 are applied after the step (every rank's IS weights see the pre-step tree, as on k real ranks with one
// shared replay).  idx: [k][B]; loss: [k] per-rank losses; td_out: [k][B].
extern "C" int dqn_sim_ranks_step(dqn_engine_t* e, const int64_t* idx, float* loss, float* grad_norm, float* td_out) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->sim_world) return fail("dqn_sim_ranks_step needs an engine created under DQN_SIM_WORLD=k");
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    const int k = e->sim_world, B = e->B;
    if (check_idx(e, idx, k * B)) return -1;
    if (build_program(e)) return -1;
    if (!e->dp_gather) return fail("DQN_SIM_WORLD needs the gather exchange (no wide dense layer in this network, or DQN_DP_ALLREDUCE is set)");
    if (!e->sim_idx) { DM(e->sim_idx, (size_t)k * B); DM(e->sim_td, (size_t)k * B); }      // per-rank index / TD copies: owned by the engine, allocated once (sim_world and B are fixed at create)
    long long* s_idx = e->sim_idx; float* s_td = e->sim_td;
    int rc = 0;
    for (int r = 0; r < k && !rc; r++) {
        HIPCHK(hipMemcpyAsync(e->idx, idx + (size_t)r * B, (size_t)B * 8, hipMemcpyHostToDevice, e->stream));
        if (e->hp.use_graph) { if (!e->g_pre[1] && capture(e, false, PH_PRE, &e->g_pre[1])) { rc = -1; break; } HIPCHK(hipGraphLaunch(e->g_pre[1], e->stream)); }
        else enqueue_step(e, false, PH_PRE);
        if (e->dp_overlap) {
            const size_t ca = e->dp_count_a, cb = e->dp_count - ca;
            HIPCHK(hipMemcpyAsync(e->dp_recv + (size_t)r * ca, e->dp_send, ca * 4, hipMemcpyDeviceToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(e->dp_recv_b + (size_t)r * cb, e->dp_send + ca, cb * 4, hipMemcpyDeviceToDevice, e->stream));
        } else
        HIPCHK(hipMemcpyAsync(e->dp_recv + (size_t)r * e->dp_count, e->dp_send, e->dp_count * 4, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(s_idx + (size_t)r * B, e->idx, (size_t)B * 8, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipMemcpyAsync(s_td + (size_t)r * B, e->td, (size_t)B * 4, hipMemcpyDeviceToDevice, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        StepState st; HIPCHK(hipMemcpy(&st, e->state, sizeof st, hipMemcpyDeviceToHost));
        if (loss) loss[r] = st.loss;
        if (r + 1 < k) { st.step -= 1; HIPCHK(hipMemcpy(e->state, &st, sizeof st, hipMemcpyHostToDevice)); }   // k_td counts train steps: the k halves are ONE step
    }
    if (!rc) {
        if (e->hp.use_graph) { if (!e->g_post && capture(e, false, PH_POST, &e->g_post)) rc = -1; else HIPCHK(hipGraphLaunch(e->g_post, e->stream)); }
        else enqueue_step(e, false, PH_POST);
    }
    if (!rc && e->hp.prioritized_replay)
        for (int r = 0; r + 1 < k; r++) launch_update_priorities(e->stream, B, e->cap2, s_idx + (size_t)r * B, s_td + (size_t)r * B, e->hp.prio_eps, e->hp.prio_alpha, e->tree, e->state, 0, 1.0, 1.0, nullptr, 0);
    if (!rc && td_out) HIPCHK(hipMemcpyAsync(td_out, s_td, (size_t)k * B * 4, hipMemcpyDeviceToHost, e->stream));
    if (!rc) rc = fetch_scalars(e, nullptr, grad_norm);
    hipStreamSynchronize(e->stream);
    return rc;
}
extern "C" int dqn_train_steps(dqn_engine_t* e, int n, float* loss, float* grad_norm) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return drqn_train_steps(e, n, loss, grad_norm);
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (build_program(e)) return -1;
    // between the steps of this call nothing else touches the replay: step i's Adam launch gathers step i+1's batch (PreGather)
    // Middle steps (no gather launch, the Adam launch gathering) also exist as ONE graph of MID_GROUP consecutive steps: a graph launch costs ~2 us
    // of stream time on top of its kernels (eager launches of the same step: 148.8 vs 150.9 us/step), which the group amortises.
    const int MID_GROUP = e->mid_group; const bool pg = e->pg_ok;
    const bool single = e->world <= 1 && !(e->comm && e->force_comm) && e->hp.use_graph && !e->profiling;
    if (single) {
        // every graph this call can need is captured up front (capturing executes nothing): a short timed call -- the driver's 20 steps -- must
        // not pay an instantiate in the middle
        auto cap1 = [&](bool tp, bool pgth, hipGraphExec_t* g, int rep) { if (*g) return 0; e->step_take_pre = tp; e->step_pregather = pgth; const int rc = capture(e, true, PH_ALL, g, rep);
                                                                         e->step_take_pre = e->step_pregather = false; return rc; };
        if (pg) { if (cap1(false, true, &e->g_pgv[0][1], 1) || cap1(true, true, &e->g_pgv[1][1], 1) || cap1(true, false, &e->g_pgv[1][0], 1)) return -1; }
        if (pg && mailbox_ok(e) && !e->g_pgv_pub) { e->step_publish = true; const int rc = cap1(true, false, &e->g_pgv_pub, 1); e->step_publish = false; if (rc) return -1; }      // last step + publish
        if (MID_GROUP > 1 && cap1(pg, pg, &e->g_mid, MID_GROUP)) return -1;
        // (ADVICE r03 suggested capturing the grouped graphs only in calls long enough to use them.  Measured, r04: the driver's `--steps 20 --warmup 5` then pays the capture
        // + instantiate of both grouped graphs INSIDE its 3 ms timed region -- 7100 -> 5984 steps/s.  They are captured by the first call, whatever its length: ~0.6 ms, once.)
        if (e->mid_big > MID_GROUP && cap1(pg, pg, &e->g_mid_big, e->mid_big)) return -1;
    }
    for (int i = 0; i < n;) {
        // a run of identical steps: middle steps (pipelined gather) or, where that does not apply, any steps
        const int BIG = e->mid_big;
        // (the FIRST launch of a graph exec costs more than the later ones, uploaded or not, and the more the bigger the graph -- tools/r05_run_y2.sh: a 20-step call that is the
        //  first to use the 16-step group reads 129.5 us/step, with the 4-step group only 128.1.  So a short call stays with the small group until some long call -- which
        //  amortises the ~28 us -- has launched the big one)
        if (single && BIG > MID_GROUP && e->g_mid_big && (e->mid_big_warm || n >= 4 * BIG) && (pg ? (i >= 1 && i + BIG <= n - 1) : (i + BIG <= n))) {
            HIPCHK(hipGraphLaunch(e->g_mid_big, e->stream)); e->mid_big_warm = true;
            i += BIG; continue;
        }
        if (single && MID_GROUP > 1 && e->g_mid && (pg ? (i >= 1 && i + MID_GROUP <= n - 1) : (i + MID_GROUP <= n))) {
            HIPCHK(hipGraphLaunch(e->g_mid, e->stream));
            i += MID_GROUP; continue;
        }
        // the last step of a call that returns scalars publishes them from its own last launch into the host mailbox (no fold launch / D2H copy / stream synchronize)
        const unsigned long long pub0 = e->pub_issued;
        e->step_publish = (loss || grad_norm) && i + 1 == n && mailbox_ok(e);
        if (e->step_publish && mail_reserve(e)) return -1;
        if (run_step(e, true, pg && i > 0, pg && i + 1 < n)) return -1;
        i++;
        if (i == n && e->pub_issued != pub0) return wait_mail(e, e->pub_issued, true, loss, grad_norm, nullptr);
    }
    if (loss || grad_norm) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
extern "C" int dqn_get_last_q(dqn_engine_t* e, float* qs, float* qsp, float* qt, int32_t* best, float* y) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    const size_t n = (size_t)e->B * e->nA * 4;
    if (qs) HIPCHK(hipMemcpy(qs, e->q_on_s, n, hipMemcpyDeviceToHost));
    if (qsp) HIPCHK(hipMemcpy(qsp, e->q_on_sp, n, hipMemcpyDeviceToHost));
    if (qt) HIPCHK(hipMemcpy(qt, e->q_tg_sp, n, hipMemcpyDeviceToHost));
    if (best) HIPCHK(hipMemcpy(best, e->best, (size_t)e->B * 4, hipMemcpyDeviceToHost));
    if (y) HIPCHK(hipMemcpy(y, e->ytarget, (size_t)e->B * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int dqn_get_last_indices(dqn_engine_t* e, int64_t* idx) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(idx, e->idx, (size_t)e->B * 8, hipMemcpyDeviceToHost)); return 0;
}

// ---------------------------------------------------------------- policy (src/policy.jl:38-64)
int policy_ws(dqn_engine* e, int n) {
    if (n <= e->pol_n) return 0;
    HIPCHK(hipStreamSynchronize(e->stream)); free_policy_ws(e); drop_act(e, e->act); drop_act(e, e->evalp);
    size_t need = 1;   // split-K partials of the widest forward at n columns
    for (int i = 0; i < e->nl; i++) { const size_t sf = dqn_nchunks(e->L[i].K, e->L[i].fwd_kc); if (sf > 1) need = std::max(need, sf * (size_t)e->L[i].out_feat * n); }
    if (need > e->partials_elems) { drop_graphs(e); hipFree(e->partials); e->partials = nullptr; DM(e->partials, 2 * need); e->partials_elems = need; }
    DM(e->pol_obs, (size_t)n * e->E); DM(e->pol_x, (size_t)n * e->E); DM(e->pol_q, (size_t)n * e->nA); DM(e->pol_a, n);
    for (int i = 0; i < e->nl; i++) DM(e->pol_act[i], (size_t)e->L[i].out_feat * n);
    e->pol_n = n; return 0;
}
int policy_state(dqn_engine* e, int n, bool force_reset) {
    // Recur state of the policy network: one (h, c) column per observation stream; reset = state0 of the ONLINE net (policy.jl:32-34)
    if (!e->hp.recurrence) return 0;
    if (n != e->pol_state_n) {
        for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
            for (int k = 0; k < 2; k++) { hipFree(e->pol_h[i][k]); hipFree(e->pol_c[i][k]); e->pol_h[i][k] = e->pol_c[i][k] = nullptr; DM(e->pol_h[i][k], (size_t)e->L[i].H * n); DM(e->pol_c[i][k], (size_t)e->L[i].H * n); }
            hipFree(e->pol_gx[i]); e->pol_gx[i] = nullptr; DM(e->pol_gx[i], (size_t)e->L[i].N * n);
        }
        e->pol_state_n = n; force_reset = true;
    }
    if (force_reset) {
        for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
            launch_bcast_state(e->stream, e->p_on + e->L[i].h0_off, e->L[i].H, n, e->pol_h[i][e->pol_flip]);
            launch_bcast_state(e->stream, e->p_on + e->L[i].c0_off, e->L[i].H, n, e->pol_c[i][e->pol_flip]);
        }
    }
    return 0;
}
static int policy_forward(dqn_engine* e, int which, const float* obs, int n) {
    if (n < 1) return fail("n must be >= 1");
    HIPCHK(hipSetDevice(e->device));
    if (policy_ws(e, n)) return -1;
    if (policy_state(e, n, false)) return -1;
    if (obs) {      // obs == nullptr: pol_x was filled on the device (vectorised envs)
        HIPCHK(hipMemcpyAsync(e->pol_obs, obs, (size_t)n * e->E * 4, hipMemcpyHostToDevice, e->stream));
        launch_transpose_obs(e->stream, e->pol_obs, e->E, n, e->pol_x);
    }
    const float* P = which == DQN_NET_TARGET ? e->p_tg : e->p_on;
    // the policy workspace has leading dimension n (not pol_n): layers are dense in the batch column
    const int fl = e->pol_flip;
    for (int i = 0; i < e->nl; i++) {
        const LayerDev& l = e->L[i]; const float* X = l.src < 0 ? e->pol_x : e->pol_act[l.src];
        if (l.kind == DQN_LAYER_LSTM) {      // one Recur step: Gx = Wi*x (bias-free view), then the cell with the carried (h, c)
            LayerDev V = l; V.kind = DQN_LAYER_DENSE; V.out_feat = l.N; V.b_off = l.z_off; V.act = DQN_ACT_IDENTITY;
            fwd_layer(e, V, P, X, n, 0, n, e->pol_gx[i], "policy_fwd");
            LstmStepArgs a; memset(&a, 0, sizeof a); a.H = l.H; a.B = n; a.nseq = 1;
            LstmSeq& q = a.s[0]; q.Gx = e->pol_gx[i]; q.Hout = e->pol_act[i]; q.Cst = e->pol_c[i][fl ^ 1]; q.ld = n; q.c0 = 0; q.Wh = P + l.wh_off; q.bias = P + l.b_off;
            q.hprev = e->pol_h[i][fl]; q.hp_ld = n; q.hp_bs = 1; q.cprev = e->pol_c[i][fl]; q.cp_ld = n; q.cp_bs = 1;
            launch_lstm_step_t(e->stream, a, 0);
            HIPCHK(hipMemcpyAsync(e->pol_h[i][fl ^ 1], e->pol_act[i], (size_t)l.H * n * 4, hipMemcpyDeviceToDevice, e->stream));
        } else fwd_layer(e, l, P, X, n, 0, n, e->pol_act[i], "policy_fwd");
    }
    if (e->hp.recurrence) e->pol_flip ^= 1;
    const int lq = e->hp.dueling ? e->last_adv : e->last_base;
    launch_q_columns(e->stream, n, e->nA, e->hp.dueling, e->hp.dueling ? e->pol_act[e->last_val] : nullptr, e->pol_act[lq], e->pol_q, e->pol_a);
    return 0;
}
extern "C" int dqn_forward(dqn_engine_t* e, int which, const float* obs, int n, float* q_out) { if (!e) return fail("null engine handle");
    if (!obs) return fail("obs is null");
    if (policy_forward(e, which, obs, n)) return -1;
    HIPCHK(hipMemcpyAsync(q_out, e->pol_q, (size_t)n * e->nA * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
extern "C" int dqn_greedy_action(dqn_engine_t* e, const float* obs, int n, int32_t* a_out) { if (!e) return fail("null engine handle");
    if (!obs) return fail("obs is null");
    if (policy_forward(e, DQN_NET_ONLINE, obs, n)) return -1;
    HIPCHK(hipMemcpyAsync(a_out, e->pol_a, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}

// ---------------------------------------------------------------- data-parallel replicas
extern "C" int dqn_comm_unique_id(void* id128) { if (rccl_load()) return -1; const int rc = g_rccl.GetUniqueId(id128); return rc ? fail("ncclGetUniqueId failed (%d)", rc) : 0; }
extern "C" int dqn_comm_init(dqn_engine_t* e, const void* id128, int rank, int world) { if (!e) return fail("null engine handle");
    if (rccl_load()) return -1;
    HIPCHK(hipSetDevice(e->device));
    // Recurrent engines on the fused column-parallel step (plan dw_kc = -cg, drqn_cols.hip): that step has no point at which a gradient could be exchanged.  A DEFAULTED
    // plan is recomputed without the column-group rule (the multi-launch recurrent program, all-reduce between backward and Adam); a plan the CALLER wrote is refused here,
    // with the reason, instead of at the first train step.
    bool cgp = false; for (int i = 0; i < e->nl; i++) cgp = cgp || e->L[i].dw_kc < 0;
    if (cgp && !e->plan_defaulted)
        return fail("dqn_comm_init: this engine was created with a column-group dW plan (dw_kc < 0: the fused single-device recurrent step); replicas need dw_kc >= 0 -- "
                    "create the engine with plan = NULL (the default plan is then re-derived for replicas here) or with contiguous dW chunks");
    Id128 id; memcpy(id.b, id128, 128);
    const int rc = g_rccl.CommInitRank(&e->comm, world, id, rank);
    if (rc) return fail("ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    // (the communicator first: a failed ncclCommInitRank leaves plan, partials, program and graphs as they were)
    read_opts(e->opt, /*comm_only*/ true);      // the communicator switches, as of this call
    e->rank = rank; e->world = world; e->sim_world = 0; e->force_comm = e->opt.force_allreduce != 0; e->dp_one_state = 0;
    // the launch program depends on the exchange mode and the world size: rebuild it on the next step
    drop_graphs(e); HIPCHK(hipStreamSynchronize(e->stream));
    for (void* p : e->prog_allocs) hipFree(p);
    e->prog_allocs.clear(); e->prog.clear(); e->prog_built = false; e->prog_post_begin = 0; e->final_reduce_step = -1; memset(&e->adam_segs, 0, sizeof e->adam_segs);
    hipFree(e->dp_send); hipFree(e->dp_recv); e->dp_send = e->dp_recv = nullptr; e->dp_gather = e->dp_pack_folds = e->dp_adam_folds = false; e->dp_count = 0;
    if (cgp) {
        HIPCHK(hipStreamSynchronize(e->stream));
        dqn_layer_plan defp[DQN_MAX_LAYERS]; default_plan(e->L, e->nl, e->B, defp, &e->hp, /*allow_cg*/ false);
        size_t pmax = e->partials_elems;
        for (int i = 0; i < e->nl; i++) {
            LayerDev& l = e->L[i]; l.fwd_kc = defp[i].fwd_kc; l.dx_kc = defp[i].dx_kc; l.dw_kc = defp[i].dw_kc;
            const size_t sf = dqn_nchunks(l.K, l.fwd_kc); if (sf > 1) pmax = std::max(pmax, sf * (size_t)l.out_feat * e->ncon);
            const size_t sw = dqn_nchunks(l.npos * e->Bc, l.dw_kc); if (sw > 1) pmax = std::max(pmax, sw * (size_t)(l.K + 1) * l.N);
            const size_t sx = l.kind != DQN_LAYER_CONV ? dqn_nchunks(l.N, l.dx_kc) : 1; if (sx > 1) pmax = std::max(pmax, sx * (size_t)l.in_feat * e->Bc);
        }
        HIPCHK(hipMemcpy(e->L_dev, e->L, sizeof(LayerDev) * e->nl, hipMemcpyHostToDevice));
        if (pmax > e->partials_elems) { hipFree(e->partials); e->partials = nullptr; DM(e->partials, 2 * pmax); e->partials_elems = pmax; }
    }
    return 0;
}

// what the COMMUNICATOR itself says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), so that a bench line can prove how many ranks RCCL saw;
// without a communicator: nranks = 0 (sim_world engines report it in `sim_world`).  exchange: 0 none, 1 all-gather of operands + small gradients, 2 all-reduce
extern "C" int dqn_comm_info(dqn_engine_t* e, dqn_comm_info_t* out) { if (!e || !out) return fail("null argument");
    memset(out, 0, sizeof *out); out->rccl_rank = -1; out->rccl_device = -1;
    out->engine_world = e->world; out->engine_rank = e->rank; out->sim_world = e->sim_world; out->dp_overlap = e->dp_overlap ? 1 : 0;
    out->exchange = (e->comm || e->sim_world) ? (e->prog_built ? (e->dp_gather ? 1 : 2) : -1) : 0;      // -1: decided when the step program is built (first train step)
    if (e->comm) {
        if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail("librccl has no ncclCommCount / ncclCommUserRank");
        int rc = g_rccl.CommCount(e->comm, &out->rccl_nranks); if (!rc) rc = g_rccl.CommUserRank(e->comm, &out->rccl_rank);
        if (!rc && g_rccl.CommCuDevice) rc = g_rccl.CommCuDevice(e->comm, &out->rccl_device);
        if (rc) return fail("ncclCommCount / ncclCommUserRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    }
    return 0;
}

extern "C" int dqn_comm_exchange_bytes(dqn_engine_t* e, int64_t* bytes) { if (!e || !bytes) return fail("null argument");
    *bytes = 0;
    if ((e->comm || e->sim_world) && e->prog_built) *bytes = (int64_t)(e->dp_gather ? e->dp_count : e->Pint) * 4;
    return 0;
}

// ---------------------------------------------------------------- misc
extern "C" int dqn_stream_sync(dqn_engine_t* e) { if (!e) return fail("null engine handle"); HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream)); return 0; }
extern "C" int dqn_stream_handle(dqn_engine_t* e, void** s) { if (!e) return fail("null engine handle"); *s = (void*)e->stream; return 0; }
// debug aid: per-workgroup s_memtime records of the forward GEMM kernels (nn_gemm.hip, KTRACE).  out == NULL: start recording (room for 65536
// records); out != NULL: stop and copy counter + records (n 64-bit words) to the host.  Process-wide (one engine at a time).
// (not declared in the public header: a development hook of trace builds, bound by tools/ktrace*.py only)
extern "C" __attribute__((visibility("default"))) int dqn_debug_ktrace(dqn_engine_t* e, uint64_t* out, size_t n) { if (!e) return fail("null engine handle");
    const size_t words = 1 + 8 * 65536ull;
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    if (!out) {
        if (!e->ktrace_buf) HIPCHK(hipMalloc((void**)&e->ktrace_buf, words * 8));
        HIPCHK(hipMemset(e->ktrace_buf, 0, words * 8));
        if (gemm_set_ktrace(e->ktrace_buf)) return fail("this library was built without -DDQN_KTRACE (DQN_EXTRA_DEF=DQN_KTRACE python __graft_entry__.py --force)");
        return 0;
    }
    gemm_set_ktrace(nullptr);
    if (!e->ktrace_buf) return fail("ktrace was not started");
    HIPCHK(hipMemcpy(out, e->ktrace_buf, std::min(n, words) * 8, hipMemcpyDeviceToHost)); return 0;
}
// Holds the stream until the host has enqueued the whole profiled step, so that the HIP events around each kernel time
// the kernel and not the host's launch latency.  Bounded spin (~0.2 s) so a dead host can never hang the GPU.
__global__ void k_gate(volatile int* flag) {
    for (long i = 0; i < 2000000 && *flag == 0; i++) __builtin_amdgcn_s_sleep(32);
}
static int profile_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries, bool steady);
extern "C" int dqn_profile_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries) { return profile_step(e, max_entries, names, ms, n_entries, false); }
// as dqn_profile_step, but the step timed is a MIDDLE step of dqn_train_steps(n) (TWO train steps run; the second is timed): without its
// gather launch and with the next batch's gather inside its Adam launch, when the engine supports that (PreGather) -- else a plain step
extern "C" int dqn_profile_steady_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries) { return profile_step(e, max_entries, names, ms, n_entries, true); }
static int profile_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries, bool steady) { if (!e) return fail("null engine handle");
    HIPCHK(hipSetDevice(e->device));
    if (!e->hp.recurrence && e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (e->hp.recurrence && e->ep_size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    HIPCHK(hipStreamSynchronize(e->stream));
    static int* gate = nullptr;
    if (!gate) HIPCHK(hipHostMalloc((void**)&gate, sizeof(int), hipHostMallocMapped));
    *gate = 0;
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, e->stream, (volatile int*)gate);
    e->profiling = true; e->prof.clear();
    int rc = 0;
    if (e->hp.recurrence) rc = dqn_train_step_drqn(e, nullptr, nullptr, nullptr, nullptr);
    else if (steady && !build_program(e)) {      // a MIDDLE step of dqn_train_steps(n): one un-timed step first (its Adam launch gathers the timed step's batch)
        e->profiling = false; rc = run_step(e, true, false, e->pg_ok); e->profiling = true;
        if (!rc) rc = run_step(e, true, e->pg_ok, e->pg_ok);
        // the timed step's own pre-gather filled the arena for a step that will not come: drop it (2 -> 1: the indices in idx_pre stay valid, the
        // next gather launch overwrites the arena)
        if (!rc && e->pg_ok) HIPCHK(hipMemsetD32Async((hipDeviceptr_t)&e->state->pre_valid, 1, 1, e->stream));      // (a memset node: no host buffer, safe behind the gate kernel)
    }
    else rc = run_step(e, true);
    e->profiling = false;
    __atomic_store_n(gate, 1, __ATOMIC_SEQ_CST);
    if (rc) return -1;
    HIPCHK(hipStreamSynchronize(e->stream));
    int n = 0;
    for (auto& pe : e->prof) {
        float t = 0; hipEventElapsedTime(&t, pe.a, pe.b);
        if (n < max_entries) { names[n] = pe.name; ms[n] = t; n++; }
        hipEventDestroy(pe.a); hipEventDestroy(pe.b);
    }
    e->prof.clear(); *n_entries = n; return 0;
}
