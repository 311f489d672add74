// engine.hip -- host side of libdqn_mi355x.so: one engine per GPU owns replay storage, sum-tree, online/target
// parameters, gradients, Adam state, the batch arena and ONE HIP stream; the train step
// (batch_train!, src/solver.jl:191-236) is enqueued as a fixed kernel sequence and replayed from a hipGraph.
// C ABI: include/dqn_mi355x.h.  No CPU fallback exists: every entry point that computes needs the HIP device.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <string>
#include <vector>

#include "common.h"

static thread_local char g_err[1024] = "";
static int fail(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); return -1;
}
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #x); } while (0)

// ---------------------------------------------------------------- RCCL (dlopen'ed; only for data-parallel replicas)
struct Id128 { char b[128]; };   // ncclUniqueId (passed by value)
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
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
    g_rccl.CommDestroy = (int (*)(void*))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce) return fail("librccl is missing nccl symbols");
    return 0;
}

// ---------------------------------------------------------------- engine
struct ProfEntry { const char* name; hipEvent_t a, b; };

struct dqn_engine {
    int device = 0; hipStream_t stream = nullptr, stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int nl = 0; LayerDev L[DQN_MAX_LAYERS]; LayerDev* L_dev = nullptr;
    dqn_hparams hp; int B = 0, nA = 0, E = 0, ncon = 0;
    int last_base = -1, last_val = -1, last_adv = -1;
    size_t P = 0, Pint = 0;   // external (Flux.params) and internal (16-B aligned arrays) parameter counts
    float *p_on = nullptr, *p_tg = nullptr, *grad = nullptr, *m = nullptr, *v = nullptr, *io_tmp = nullptr;
    StepState* state = nullptr;
    // replay
    long long cap = 0, cap2 = 1, widx = 0, size = 0;
    void *s_rows = nullptr, *sp_rows = nullptr; int* ra = nullptr; float* rr = nullptr; unsigned char* rdone = nullptr; float* tree = nullptr;
    static const int ADD_CHUNK = 1024;
    int* st_a = nullptr; float* st_r = nullptr; unsigned char* st_done = nullptr; float* st_td = nullptr;
    // step workspace
    long long* idx = nullptr; float* x0 = nullptr;
    float *act_on[DQN_MAX_LAYERS] = {}, *act_tg[DQN_MAX_LAYERS] = {}, *dact[DQN_MAX_LAYERS] = {};
    float *join_tmp = nullptr, *partials = nullptr, *gmax_part = nullptr; size_t partials_elems = 0;
    float *w_is = nullptr, *td = nullptr, *q_on_s = nullptr, *q_on_sp = nullptr, *q_tg_sp = nullptr, *ytarget = nullptr; int* best = nullptr;
    // get_batch seam workspace
    float *gb_rows = nullptr, *gb_r = nullptr, *gb_done = nullptr, *gb_w = nullptr; int* gb_a = nullptr; long long* gb_idx = nullptr;
    // policy workspace
    EnvDev env{}; bool has_envs = false; unsigned char* env_images = nullptr;
    int pol_n = 0; float *pol_obs = nullptr, *pol_x = nullptr, *pol_act[DQN_MAX_LAYERS] = {}, *pol_q = nullptr; int* pol_a = nullptr;
    // graphs: [0] = step with sampling, [1] = step on given indices; with a communicator the step is cut in two
    hipGraphExec_t g_full[2] = {nullptr, nullptr}, g_pre[2] = {nullptr, nullptr}, g_post = nullptr;
    // comm
    void* comm = nullptr; int rank = 0, world = 1; bool force_comm = false;   // force_comm: run the all-reduce path even at world == 1 (tests)
    // DRQN (recurrence = true): column count per sequence set Bc = T*B (B otherwise); EpisodeReplayBuffer storage; LSTM workspaces
    int Bc = 0, T = 1; long long ep_cap = 0, ep_size = 0, ep_widx = 0, ep_cur_len = 0; std::vector<int> ep_len_host; std::vector<int64_t> ep_perm;
    float *ep_s = nullptr, *ep_sp = nullptr, *ep_r = nullptr; int* ep_a = nullptr; unsigned char* ep_done = nullptr; int* ep_len = nullptr;
    long long* ep_idx = nullptr; int* ep_start = nullptr; int* r_a = nullptr; float *r_r = nullptr, *r_done = nullptr, *r_mask = nullptr;
    float *gx_on[DQN_MAX_LAYERS] = {}, *gx_tg[DQN_MAX_LAYERS] = {}, *cst_on[DQN_MAX_LAYERS] = {}, *cst_tg[DQN_MAX_LAYERS] = {}, *gates[DQN_MAX_LAYERS] = {}, *tcb[DQN_MAX_LAYERS] = {},
          *hprev_buf[DQN_MAX_LAYERS] = {}, *cprev_buf[DQN_MAX_LAYERS] = {}, *dG[DQN_MAX_LAYERS] = {}, *dhn[DQN_MAX_LAYERS] = {}, *dcn[DQN_MAX_LAYERS] = {};
    float *pol_h[DQN_MAX_LAYERS][2] = {}, *pol_c[DQN_MAX_LAYERS][2] = {}, *pol_gx[DQN_MAX_LAYERS] = {}; int pol_flip = 0, pol_state_n = 0; uint64_t drqn_draws = 0;
    hipGraphExec_t g_drqn = nullptr;
    // static launch program
    struct Step { const char* name; std::function<void(dqn_engine*)> fn; };
    // acting programs (forward on n columns + env kernels), one for the training envs and one for the evaluation envs
    struct ActProg { std::vector<Step> steps; int n = 0; hipGraphExec_t graph = nullptr; std::vector<void*> allocs; };
    ActProg act, evalp; std::vector<Step>* sink = nullptr; std::vector<void*>* alloc_sink = nullptr; RolloutDev *roll = nullptr, *eval_roll = nullptr;
    EnvDev eval_env{}; int eval_n = 0;
    std::vector<Step> prog; size_t prog_post_begin = 0; bool prog_built = false, step_sampled = true;
    AdamSegs adam_segs; long final_reduce_step = -1;   // deferred dW slabs: reduced inside k_adam unless a communicator needs the materialised gradient
    std::vector<void*> prog_allocs; std::vector<std::string> prog_names;
    // profiling
    bool profiling = false; std::vector<ProfEntry> prof;
};

static void prof_begin(dqn_engine* e, const char* name) {
    if (!e->profiling) return;
    ProfEntry pe; pe.name = name; hipEventCreate(&pe.a); hipEventCreate(&pe.b); hipEventRecord(pe.a, e->stream); e->prof.push_back(pe);
}
static void prof_end(dqn_engine* e) { if (e->profiling) hipEventRecord(e->prof.back().b, e->stream); }
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
static void default_plan(const LayerDev* L, int n, int B, dqn_layer_plan* out) {
    for (int i = 0; i < n; i++) {
        out[i].fwd_kc = 0;
        if (L[i].K > 1024) { const int s = (L[i].K + 511) / 512; int kc = (L[i].K + s - 1) / s; kc = (kc + 3) / 4 * 4; out[i].fwd_kc = kc; }
        else if (L[i].kind == DQN_LAYER_DENSE && L[i].N < 16 && L[i].K >= 128) out[i].fwd_kc = 32;   // heads: 16+ short chains instead of one long one
        out[i].dx_kc = (L[i].kind != DQN_LAYER_CONV && L[i].N > 512) ? 256 : 0;
        out[i].dw_kc = 0;
        if (L[i].kind == DQN_LAYER_CONV) {   // positions per chunk so that (K/64 row tiles) x chunks >= ~512 workgroups
            const int st = (512 + (L[i].K + 63) / 64 - 1) / ((L[i].K + 63) / 64); int ppc = L[i].npos / st; if (ppc < 1) ppc = 1; out[i].dw_kc = ppc * B;
        }
    }
}
extern "C" int dqn_plan_default(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, dqn_layer_plan* plan_out) {
    LayerDev L[DQN_MAX_LAYERS]; int lb, lv, la; size_t P, Pi;
    if (build_layers(layers, n_layers, hp, L, &lb, &lv, &la, &P, &Pi)) return -1;
    default_plan(L, n_layers, hp->batch_size, plan_out); return 0;
}

template <class T> static int dmalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}
#define DM(p, n) do { if (dmalloc(&(p), (n))) return -1; } while (0)

extern "C" int dqn_engine_destroy(dqn_engine_t* e);

extern "C" int dqn_engine_create(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp, const dqn_layer_plan* plan, int device,
                                 dqn_engine_t** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail("no HIP device: libdqn_mi355x has no CPU fallback (hipGetDeviceCount found %d devices)", ndev);
    if (device < 0 || device >= ndev) return fail("device %d out of range (0..%d)", device, ndev - 1);
    if (hp->batch_size < 1 || hp->batch_size > 1024) return fail("batch_size %d unsupported (1..1024)", hp->batch_size);
    if (hp->buffer_size < hp->batch_size) return fail("AssertionError: r.max_size >= r.batch_size");   // ...replay.jl:84
    dqn_engine* e = new dqn_engine();
    e->device = device; e->hp = *hp; e->B = hp->batch_size; e->nA = hp->n_actions; e->E = hp->obs_c * hp->obs_h * hp->obs_w;
    if (build_layers(layers, n_layers, hp, e->L, &e->last_base, &e->last_val, &e->last_adv, &e->P, &e->Pint)) { delete e; return -1; }
    e->nl = n_layers;
    dqn_layer_plan defp[DQN_MAX_LAYERS];
    if (!plan) { default_plan(e->L, e->nl, e->B, defp); plan = defp; }
    for (int i = 0; i < e->nl; i++) {
        e->L[i].fwd_kc = plan[i].fwd_kc; e->L[i].dx_kc = plan[i].dx_kc; e->L[i].dw_kc = plan[i].dw_kc;
        if (e->L[i].kind == DQN_LAYER_CONV && e->L[i].dw_kc > 0 && e->L[i].dw_kc % e->B) { delete e; return fail("plan: conv dw_kc must be a multiple of batch_size"); }
    }
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&e->stream2, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    const int B = e->B;
    e->T = hp->recurrence ? hp->trace_length : 1;
    if (hp->recurrence && (e->T < 1 || (long long)e->T * B > 65536)) return fail("trace_length %d unsupported", e->T);
    const int Bc = e->Bc = e->T * B;            // columns per sequence set: B, or T*B time-major columns for DRQN
    e->ncon = hp->double_q ? 2 * Bc : Bc;
    DM(e->L_dev, e->nl); HIPCHK(hipMemcpy(e->L_dev, e->L, sizeof(LayerDev) * e->nl, hipMemcpyHostToDevice));
    DM(e->p_on, e->Pint); DM(e->p_tg, e->Pint); DM(e->grad, e->Pint); DM(e->m, e->Pint); DM(e->v, e->Pint); DM(e->io_tmp, e->P);
    HIPCHK(hipMemset(e->p_on, 0, e->Pint * 4)); HIPCHK(hipMemset(e->p_tg, 0, e->Pint * 4)); HIPCHK(hipMemset(e->grad, 0, e->Pint * 4));
    HIPCHK(hipMemset(e->m, 0, e->Pint * 4)); HIPCHK(hipMemset(e->v, 0, e->Pint * 4));
    DM(e->state, 1);
    StepState s0; memset(&s0, 0, sizeof s0); s0.bp[0][0] = s0.bp[1][0] = hp->adam_beta1; s0.bp[0][1] = s0.bp[1][1] = hp->adam_beta2;
    HIPCHK(hipMemcpy(e->state, &s0, sizeof s0, hipMemcpyHostToDevice));
    e->cap = hp->recurrence ? B : hp->buffer_size; while (e->cap2 < e->cap) e->cap2 <<= 1;   // DRQN keeps episodes instead (below)
    const size_t osz = hp->obs_dtype == DQN_OBS_U8 ? 1 : 4;
    { unsigned char *a = nullptr, *b = nullptr; DM(a, (size_t)e->cap * e->E * osz); DM(b, (size_t)e->cap * e->E * osz); e->s_rows = a; e->sp_rows = b; }
    DM(e->ra, e->cap); DM(e->rr, e->cap); DM(e->rdone, e->cap); DM(e->tree, 2 * (size_t)e->cap2);
    HIPCHK(hipMemset(e->tree, 0, 2 * (size_t)e->cap2 * 4));
    DM(e->st_a, dqn_engine::ADD_CHUNK); DM(e->st_r, dqn_engine::ADD_CHUNK); DM(e->st_done, dqn_engine::ADD_CHUNK); DM(e->st_td, dqn_engine::ADD_CHUNK);
    DM(e->idx, B); HIPCHK(hipMemset(e->idx, 0, B * 8)); DM(e->x0, (size_t)e->E * 2 * Bc);
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
    DM(e->join_tmp, jmax); DM(e->gmax_part, adam_blocks(e->Pint) + 4096); HIPCHK(hipMemset(e->gmax_part, 0, (adam_blocks(e->Pint) + 4096) * 4));
    DM(e->w_is, B); DM(e->td, Bc); DM(e->q_on_s, (size_t)B * e->nA); DM(e->q_on_sp, (size_t)B * e->nA); DM(e->q_tg_sp, (size_t)B * e->nA);
    DM(e->ytarget, B); DM(e->best, B);
    DM(e->gb_rows, (size_t)B * e->E); DM(e->gb_r, B); DM(e->gb_done, B); DM(e->gb_w, B); DM(e->gb_a, B); DM(e->gb_idx, B);
    HIPCHK(hipStreamSynchronize(e->stream));
    *out = e; return 0;
}

static void drop_graphs(dqn_engine* e) {
    for (int i = 0; i < 2; i++) {
        if (e->g_full[i]) { hipGraphExecDestroy(e->g_full[i]); e->g_full[i] = nullptr; }
        if (e->g_pre[i]) { hipGraphExecDestroy(e->g_pre[i]); e->g_pre[i] = nullptr; }
    }
    if (e->g_post) { hipGraphExecDestroy(e->g_post); e->g_post = nullptr; }
    for (dqn_engine::ActProg* a : {&e->act, &e->evalp}) if (a->graph) { hipGraphExecDestroy(a->graph); a->graph = nullptr; }
}
static void free_envs(dqn_engine* e);
static void drop_act(dqn_engine* e, dqn_engine::ActProg& a) {
    if (a.graph) { hipGraphExecDestroy(a.graph); a.graph = nullptr; }
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
    hipFree(e->L_dev); hipFree(e->p_on); hipFree(e->p_tg); hipFree(e->grad); hipFree(e->m); hipFree(e->v); hipFree(e->io_tmp); hipFree(e->state);
    hipFree(e->s_rows); hipFree(e->sp_rows); hipFree(e->ra); hipFree(e->rr); hipFree(e->rdone); hipFree(e->tree);
    hipFree(e->st_a); hipFree(e->st_r); hipFree(e->st_done); hipFree(e->st_td); hipFree(e->idx); hipFree(e->x0);
    for (int i = 0; i < e->nl; i++) { hipFree(e->act_on[i]); hipFree(e->act_tg[i]); hipFree(e->dact[i]); }
    hipFree(e->join_tmp); hipFree(e->partials); hipFree(e->gmax_part); hipFree(e->w_is); hipFree(e->td); hipFree(e->q_on_s); hipFree(e->q_on_sp); hipFree(e->q_tg_sp);
    hipFree(e->ytarget); hipFree(e->best); hipFree(e->gb_rows); hipFree(e->gb_r); hipFree(e->gb_done); hipFree(e->gb_w); hipFree(e->gb_a); hipFree(e->gb_idx);
    free_policy_ws(e); free_envs(e);
    for (void* p : e->prog_allocs) hipFree(p);
    hipFree(e->ep_s); hipFree(e->ep_sp); hipFree(e->ep_a); hipFree(e->ep_r); hipFree(e->ep_done); hipFree(e->ep_len); hipFree(e->ep_idx); hipFree(e->ep_start);
    hipFree(e->r_a); hipFree(e->r_r); hipFree(e->r_done); hipFree(e->r_mask);
    for (int i = 0; i < e->nl; i++) { hipFree(e->gx_on[i]); hipFree(e->gx_tg[i]); hipFree(e->cst_on[i]); hipFree(e->cst_tg[i]); hipFree(e->gates[i]); hipFree(e->tcb[i]); hipFree(e->hprev_buf[i]);
        hipFree(e->cprev_buf[i]); hipFree(e->dG[i]); hipFree(e->dhn[i]); hipFree(e->dcn[i]); for (int k = 0; k < 2; k++) { hipFree(e->pol_h[i][k]); hipFree(e->pol_c[i][k]); } hipFree(e->pol_gx[i]); }
    if (e->g_drqn) hipGraphExecDestroy(e->g_drqn);
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->stream2) hipStreamDestroy(e->stream2);
    if (e->ev_fork) hipEventDestroy(e->ev_fork);
    if (e->ev_join) hipEventDestroy(e->ev_join);
    delete e; return 0;
}
extern "C" int dqn_engine_get_plan(dqn_engine_t* e, dqn_layer_plan* p) {
    for (int i = 0; i < e->nl; i++) { p[i].fwd_kc = e->L[i].fwd_kc; p[i].dx_kc = e->L[i].dx_kc; p[i].dw_kc = e->L[i].dw_kc; } return 0;
}
extern "C" int dqn_n_params(dqn_engine_t* e, size_t* n) { *n = e->P; return 0; }

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
extern "C" int dqn_set_params(dqn_engine_t* e, int which, const float* flat, size_t n) {
    if (n != e->P) return fail("set_params: got %zu values, the network has %zu parameters", n, e->P);
    return put_vec(e, flat, which == DQN_NET_TARGET ? e->p_tg : e->p_on);
}
extern "C" int dqn_get_params(dqn_engine_t* e, int which, float* flat, size_t n) {
    if (n != e->P) return fail("get_params: size mismatch (%zu vs %zu)", n, e->P);
    return get_vec(e, which == DQN_NET_TARGET ? e->p_tg : e->p_on, flat);
}
extern "C" int dqn_get_grads(dqn_engine_t* e, float* flat, size_t n) {
    if (n != e->P) return fail("get_grads: size mismatch"); return get_vec(e, e->grad, flat);
}
extern "C" int dqn_sync_target(dqn_engine_t* e) {   // Flux.loadparams!(target_q, params(active_q)), src/solver.jl:142-145
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipMemcpyAsync(e->p_tg, e->p_on, e->Pint * 4, hipMemcpyDeviceToDevice, e->stream)); return 0;
}
extern "C" int dqn_get_adam_state(dqn_engine_t* e, float* m, float* v, double* bp, size_t n) {
    if (n != e->P) return fail("size mismatch");
    if (m && get_vec(e, e->m, m)) return -1;
    if (v && get_vec(e, e->v, v)) return -1;
    if (bp) { StepState s; HIPCHK(hipStreamSynchronize(e->stream)); HIPCHK(hipMemcpy(&s, e->state, sizeof s, hipMemcpyDeviceToHost)); const int sl = (int)((s.step + 1) & 1); bp[0] = s.bp[sl][0]; bp[1] = s.bp[sl][1]; }
    return 0;
}
extern "C" int dqn_set_adam_state(dqn_engine_t* e, const float* m, const float* v, const double* bp, size_t n) {
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
                              const float* td_err, int n) {
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
extern "C" int dqn_replay_size(dqn_engine_t* e, int64_t* cur, int64_t* cap) { if (cur) *cur = e->size; if (cap) *cap = e->cap; return 0; }
extern "C" int dqn_replay_get_priorities(dqn_engine_t* e, float* prio, int64_t n) {
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(prio, e->tree + e->cap2, (size_t)n * 4, hipMemcpyDeviceToHost)); return 0;
}
static int check_idx(dqn_engine* e, const int64_t* idx, int n) {
    for (int i = 0; i < n; i++) if (idx[i] < 0 || idx[i] >= e->size) return fail("BoundsError: index %lld outside 0..%lld", (long long)idx[i], (long long)e->size - 1);
    return 0;
}
extern "C" int dqn_replay_sample(dqn_engine_t* e, int64_t* idx_out) {
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");   // ...replay.jl:83
    HIPCHK(hipSetDevice(e->device));
    launch_sample(e->stream, e->B, e->cap2, e->tree, e->hp.seed, e->idx, e->state, 1);
    if (idx_out) { HIPCHK(hipMemcpyAsync(idx_out, e->idx, (size_t)e->B * 8, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); }
    return 0;
}
extern "C" int dqn_replay_get_batch(dqn_engine_t* e, const int64_t* idx, float* s, int32_t* a, float* r, float* sp, float* done, float* w) {
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
extern "C" int dqn_update_priorities(dqn_engine_t* e, const int64_t* idx, const float* td, int n) {
    if (check_idx(e, idx, n)) return -1;
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
static void fwd_layer(dqn_engine* e, const LayerDev& l, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, const char* name) {
    prof_begin(e, name);
    if (!(e->hp.use_mfma && launch_mfma_fwd(e->stream, l, P, X, ldx, col0, ncols, Y, e->partials)))
        launch_valu_fwd(e->stream, l, P, X, ldx, col0, ncols, Y, e->partials);
    prof_end(e);
}
enum { PH_ALL = 0, PH_PRE = 1, PH_POST = 2 };

// ---------------------------------------------------------------- static launch program
// Every pointer, shape and plan is fixed at engine creation, so the train step is compiled ONCE into a list of launches
// (closures) and merely replayed (and captured into a hipGraph).  Small independent kernels are batched: one k_valu_multi
// launch per network level, one k_reduce_multi per level, head reductions folded into k_td.
template <class T> static T* upload(dqn_engine* e, const std::vector<T>& v) {
    T* d = nullptr; hipMalloc((void**)&d, sizeof(T) * v.size()); hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
    (e->alloc_sink ? *e->alloc_sink : e->prog_allocs).push_back(d); return d;
}
static float* palloc(dqn_engine* e, size_t n) { float* d = nullptr; hipMalloc((void**)&d, n * 4); (e->alloc_sink ? *e->alloc_sink : e->prog_allocs).push_back(d); return d; }
static bool same_geo(const LayerDev& a, const LayerDev& b) {
    return a.kind == b.kind && a.act == b.act && a.K == b.K && a.N == b.N && a.npos == b.npos && a.cin == b.cin && a.kh == b.kh && a.kw == b.kw &&
           a.sh == b.sh && a.sw == b.sw && a.ih == b.ih && a.iw == b.iw && a.fwd_kc == b.fwd_kc && a.src == b.src;
}
static void add_valu(dqn_engine* e, std::vector<VTask>& pend, const VTask& t) { pend.push_back(t); }
static void flush_valu(dqn_engine* e, std::vector<VTask>& pend, const char* name) {
    if (pend.empty()) return;
    unsigned blocks = 0;
    for (auto& t : pend) { t.first_block = blocks; blocks += valu_task_blocks(t); }
    VTask* dev = upload(e, pend); const int n = (int)pend.size();
    (e->sink ? *e->sink : e->prog).push_back({name, [=](dqn_engine* en) { launch_valu_multi(en->stream, dev, n, blocks); }});
    pend.clear();
}
static void emit_reduce(dqn_engine* e, std::vector<RSeg>& segs, const char* name) {
    if (segs.empty()) return;
    unsigned blocks = 0;
    for (auto& r : segs) { r.first_block = blocks; blocks += (unsigned)((r.elems + 255) / 256); }
    RSeg* dev = upload(e, segs); const int n = (int)segs.size();
    (e->sink ? *e->sink : e->prog).push_back({name, [=](dqn_engine* en) { launch_reduce_multi(en->stream, dev, n, blocks); }});
    segs.clear();
}
static const char* pname(dqn_engine* e, const char* op, int kind, int i) {
    char b[32]; snprintf(b, sizeof b, "%s_%s%d", op, kind == DQN_LAYER_CONV ? "conv" : "dense", i); e->prog_names.push_back(b); return e->prog_names.back().c_str();
}
static int build_program(dqn_engine* e) {
    if (e->prog_built) return 0;
    HIPCHK(hipSetDevice(e->device));
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
    HeadSrc head[DQN_MAX_LAYERS][2];   // per (layer, net): where k_td finds the layer's output
    // ---------------- forward: online net on [s ; sp] (src/solver.jl:210,220), target net on sp (:211)
    for (size_t li = 0; li < levels.size(); li++) {
        // the head layers' split-K slabs are reduced inside the single-workgroup TD kernel only while that is cheaper than a reduce
        // launch (small batches); at B = 512 the 7680 head values x 16 slabs belong on many workgroups
        const auto& lv = levels[li]; const bool last = li + 1 == levels.size() && !rec && e->B <= 64;
        struct Prob { int l, net; const float *P, *X; int ldx, col0, ncols; float *Y, *part; int S; };
        std::vector<Prob> pr;
        for (int l : lv) for (int net = 0; net < 2; net++) {
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
            struct A { const float *W[4], *bias[4], *X[4]; int ldx[4], col0[4], ncols[4]; float* out[4]; } a;
            for (int i = 0; i < n; i++) { const Prob& q = pr[ids[i]]; const LayerDev& Lq = LV[q.l]; a.W[i] = q.P + Lq.w_off; a.bias[i] = q.P + Lq.b_off; a.X[i] = q.X; a.ldx[i] = q.ldx; a.col0[i] = q.col0; a.ncols[i] = q.ncols; a.out[i] = q.S > 1 ? q.part : q.Y; }
            e->prog.push_back({name, [=](dqn_engine* en) { launch_gemm_fwd(en->stream, L, n, a.W, a.bias, a.X, a.ldx, a.col0, a.ncols, a.out); }});
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
            if (q.S > 1) {
                if (last) { h.p = q.part; h.S = q.S; h.per_s = (unsigned long long)L.out_feat * q.ncols; }   // reduced on the fly by k_td
                else { RSeg r; memset(&r, 0, sizeof r); r.part = q.part; r.S = q.S; r.elems = (unsigned long long)L.out_feat * q.ncols; r.mode = 0; r.bias = q.P + L.b_off; r.per_n = L.npos * q.ncols; r.act = L.act; r.out = q.Y; segs.push_back(r); }
            }
            head[q.l][q.net] = h;
        }
        emit_reduce(e, segs, pname(e, "fwd_reduce", e->L[lv[0]].kind, lv[0]));
        if (e->L[lv[0]].kind == DQN_LAYER_LSTM) {
            // the recurrence: T launches, each advancing the online s-sequence, the online sp-sequence (double-Q) and the target
            // sp-sequence by one step from the reset state (Flux.reset!, src/solver.jl:249-250,271)
            const int l = lv[0]; const LayerDev L = e->L[l]; const int H = L.H;
            if (lstm_seq_fits(H, Bb)) {        // small LSTM: the whole recurrence of the three sequence sets in ONE launch
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
        t.w_is = e->w_is; t.td = e->td; t.q_on_s = e->q_on_s; t.q_on_sp = e->q_on_sp; t.q_tg_sp = e->q_tg_sp; t.ytarget = e->ytarget; t.best = e->best; t.st = e->state;
        if (!rec) e->prog.push_back({"td_huber", [=](dqn_engine* en) { TdArgs a = t; a.bump_sample_ctr = en->step_sampled ? 1 : 0; launch_td(en->stream, a); }});
        else {
            TdDrqnArgs d; memset(&d, 0, sizeof d); d.B = Bb; d.T = T; d.nA = e->nA; d.ncon = ncon; d.dueling = e->hp.dueling; d.double_q = e->hp.double_q; d.gamma = e->hp.gamma;
            d.on_val = t.on_val; d.on_adv = t.on_adv; d.tg_val = t.tg_val; d.tg_adv = t.tg_adv; d.d_val = t.d_val; d.d_adv = t.d_adv;
            d.a = e->r_a; d.r = e->r_r; d.done = e->r_done; d.mask = e->r_mask; d.td = e->td; d.st = e->state;
            e->prog.push_back({"td_huber_drqn", [=](dqn_engine* en) { launch_td_drqn(en->stream, d); }});
        }
    }
    // ---------------- backward of the online net on the s columns (Zygote through src/solver.jl:219-225)
    std::vector<RSeg> final_segs;   // dW split-K slabs: nothing reads the gradient before Adam, so ONE reduce launch at the end
    bool joined = false;
    for (int li = (int)levels.size() - 1; li >= 0; li--) {
        const auto& lv = levels[li];
        std::vector<VTask> pend;
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
                if (lstm_seq_fits(L.H, Bb)) {
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
                    if (mf && gemm_dw_eligible(V, B, ldv)) { struct A { const float* X[1]; const float* d[1]; float* o[1]; } a; a.X[0] = Xv; a.d[0] = dG; a.o[0] = dst;
                        e->prog.push_back({nm, [=](dqn_engine* en) { launch_gemm_dw(en->stream, V, 1, a.X, ldv, a.d, B, a.o); }}); }
                    else if (mf && mfma_dw_ok(V, B)) e->prog.push_back({nm, [=](dqn_engine* en) { launch_mfma_dw(en->stream, V, Xv, ldv, dG, B, grad, part, false); }});
                    else { VTask t; memset(&t, 0, sizeof t); t.kind = 1; t.L = V; t.X = Xv; t.ldx = ldv; t.dpre = dG; t.B = B; t.S = S; t.kc = dqn_chunk_len(B, V.dw_kc); t.out = dst; add_valu(e, pend, t); }
                    if (S > 1) { RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)(V.K + 1) * V.N; r.mode = 2; r.out = grad + V.w_off; final_segs.push_back(r); }
                };
                emit_dw1(Vh, e->hprev_buf[l], B, pname(e, "dw_wh", L.kind, l));      // first: its junk bias row is then overwritten by nothing that matters
                emit_dw1(Vi, X, ldx, pname(e, "dw_wi", L.kind, l));
                if (L.src >= 0) {
                    const int src = L.src; const int act_src = e->L[src].act; float* out = e->dact[src]; const float* ysrc = e->act_on[src]; const float* P = e->p_on;
                    const int S = dqn_nchunks(Vi.N, Vi.dx_kc); float* part = S > 1 ? palloc(e, (size_t)S * Vi.in_feat * B) : nullptr;
                    if (mf && gemm_dx_eligible(Vi, B, ncon)) { struct A1 { const float* W[1]; const float* d[1]; } a; a.W[0] = P + Vi.w_off; a.d[0] = dG; float* dst = S > 1 ? part : out; const float* ys = S > 1 ? nullptr : ysrc;
                        e->prog.push_back({pname(e, "dx", L.kind, l), [=](dqn_engine* en) { launch_gemm_dx(en->stream, Vi, 1, a.W, a.d, B, dst, ys, ncon, act_src); }}); }
                    else if (mf && mfma_dx_ok(Vi, B, ncon)) e->prog.push_back({pname(e, "dx", L.kind, l), [=](dqn_engine* en) { launch_mfma_dx(en->stream, Vi, P, dG, B, out, part, nullptr, ysrc, ncon, act_src, false); }});
                    else { VTask t; memset(&t, 0, sizeof t); t.kind = 2; t.L = Vi; t.P = P; t.dpre = dG; t.B = B; t.S = S; t.kc = dqn_chunk_len(Vi.N, Vi.dx_kc); t.out = S > 1 ? part : out; t.ysrc = ysrc; t.ldy = ncon; t.act_src = act_src; add_valu(e, pend, t); }
                    if (S > 1) { flush_valu(e, pend, pname(e, "bwd_valu", L.kind, l)); std::vector<RSeg> one; RSeg r; memset(&r, 0, sizeof r); r.part = part; r.S = S; r.elems = (unsigned long long)Vi.in_feat * B; r.mode = 1; r.act = act_src; r.ysrc = ysrc; r.B = B; r.ldy = ncon; r.out = out; one.push_back(r); emit_reduce(e, one, pname(e, "dx_reduce", L.kind, l)); }
                }
                continue;
            }
            {   // dW / db
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
            const bool dense = L.kind == DQN_LAYER_DENSE; const int S = dense ? dqn_nchunks(L.N, L.dx_kc) : 1;
            const float* P = e->p_on; const int act_src = e->L[src].act;
            if (is_join && lv.size() == 2 && S == 1 && mf && same_geo(e->L[lv[0]], e->L[lv[1]]) && gemm_dx_eligible(L, B, ncon)) {
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
        flush_valu(e, pend, pname(e, "bwd_valu", e->L[lv[0]].kind, lv[0]));
        if (dwl.on && dxl.on) {      // dW and dX of this level in ONE launch
            const DwL a = dwl; const DxL x = dxl; dwl.on = dxl.on = false;
            char nm[48]; snprintf(nm, sizeof nm, "%s+%s", a.name, x.name); e->prog_names.push_back(nm); const char* name = e->prog_names.back().c_str();
            e->prog.push_back({name, [=](dqn_engine* en) { launch_gemm_dwdx(en->stream, a.L, a.nprob, a.X, a.ldx, a.d, B, a.o, x.L, x.nsrc, x.W, x.d, x.out, x.ys, ncon, x.act_src); }});
        }
        flush_dw(); flush_dx();
    }
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
    emit_reduce(e, final_segs, "dw_reduce_all");
    e->prog_post_begin = e->prog.size();
    e->prog.push_back({"adam", [](dqn_engine* en) {
        PrioArgs pa; pa.n = (en->hp.prioritized_replay && !en->hp.recurrence) ? en->B : 0; pa.cap2 = en->cap2; pa.idx = en->idx; pa.td = en->td; pa.eps = en->hp.prio_eps; pa.alpha = en->hp.prio_alpha; pa.tree = en->tree;
        AdamSegs none; memset(&none, 0, sizeof none);
        const bool fold = en->adam_segs.n > 0 && !en->comm;     // with a communicator the gradient must be materialised before the all-reduce
        launch_adam(en->stream, en->Pint, en->p_on, en->m, en->v, en->grad, en->state, en->gmax_part, en->hp.adam_f64_scalars, en->hp.learning_rate,
                    en->hp.adam_beta1, en->hp.adam_beta2, en->hp.adam_eps, en->world > 1 ? 1.0f / (float)en->world : 1.0f, pa, fold ? en->adam_segs : none, en->grad); }});
    e->prog_built = true;
    return 0;
}
static void enqueue_step(dqn_engine* e, bool sample, int phase) {
    e->step_sampled = sample;
    if (phase != PH_POST) {
        if (e->hp.recurrence) {
            EpGatherArgs g; g.ep_s = e->ep_s; g.ep_sp = e->ep_sp; g.ep_a = e->ep_a; g.ep_r = e->ep_r; g.ep_done = e->ep_done; g.ep_len = e->ep_len; g.ep_idx = e->ep_idx; g.ep_start = e->ep_start;
            g.E = e->E; g.B = e->B; g.T = e->T; g.x0 = e->x0; g.a_out = e->r_a; g.r_out = e->r_r; g.done_out = e->r_done; g.mask_out = e->r_mask;
            RUN(e, "gather_episodes", launch_gather_episodes(e->stream, g));
        } else {
            // the descent is fused into the gather (every workgroup repeats it) while that is cheaper than a launch of its own:
            // small batches.  At B = 512 / 1e6 leaves the repeats cost more than the ~5 us launch, so sample once, then gather.
            const bool fused = sample && e->B <= 64;
            if (sample && !fused) RUN(e, "sample", launch_sample(e->stream, e->B, e->cap2, e->tree, e->hp.seed, e->idx, e->state, 0));    // k_td bumps the Philox counter
            RUN(e, fused ? "sample_gather" : "gather", launch_gather_fb(e->stream, e->s_rows, e->sp_rows, e->hp.obs_dtype == DQN_OBS_U8, e->E, e->B, e->idx, e->x0,
                                                                        fused ? 1 : 0, e->cap2, e->tree, e->hp.seed, e->state));
        }
        for (size_t i = 0; i < e->prog_post_begin; i++) {
            if ((long)i == e->final_reduce_step && e->adam_segs.n > 0 && !e->comm) continue;   // folded into k_adam
            RUN(e, e->prog[i].name, e->prog[i].fn(e));
        }
    }
    if (phase != PH_PRE) for (size_t i = e->prog_post_begin; i < e->prog.size(); i++) RUN(e, e->prog[i].name, e->prog[i].fn(e));
}
static int capture(dqn_engine* e, bool sample, int phase, hipGraphExec_t* out) {
    hipGraph_t g;
    HIPCHK(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
    enqueue_step(e, sample, phase);
    HIPCHK(hipStreamEndCapture(e->stream, &g));
    HIPCHK(hipGraphInstantiate(out, g, nullptr, nullptr, 0));
    HIPCHK(hipGraphDestroy(g)); return 0;
}
static int allreduce_grads(dqn_engine* e) {
    const int rc = g_rccl.AllReduce(e->grad, e->grad, e->Pint, /*ncclFloat*/ 7, /*ncclSum*/ 0, e->comm, e->stream);
    if (rc) return fail("ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return 0;
}
static int run_step(dqn_engine* e, bool sample) {
    if (build_program(e)) return -1;
    const int gi = sample ? 0 : 1;
    if (e->world > 1 || (e->comm && e->force_comm)) {
        if (e->hp.use_graph && !e->profiling) {
            if (!e->g_pre[gi] && capture(e, sample, PH_PRE, &e->g_pre[gi])) return -1;
            if (!e->g_post && capture(e, sample, PH_POST, &e->g_post)) return -1;
            HIPCHK(hipGraphLaunch(e->g_pre[gi], e->stream));
            if (allreduce_grads(e)) return -1;
            HIPCHK(hipGraphLaunch(e->g_post, e->stream));
        } else { enqueue_step(e, sample, PH_PRE); if (allreduce_grads(e)) return -1; enqueue_step(e, sample, PH_POST); }
        return 0;
    }
    if (e->hp.use_graph && !e->profiling) {
        if (!e->g_full[gi] && capture(e, sample, PH_ALL, &e->g_full[gi])) return -1;
        HIPCHK(hipGraphLaunch(e->g_full[gi], e->stream));
    } else enqueue_step(e, sample, PH_ALL);
    return 0;
}
static int fetch_scalars(dqn_engine* e, float* loss, float* gn) {
    // globalnorm (helpers.jl:38-46): fold the Adam kernel's per-block maxima only when the host asks for the scalar
    if (gn) launch_update_priorities(e->stream, 0, e->cap2, e->idx, e->td, e->hp.prio_eps, e->hp.prio_alpha, e->tree, e->state, 1, 1.0, 1.0, e->gmax_part, adam_blocks(e->Pint) + 4096);
    StepState s; HIPCHK(hipMemcpyAsync(&s, e->state, sizeof s, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
    if (s.err == 2) return fail("AssertionError: all(new_priorities .> 0f0)");
    if (loss) *loss = s.loss;
    if (gn) { float g; memcpy(&g, &s.gnorm_bits, 4); *gn = g; }
    return 0;
}
extern "C" int dqn_train_step(dqn_engine_t* e, const int64_t* idx, float* loss, float* grad_norm, float* td_out) {
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) return fail("recurrence = true: use dqn_train_step_drqn (src/solver.jl:239-287)");
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (idx) { if (check_idx(e, idx, e->B)) return -1; HIPCHK(hipMemcpyAsync(e->idx, idx, (size_t)e->B * 8, hipMemcpyHostToDevice, e->stream)); }
    if (run_step(e, idx == nullptr)) return -1;
    if (td_out) HIPCHK(hipMemcpyAsync(td_out, e->td, (size_t)e->B * 4, hipMemcpyDeviceToHost, e->stream));
    if (loss || grad_norm || td_out) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
extern "C" int dqn_train_steps(dqn_engine_t* e, int n, float* loss, float* grad_norm) {
    HIPCHK(hipSetDevice(e->device));
    if (e->hp.recurrence) { for (int i = 0; i < n; i++) if (dqn_train_step_drqn(e, nullptr, nullptr, i + 1 == n ? loss : nullptr, i + 1 == n ? grad_norm : nullptr)) return -1; return 0; }
    if (e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    for (int i = 0; i < n; i++) if (run_step(e, true)) return -1;
    if (loss || grad_norm) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
extern "C" int dqn_get_last_q(dqn_engine_t* e, float* qs, float* qsp, float* qt, int32_t* best, float* y) {
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    const size_t n = (size_t)e->B * e->nA * 4;
    if (qs) HIPCHK(hipMemcpy(qs, e->q_on_s, n, hipMemcpyDeviceToHost));
    if (qsp) HIPCHK(hipMemcpy(qsp, e->q_on_sp, n, hipMemcpyDeviceToHost));
    if (qt) HIPCHK(hipMemcpy(qt, e->q_tg_sp, n, hipMemcpyDeviceToHost));
    if (best) HIPCHK(hipMemcpy(best, e->best, (size_t)e->B * 4, hipMemcpyDeviceToHost));
    if (y) HIPCHK(hipMemcpy(y, e->ytarget, (size_t)e->B * 4, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int dqn_get_last_indices(dqn_engine_t* e, int64_t* idx) {
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(idx, e->idx, (size_t)e->B * 8, hipMemcpyDeviceToHost)); return 0;
}

// ---------------------------------------------------------------- policy (src/policy.jl:38-64)
static int policy_ws(dqn_engine* e, int n) {
    if (n <= e->pol_n) return 0;
    HIPCHK(hipStreamSynchronize(e->stream)); free_policy_ws(e); drop_act(e, e->act); drop_act(e, e->evalp);
    size_t need = 1;   // split-K partials of the widest forward at n columns
    for (int i = 0; i < e->nl; i++) { const size_t sf = dqn_nchunks(e->L[i].K, e->L[i].fwd_kc); if (sf > 1) need = std::max(need, sf * (size_t)e->L[i].out_feat * n); }
    if (need > e->partials_elems) { drop_graphs(e); hipFree(e->partials); e->partials = nullptr; DM(e->partials, 2 * need); e->partials_elems = need; }
    DM(e->pol_obs, (size_t)n * e->E); DM(e->pol_x, (size_t)n * e->E); DM(e->pol_q, (size_t)n * e->nA); DM(e->pol_a, n);
    for (int i = 0; i < e->nl; i++) DM(e->pol_act[i], (size_t)e->L[i].out_feat * n);
    e->pol_n = n; return 0;
}
static int policy_state(dqn_engine* e, int n, bool force_reset) {
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
extern "C" int dqn_forward(dqn_engine_t* e, int which, const float* obs, int n, float* q_out) {
    if (!obs) return fail("obs is null");
    if (policy_forward(e, which, obs, n)) return -1;
    HIPCHK(hipMemcpyAsync(q_out, e->pol_q, (size_t)n * e->nA * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
extern "C" int dqn_greedy_action(dqn_engine_t* e, const float* obs, int n, int32_t* a_out) {
    if (!obs) return fail("obs is null");
    if (policy_forward(e, DQN_NET_ONLINE, obs, n)) return -1;
    HIPCHK(hipMemcpyAsync(a_out, e->pol_a, (size_t)n * 4, hipMemcpyDeviceToHost, e->stream)); HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}

// ---------------------------------------------------------------- DRQN: EpisodeReplayBuffer + recurrent batch_train!
#define NEED_REC(e) do { if (!(e)->hp.recurrence) return fail("this engine was created with recurrence = false"); } while (0)
extern "C" int dqn_episode_commit(dqn_engine_t* e) {          // add_episode! (src/episode_replay.jl:54-60)
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    const int len = (int)e->ep_cur_len;
    e->ep_len_host[(size_t)e->ep_widx] = len;
    HIPCHK(hipMemcpyAsync(e->ep_len + e->ep_widx, &e->ep_len_host[(size_t)e->ep_widx], 4, hipMemcpyHostToDevice, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    e->ep_widx = (e->ep_widx + 1) % e->ep_cap; if (e->ep_size < e->ep_cap) e->ep_size++;
    e->ep_cur_len = 0; return 0;
}
extern "C" int dqn_episode_add(dqn_engine_t* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done, int n) {
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
extern "C" int dqn_episode_count(dqn_engine_t* e, int64_t* cur, int64_t* cap) { NEED_REC(e); if (cur) *cur = e->ep_size; if (cap) *cap = e->ep_cap; return 0; }
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
extern "C" int dqn_episode_get_batch(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* s, int32_t* a, float* r, float* sp, float* done, int32_t* mask) {
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
extern "C" int dqn_train_step_drqn(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* loss, float* grad_norm) {
    NEED_REC(e); HIPCHK(hipSetDevice(e->device));
    std::vector<int64_t> di; std::vector<int32_t> ds;
    if (!ep_idx) {   // sample(rng, 1:n, B, replace=false); ep_start = rand(rng, 1:length(ep))  (src/episode_replay.jl:75,81) -- host-side SplitMix draws
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
        ep_idx = di.data(); ep_start = ds.data();
    }
    if (drqn_check(e, ep_idx, ep_start) || drqn_upload_draws(e, ep_idx, ep_start)) return -1;
    if (build_program(e)) return -1;
    if (e->hp.use_graph && !e->profiling && e->world == 1) {
        if (!e->g_drqn && capture(e, false, PH_ALL, &e->g_drqn)) return -1;
        HIPCHK(hipGraphLaunch(e->g_drqn, e->stream));
    } else { enqueue_step(e, false, PH_PRE); if (e->world > 1 && allreduce_grads(e)) return -1; enqueue_step(e, false, PH_POST); }
    if (loss || grad_norm) return fetch_scalars(e, loss, grad_norm);
    return 0;
}
extern "C" int dqn_reset_state(dqn_engine_t* e) {             // resetstate!(policy) (src/policy.jl:32-34)
    HIPCHK(hipSetDevice(e->device));
    if (!e->hp.recurrence) return 0;
    return policy_state(e, e->pol_state_n > 0 ? e->pol_state_n : 1, true);
}
extern "C" int dqn_get_hidden(dqn_engine_t* e, float* hc, size_t n) {   // hiddenstates(m) (src/helpers.jl:61-63): per LSTM layer h then c, [out][streams]
    HIPCHK(hipSetDevice(e->device)); size_t off = 0;
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_state_n; if (off + 2 * m > n) return fail("get_hidden: buffer too small");
        HIPCHK(hipMemcpyAsync(hc + off, e->pol_h[i][e->pol_flip], m * 4, hipMemcpyDeviceToHost, e->stream)); off += m;
        HIPCHK(hipMemcpyAsync(hc + off, e->pol_c[i][e->pol_flip], m * 4, hipMemcpyDeviceToHost, e->stream)); off += m;
    }
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}
extern "C" int dqn_set_hidden(dqn_engine_t* e, const float* hc, size_t n) {   // sethiddenstates!(m, hs) (src/helpers.jl:71-79)
    HIPCHK(hipSetDevice(e->device)); size_t off = 0;
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_state_n; if (off + 2 * m > n) return fail("set_hidden: buffer too small");
        HIPCHK(hipMemcpyAsync(e->pol_h[i][e->pol_flip], hc + off, m * 4, hipMemcpyHostToDevice, e->stream)); off += m;
        HIPCHK(hipMemcpyAsync(e->pol_c[i][e->pol_flip], hc + off, m * 4, hipMemcpyHostToDevice, e->stream)); off += m;
    }
    HIPCHK(hipStreamSynchronize(e->stream)); return 0;
}

// ---------------------------------------------------------------- vectorised environments on the device
static void free_env_arrays(EnvDev& V) {      // the per-copy arrays of an evaluation env set (images and spec are shared with the training set)
    hipFree(V.tm_s); hipFree(V.tm_prev); hipFree(V.tm_t); hipFree(V.gw_pos); hipFree(V.gw_prev);
    hipFree(V.actions); hipFree(V.rewards); hipFree(V.dones); hipFree(V.pending); hipFree(V.ep_reward); hipFree(V.ep_step); hipFree(V.fin_eps); hipFree(V.fin_reward);
    memset(&V, 0, sizeof V);
}
static void free_envs(dqn_engine* e) {
    EnvDev& V = e->env;
    hipFree(e->env_images); hipFree(V.tm_s); hipFree(V.tm_prev); hipFree(V.tm_t); hipFree(V.gw_pos); hipFree(V.gw_prev); hipFree(e->roll);
    hipFree(V.actions); hipFree(V.rewards); hipFree(V.dones); hipFree(V.pending); hipFree(V.ep_reward); hipFree(V.ep_step); hipFree(V.fin_eps); hipFree(V.fin_reward);
    e->env_images = nullptr; e->roll = nullptr; memset(&V, 0, sizeof V); e->has_envs = false;
    free_env_arrays(e->eval_env); hipFree(e->eval_roll); e->eval_roll = nullptr; e->eval_n = 0;
    drop_act(e, e->act); drop_act(e, e->evalp);
}
extern "C" int dqn_envs_create(dqn_engine_t* e, const dqn_env_spec* sp) {
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
    DM(V.actions, n); DM(V.rewards, n); DM(V.dones, n); DM(V.pending, n); DM(V.ep_reward, n); DM(V.ep_step, n); DM(V.fin_eps, n); DM(V.fin_reward, n); DM(e->roll, 1);
    HIPCHK(hipMemsetAsync(V.fin_eps, 0, (size_t)n * 8, e->stream)); HIPCHK(hipMemsetAsync(V.fin_reward, 0, (size_t)n * 8, e->stream));
    HIPCHK(hipMemsetAsync(V.actions, 0, (size_t)n * 4, e->stream)); HIPCHK(hipMemsetAsync(V.rewards, 0, (size_t)n * 4, e->stream));
    HIPCHK(hipMemsetAsync(e->roll, 0, sizeof(RolloutDev), e->stream));
    e->has_envs = true;
    return dqn_envs_reset(e);
}
extern "C" int dqn_envs_reset(dqn_engine_t* e) {
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
    for (size_t li = 0; li < levels.size(); li++) {
        const auto& lv = levels[li]; const bool last = li + 1 == levels.size();
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
            ap.steps.push_back({name, [=](dqn_engine* en) { launch_gemm_fwd(en->stream, L, np, a.W, a.bias, a.X, a.ldx, a.col0, a.ncols, a.out); }});
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
            if (q.S > 1) {
                if (last) { h.p = q.part; h.S = q.S; h.per_s = (unsigned long long)L.out_feat * n; }      // reduced on the fly by k_env_step
                else { RSeg r; memset(&r, 0, sizeof r); r.part = q.part; r.S = q.S; r.elems = (unsigned long long)L.out_feat * n; r.mode = 0; r.bias = P + L.b_off; r.per_n = L.npos * n; r.act = L.act; r.out = q.Y; segs.push_back(r); }
            }
            head[q.l] = h;
        }
        emit_reduce(e, segs, pname(e, "act_reduce", e->L[lv[0]].kind, lv[0]));
    }
    e->sink = nullptr; e->alloc_sink = nullptr;
    const int lq = e->hp.dueling ? e->last_adv : e->last_base;
    ActHeads Hd; memset(&Hd, 0, sizeof Hd); Hd.adv = head[lq]; if (e->hp.dueling) Hd.val = head[e->last_val]; Hd.dueling = e->hp.dueling; Hd.q_out = e->pol_q; Hd.amax = e->pol_a;
    // act!, add_exp!, observe, episode bookkeeping
    const bool u8 = e->hp.obs_dtype == DQN_OBS_U8;
    ReplayMeta R; R.cap = e->cap; R.cap2 = e->cap2; R.a = e->ra; R.r = e->rr; R.done = e->rdone; R.tree = e->tree; R.state = e->state; R.eps = e->hp.prio_eps; R.alpha = e->hp.prio_alpha;
    void *srows = e->s_rows, *sprows = e->sp_rows; float* px = e->pol_x; const long long cap = e->cap; const EnvDev Vc = V;
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
extern "C" int dqn_rollout(dqn_engine_t* e, int n_steps, const dqn_rollout_cfg* cfg, dqn_rollout_stats* out) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->has_envs) return fail("no device environments: call dqn_envs_create");
    if (cfg->t0 < 1) return fail("t0 counts from 1 (src/solver.jl:82)");
    EnvDev& V = e->env; const int n = V.n;
    if (build_act_program(e, e->act, V, e->roll)) return -1;
    if (cfg->train_freq > 0 && build_program(e)) return -1;       // may reallocate split-K workspaces: before any capture
    RolloutDev h; h.t = cfg->t0 - 1; h.widx = ((e->widx - n) % e->cap + e->cap) % e->cap; h.eps_start = cfg->eps_start; h.eps_stop = cfg->eps_stop; h.eps_steps = cfg->eps_steps; h.pad = 0;
    HIPCHK(hipMemcpyAsync(e->roll, &h, sizeof h, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));   // h lives on this stack frame
    launch_env_observe(e->stream, V, nullptr, 0, e->pol_x);
    const bool graph = e->hp.use_graph && !e->profiling;
    if (graph && act_graph(e, e->act)) return -1;
    long long trained = 0;
    for (int k = 0; k < n_steps; k++) {
        const long long t = cfg->t0 + k;
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
extern "C" int dqn_evaluate(dqn_engine_t* e, int n_eval, int max_episode_length, uint64_t seed, double* avg_reward, double* avg_steps) {
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
        DM(e->eval_roll, 1);
        e->eval_n = n_eval;
    }
    if (W.seed != seed || W.max_episode_length != max_episode_length) { W.seed = seed; W.max_episode_length = max_episode_length; drop_act(e, e->evalp); }   // baked into the program
    if (build_act_program(e, e->evalp, W, e->eval_roll)) return -1;
    RolloutDev h; memset(&h, 0, sizeof h);                                   // t = 0; eps schedule (0, 0, 1): always greedy
    h.eps_steps = 1.0f;
    HIPCHK(hipMemcpyAsync(e->eval_roll, &h, sizeof h, hipMemcpyHostToDevice, e->stream)); HIPCHK(hipStreamSynchronize(e->stream));
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
extern "C" int dqn_envs_peek(dqn_engine_t* e, float* obs, int32_t* actions, float* rewards, uint8_t* dones) {
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

// ---------------------------------------------------------------- data-parallel replicas
extern "C" int dqn_comm_unique_id(void* id128) { if (rccl_load()) return -1; const int rc = g_rccl.GetUniqueId(id128); return rc ? fail("ncclGetUniqueId failed (%d)", rc) : 0; }
extern "C" int dqn_comm_init(dqn_engine_t* e, const void* id128, int rank, int world) {
    if (rccl_load()) return -1;
    HIPCHK(hipSetDevice(e->device));
    Id128 id; memcpy(id.b, id128, 128);
    const int rc = g_rccl.CommInitRank(&e->comm, world, id, rank);
    if (rc) return fail("ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    e->rank = rank; e->world = world; e->force_comm = getenv("DQN_FORCE_ALLREDUCE") != nullptr; drop_graphs(e); return 0;
}

// ---------------------------------------------------------------- misc
extern "C" int dqn_stream_sync(dqn_engine_t* e) { HIPCHK(hipSetDevice(e->device)); HIPCHK(hipStreamSynchronize(e->stream)); return 0; }
extern "C" int dqn_stream_handle(dqn_engine_t* e, void** s) { *s = (void*)e->stream; return 0; }
// Holds the stream until the host has enqueued the whole profiled step, so that the HIP events around each kernel time
// the kernel and not the host's launch latency.  Bounded spin (~0.2 s) so a dead host can never hang the GPU.
__global__ void k_gate(volatile int* flag) {
    for (long i = 0; i < 2000000 && *flag == 0; i++) __builtin_amdgcn_s_sleep(32);
}
extern "C" int dqn_profile_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries) {
    HIPCHK(hipSetDevice(e->device));
    if (!e->hp.recurrence && e->size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    if (e->hp.recurrence && e->ep_size < e->B) return fail("AssertionError: r._curr_size >= r.batch_size");
    HIPCHK(hipStreamSynchronize(e->stream));
    static int* gate = nullptr;
    if (!gate) HIPCHK(hipHostMalloc((void**)&gate, sizeof(int), hipHostMallocMapped));
    *gate = 0;
    hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, e->stream, (volatile int*)gate);
    e->profiling = true; e->prof.clear();
    const int rc = e->hp.recurrence ? dqn_train_step_drqn(e, nullptr, nullptr, nullptr, nullptr) : run_step(e, true);
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
