// engine.h -- internal header of libdqn_mi355x.so's host side: the engine record, error/alloc helpers and the functions the
// translation units engine.hip (ABI core, graphs, policy), engine_program.hip (static launch program of the train step),
// engine_drqn.hip (EpisodeReplayBuffer + recurrent step) and engine_envs.hip (device-resident environments) share.
#pragma once
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <functional>
#include <string>
#include <vector>
#include "common.h"

int fail(const char* fmt, ...);      // records the message for dqn_last_error(), returns -1
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return fail("HIP error %s at %s:%d (%s)", hipGetErrorString(_e), __FILE__, __LINE__, #x); } while (0)

// ---------------------------------------------------------------- engine
struct ProfEntry { const char* name; hipEvent_t a, b; };

// Experiment / test switches.  Read ONCE from the environment of the call that creates the engine (read_opts, engine.hip -- the only getenv site of the
// library besides trace builds), kept in the engine: no function-local statics, nothing process-wide, a second engine in the same process never inherits the
// first one's switches.  The communicator switches (force_allreduce, dp_*) are re-read by dqn_comm_init, the call that makes them meaningful.
struct EngineOpts {
    int no_tiny = 0;              // DQN_NO_TINY: networks that fit in LDS take the multi-launch program
    int fwd_m32 = -1 /* -1: large launches only */, no_dx_wide = 0, no_fwd_wres = 0;      // DQN_FWD_M32 / DQN_NO_DX_WIDE / DQN_NO_FWD_WRES -> LayerDev::opt bits
    int mid_group = 4, mid_big = 16;                   // DQN_MID_GROUP / DQN_MID_BIG: middle steps of dqn_train_steps per graph (mid_big also needs mid_group > 1)
    int sim_world = 0;            // DQN_SIM_WORLD=k: one process plays k ranks (tests)
    int no_graph_upload = 0;      // DQN_NO_GRAPH_UPLOAD
    int no_rollout_cycle = 0;     // DQN_NO_ROLLOUT_CYCLE
    int no_act_head = 0;          // DQN_NO_ACT_HEAD (A/B, both schedules under test): the acting step keeps k_reduce_multi + heads + k_env_step where the fused tail (act_head.hip) would apply
    int no_rh_pm = 0;             // DQN_NO_RH_PM (A/B, r06): the hidden layers' split-K slabs stay [S][N][columns] where k_red_head reads them (default: piece-major, GFwdProb::pm)
    int dw_split = 128;            // DQN_DW_SPLIT=n: the last n units of a large-batch dW section run as two halves along N (nn_gemm.hip dw_section; 0 = off; LayerDev::opt bits 8..15 in units of 16)
    int no_st_wt = 0;             // DQN_NO_ST_WT: small-batch engines keep plain / non-temporal output stores in the GEMM launches (A/B)
    int no_head_cols4 = 0;        // DQN_NO_HEAD_COLS4=1: keep k_head_td (one workgroup per column) at large batches where k_head_cols4 (red_head.hip) would apply (A/B); =2: k_head_cols4 without the transposed copies (its fallback loader, under test)
    int no_red_head = 0;          // DQN_NO_RED_HEAD: keep k_reduce_multi + k_head_td where the fused reduce + head launch (red_head.hip) would apply (A/B, both schedules under test)
    int no_u8_arena = 0, head_fuse_maxb = 1024, no_head_fuse = 0, head_dbg = 0, prio_fork = 0, no_pregather = 0, lstm_dw_mfma = 0;
    int force_allreduce = 0, dp_allreduce = 0, dp_overlap = -1 /* -1: decided from world size and bytes (engine_program.hip) */, dp_no_one_graph = 0;      // DQN_FORCE_ALLREDUCE / DQN_DP_ALLREDUCE / DQN_DP_OVERLAP / DQN_DP_NO_ONE_GRAPH
    // timing probes (wrong numbers, right schedule) and stamps
    int probe_no_tg = 0, drqn_probe = 0, drqn_stamps = 0, tiny_stop = 0;
};
void read_opts(EngineOpts& o, bool comm_only = false);

struct dqn_engine {
    EngineOpts opt;
    int device = 0; hipStream_t stream = nullptr, stream2 = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t stream3 = nullptr; hipEvent_t ev_xa = nullptr, ev_xb = nullptr, ev_xc = nullptr;      // replicas: the exchange runs on its own stream (dp_overlap)
    int nl = 0; LayerDev L[DQN_MAX_LAYERS]; LayerDev* L_dev = nullptr; bool plan_defaulted = false;      // plan == NULL at dqn_engine_create (dqn_comm_init may re-derive it for replicas)
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
    long long* idx = nullptr; long long* idx_pre = nullptr; float* x0 = nullptr;      // idx_pre: the next sample()'s indices, drawn by the priority block (StepState::pre_valid)
    float *act_on[DQN_MAX_LAYERS] = {}, *act_tg[DQN_MAX_LAYERS] = {}, *dact[DQN_MAX_LAYERS] = {};
    float *join_tmp = nullptr, *partials = nullptr, *gmax_part = nullptr; size_t partials_elems = 0;
    float *w_is = nullptr, *td = nullptr, *q_on_s = nullptr, *q_on_sp = nullptr, *q_tg_sp = nullptr, *ytarget = nullptr; int* best = nullptr;
    // get_batch seam workspace
    float *gb_rows = nullptr, *gb_r = nullptr, *gb_done = nullptr, *gb_w = nullptr; int* gb_a = nullptr; long long* gb_idx = nullptr;
    // train-step batch scalars (written by the gather launch, read by the fused head kernel)
    float *gb_r2 = nullptr, *gb_done2 = nullptr, *gb_w2 = nullptr; int* gb_a2 = nullptr;
    // policy workspace
    EnvDev env{}; bool has_envs = false; unsigned char* env_images = nullptr;
    int pol_n = 0; float *pol_obs = nullptr, *pol_x = nullptr, *pol_act[DQN_MAX_LAYERS] = {}, *pol_q = nullptr; int* pol_a = nullptr;
    // graphs: [0] = step with sampling, [1] = step on given indices; with a communicator the step is cut in two
    hipGraphExec_t g_full[2] = {nullptr, nullptr}, g_pre[2] = {nullptr, nullptr}, g_post = nullptr;
    bool arena_u8 = false;  // the observation arena x0 holds bytes (u8 replay, first layer converts in its tile loads): set by build_program
    unsigned long long* ktrace_buf = nullptr;     // dqn_debug_ktrace (trace builds): owned by the engine, freed with it
    // comm
    void* comm = nullptr; int rank = 0, world = 1; bool force_comm = false;
    // exchange mode of the replicas: gather = all-gather of the wide dense layers' operands + small gradients (dp.hip); else all-reduce of the gradient.
    // sim_world = k (env DQN_SIM_WORLD, tests): one process plays k identical ranks, the collective is k local copies.
    bool dp_gather = false, dp_pack_folds = false, dp_adam_folds = false; AdamSegs dp_adam_segs; int sim_world = 0; float *dp_send = nullptr, *dp_recv = nullptr; size_t dp_count = 0;
    // dp_overlap: the block is exchanged in TWO collectives -- [0, dp_count_a): X | dpre of the wide dense layers, final right after the head level, gathered (into dp_recv) on
    // stream3 WHILE the conv backward runs; [dp_count_a, dp_count): the small gradients, gathered (into dp_recv_b) after the backward pass
    long long* sim_idx = nullptr; float* sim_td = nullptr;      // dqn_sim_ranks_step (test hook): per-rank copies of the index lists and TD errors
    bool dp_overlap = false; size_t dp_count_a = 0; float* dp_recv_b = nullptr; DpPackArgs dp_pk_a;   // force_comm: run the all-reduce path even at world == 1 (tests)
    // DRQN (recurrence = true): column count per sequence set Bc = T*B (B otherwise); EpisodeReplayBuffer storage; LSTM workspaces
    int Bc = 0, T = 1; long long ep_cap = 0, ep_size = 0, ep_widx = 0, ep_cur_len = 0; std::vector<int> ep_len_host; std::vector<int64_t> ep_perm;
    float *ep_s = nullptr, *ep_sp = nullptr, *ep_r = nullptr; int* ep_a = nullptr; unsigned char* ep_done = nullptr; int* ep_len = nullptr;
    long long* ep_idx = nullptr; int* ep_start = nullptr; int* r_a = nullptr; float *r_r = nullptr, *r_done = nullptr, *r_mask = nullptr;
    float *gx_on[DQN_MAX_LAYERS] = {}, *gx_tg[DQN_MAX_LAYERS] = {}, *cst_on[DQN_MAX_LAYERS] = {}, *cst_tg[DQN_MAX_LAYERS] = {}, *gates[DQN_MAX_LAYERS] = {}, *tcb[DQN_MAX_LAYERS] = {},
          *hprev_buf[DQN_MAX_LAYERS] = {}, *cprev_buf[DQN_MAX_LAYERS] = {}, *dG[DQN_MAX_LAYERS] = {}, *dhn[DQN_MAX_LAYERS] = {}, *dcn[DQN_MAX_LAYERS] = {};
    float *pol_h[DQN_MAX_LAYERS][2] = {}, *pol_c[DQN_MAX_LAYERS][2] = {}, *pol_gx[DQN_MAX_LAYERS] = {}; int pol_flip = 0, pol_state_n = 0; uint64_t drqn_draws = 0;
    hipGraphExec_t g_drqn[2] = {nullptr, nullptr}, g_drqn_k[2] = {nullptr, nullptr};      // the recurrent step as a graph (fused step: two alternating instances, see draw_ev); g_drqn_k: runs of 8 fused steps
    // static launch program
    struct Step { const char* name; std::function<void(dqn_engine*)> fn; };
    // acting programs (forward on n columns + env kernels), one for the training envs and one for the evaluation envs
    // cycle: ONE graph of `cycle_F` acting steps (+ a plain sampled train step when cycle_train) -- the device loop's unit of work between two
    // train steps; a graph launch costs ~5 us of stream time, a GridWorld vector step 28
    struct ActProg { std::vector<Step> steps; int n = 0; bool fused_tail = false; hipGraphExec_t graph = nullptr; std::vector<void*> allocs;
                     hipGraphExec_t cycle = nullptr; int cycle_F = 0; bool cycle_train = false;
                     hipGraphExec_t envc = nullptr; int envc_due = 0; };      // envc: one vector step of the reference's cadence (the acting step + its `envc_due` pipelined train steps) as ONE graph
    ActProg act, evalp; std::vector<Step>* sink = nullptr; std::vector<void*>* alloc_sink = nullptr; RolloutDev *roll = nullptr, *eval_roll = nullptr;
    EnvDev eval_env{}; int eval_n = 0;
    std::vector<Step> prog; size_t prog_post_begin = 0, prog_pre1_end = 0; bool prog_built = false, step_sampled = true, prio_forked = false, prio_in_bwd = false;
    // pre-gather (common.h PreGather), only between the steps of one dqn_train_steps(n) call: step_pregather = this step's Adam launch gathers
    // the next batch; step_take_pre = this step runs without its gather launch
    int gmax_used = 0;                    // live slots of gmax_part (per-block max |g| of the step's Adam jobs): what the on-demand fold reads
    StepState* state_host = nullptr;      // pinned landing buffer of fetch_scalars
    // scalar mailbox (StepMail, common.h): mapped pinned host ring the step's last launch writes (loss, grad_norm) into; pub_issued = publishes enqueued by the host,
    // pub_ctr = publishes executed by the device (the record of ticket t sits in slot t % DQN_MAIL_SLOTS once its seq == t)
    StepMail *mail_host = nullptr, *mail_dev = nullptr; unsigned long long* pub_ctr = nullptr; unsigned long long pub_issued = 0; bool step_publish = false;
    unsigned long long mail_swept = 0; bool mail_plain = false;      // mail_swept: records [1, mail_swept] have been checked for device-side errors (mail_sweep); mail_plain: the ring is ordinary host memory (mapped allocation refused)
    hipGraphExec_t g_full_pub[2] = {nullptr, nullptr};      // g_full + the publish launch
    hipGraphExec_t g_pgv_pub = nullptr;                      // the LAST step of dqn_train_steps (takes the pre-gathered batch, gathers nothing) + the publish launch
    bool pg_ok = false, step_pregather = false, step_take_pre = false; PreGather pg; long adam_step = -1;
    bool no_tiny = false;   // DQN_NO_TINY at dqn_engine_create: always the multi-launch program
    bool tiny = false;      // the whole step is ONE single-workgroup launch that samples and gathers itself (tiny_step.hip)
    // fused recurrent step: episode draws travel through a mapped pinned host buffer of DQN_DRAW_SLOTS slots the kernel reads directly.  A slot is a LAUNCH PARAMETER, fixed per graph
    // node: two 8-step graphs (slots 0-7 / 8-15) and two single-step graphs (16 / 17) alternate, and before the host rewrites the slots of one it waits for the event recorded
    // behind that graph's previous launch
    long long *draw_idx_h = nullptr, *draw_idx_d = nullptr; int *draw_start_h = nullptr, *draw_start_d = nullptr;
    hipEvent_t draw_ev[4] = {nullptr, nullptr, nullptr, nullptr}; bool draw_ev_used[4] = {false, false, false, false}; int drqn_grp_par = 0, drqn_one_par = 0, drqn_slot_next = 0;
    unsigned long long* drqn_stamps = nullptr;      // timing probe of the fused recurrent step (DQN_DRQN_STAMPS)
    bool launch_failed = false;      // a launcher refused (an LDS attribute the device would not grant): reported by the entry point that enqueued the step
    bool drqn_fused = false;      // recurrent step = the column-parallel launch (which gathers its own episode rows) + the Adam launch (drqn_cols.hip)
    hipGraphExec_t g_pgv[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [take_pre][pregather] variants of the sampled single-device step
    hipGraphExec_t g_mid = nullptr; int mid_group = 4;                          // mid_group consecutive middle steps of dqn_train_steps as one graph
    bool mid_big_warm = false;                                                  // g_mid_big has been launched at least once (its first launch is the expensive one)
    hipGraphExec_t g_mid_big = nullptr; int mid_big = 16;                       // ... and runs of mid_big of them (DQN_MID_BIG; 0 = off): 20 steps = first + 16 + 2 single + last
    hipGraphExec_t g_pre1[3] = {nullptr, nullptr, nullptr}, g_pre2 = nullptr;
    hipGraphExec_t g_dp_one[4] = {nullptr, nullptr, nullptr, nullptr}; int dp_one_state = 0;      // replicas: the WHOLE step incl. its collective(s) as ONE graph ([take_pre][pregather]); state 0 untried, 1 works, -1 RCCL refused the capture      // dp_overlap: first half cut in two ([0] sampled, [1] given indices, [2] without the gather launch)
    hipGraphExec_t g_pre_tp = nullptr, g_post_pg = nullptr;                      // replicas: first half without the gather launch / second half whose Adam launch gathers
    AdamSegs adam_segs; long final_reduce_step = -1;   // deferred dW slabs: reduced inside k_adam unless a communicator needs the materialised gradient
    std::vector<void*> prog_allocs; std::vector<std::string> prog_names;
    // profiling
    bool profiling = false; std::vector<ProfEntry> prof;
};

void prof_begin(dqn_engine* e, const char* name);
void prof_end(dqn_engine* e);
#define RUN(e, name, call) do { prof_begin(e, name); call; prof_end(e); } while (0)
enum { PH_ALL = 0, PH_PRE = 1, PH_POST = 2, PH_PRE1 = 3, PH_PRE2 = 4, PH_DP_ONE = 5 };      // PRE = PRE1 (up to the point where the wide layers' operands are final) + PRE2 (the rest of the backward pass)
template <class T> static int dmalloc(T** p, size_t n) {
    hipError_t e = hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return fail("hipMalloc of %zu bytes failed: %s", n * sizeof(T), hipGetErrorString(e));
    return 0;
}
#define DM(p, n) do { if (dmalloc(&(p), (n))) return -1; } while (0)
#define NEED_REC(e) do { if (!(e)->hp.recurrence) return fail("this engine was created with recurrence = false"); } while (0)

// engine.hip
void drop_graphs(dqn_engine* e);
void drop_act(dqn_engine* e, dqn_engine::ActProg& a);
void fwd_layer(dqn_engine* e, const LayerDev& l, const float* P, const float* X, int ldx, int col0, int ncols, float* Y, const char* name);
void enqueue_step(dqn_engine* e, bool sample, int phase);
int capture(dqn_engine* e, bool sample, int phase, hipGraphExec_t* out, int repeat = 1);
int exchange_grads(dqn_engine* e);      // the one collective of a data-parallel step (all-gather or all-reduce)
int run_step(dqn_engine* e, bool sample, bool take_pre = false, bool pregather = false);
int fetch_scalars(dqn_engine* e, float* loss, float* gn);
int policy_ws(dqn_engine* e, int n);
int policy_state(dqn_engine* e, int n, bool force_reset);
// engine_drqn.hip
int drqn_train_steps(dqn_engine* e, int n, float* loss, float* grad_norm);
// engine_program.hip
template <class T> static T* upload(dqn_engine* e, const std::vector<T>& v) {
    T* d = nullptr; hipMalloc((void**)&d, sizeof(T) * v.size()); hipMemcpy(d, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice);
    (e->alloc_sink ? *e->alloc_sink : e->prog_allocs).push_back(d); return d;
}
float* palloc(dqn_engine* e, size_t n);
bool same_geo(const LayerDev& a, const LayerDev& b);
void add_valu(dqn_engine* e, std::vector<VTask>& pend, const VTask& t);
void flush_valu(dqn_engine* e, std::vector<VTask>& pend, const char* name, const PrioArgs* prio = nullptr);
void emit_reduce(dqn_engine* e, std::vector<RSeg>& segs, const char* name);
const char* pname(dqn_engine* e, const char* op, int kind, int i);
int build_program(dqn_engine* e);
// engine_envs.hip
void free_envs(dqn_engine* e);
