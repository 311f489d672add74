/*
 * dqn_mi355x.h -- C ABI of libdqn_mi355x.so, the MI355X-native (gfx950) engine for
 * the DeepQLearning.jl hot path: PrioritizedReplayBuffer + batch_train!.
 *
 * The reference (JuliaPOMDP/DeepQLearning.jl v0.7.1) has NO FFI: its extension
 * mechanism is Julia multiple dispatch.  Each entry point below replaces one
 * dispatch seam of the reference; the `ccall` stubs a maintainer adds are in
 * INTEGRATION.md and deepqlearning.jl_amd/julia/DeepQLearningMI355X.jl.
 *
 * Conventions
 *  - every function returns 0 on success, <0 on error; dqn_last_error() gives the
 *    message (thread-local).  Reference errors are thrown Strings
 *    ("DeepQLearningError: ...", src/solver.jl:46, src/dueling.jl:47) and @assert
 *    (src/prioritized_experience_replay.jl:66,78,83-84,90); the shim re-throws.
 *  - host arrays are taken AS JULIA LAYS THEM OUT (column-major, batch last):
 *      obs (W,H,C)  == C float[C][H][W];    batch (W,H,C,B) == float[B][C][H][W]
 *      Dense weight (out,in) == float[in][out];  Conv weight (kw,kh,cin,cout) ==
 *      float[cout][cin][kh][kw], UN-flipped (the engine applies NNlib's true
 *      convolution itself);  Q-values (nA,B) == float[B][nA].
 *  - flat parameter vectors are in Flux.params order: base, val, adv streams
 *    (src/dueling.jl:2-6,13), per layer weight then bias.
 *  - action indices are 0-BASED at the ABI (the Julia shim subtracts 1 from the
 *    reference's 1-based index, src/solver.jl:84).
 *  - the caller owns every host pointer and may free it when the call returns.
 *  - an engine is NOT thread-safe (the reference is strictly sequential); distinct
 *    engines (one per GPU) are independent.
 *  - no torch / HIP types appear in any signature.
 */
#ifndef DQN_MI355X_H
#define DQN_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only these entry points are exported */
#endif

typedef struct dqn_engine dqn_engine_t;

enum { DQN_LAYER_DENSE = 0, DQN_LAYER_CONV = 1, DQN_LAYER_LSTM = 2 };
enum { DQN_ACT_IDENTITY = 0, DQN_ACT_RELU = 1, DQN_ACT_TANH = 2, DQN_ACT_SIGMOID = 3 };
enum { DQN_STREAM_BASE = 0, DQN_STREAM_VAL = 1, DQN_STREAM_ADV = 2 };
enum { DQN_OBS_F32 = 0, DQN_OBS_U8 = 1 }; /* u8: stored byte, consumed as (float)byte/255f0 (test/test_env.jl:59) */
enum { DQN_NET_ONLINE = 0, DQN_NET_TARGET = 1 };

/* One layer of solver.qnetwork (a Flux.Chain of Conv / Dense, src/solver.jl:2,
 * README.md:64-70) AFTER create_dueling_network (src/dueling.jl:36-58) has split
 * it: the shim passes base layers, then val layers, then adv layers.
 * flattenbatch (src/helpers.jl:6-8) is implicit between a Conv and a Dense. */
typedef struct {
    int32_t kind;   /* DQN_LAYER_* */
    int32_t act;    /* DQN_ACT_* */
    int32_t stream; /* DQN_STREAM_*; all BASE when dueling == 0 */
    int32_t n_in, n_out;               /* Dense(in,out); LSTM(in,out) = Flux Recur(LSTMCell): params Wi (4out,in), Wh (4out,out), b (4out), state0 h0, c0 */
    int32_t cin, cout, kh, kw, sh, sw; /* Conv((kh,kw), cin=>cout; stride=(sh,sw)), pad 0 */
} dqn_layer_desc;

/* Summation-order plan of one layer: the K dimension of each contraction is cut
 * into chunks of `*_kc` elements; inside a chunk products are accumulated as ONE
 * k-ascending fp32 fma chain starting at +0 (== gfx950 fp32 MFMA numerics), and
 * chunk sums are added in ascending chunk order.  0 = no split.  The plan only
 * fixes rounding order; DESIGN.md section 4 states it. */
typedef struct {
    int32_t fwd_kc; /* forward:  K = n_in            | cin*kh*kw           */
    int32_t dx_kc;  /* dX:       K = n_out           | conv: RAW kernel taps (ky*kw + kx ascending) per chunk; each chunk chains its VALID taps, co innermost */
    int32_t dw_kc;  /* dW, db:   K = batch           | out_positions*batch (position-major, sample-minor); conv: a multiple of batch_size, or -- when batch_size % 32 == 0 -- of 32 */
} dqn_layer_plan;

/* DeepQLearningSolver fields that reach the hot path (src/solver.jl:1-28) plus the
 * PrioritizedReplayBuffer constructor defaults (src/prioritized_experience_replay.jl:39-45;
 * NB the reference never forwards the solver's alpha/beta/epsilon, src/solver.jl:185). */
typedef struct {
    int32_t batch_size;        /* solver.batch_size (32) */
    int32_t n_actions;         /* length(actions(env)) */
    int32_t obs_c, obs_h, obs_w; /* obs as float[C][H][W]; vector obs of length n: (n,1,1) */
    int32_t obs_dtype;         /* DQN_OBS_* replay storage type */
    float   learning_rate;     /* solver.learning_rate (1f-4) -> Adam eta */
    double  adam_beta1, adam_beta2, adam_eps; /* Flux Adam defaults 0.9, 0.999, 1e-8 */
    int32_t adam_f64_scalars;  /* 1: Flux 0.14 semantics (Float64 eta/beta/eps, evaluate in f64, round on store) */
    float   gamma;             /* discount (src/solver.jl:208, helpers.jl:83-85) */
    int32_t double_q;          /* solver.double_q */
    int32_t dueling;           /* solver.dueling */
    int32_t prioritized_replay;/* solver.prioritized_replay: 0 => priorities are not updated (src/solver.jl:231) */
    int64_t buffer_size;       /* solver.buffer_size */
    float   prio_alpha, prio_beta, prio_eps; /* 0.6, 0.4, 1e-3 */
    uint64_t seed;             /* sampler key (Philox4x32-10 counter RNG) */
    int32_t use_graph;         /* 1: replay the train step from a captured hipGraph */
    int32_t use_mfma;          /* 1: fp32 MFMA kernels where shapes allow (bit-identical to the VALU path) */
    int32_t recurrence;        /* solver.recurrence: DRQN on an EpisodeReplayBuffer (src/solver.jl:12,182-183,239-287) */
    int32_t trace_length;      /* solver.trace_length (40) */
    int32_t sample_distinct;   /* 0 (default): stratified sum-tree draws, duplicates possible within a batch.  1: B DISTINCT indices like the reference's
                                * sample(rng, 1:n, Weights(p), B, replace=false) (...replay.jl:85): after the stratified draws, every later duplicate is redrawn
                                * from the tree with the mass of all taken leaves removed (successive sampling on the residual priorities) */
    int32_t reserved[3];
} dqn_hparams;

const char* dqn_last_error(void);
int dqn_version(void);

/* Fill `hp` with the reference defaults (src/solver.jl:3-27, ...replay.jl:42-45). */
int dqn_hparams_default(dqn_hparams* hp);

/* Host-only (no GPU needed): the default summation-order plan the engine would use
 * for this network, one entry per layer. */
/* Version of the plan SEMANTICS (what a dqn_layer_plan value means for the rounding order) and of dqn_plan_default's choices.  Results are bit-identical only
 * between builds of the same version under the same plan; a checkpoint records both and a resume across versions is refused.
 *   1  rounds 1-2      2  round 3: conv dx_kc = RAW taps per chunk, conv dw_kc may be sample-granular      3  round 4: dw_kc < 0 = column-group chunks
 *   (recurrent networks; the default for the networks the fused recurrent step covers) */
#define DQN_PLAN_VERSION 3
int dqn_plan_version(void);
int dqn_plan_default(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp,
                     dqn_layer_plan* plan_out);

/* solve(): src/solver.jl:40-57 builds replay + (dueling) network + policy; the
 * engine owns their device state.  plan_or_null overrides the default plan. */
int dqn_engine_create(const dqn_layer_desc* layers, int n_layers, const dqn_hparams* hp,
                      const dqn_layer_plan* plan_or_null, int device, dqn_engine_t** out);
int dqn_engine_destroy(dqn_engine_t* e);
int dqn_engine_get_plan(dqn_engine_t* e, dqn_layer_plan* plan_out /* n_layers entries */);
int dqn_n_params(dqn_engine_t* e, size_t* n);
/* bytes per element of the batch arena the train step gathers into: 4 (fp32 X0[f][2B]) or 1 (byte arena: u8 replay whose first layer converts
 * byte/255 inside its tile loads).  Decided when the step program is built (first train call); measurement tools price the gather with it. */
int dqn_batch_arena_elem_bytes(dqn_engine_t* e, int* bytes);

/* Flux.params(active_q) / Flux.loadparams! (src/solver.jl:143-144, :292, :314-315). */
int dqn_set_params(dqn_engine_t* e, int which, const float* flat, size_t n);
int dqn_get_params(dqn_engine_t* e, int which, float* flat, size_t n);
/* target_q <- active_q every target_update_freq steps (src/solver.jl:142-145). */
int dqn_sync_target(dqn_engine_t* e);
/* Adam state (no reference equivalent: the reference cannot resume an optimizer). */
int dqn_get_adam_state(dqn_engine_t* e, float* m, float* v, double* beta_pow /*[2]*/, size_t n);
int dqn_set_adam_state(dqn_engine_t* e, const float* m, const float* v, const double* beta_pow, size_t n);

/* add_exp!(r, DQExperience(s,a,r,sp,done), td_err) for n transitions
 * (src/prioritized_experience_replay.jl:65-74; n > 1 serves vectorised envs).
 * s, sp: n rows of obs in the replay storage dtype; td_err may be NULL => |r| (:65). */
int dqn_replay_add(dqn_engine_t* e, const void* s, const int32_t* a, const float* r,
                   const void* sp, const uint8_t* done, const float* td_err, int n);
int dqn_replay_size(dqn_engine_t* e, int64_t* cur, int64_t* cap); /* _curr_size, max_size (:61-63) */
int dqn_replay_get_priorities(dqn_engine_t* e, float* prio, int64_t n); /* r._priorities[1:n] */

/* StatsBase.sample(r) (:82-87).  The reference draws B distinct indices with
 * probability proportional to priority (StatsBase A-ExpJ, O(n)); the engine uses a
 * stratified sum-tree descent (O(B log n)); parity is defined on GIVEN indices via
 * the two calls below.  idx_out (host, B int64, 0-based) may be NULL. */
int dqn_replay_sample(dqn_engine_t* e, int64_t* idx_out);
/* get_batch(r, idx) (:89-104): the deterministic seam.  Any output may be NULL.
 * s, sp: float[B][C][H][W]; a: int32[B]; r, done, w: float[B]. */
int dqn_replay_get_batch(dqn_engine_t* e, const int64_t* idx, float* s, int32_t* a, float* r,
                         float* sp, float* done, float* w);
/* update_priorities!(r, idx, td) (:76-80). */
int dqn_update_priorities(dqn_engine_t* e, const int64_t* idx, const float* td, int n);

/* batch_train!(solver, env, policy, optimizer, target_q, replay) (src/solver.jl:191-236):
 * sample (or use idx) -> gather + IS weights -> double-Q Bellman target -> Huber(w*td)/B ->
 * backward -> max-abs grad norm -> Adam -> priority update.  Returns (loss_val, grad_norm)
 * like the reference (:235).  All outputs optional; with every output NULL the call only
 * enqueues work on the engine's stream (no host sync). */
int dqn_train_step(dqn_engine_t* e, const int64_t* idx_or_null, float* loss, float* grad_norm,
                   float* td_out /* B */);
/* The same step without waiting for it: returns as soon as the step is enqueued on the engine's stream.  *ticket names the (loss, grad_norm) record the step's
 * last launch publishes into a mapped pinned host ring (no fold launch, no D2H copy, no stream synchronize).  The reference reads batch_train!'s return values
 * only every log_freq env steps (src/solver.jl:154-167); the shim's dqn_train! loop fetches them there.  dqn_train_step(e, idx, &loss, &gn, NULL) itself is
 * this call followed by dqn_step_scalars(e, ticket, 1, ...).  Single-device feed-forward engines. */
int dqn_train_step_async(dqn_engine_t* e, const int64_t* idx_or_null, uint64_t* ticket);
/* (loss, grad_norm) of the step that returned `ticket` -- one of the 64 newest.  wait != 0: blocks (host spin on the record, stream synchronize as a fallback)
 * until it has arrived.  wait == 0: returns at once; *published = ticket and the outputs are written if the record was there, *published = 0 otherwise.
 * Device-side assertion failures of that step (AssertionError: all(new_priorities .> 0f0)) are reported here. */
int dqn_step_scalars(dqn_engine_t* e, uint64_t ticket, int wait, float* loss, float* grad_norm, uint64_t* published);
/* run `n_steps` sampled train steps back to back; returns the last step's scalars.  Bit-identical to n_steps calls of dqn_train_step(e, NULL, ...)
 * (tests/test_gpu_parity.py::test_train_steps_pipelined_gather_bit_exact), but inside the call nothing else can touch the replay, so step i's last
 * launch already gathers step i+1's batch and steps 2..n run without a sample / gather launch (prioritized replay; f32 observations, or u8
 * observations on the byte arena; any batch size; replicas too -- the gather is rank-local). */
int dqn_train_steps(dqn_engine_t* e, int n_steps, float* loss, float* grad_norm);

/* Results of the last train step (parity checks; not on the hot path). */
int dqn_get_last_q(dqn_engine_t* e, float* q_on_s, float* q_on_sp, float* q_tg_sp /* each [B][nA] */,
                   int32_t* best_a /* B */, float* q_targets /* B */);
int dqn_get_last_indices(dqn_engine_t* e, int64_t* idx /* B */);
int dqn_get_grads(dqn_engine_t* e, float* flat, size_t n); /* Flux.params order/layout */

/* ---- DRQN (solver.recurrence = true): EpisodeReplayBuffer (src/episode_replay.jl) and batch_train!(..., ::EpisodeReplayBuffer)
 * (src/solver.jl:239-287).  buffer_size counts EPISODES.  Only the first trace_length transitions of an episode are ever
 * read by the reference's sampler (episode_replay.jl:82-92 copies the episode PREFIX), so only those are kept. */
/* add_exp!(r::EpisodeReplayBuffer, exp) (:46-52): append n transitions to the open episode; an episode is stored when done. */
int dqn_episode_add(dqn_engine_t* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done, int n);
/* add_episode!(r, ep) (:54-60) for populate_replay_buffer!/generate_episode (:97-130): store the open episode now. */
int dqn_episode_commit(dqn_engine_t* e);
int dqn_episode_count(dqn_engine_t* e, int64_t* cur, int64_t* cap);
/* sample(r::EpisodeReplayBuffer) (:71-95) for GIVEN draws: ep_idx[B] distinct episodes (0-based), ep_start[B] in [0,len).
 * Outputs (all optional): s, sp: float[T][B][C][H][W]; a: int32[T][B] (0-based); r, done: float[T][B]; mask: int32[T][B]. */
int dqn_episode_get_batch(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* s, int32_t* a, float* r,
                          float* sp, float* done, int32_t* mask);
/* recurrent batch_train! (src/solver.jl:239-287): targets with hidden state carried over sp[1..T] for both nets, BPTT over
 * s[1..T] of the masked Huber loss / B / T, max-abs grad norm, Adam; no IS weights, no priority update.
 * ep_idx/ep_start NULL => the engine draws them (uniform without replacement; start uniform in [0,len)). */
int dqn_train_step_drqn(dqn_engine_t* e, const int64_t* ep_idx, const int32_t* ep_start, float* loss, float* grad_norm);
/* Recur state of the POLICY network (src/policy.jl:32-34 resetstate!, src/helpers.jl:61-79 hiddenstates/sethiddenstates!):
 * dqn_forward / dqn_greedy_action advance it for recurrent networks (n streams, one per observation row). */
int dqn_reset_state(dqn_engine_t* e);
/* n = sum over LSTM layers of 2 x out x streams floats: per layer h then c, each [out][streams] (streams = observations per dqn_forward call, 1 before the first) */
int dqn_get_hidden(dqn_engine_t* e, float* hc, size_t n);
int dqn_set_hidden(dqn_engine_t* e, const float* hc, size_t n);

/* NNPolicy: actionvalues / action / value (src/policy.jl:38-64) for n observations
 * (n = 1 in the reference; n > 1 serves vectorised envs).  obs: float[n][C][H][W]. */
int dqn_forward(dqn_engine_t* e, int which, const float* obs, int n, float* q_out /* [n][nA] */);
int dqn_greedy_action(dqn_engine_t* e, const float* obs, int n, int32_t* a_out /* first-max tie rule */);

/* ---- vectorised environments on the device (north star: "vectorised parallel environments (SimpleGridWorld /
 * image-obs MDPs)").  n lock-stepped copies of a built-in MDP live in HBM; one vector step is one iteration of the dqn_train!
 * loop (src/solver.jl:82-145) for all copies: eps-greedy action from the online net (POMDPTools EpsGreedyPolicy:
 * rand < eps ? random action : greedy), act!, observe, add_exp!(replay, exp, |r|) straight into the replay ring, episode
 * bookkeeping (reset on done or max_episode_length), every train_freq vector steps one batch_train!, every
 * target_update_freq a target sync.  No observation crosses PCIe.  Randomness: Philox4x32-10 keyed by seed, counter =
 * (global vector step, env, purpose), so the CPU twin reproduces every trajectory bit for bit. */
enum { DQN_ENV_TESTMDP = 0, DQN_ENV_GRIDWORLD = 1 };
typedef struct {
    int32_t kind;               /* DQN_ENV_* */
    int32_t n_envs;
    int32_t max_episode_length; /* solver.max_episode_length (100) */
    uint64_t seed;
    /* TestMDP (test/test_env.jl:10-87): obs = stack of o_stack of three fixed H x W integer images / 255, chosen by the last
     * actions; rewards [-0.1, 0, 0.1][sp[end]] x (-10 if s[end] == 2); terminal at t >= max_time; 4 actions */
    int32_t o_stack, max_time;
    const uint8_t* images;      /* uint8[3][H*W]: bad, normal, good; H, W from hparams.obs_h/obs_w; obs_c == o_stack */
    /* SimpleGridWorld (POMDPModels defaults, third-party; recalled): size_x x size_y grid, 4 actions up/down/left/right,
     * reward cells are terminal, intended move with probability tprob, obs = Float32[x, y] */
    int32_t size_x, size_y; float tprob; int32_t n_reward_cells; int32_t reward_xy[8][2]; float reward_val[8];
} dqn_env_spec;
typedef struct {
    int32_t train_freq, target_update_freq;     /* solver.train_freq (4), solver.target_update_freq (500); 0 = never */
    float eps_start, eps_stop, eps_steps;       /* LinearDecaySchedule(start, stop, steps) of the exploration policy */
    int32_t cadence_env_steps;                  /* (r05; occupies what was padding: offsets and size unchanged)  0: train_freq / target_update_freq count VECTOR steps -- one
                                                 * train step per train_freq steps of all n copies.  1: they count ENV steps as the reference's loop does (src/solver.jl:136-145):
                                                 * after vector step t, floor(t*n / train_freq) - floor((t-1)*n / train_freq) train steps run back to back (8 per vector step
                                                 * for 32 copies at train_freq = 4) and the target network is synced whenever a multiple of target_update_freq env steps
                                                 * was crossed.  The n transitions of a vector step are added BEFORE its train steps (the reference interleaves them). */
    int64_t t0;                                 /* global index of the first vector step of this call (t counts from 1) */
} dqn_rollout_cfg;
typedef struct { int64_t episodes; double reward_sum; int64_t train_steps; float last_loss, last_grad_norm; } dqn_rollout_stats;
int dqn_envs_create(dqn_engine_t* e, const dqn_env_spec* spec);
int dqn_envs_reset(dqn_engine_t* e);
int dqn_rollout(dqn_engine_t* e, int n_vector_steps, const dqn_rollout_cfg* cfg, dqn_rollout_stats* stats_or_null);
/* basic_evaluation (src/evaluation_policy.jl:17-42; cadence src/solver.jl:101-122) batched on the device: n_eval
 * further copies of the MDP given to dqn_envs_create each run ONE greedy episode (while !done && step <= max_episode_length); returns
 * the average undiscounted return (Float64 sum of the Float32 rewards, as r_tot) and the average step count. */
int dqn_evaluate(dqn_engine_t* e, int n_eval, int max_episode_length, uint64_t seed, double* avg_reward, double* avg_steps);
/* inspection (parity tests): current observations float[n][C][H][W], last actions int32[n], last rewards float[n], last dones uint8[n] */
int dqn_envs_peek(dqn_engine_t* e, float* obs, int32_t* actions, float* rewards, uint8_t* dones);
/* inspection (parity tests, which must know WHICH schedule they compared): n_envs of the training set; fused_tail = 1 when the acting step of that set runs its tail as one
 * launch (act_head.hip: reduce + heads + Q / argmax + eps-greedy + act! + add_exp!'s per-experience part), 0 for the general four-launch tail.  Builds the acting program
 * (as dqn_rollout would) if it does not exist yet. */
int dqn_envs_info(dqn_engine_t* e, int* n_envs, int* fused_tail);

/* ---- checkpoint / resume (absent in the reference, which only saves the best network: src/solver.jl:290-318).
 * Together with dqn_get/set_params, dqn_get/set_adam_state these make a run resumable bit for bit: the replay in its storage type
 * (rows `first .. first+n-1` of the ring, slot order), stored priorities, ring cursor, and the sampler / optimizer counters. */
typedef struct { int64_t size, widx; uint64_t sample_ctr, train_steps; } dqn_counters;
int dqn_replay_export(dqn_engine_t* e, int64_t first, int64_t n, void* s, void* sp, int32_t* a, float* r, uint8_t* done, float* priorities);
/* replaces the whole replay: n <= capacity transitions go to slots 0..n-1 with the given priorities (NOT td errors); the sum-tree is rebuilt */
int dqn_replay_import(dqn_engine_t* e, int64_t n, const void* s, const void* sp, const int32_t* a, const float* r, const uint8_t* done, const float* priorities);
/* ... and for the episode replay of a recurrent engine (src/episode_replay.jl; config 4): episodes first..first+n-1 in slot order, rows
 * [n][trace_length][obs] float (only the first trace_length transitions of an episode can ever be sampled, :82-92, so only they are stored),
 * a / r / done [n][trace_length], len[n] = the episode's true length.  dqn_get/set_counters then mean: size = episodes, widx = ring cursor,
 * sample_ctr = the host sampler's draw counter.  An episode still being collected is not part of a checkpoint. */
int dqn_episode_export(dqn_engine_t* e, int64_t first, int64_t n, float* s, float* sp, int32_t* a, float* r, uint8_t* done, int32_t* len);
int dqn_episode_import(dqn_engine_t* e, int64_t n, const float* s, const float* sp, const int32_t* a, const float* r, const uint8_t* done, const int32_t* len);
int dqn_get_counters(dqn_engine_t* e, dqn_counters* out);
int dqn_set_counters(dqn_engine_t* e, const dqn_counters* in);   /* size must equal the imported n; widx < capacity */

/* data-parallel replicas (no reference equivalent): after dqn_comm_init every train step exchanges gradients between backward and Adam
 * with ONE collective over RCCL (dlopen'ed librccl) on the engine's stream -- an all-gather of the wide dense layers' operands and of the
 * small gradients, or an all-reduce of the flat gradient (no qualifying layer, recurrent engines, DQN_DP_ALLREDUCE=1);
 * Adam then applies the sum * 1/world, identically on every rank.  unique_id: 128 bytes from dqn_comm_unique_id on rank 0.
 * Every rank must issue the same sequence of train steps. */
int dqn_comm_unique_id(void* id128);
int dqn_comm_init(dqn_engine_t* e, const void* id128, int rank, int world);
/* what the communicator ITSELF reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) next to what the engine was told: a bench line can prove how
 * many ranks RCCL saw.  No communicator: rccl_nranks = 0, rccl_rank = rccl_device = -1.  exchange: 0 none, 1 all-gather of the wide dense layers' operands +
 * small gradients, 2 all-reduce of the flat gradient, -1 not decided yet (the step program is built at the first train step). */
typedef struct dqn_comm_info_t { int32_t rccl_nranks, rccl_rank, rccl_device, engine_world, engine_rank, sim_world, exchange, dp_overlap; } dqn_comm_info_t;
int dqn_comm_info(dqn_engine_t* e, dqn_comm_info_t* out);
/* bytes ONE rank contributes to the step's collective (the packed block of the all-gather: X | dpre of the wide dense layers + every other gradient range; or the flat
 * gradient of the all-reduce); 0 without a communicator / before the step program exists (it is built by the first train step) */
int dqn_comm_exchange_bytes(dqn_engine_t* e, int64_t* bytes_per_rank);

/* TEST HOOK for the exchange above on ONE GPU: an engine created with the environment variable DQN_SIM_WORLD=k plays k ranks; this call runs
 * one data-parallel step with k DISTINCT batches idx[k][B] (rank r's packed block lands in slot r of the gathered buffer, as ncclAllGather
 * would deliver it).  loss[k]: per-rank losses; td_out[k][B].  Equivalent single-device step: the concatenated batch of k*B samples. */
int dqn_sim_ranks_step(dqn_engine_t* e, const int64_t* idx, float* loss, float* grad_norm, float* td_out);

/* raw access for harnesses that time kernels on the engine's own stream. */
int dqn_stream_sync(dqn_engine_t* e);
int dqn_stream_handle(dqn_engine_t* e, void** hip_stream);
/* per-kernel timing of the last dqn_profile_step call: names (static strings) and
 * milliseconds measured with HIP events on the engine stream. */
int dqn_profile_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries);
/* as dqn_profile_step, but TWO train steps run and the second is timed as a MIDDLE step of dqn_train_steps(n): inside that call step i's last
 * launch (Adam) already gathers step i+1's batch, so a middle step has no gather launch of its own (where dqn_train_steps pipelines it;
 * otherwise both steps are plain ones).  Results of dqn_train_steps(n) are bit-identical to n dqn_train_step calls either way. */
int dqn_profile_steady_step(dqn_engine_t* e, int max_entries, const char** names, float* ms, int* n_entries);


#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DQN_MI355X_H */
