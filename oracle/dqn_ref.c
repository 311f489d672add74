/*
 * oracle/dqn_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * "libdqn_ref": a plain-C, CPU, canonical-summation-order restatement of the
 * DeepQLearning.jl hot path, used (a) as the BIT-EXACT checker of the HIP engine
 * (TD loss, greedy actions, parameters after N steps) and (b) as the timed CPU
 * baseline ("port") in bench.py.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product never does.
 *
 * PARITY UNPINNED BY THE REFERENCE (no Julia in the image, no golden vectors in
 * the reference's tests).  This twin is pinned against
 * oracle/dqn_oracle.py (NumPy fp64, itself cross-checked with torch autograd) to
 * fp32 round-off, and against the committed fixtures in tests/golden/.
 *
 * What it restates (reference file:line):
 *   add_exp!            src/prioritized_experience_replay.jl:65-74
 *   update_priorities!  src/prioritized_experience_replay.jl:76-80
 *   get_batch           src/prioritized_experience_replay.jl:89-104
 *   DuelingNetwork      src/dueling.jl:8-11
 *   huber_loss          src/helpers.jl:14-19
 *   globalnorm          src/helpers.jl:38-46
 *   batch_train!        src/solver.jl:191-236
 *   NNPolicy forward    src/policy.jl:38-64
 *   Flux Dense/Conv/Adam (third-party; semantics recalled)
 *
 * Canonical order: every contraction is a k-ascending fp32
 * fma chain from +0 per chunk of the layer plan, chunk sums added in ascending
 * order, then bias, then activation.  This is exactly what gfx950's fp32 MFMA
 * computes, which is why the GPU can match this file bit for bit.
 * Build: see oracle/Makefile (-O2 -ffp-contract=off -mavx2 -mfma -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/dqn_mi355x.h"

#define MAXL 32

typedef struct {
    int kind, act, stream;
    int K, N;                 /* contraction dims of the forward GEMM */
    int cin, cout, kh, kw, sh, sw, ih, iw, oh, ow;
    int in_feat, out_feat;
    int src;                  /* producing layer index, -1 = observation */
    size_t w_off, b_off;      /* offsets in the flat parameter vector */
    size_t wh_off, h0_off, c0_off; int H; /* LSTM: Flux.params order Wi (w_off), Wh, b (b_off), h0, c0 */
    dqn_layer_plan plan;
} RLayer;

typedef struct ref_engine {
    int nl; RLayer L[MAXL];
    dqn_hparams hp;
    int B, nA, obs_elems;
    int last_base, last_val, last_adv; /* layer indices (-1 if none) */
    size_t P;
    float *p_on, *p_tg, *grad, *m, *v;
    double bp1, bp2;
    /* replay */
    int64_t cap, cap2, size, widx; uint64_t sample_ctr;
    float *s_f32, *sp_f32; uint8_t *s_u8, *sp_u8;
    int32_t* a; float* r; uint8_t* done; float* tree; /* tree[1]=root, leaves at cap2+i */
    /* step workspace */
    int64_t* idx; float *x0; /* [obs][2B] */
    float *act_on[MAXL], *act_tg[MAXL], *dact[MAXL]; /* dact: grad wrt layer output, [out_feat][B] */
    float *w_is, *rew, *donef, *td, *qon_s, *qon_sp, *qtg_sp, *ytarget; int32_t *abatch, *best;
    float loss, gnorm;
    int nthreads;
    /* DRQN: EpisodeReplayBuffer (src/episode_replay.jl) -- only the first T transitions of an episode are ever sampled */
    int T; int64_t ep_cap, ep_size, ep_widx, ep_cur_len;
    float *ep_s, *ep_sp; int32_t* ep_a; float* ep_r; uint8_t* ep_done; int32_t* ep_len;
    float *rx0, *racc_on[MAXL], *racc_tg[MAXL], *rdact[MAXL];            /* [feat][2TB] / [feat][TB] */
    float *gx_on[MAXL], *gx_tg[MAXL], *gates[MAXL], *cst[MAXL], *hprev[MAXL], *dgates[MAXL]; /* LSTM workspaces (online s-sequence keeps gates) */
    float *r_a_f, *r_r, *r_done, *r_mask; int32_t* r_a; uint64_t drqn_ctr;
    float* pol_h[MAXL]; float* pol_c[MAXL]; int pol_n;
    struct ref_envs* envs;
} ref_engine;

static char g_err[512];
const char* ref_last_error(void) { return g_err; }
#define FAIL(...) do { snprintf(g_err, sizeof g_err, __VA_ARGS__); return -1; } while (0)

/* ---------------------------------------------------------------- helpers */
static inline float act_f(float y, int act) {
    switch (act) {
    case DQN_ACT_RELU: return y > 0.0f ? y : 0.0f;
    case DQN_ACT_TANH: return (float)tanh((double)y);          /* f64-evaluated, rounded once */
    case DQN_ACT_SIGMOID: return (float)(1.0 / (1.0 + exp(-(double)y)));
    default: return y;
    }
}
static inline float dact_f(float dy, float y, int act) {
    switch (act) {
    case DQN_ACT_RELU: return y > 0.0f ? dy : 0.0f;
    case DQN_ACT_TANH: { float t = y * y; float u = 1.0f - t; return dy * u; }
    case DQN_ACT_SIGMOID: { float u = 1.0f - y; float t = y * u; return dy * t; }
    default: return dy;
    }
}
static inline float prio_f(float td_abs, float eps, float alpha) {
    /* (td + eps)^alpha, Julia Float32^Float32 evaluates through Float64 (...replay.jl:67,77) */
    float base = td_abs + eps;
    return (float)pow((double)base, (double)alpha);
}
static inline int nchunks(int K, int kc) { return (kc <= 0 || kc >= K) ? 1 : (K + kc - 1) / kc; }
static inline int chunk_len(int K, int kc) { return (kc <= 0 || kc >= K) ? K : kc; }

/* Philox4x32-10 */
static inline void philox(uint32_t k0, uint32_t k1, uint32_t c[4]) {
    for (int i = 0; i < 10; i++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* ---------------------------------------------------------------- plan */
/* Column-group dW chunks of recurrent networks (plan.dw_kc = -cg): the T*B sample columns of a dW / db contraction are cut by
 * BATCH column -- chunk g = columns {t*B + b : b in [g*cg, (g+1)*cg), t = 0..T-1} chained in ascending column order -- instead of into contiguous column
 * ranges: batch columns never interact before the gradient sum, so a group of cg columns is what ONE workgroup of the fused recurrent step owns.
 * This is the independent restatement of drqn_fused_cg (deepqlearning.jl_amd/csrc/common.h): the group size the default plan picks, 0 = not eligible. */
static int fused_cg(const dqn_layer_desc* d, int n, const dqn_hparams* hp) {
    if (!hp->recurrence) return 0;
    const int duel = hp->dueling ? 1 : 0;
    if (n != (duel ? 3 : 2) || d[0].kind != DQN_LAYER_LSTM || d[0].stream != DQN_STREAM_BASE) return 0;
    const int E = hp->obs_c * hp->obs_h * hp->obs_w, H = d[0].n_out, N = 4 * H, T = hp->trace_length, B = hp->batch_size, nA = hp->n_actions;
    if (d[0].n_in != E || H < 8 || H > 64 || H % 8 || T < 1 || T > 64 || nA > 16 || E > 512) return 0;
    if (!duel) { if (d[1].kind != DQN_LAYER_DENSE || d[1].stream != DQN_STREAM_BASE || d[1].n_in != H || d[1].n_out != nA) return 0; }
    else if (d[1].kind != DQN_LAYER_DENSE || d[1].stream != DQN_STREAM_VAL || d[1].n_in != H || d[1].n_out != 1 ||
             d[2].kind != DQN_LAYER_DENSE || d[2].stream != DQN_STREAM_ADV || d[2].n_in != H || d[2].n_out != nA) return 0;
    const int nset = hp->double_q ? 3 : 2, Ep = (E + 3) / 4 * 4, no = nA + duel;
    const long pp = (long)(E + 1) * N + (long)(H + 1) * N + 2 * H + N + (long)(H + 1) * no + 32;
    for (int c = 4; c >= 1; c >>= 1) {
        if (B % c || nset * 4 * H * c > 1024) continue;
        const long lds = 2 * pp + 2L * T * c * Ep + 4L * T * c + (long)nset * T * H * c + (long)nset * 7 * H * c + (long)T * N * c + 3L * T * H * c +
                         (long)(nset + 1) * T * c * (no + 1) + (long)T * H * c + 2L * H * c + (long)H * (N + 4) + (long)nset * ((T * c + 15) / 16 * 16) * N + 64;
        if (lds <= 36000) return c;
    }
    return 0;
}
int ref_plan_version(void) { return DQN_PLAN_VERSION; }
int ref_plan_default(const dqn_layer_desc* d, int n, const dqn_hparams* hp, dqn_layer_plan* out) {
    /* independent restatement of the rule in DESIGN.md section 4 (the tests check it
     * equals dqn_plan_default of the product) */
    int c = hp->obs_c, h = hp->obs_h, w = hp->obs_w; int bc = c, bh = h, bw = w; int seen_val = 0, seen_adv = 0, has_base = 0, rec = 0;
    for (int i = 0; i < n; i++) if (d[i].kind == DQN_LAYER_LSTM) rec = 1;
    for (int i = 0; i < n; i++) {
        int join = 0;      /* first layer of a dueling stream: its dX meets the other stream's at the base output */
        if (d[i].stream == DQN_STREAM_VAL && !seen_val) { seen_val = 1; c = bc; h = bh; w = bw; join = has_base; }
        if (d[i].stream == DQN_STREAM_ADV && !seen_adv) { seen_adv = 1; c = bc; h = bh; w = bw; join = has_base; }
        int K, posB;
        if (d[i].kind == DQN_LAYER_CONV) {
            int oh = (h - d[i].kh) / d[i].sh + 1, ow = (w - d[i].kw) / d[i].sw + 1;
            K = d[i].cin * d[i].kh * d[i].kw; c = d[i].cout; h = oh; w = ow; posB = 1;
        } else { K = d[i].n_in; c = d[i].n_out; h = 1; w = 1; posB = 0; }
        if (d[i].stream == DQN_STREAM_BASE) { bc = c; bh = h; bw = w; has_base = 1; }
        int B = hp->batch_size; const int nout = d[i].kind == DQN_LAYER_LSTM ? 4 * d[i].n_out : d[i].n_out;
        out[i].fwd_kc = 0;
        if (K > 1024 && B < 128) { int s = (K + 511) / 512; int kc = (K + s - 1) / s; kc = (kc + 3) / 4 * 4; out[i].fwd_kc = kc; }
        else if (d[i].kind == DQN_LAYER_DENSE && d[i].n_out < 16 && K >= 128) out[i].fwd_kc = 32;
        out[i].dx_kc = 0;
        if (d[i].kind != DQN_LAYER_CONV && nout > 512) out[i].dx_kc = 256;
        else if (d[i].kind == DQN_LAYER_DENSE && B <= 64 && nout >= 128) {      /* small batches: 4 concurrent chains per output tile (2 per stream at the dueling join) */
            int S = join ? 2 : 4; int kc = ((nout + S - 1) / S + 31) / 32 * 32; if (kc < nout) out[i].dx_kc = kc;
        } else if (d[i].kind == DQN_LAYER_CONV && B <= 64) {                    /* RAW taps per chunk so that an interior position has <= 4 non-empty chunks */
            int valid = ((d[i].kh + d[i].sh - 1) / d[i].sh) * ((d[i].kw + d[i].sw - 1) / d[i].sw); int raw = ((valid + 3) / 4) * d[i].sw;
            if (raw < d[i].kh * d[i].kw && valid * d[i].cout > 256) out[i].dx_kc = raw;      /* ... where an element's chain is longer than eight K tiles */
        }
        out[i].dw_kc = 0;
        if (d[i].kind == DQN_LAYER_DENSE && d[i].n_out < 16 && B >= 128 && rec) out[i].dw_kc = 64;      /* recurrent networks only */
        if (posB) { int mrows = (K + 63) / 64; int st = (512 + mrows - 1) / mrows; int ppc = (h * w) / st; if (ppc < 1) ppc = 1;
            /* small batches: an even number of 32-sample K tiles per chunk of 3+ tiles, while >= 3/4 of the 512 target workgroups remain */
            if (B % 32 == 0 && B < 128) { int kt = ppc * (B / 32); if (kt >= 3 && (kt & 1) && (B / 32) % 2 == 1 && ((h * w + ppc) / (ppc + 1)) * mrows * 4 >= 3 * 512) ppc += 1; }
            /* small batches, every conv layer but the network's first (its dW shares a launch with its dX): three K tiles per chunk */
            if (B % 32 == 0 && B < 128 && i > 0) { int tpp = B / 32; ppc = (3 + tpp - 1) / tpp; if (ppc > h * w) ppc = h * w; }
            out[i].dw_kc = ppc * B;
            if (B >= 128 && B % 32 == 0) {      /* large batches: sample-granular chunks, <= 1024 workgroups */
                int KK = h * w * B; int ch = 1024 / mrows; if (ch < 1) ch = 1;
                int kc = ((KK + ch - 1) / ch + 31) / 32 * 32; out[i].dw_kc = kc < KK ? kc : 0;
            } }
    }
    const int cg = fused_cg(d, n, hp);      /* recurrent networks the fused column-parallel step covers: column-group dW chunks */
    if (cg) for (int i = 0; i < n; i++) out[i].dw_kc = -cg;
    return 0;
}

/* ---------------------------------------------------------------- create */
int ref_create(const dqn_layer_desc* d, int n, const dqn_hparams* hp, const dqn_layer_plan* plan, ref_engine** out) {
    if (n <= 0 || n > MAXL) FAIL("bad layer count %d", n);
    if (hp->recurrence && hp->obs_dtype == DQN_OBS_U8) FAIL("DeepQLearningError: obs_dtype = u8 is not supported with recurrence = true (the episode replay stores Float32 rows, src/episode_replay.jl:3-20)");
    ref_engine* e = (ref_engine*)calloc(1, sizeof *e);
    e->nl = n; e->hp = *hp; e->B = hp->batch_size; e->nA = hp->n_actions;
    e->obs_elems = hp->obs_c * hp->obs_h * hp->obs_w;
    e->last_base = e->last_val = e->last_adv = -1;
    dqn_layer_plan defp[MAXL];
    if (!plan) { ref_plan_default(d, n, hp, defp); plan = defp; }
    /* column-group dW chunks (dw_kc = -cg): recurrent engines only, every layer alike, cg divides the batch */
    for (int i = 0; i < n; i++) if (plan[i].dw_kc < 0) {
        if (!hp->recurrence) { free(e); FAIL("plan: dw_kc < 0 (column-group chunks) needs recurrence = true"); }
        for (int j = 0; j < n; j++) if (plan[j].dw_kc != plan[i].dw_kc) { free(e); FAIL("plan: column-group dw_kc must be the same for every layer"); }
        if (hp->batch_size % (-plan[i].dw_kc)) { free(e); FAIL("plan: column group %d does not divide batch_size %d", -plan[i].dw_kc, hp->batch_size); }
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        RLayer* L = &e->L[i];
        L->kind = d[i].kind; L->act = d[i].act; L->stream = d[i].stream; L->plan = plan[i];
        int prev;
        if (d[i].stream == DQN_STREAM_BASE) prev = e->last_base;
        else if (d[i].stream == DQN_STREAM_VAL) prev = e->last_val >= 0 ? e->last_val : e->last_base;
        else prev = e->last_adv >= 0 ? e->last_adv : e->last_base;
        L->src = prev;
        int c, h, w;
        if (prev < 0) { c = hp->obs_c; h = hp->obs_h; w = hp->obs_w; }
        else if (e->L[prev].kind == DQN_LAYER_CONV) { c = e->L[prev].cout; h = e->L[prev].oh; w = e->L[prev].ow; }
        else { c = e->L[prev].out_feat; h = 1; w = 1; }
        L->in_feat = c * h * w;
        if (L->kind == DQN_LAYER_CONV) {
            L->cin = d[i].cin; L->cout = d[i].cout; L->kh = d[i].kh; L->kw = d[i].kw; L->sh = d[i].sh; L->sw = d[i].sw;
            if (L->cin != c) { free(e); FAIL("layer %d: conv cin %d != incoming channels %d", i, L->cin, c); }
            L->ih = h; L->iw = w; L->oh = (h - L->kh) / L->sh + 1; L->ow = (w - L->kw) / L->sw + 1;
            L->K = L->cin * L->kh * L->kw; L->N = L->cout; L->out_feat = L->cout * L->oh * L->ow;
        } else if (L->kind == DQN_LAYER_LSTM) {
            if (d[i].n_in != L->in_feat) { free(e); FAIL("layer %d: LSTM n_in %d != incoming features %d", i, d[i].n_in, L->in_feat); }
            if (!hp->recurrence) { free(e); FAIL("DeepQLearningError: you passed in a recurrent model but recurrence is set to false"); }
            if (d[i].stream != DQN_STREAM_BASE) { free(e); FAIL("LSTM layers are supported in the base chain only"); }
            L->H = d[i].n_out; L->K = d[i].n_in; L->N = 4 * L->H; L->out_feat = L->H; L->act = DQN_ACT_IDENTITY;
        } else {
            if (d[i].n_in != L->in_feat) { free(e); FAIL("layer %d: dense n_in %d != incoming features %d", i, d[i].n_in, L->in_feat); }
            L->K = d[i].n_in; L->N = d[i].n_out; L->out_feat = L->N;
        }
        if (L->kind == DQN_LAYER_LSTM) {   /* Flux.params order: Wi, Wh, b, h0, c0 */
            L->w_off = off; off += (size_t)L->K * L->N; L->wh_off = off; off += (size_t)L->H * L->N; L->b_off = off; off += L->N;
            L->h0_off = off; off += L->H; L->c0_off = off; off += L->H;
        } else { L->w_off = off; off += (size_t)L->K * L->N; L->b_off = off; off += L->N; }
        if (d[i].stream == DQN_STREAM_BASE) e->last_base = i;
        else if (d[i].stream == DQN_STREAM_VAL) e->last_val = i; else e->last_adv = i;
    }
    e->P = off;
    if (hp->dueling) {
        if (e->last_val < 0 || e->last_adv < 0 || e->L[e->last_val].out_feat != 1 || e->L[e->last_adv].out_feat != e->nA) {
            free(e); FAIL("DeepQLearningError: the qnetwork provided is incompatible with dueling");
        }
    } else if (e->last_base < 0 || e->L[e->last_base].out_feat != e->nA) { free(e); FAIL("network output != n_actions"); }
    int B = e->B;
    e->p_on = calloc(e->P, 4); e->p_tg = calloc(e->P, 4); e->grad = calloc(e->P, 4); e->m = calloc(e->P, 4); e->v = calloc(e->P, 4);
    e->bp1 = hp->adam_beta1; e->bp2 = hp->adam_beta2;
    e->cap = hp->buffer_size; e->cap2 = 1; while (e->cap2 < e->cap) e->cap2 <<= 1;
    if (hp->obs_dtype == DQN_OBS_U8) { e->s_u8 = calloc((size_t)e->cap * e->obs_elems, 1); e->sp_u8 = calloc((size_t)e->cap * e->obs_elems, 1); }
    else { e->s_f32 = calloc((size_t)e->cap * e->obs_elems, 4); e->sp_f32 = calloc((size_t)e->cap * e->obs_elems, 4); }
    e->a = calloc(e->cap, 4); e->r = calloc(e->cap, 4); e->done = calloc(e->cap, 1); e->tree = calloc(2 * (size_t)e->cap2, 4);
    e->idx = calloc(B, 8); e->x0 = calloc((size_t)e->obs_elems * 2 * B, 4);
    for (int i = 0; i < n; i++) {
        e->act_on[i] = calloc((size_t)e->L[i].out_feat * 2 * B, 4);
        e->act_tg[i] = calloc((size_t)e->L[i].out_feat * B, 4);
        e->dact[i] = calloc((size_t)e->L[i].out_feat * B, 4);
    }
    e->w_is = calloc(B, 4); e->rew = calloc(B, 4); e->donef = calloc(B, 4); e->td = calloc(B, 4);
    e->qon_s = calloc((size_t)B * e->nA, 4); e->qon_sp = calloc((size_t)B * e->nA, 4); e->qtg_sp = calloc((size_t)B * e->nA, 4);
    e->ytarget = calloc(B, 4); e->abatch = calloc(B, 4); e->best = calloc(B, 4);
    e->nthreads = 1;
    if (hp->recurrence) {
        const int T = hp->trace_length, TB = T * B; e->T = T; e->ep_cap = hp->buffer_size;
        if (T < 1) { FAIL("trace_length must be >= 1"); }
        e->ep_s = calloc((size_t)e->ep_cap * T * e->obs_elems, 4); e->ep_sp = calloc((size_t)e->ep_cap * T * e->obs_elems, 4);
        e->ep_a = calloc((size_t)e->ep_cap * T, 4); e->ep_r = calloc((size_t)e->ep_cap * T, 4); e->ep_done = calloc((size_t)e->ep_cap * T, 1); e->ep_len = calloc(e->ep_cap, 4);
        e->rx0 = calloc((size_t)e->obs_elems * 2 * TB, 4);
        for (int i = 0; i < n; i++) {
            const RLayer* L = &e->L[i];
            e->racc_on[i] = calloc((size_t)L->out_feat * 2 * TB, 4); e->racc_tg[i] = calloc((size_t)L->out_feat * TB, 4); e->rdact[i] = calloc((size_t)L->out_feat * TB, 4);
            if (L->kind == DQN_LAYER_LSTM) {
                e->gx_on[i] = calloc((size_t)L->N * 2 * TB, 4); e->gx_tg[i] = calloc((size_t)L->N * TB, 4);
                e->gates[i] = calloc((size_t)L->N * TB, 4); e->cst[i] = calloc((size_t)L->H * TB * 2, 4);   /* c and tanh(c) */
                e->hprev[i] = calloc((size_t)L->H * TB * 2, 4);                                           /* h_{t-1} and c_{t-1} */
                e->dgates[i] = calloc((size_t)L->N * TB, 4);
            }
        }
        e->r_a = calloc(TB, 4); e->r_r = calloc(TB, 4); e->r_done = calloc(TB, 4); e->r_mask = calloc(TB, 4);
    }
    *out = e; return 0;
}
int ref_set_threads(ref_engine* e, int n) {
    e->nthreads = n < 1 ? 1 : n;
#ifdef _OPENMP
    omp_set_num_threads(e->nthreads);
#endif
    return 0;
}
static void envs_free(struct ref_envs* v);
int ref_destroy(ref_engine* e) {
    if (!e) return 0;
    envs_free(e->envs);
    free(e->p_on); free(e->p_tg); free(e->grad); free(e->m); free(e->v);
    free(e->s_f32); free(e->sp_f32); free(e->s_u8); free(e->sp_u8); free(e->a); free(e->r); free(e->done); free(e->tree);
    free(e->idx); free(e->x0);
    for (int i = 0; i < e->nl; i++) { free(e->act_on[i]); free(e->act_tg[i]); free(e->dact[i]); }
    free(e->w_is); free(e->rew); free(e->donef); free(e->td); free(e->qon_s); free(e->qon_sp); free(e->qtg_sp);
    free(e->ytarget); free(e->abatch); free(e->best); free(e); return 0;
}
int ref_n_params(ref_engine* e, size_t* n) { *n = e->P; return 0; }
int ref_get_plan(ref_engine* e, dqn_layer_plan* p) { for (int i = 0; i < e->nl; i++) p[i] = e->L[i].plan; return 0; }

/* external (Flux.params, Julia memory order) <-> internal ([K][N], conv kernels flipped) */
static void convert_params(const ref_engine* e, const float* src, float* dst, int to_internal) {
    for (int i = 0; i < e->nl; i++) {
        const RLayer* L = &e->L[i];
        if (L->kind == DQN_LAYER_LSTM) { memcpy(dst + L->w_off, src + L->w_off, ((size_t)L->K * L->N + (size_t)L->H * L->N + L->N + 2 * L->H) * 4); continue; }
        if (L->kind == DQN_LAYER_DENSE) memcpy(dst + L->w_off, src + L->w_off, (size_t)L->K * L->N * 4);
        else for (int co = 0; co < L->cout; co++) for (int ci = 0; ci < L->cin; ci++)
            for (int ky = 0; ky < L->kh; ky++) for (int kx = 0; kx < L->kw; kx++) {
                size_t ext = L->w_off + (((size_t)co * L->cin + ci) * L->kh + (L->kh - 1 - ky)) * L->kw + (L->kw - 1 - kx);
                size_t in = L->w_off + ((size_t)(ci * L->kh + ky) * L->kw + kx) * L->cout + co;
                if (to_internal) dst[in] = src[ext]; else dst[ext] = src[in];
            }
        memcpy(dst + L->b_off, src + L->b_off, (size_t)L->N * 4);
    }
}
int ref_set_params(ref_engine* e, int which, const float* flat, size_t n) {
    if (n != e->P) FAIL("set_params: got %zu values, network has %zu", n, e->P);
    convert_params(e, flat, which == DQN_NET_TARGET ? e->p_tg : e->p_on, 1); return 0;
}
int ref_get_params(ref_engine* e, int which, float* flat, size_t n) {
    if (n != e->P) FAIL("get_params: size mismatch");
    convert_params(e, which == DQN_NET_TARGET ? e->p_tg : e->p_on, flat, 0); return 0;
}
int ref_get_grads(ref_engine* e, float* flat, size_t n) { if (n != e->P) FAIL("size"); convert_params(e, e->grad, flat, 0); return 0; }
int ref_sync_target(ref_engine* e) { memcpy(e->p_tg, e->p_on, e->P * 4); return 0; }
int ref_get_adam_state(ref_engine* e, float* m, float* v, double* bp, size_t n) {
    if (n != e->P) FAIL("size");
    if (m) convert_params(e, e->m, m, 0);
    if (v) convert_params(e, e->v, v, 0);
    if (bp) { bp[0] = e->bp1; bp[1] = e->bp2; } return 0;
}

/* ---------------------------------------------------------------- replay */
static void tree_fix_leaf(ref_engine* e, int64_t leaf) {
    for (int64_t node = (e->cap2 + leaf) >> 1; node >= 1; node >>= 1) e->tree[node] = e->tree[2 * node] + e->tree[2 * node + 1];
}
int ref_replay_add(ref_engine* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done,
                   const float* td_err, int n) {
    size_t row = (size_t)e->obs_elems * (e->hp.obs_dtype == DQN_OBS_U8 ? 1 : 4);
    for (int i = 0; i < n; i++) {
        float td = td_err ? td_err[i] : fabsf(r[i]);
        if (!(td + e->hp.prio_eps > 0.0f)) FAIL("AssertionError: td_err + eps > 0");
        int64_t w = e->widx;
        if (a[i] < 0 || a[i] >= e->nA) FAIL("action index %d out of range", a[i]);
        if (e->hp.obs_dtype == DQN_OBS_U8) { memcpy(e->s_u8 + w * row, (const uint8_t*)s + i * row, row); memcpy(e->sp_u8 + w * row, (const uint8_t*)sp + i * row, row); }
        else { memcpy((uint8_t*)e->s_f32 + w * row, (const uint8_t*)s + i * row, row); memcpy((uint8_t*)e->sp_f32 + w * row, (const uint8_t*)sp + i * row, row); }
        e->a[w] = a[i]; e->r[w] = r[i]; e->done[w] = done[i] ? 1 : 0;
        e->tree[e->cap2 + w] = prio_f(td, e->hp.prio_eps, e->hp.prio_alpha);
        tree_fix_leaf(e, w);
        e->widx = (w + 1) % e->cap; if (e->size < e->cap) e->size++;
    }
    return 0;
}
int ref_replay_size(ref_engine* e, int64_t* cur, int64_t* cap) { if (cur) *cur = e->size; if (cap) *cap = e->cap; return 0; }
int ref_replay_get_priorities(ref_engine* e, float* p, int64_t n) { memcpy(p, e->tree + e->cap2, (size_t)n * 4); return 0; }

/* hp.sample_distinct: B DISTINCT indices (sample(..., replace=false), ...replay.jl:85).  Positions are visited in ascending order; a position
 * whose index was already taken by an earlier one is redrawn by SUCCESSIVE SAMPLING on the residual priorities: u * (total - taken mass) walked
 * down the tree, where a child's mass is its stored sum minus the priorities of the taken leaves below it (subtracted in the order they were
 * taken), and 0 when no untaken leaf is left below it.  Philox lane B + i, word 3 offset by the attempt; after 8 attempts (float round-off can
 * leave a sliver of mass on a taken leaf) the first untaken leaf in index order is used. */
static void distinct_fix(ref_engine* e, uint64_t ctr) {
    const int B = e->B; int L = 0; for (int64_t w = e->cap2; w > 1; w >>= 1) L++;
    int64_t taken[4096]; float tp[4096]; int nt = 0;
    if (B > 4096 || e->size < B) return;
    for (int i = 0; i < B; i++) {
        int dup = 0; for (int j = 0; j < nt; j++) if (taken[j] == e->idx[i]) { dup = 1; break; }
        int64_t leaf = e->idx[i];
        if (dup) {
            int ok = 0;
            for (int att = 0; att < 8 && !ok; att++) {
                float R = e->tree[1]; for (int j = 0; j < nt; j++) R = R - tp[j];
                uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)(B + i), 0x5A4D504Cu + (uint32_t)(att + 1)};
                philox((uint32_t)e->hp.seed, (uint32_t)(e->hp.seed >> 32), c);
                float t = (float)(c[0] >> 8) * (1.0f / 16777216.0f) * R;
                int64_t node = 1;
                for (int lev = 0; lev < L; lev++) {
                    float m[2];
                    for (int ch = 0; ch < 2; ch++) {
                        const int64_t cn = 2 * node + ch; const int sh = L - lev - 1;          /* leaves below cn: [cn << sh, (cn + 1) << sh) - cap2 */
                        int64_t lo = (cn << sh) - e->cap2, hi = ((cn + 1) << sh) - e->cap2; if (hi > e->size) hi = e->size;
                        int64_t cnt = hi > lo ? hi - lo : 0; float v = e->tree[cn];
                        for (int j = 0; j < nt; j++) if (((taken[j] + e->cap2) >> sh) == cn) { v = v - tp[j]; cnt--; }
                        m[ch] = (cnt > 0 && v > 0.0f) ? v : 0.0f;
                    }
                    if (t < m[0] || !(m[1] > 0.0f)) node = 2 * node; else { t -= m[0]; node = 2 * node + 1; }
                }
                leaf = node - e->cap2; if (leaf >= e->size) leaf = e->size - 1;
                ok = 1; for (int j = 0; j < nt; j++) if (taken[j] == leaf) { ok = 0; break; }
            }
            if (!ok) for (leaf = 0; leaf < e->size; leaf++) { int tk = 0; for (int j = 0; j < nt; j++) if (taken[j] == leaf) { tk = 1; break; } if (!tk) break; }
            e->idx[i] = leaf;
        }
        taken[nt] = leaf; tp[nt] = e->tree[e->cap2 + leaf]; nt++;
    }
}
int ref_replay_sample(ref_engine* e, int64_t* idx_out) {
    int B = e->B;
    if (e->size < B) FAIL("AssertionError: r._curr_size >= r.batch_size");
    float total = e->tree[1], seg = total / (float)B;
    uint64_t ctr = e->sample_ctr++;
    for (int i = 0; i < B; i++) {
        uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)i, 0x5A4D504Cu};
        philox((uint32_t)e->hp.seed, (uint32_t)(e->hp.seed >> 32), c);
        float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f);
        float t = ((float)i + u) * seg;
        int64_t node = 1;
        while (node < e->cap2) {
            float l = e->tree[2 * node], rg = e->tree[2 * node + 1];
            if (t < l || !(rg > 0.0f)) node = 2 * node; else { t -= l; node = 2 * node + 1; }
        }
        int64_t leaf = node - e->cap2; if (leaf >= e->size) leaf = e->size - 1;
        e->idx[i] = leaf;
    }
    if (e->hp.sample_distinct) distinct_fix(e, ctr);
    if (idx_out) memcpy(idx_out, e->idx, (size_t)B * 8);
    return 0;
}
static inline float obs_at(const ref_engine* e, int sp, int64_t row, int f) {
    if (e->hp.obs_dtype == DQN_OBS_U8) { const uint8_t* p = sp ? e->sp_u8 : e->s_u8; return (float)p[row * e->obs_elems + f] / 255.0f; }
    const float* p = sp ? e->sp_f32 : e->s_f32; return p[row * e->obs_elems + f];
}
static int check_idx(ref_engine* e, const int64_t* idx, int n) {
    for (int i = 0; i < n; i++) if (idx[i] < 0 || idx[i] >= e->size) FAIL("BoundsError: index %lld outside 0..%lld", (long long)idx[i], (long long)e->size - 1);
    return 0;
}
static void is_weights(ref_engine* e, const int64_t* idx, float* w) {
    /* p = prio ./ sum(prio[1:n]); w = (n .* p) .^ (-beta)   (...replay.jl:101-102); the sum is the
     * canonical sum-tree root */
    float total = e->tree[1];
    for (int i = 0; i < e->B; i++) {
        float p = e->tree[e->cap2 + idx[i]] / total;
        float x = (float)e->size * p;
        w[i] = (float)pow((double)x, -(double)e->hp.prio_beta);
    }
}
int ref_replay_get_batch(ref_engine* e, const int64_t* idx, float* s, int32_t* a, float* r, float* sp, float* done, float* w) {
    if (check_idx(e, idx, e->B)) return -1;
    for (int i = 0; i < e->B; i++) {
        if (s) for (int f = 0; f < e->obs_elems; f++) s[(size_t)i * e->obs_elems + f] = obs_at(e, 0, idx[i], f);
        if (sp) for (int f = 0; f < e->obs_elems; f++) sp[(size_t)i * e->obs_elems + f] = obs_at(e, 1, idx[i], f);
        if (a) a[i] = e->a[idx[i]]; if (r) r[i] = e->r[idx[i]]; if (done) done[i] = (float)e->done[idx[i]];
    }
    if (w) is_weights(e, idx, w);
    return 0;
}
int ref_update_priorities(ref_engine* e, const int64_t* idx, const float* td, int n) {
    if (check_idx(e, idx, n)) return -1;
    for (int i = 0; i < n; i++) { /* duplicates: last write wins (...replay.jl:79) */
        float p = prio_f(fabsf(td[i]), e->hp.prio_eps, e->hp.prio_alpha);
        if (!(p > 0.0f)) FAIL("AssertionError: all(new_priorities .> 0f0)");
        e->tree[e->cap2 + idx[i]] = p; tree_fix_leaf(e, idx[i]);
    }
    return 0;
}

/* ---------------------------------------------------------------- layers */
/* input X[in_feat][ldx] at column col0, ncols columns -> Y[out_feat][ncols] */
static void layer_forward(const RLayer* L, const float* P, const float* X, int ldx, int col0, int ncols, float* Y) {
    const float* W = P + L->w_off; const float* bias = P + L->b_off;
    const int K = L->K, N = L->N, kc = chunk_len(K, L->plan.fwd_kc), S = nchunks(K, L->plan.fwd_kc);
    const int npos = L->kind == DQN_LAYER_CONV ? L->oh * L->ow : 1;
    int* koff = (int*)malloc(sizeof(int) * K);
    for (int k = 0; k < K; k++) {
        if (L->kind == DQN_LAYER_CONV) { int ci = k / (L->kh * L->kw), ky = (k / L->kw) % L->kh, kx = k % L->kw; koff[k] = (ci * L->ih + ky) * L->iw + kx; }
        else koff[k] = k;
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; n++) for (int pos = 0; pos < npos; pos++) {
        float acc[1024], tot[1024];
        int xb = 0; if (L->kind == DQN_LAYER_CONV) { int oy = pos / L->ow, ox = pos % L->ow; xb = oy * L->sh * L->iw + ox * L->sw; }
        for (int c0 = 0; c0 < ncols; c0 += 1024) {
            int nc = ncols - c0 < 1024 ? ncols - c0 : 1024;
            for (int s = 0; s < S; s++) {
                for (int b = 0; b < nc; b++) acc[b] = 0.0f;
                int k1 = (s + 1) * kc < K ? (s + 1) * kc : K;
                for (int k = s * kc; k < k1; k++) {
                    const float w = W[(size_t)k * N + n]; const float* xr = X + (size_t)(xb + koff[k]) * ldx + col0 + c0;
                    for (int b = 0; b < nc; b++) acc[b] = fmaf(xr[b], w, acc[b]);
                }
                if (s == 0) for (int b = 0; b < nc; b++) tot[b] = acc[b]; else for (int b = 0; b < nc; b++) tot[b] = tot[b] + acc[b];
            }
            float* yr = Y + ((size_t)n * npos + pos) * ncols + c0;
            for (int b = 0; b < nc; b++) yr[b] = act_f(tot[b] + bias[n], L->act);
        }
    }
    free(koff);
}

/* dpre[out_feat][B] (already multiplied by act') ; X = layer input [in_feat][ldx] cols 0..B-1 */
static void layer_backward_w(const RLayer* L, const float* X, int ldx, const float* dpre, int B, float* G) {
    float* dW = G + L->w_off; float* db = G + L->b_off;
    const int K = L->K, N = L->N; const int npos = L->kind == DQN_LAYER_CONV ? L->oh * L->ow : 1;
    const int KK = npos * B; const int kc = chunk_len(KK, L->plan.dw_kc), S = nchunks(KK, L->plan.dw_kc);
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) {
        int ko = k; if (L->kind == DQN_LAYER_CONV) { int ci = k / (L->kh * L->kw), ky = (k / L->kw) % L->kh, kx = k % L->kw; ko = (ci * L->ih + ky) * L->iw + kx; }
        float acc[512], tot[512];
        for (int n0 = 0; n0 < N; n0 += 512) {
            int nn = N - n0 < 512 ? N - n0 : 512;
            for (int s = 0; s < S; s++) {
                for (int n = 0; n < nn; n++) acc[n] = 0.0f;
                int j1 = (s + 1) * kc < KK ? (s + 1) * kc : KK;
                for (int j = s * kc; j < j1; j++) {
                    int pos = j / B, b = j % B; int xb = 0;
                    if (L->kind == DQN_LAYER_CONV) { int oy = pos / L->ow, ox = pos % L->ow; xb = oy * L->sh * L->iw + ox * L->sw; }
                    const float x = X[(size_t)(xb + ko) * ldx + b];
                    for (int n = 0; n < nn; n++) acc[n] = fmaf(x, dpre[((size_t)(n0 + n) * npos + pos) * B + b], acc[n]);
                }
                if (s == 0) for (int n = 0; n < nn; n++) tot[n] = acc[n]; else for (int n = 0; n < nn; n++) tot[n] = tot[n] + acc[n];
            }
            for (int n = 0; n < nn; n++) dW[(size_t)k * N + n0 + n] = tot[n];
        }
    }
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        float tot = 0.0f;
        for (int s = 0; s < S; s++) {
            float acc = 0.0f; int j1 = (s + 1) * kc < KK ? (s + 1) * kc : KK;
            for (int j = s * kc; j < j1; j++) { int pos = j / B, b = j % B; acc = acc + dpre[((size_t)n * npos + pos) * B + b]; }
            tot = s == 0 ? acc : tot + acc;
        }
        db[n] = tot;
    }
}
/* the same contraction with COLUMN-GROUP chunks (plan.dw_kc = -cg, recurrent networks, dense views only): TB = T*Bb columns j = t*Bb + b; chunk g holds the
 * columns of batch columns [g*cg, (g+1)*cg) for all t, chained in ascending j; chunk sums are added in ascending g */
static void layer_backward_w_cg(const RLayer* L, const float* X, int ldx, const float* dpre, int T, int Bb, int cg, float* G) {
    float* dW = G + L->w_off; float* db = G + L->b_off;
    const int K = L->K, N = L->N, TB = T * Bb, S = Bb / cg;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++) for (int n = 0; n < N; n++) {
        float tot = 0.0f;
        for (int g = 0; g < S; g++) {
            float acc = 0.0f;
            for (int t = 0; t < T; t++) for (int b = g * cg; b < (g + 1) * cg; b++) { const int j = t * Bb + b; acc = fmaf(X[(size_t)k * ldx + j], dpre[(size_t)n * TB + j], acc); }
            tot = g == 0 ? acc : tot + acc;
        }
        dW[(size_t)k * N + n] = tot;
    }
    for (int n = 0; n < N; n++) {
        float tot = 0.0f;
        for (int g = 0; g < S; g++) {
            float acc = 0.0f;
            for (int t = 0; t < T; t++) for (int b = g * cg; b < (g + 1) * cg; b++) acc = acc + dpre[(size_t)n * TB + t * Bb + b];
            tot = g == 0 ? acc : tot + acc;
        }
        db[n] = tot;
    }
}
/* dX[in_feat][B] = W * dpre (dense: chunks over n; conv: valid taps (ky,kx) ascending, co innermost, chunks of RAW taps per plan.dx_kc) */
static void layer_backward_x(const RLayer* L, const float* P, const float* dpre, int B, float* dX) {
    const float* W = P + L->w_off; const int N = L->N;
    if (L->kind == DQN_LAYER_DENSE) {
        const int kc = chunk_len(N, L->plan.dx_kc), S = nchunks(N, L->plan.dx_kc);
#pragma omp parallel for schedule(static)
        for (int k = 0; k < L->K; k++) {
            float acc[1024], tot[1024];
            for (int c0 = 0; c0 < B; c0 += 1024) {
                int nc = B - c0 < 1024 ? B - c0 : 1024;
                for (int s = 0; s < S; s++) {
                    for (int b = 0; b < nc; b++) acc[b] = 0.0f;
                    int n1 = (s + 1) * kc < N ? (s + 1) * kc : N;
                    for (int n = s * kc; n < n1; n++) { const float w = W[(size_t)k * N + n]; const float* d = dpre + (size_t)n * B + c0; for (int b = 0; b < nc; b++) acc[b] = fmaf(d[b], w, acc[b]); }
                    if (s == 0) for (int b = 0; b < nc; b++) tot[b] = acc[b]; else for (int b = 0; b < nc; b++) tot[b] = tot[b] + acc[b];
                }
                for (int b = 0; b < nc; b++) dX[(size_t)k * B + c0 + b] = tot[b];
            }
        }
        return;
    }
    const int npos = L->oh * L->ow;
    /* plan.dx_kc > 0 cuts the RAW tap list (index ky*kw + kx, ascending) into chunks of dx_kc taps: each chunk is one fma chain from +0 over its
     * VALID taps (co innermost), chunk sums are added in ascending chunk order; chunks without a valid tap contribute nothing */
    const int tc = (L->plan.dx_kc > 0 && L->plan.dx_kc < L->kh * L->kw) ? L->plan.dx_kc : L->kh * L->kw;
#pragma omp parallel for collapse(2) schedule(static)
    for (int ci = 0; ci < L->cin; ci++) for (int ip = 0; ip < L->ih * L->iw; ip++) {
        int iy = ip / L->iw, ix = ip % L->iw; float acc[1024], tot[1024];
        for (int c0 = 0; c0 < B; c0 += 1024) {
            int nc = B - c0 < 1024 ? B - c0 : 1024;
            int cur = -1, have = 0;
            for (int b = 0; b < nc; b++) acc[b] = 0.0f;
            for (int ky = 0; ky < L->kh; ky++) {
                int ty = iy - ky; if (ty < 0 || ty % L->sh) continue; int oy = ty / L->sh; if (oy >= L->oh) continue;
                for (int kx = 0; kx < L->kw; kx++) {
                    int tx = ix - kx; if (tx < 0 || tx % L->sw) continue; int ox = tx / L->sw; if (ox >= L->ow) continue;
                    const int cid = (ky * L->kw + kx) / tc;
                    if (cid != cur) {
                        if (cur >= 0) { if (have) for (int b = 0; b < nc; b++) tot[b] = tot[b] + acc[b]; else for (int b = 0; b < nc; b++) tot[b] = acc[b]; have = 1; for (int b = 0; b < nc; b++) acc[b] = 0.0f; }
                        cur = cid;
                    }
                    const size_t krow = (size_t)((ci * L->kh + ky) * L->kw + kx) * N; const int pos = oy * L->ow + ox;
                    for (int co = 0; co < N; co++) { const float w = W[krow + co]; const float* d = dpre + ((size_t)co * npos + pos) * B + c0; for (int b = 0; b < nc; b++) acc[b] = fmaf(d[b], w, acc[b]); }
                }
            }
            if (have) for (int b = 0; b < nc; b++) acc[b] = tot[b] + acc[b];
            for (int b = 0; b < nc; b++) dX[((size_t)ci * L->ih * L->iw + ip) * B + c0 + b] = acc[b];
        }
    }
}

static void net_forward(ref_engine* e, const float* P, float** act, const float* X0, int ld0, int col0, int ncols) {
    for (int i = 0; i < e->nl; i++) {
        const RLayer* L = &e->L[i];
        if (L->src < 0) layer_forward(L, P, X0, ld0, col0, ncols, act[i]);
        else layer_forward(L, P, act[L->src], ncols, 0, ncols, act[i]);
    }
}
/* Q[a] for column b: dueling (val .+ adv) .- mean(adv)  with mean = (sum ascending)/nA  (dueling.jl:10) */
static void q_column(const ref_engine* e, float** act, int ld, int col, float* q) {
    int nA = e->nA;
    if (!e->hp.dueling) { for (int a = 0; a < nA; a++) q[a] = act[e->last_base][(size_t)a * ld + col]; return; }
    const float* A = act[e->last_adv]; float v = act[e->last_val][col];
    float sum = A[col]; for (int a = 1; a < nA; a++) sum = sum + A[(size_t)a * ld + col];
    float mean = sum / (float)nA;
    for (int a = 0; a < nA; a++) q[a] = (v + A[(size_t)a * ld + col]) - mean;
}
static inline int argmax_first(const float* q, int n) { int bi = 0; for (int a = 1; a < n; a++) if (q[a] > q[bi]) bi = a; return bi; }

/* ---------------------------------------------------------------- train step (solver.jl:191-236) */
int ref_train_step(ref_engine* e, const int64_t* idx_or_null, float* loss_out, float* gnorm_out, float* td_out) {
    const int B = e->B, nA = e->nA, E = e->obs_elems, ld0 = 2 * B;
    if (idx_or_null) { if (check_idx(e, idx_or_null, B)) return -1; memcpy(e->idx, idx_or_null, (size_t)B * 8); }
    else if (ref_replay_sample(e, NULL)) return -1;
    /* get_batch: gather into the batch-innermost arena X0[f][2B]: cols 0..B-1 = s, B..2B-1 = sp */
#pragma omp parallel for schedule(static)
    for (int f = 0; f < E; f++) for (int b = 0; b < B; b++) {
        e->x0[(size_t)f * ld0 + b] = obs_at(e, 0, e->idx[b], f);
        e->x0[(size_t)f * ld0 + B + b] = obs_at(e, 1, e->idx[b], f);
    }
    for (int b = 0; b < B; b++) { e->abatch[b] = e->a[e->idx[b]]; e->rew[b] = e->r[e->idx[b]]; e->donef[b] = (float)e->done[e->idx[b]]; }
    is_weights(e, e->idx, e->w_is);
    /* forwards: online on [s ; sp] (or just s), target on sp */
    const int ncon = e->hp.double_q ? 2 * B : B;
    net_forward(e, e->p_on, e->act_on, e->x0, ld0, 0, ncon);
    net_forward(e, e->p_tg, e->act_tg, e->x0, ld0, B, B);
    /* TD, Huber, dL/dQ */
    const float gamma = e->hp.gamma, invB = 1.0f / (float)B;
    float lsum = 0.0f; float q[64], qt[64];
    if (nA > 64) FAIL("n_actions > 64 unsupported in the twin");
    const int lastv = e->last_val, lasta = e->hp.dueling ? e->last_adv : e->last_base;
    for (int b = 0; b < B; b++) {
        q_column(e, e->act_tg, B, b, qt); memcpy(e->qtg_sp + (size_t)b * nA, qt, nA * 4);
        float qsp; int best;
        if (e->hp.double_q) { q_column(e, e->act_on, ncon, B + b, q); memcpy(e->qon_sp + (size_t)b * nA, q, nA * 4); best = argmax_first(q, nA); qsp = qt[best]; }
        else { best = argmax_first(qt, nA); qsp = qt[best]; memcpy(e->qon_sp + (size_t)b * nA, qt, nA * 4); }
        e->best[b] = best;
        float t1 = 1.0f - e->donef[b]; float t2 = t1 * gamma; float t3 = t2 * qsp; float y = e->rew[b] + t3;  /* :217 */
        e->ytarget[b] = y;
        q_column(e, e->act_on, ncon, b, q); memcpy(e->qon_s + (size_t)b * nA, q, nA * 4);
        float td = q[e->abatch[b]] - y; e->td[b] = td;                                           /* :220-222 */
        float x = e->w_is[b] * td; float ab = fabsf(x); float qd = ab < 1.0f ? ab : 1.0f; float lin = ab - qd;
        float hl = (0.5f * qd) * qd + lin;                                                        /* helpers.jl:14-19 */
        lsum = lsum + hl;
        float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
        float g = (invB * cl) * e->w_is[b];
        /* dL/d(last layer outputs) */
        if (e->hp.dueling) {
            e->dact[lastv][b] = g;
            float gm = g / (float)nA;
            for (int a = 0; a < nA; a++) e->dact[lasta][(size_t)a * B + b] = (a == e->abatch[b] ? g : 0.0f) - gm;
        } else for (int a = 0; a < nA; a++) e->dact[lasta][(size_t)a * B + b] = (a == e->abatch[b] ? g : 0.0f);
    }
    e->loss = lsum / (float)B;                                                                    /* :223-224 */
    /* backward (reverse layer order; val/adv first-layer input grads are joined into the base output grad) */
    int joined = 0;
    for (int i = e->nl - 1; i >= 0; i--) {
        const RLayer* L = &e->L[i];
        float* d = e->dact[i]; const float* y = e->act_on[i];
        for (size_t t = 0; t < (size_t)L->out_feat; t++) for (int b = 0; b < B; b++) d[t * B + b] = dact_f(d[t * B + b], y[t * ncon + b], L->act);
        const float* X = L->src < 0 ? e->x0 : e->act_on[L->src]; int ldx = L->src < 0 ? ld0 : ncon;
        layer_backward_w(L, X, ldx, d, B, e->grad);
        if (L->src >= 0) {
            int src = L->src; size_t n = (size_t)e->L[src].out_feat * B;
            int is_join = e->hp.dueling && src == e->last_base && L->stream != DQN_STREAM_BASE;
            if (!is_join) layer_backward_x(L, e->p_on, d, B, e->dact[src]);
            else {
                /* processed in reverse order: the adv stream reaches the join first, then val; canonical
                 * join order is dX_val + dX_adv */
                float* tmp = (float*)malloc(n * 4); layer_backward_x(L, e->p_on, d, B, tmp);
                if (!joined) { memcpy(e->dact[src], tmp, n * 4); joined = 1; }
                else for (size_t t = 0; t < n; t++) e->dact[src][t] = tmp[t] + e->dact[src][t];
                free(tmp);
            }
        }
    }
    /* globalnorm (helpers.jl:38-46) and Flux Adam */
    float gn = 0.0f; for (size_t i = 0; i < e->P; i++) { float a = fabsf(e->grad[i]); if (a > gn) gn = a; }
    e->gnorm = gn;
    if (e->hp.adam_f64_scalars) {
        const double b1 = e->hp.adam_beta1, b2 = e->hp.adam_beta2, eps = e->hp.adam_eps, eta = (double)e->hp.learning_rate;
        const double omb1 = 1.0 - b1, omb2 = 1.0 - b2, c1 = 1.0 - e->bp1, c2 = 1.0 - e->bp2;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < e->P; i++) {
            double g = (double)e->grad[i];
            double t1 = b1 * (double)e->m[i]; double t2 = omb1 * g; float mn = (float)(t1 + t2);
            double u1 = b2 * (double)e->v[i]; double u2 = omb2 * g; double u3 = u2 * g; float vn = (float)(u1 + u3);
            double mh = (double)mn / c1; double vh = (double)vn / c2; double den = sqrt(vh) + eps; double q1 = mh / den; float dl = (float)(q1 * eta);
            e->m[i] = mn; e->v[i] = vn; e->p_on[i] = e->p_on[i] - dl;
        }
    } else {
        const float b1 = (float)e->hp.adam_beta1, b2 = (float)e->hp.adam_beta2, eps = (float)e->hp.adam_eps, eta = e->hp.learning_rate;
        const float omb1 = 1.0f - b1, omb2 = 1.0f - b2, c1 = 1.0f - (float)e->bp1, c2 = 1.0f - (float)e->bp2;
        for (size_t i = 0; i < e->P; i++) {
            float g = e->grad[i];
            float t1 = b1 * e->m[i]; float t2 = omb1 * g; float mn = t1 + t2;
            float u1 = b2 * e->v[i]; float u2 = omb2 * g; float u3 = u2 * g; float vn = u1 + u3;
            float mh = mn / c1; float vh = vn / c2; float den = sqrtf(vh) + eps; float q1 = mh / den; float dl = q1 * eta;
            e->m[i] = mn; e->v[i] = vn; e->p_on[i] = e->p_on[i] - dl;
        }
    }
    e->bp1 *= e->hp.adam_beta1; e->bp2 *= e->hp.adam_beta2;
    if (e->hp.prioritized_replay) if (ref_update_priorities(e, e->idx, e->td, B)) return -1;   /* :231-233, unweighted td */
    if (loss_out) *loss_out = e->loss; if (gnorm_out) *gnorm_out = e->gnorm; if (td_out) memcpy(td_out, e->td, (size_t)B * 4);
    return 0;
}
int ref_train_steps(ref_engine* e, int n, float* loss, float* gn) {
    for (int i = 0; i < n; i++) if (ref_train_step(e, NULL, loss, gn, NULL)) return -1; return 0;
}
int ref_get_last_q(ref_engine* e, float* qs, float* qsp, float* qt, int32_t* best, float* y) {
    size_t n = (size_t)e->B * e->nA * 4;
    if (qs) memcpy(qs, e->qon_s, n); if (qsp) memcpy(qsp, e->qon_sp, n); if (qt) memcpy(qt, e->qtg_sp, n);
    if (best) memcpy(best, e->best, (size_t)e->B * 4); if (y) memcpy(y, e->ytarget, (size_t)e->B * 4); return 0;
}
int ref_get_last_indices(ref_engine* e, int64_t* idx) { memcpy(idx, e->idx, (size_t)e->B * 8); return 0; }

/* ---------------------------------------------------------------- policy (policy.jl:38-64) */
static void lstm_input_proj(const RLayer* L, const float* P, const float* X, int ldx, int col0, int ncols, float* Gx);
static inline float sigm_f(float x); static inline float tanh_f(float x);
/* Recur state of the policy network: one (h, c) column per observation stream; reset = state0 of the ONLINE net (policy.jl:32-34) */
static void policy_state(ref_engine* e, int n, int force_reset) {
    if (!e->hp.recurrence) return;
    if (n != e->pol_n) {
        for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
            free(e->pol_h[i]); free(e->pol_c[i]);
            e->pol_h[i] = (float*)malloc((size_t)e->L[i].H * n * 4); e->pol_c[i] = (float*)malloc((size_t)e->L[i].H * n * 4);
        }
        e->pol_n = n; force_reset = 1;
    }
    if (force_reset) for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const RLayer* L = &e->L[i];
        for (int u = 0; u < L->H; u++) for (int b = 0; b < n; b++) { e->pol_h[i][u * n + b] = e->p_on[L->h0_off + u]; e->pol_c[i][u * n + b] = e->p_on[L->c0_off + u]; }
    }
}
int ref_reset_state(ref_engine* e) { policy_state(e, e->pol_n > 0 ? e->pol_n : 1, 1); return 0; }      /* resetstate!(policy) */
/* hiddenstates(m) / sethiddenstates!(m, hs) (src/helpers.jl:61-79; used around batch_train!, src/solver.jl:137-139): per LSTM layer h then c,
   each [out][streams]; the same flat layout as dqn_get_hidden / dqn_set_hidden */
int ref_get_hidden(ref_engine* e, float* hc, size_t n) {
    if (e->hp.recurrence && e->pol_n == 0) policy_state(e, 1, 1);
    size_t off = 0;
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_n; if (off + 2 * m > n) FAIL("get_hidden: buffer too small");
        memcpy(hc + off, e->pol_h[i], m * 4); off += m; memcpy(hc + off, e->pol_c[i], m * 4); off += m;
    }
    return 0;
}
int ref_set_hidden(ref_engine* e, const float* hc, size_t n) {
    if (e->hp.recurrence && e->pol_n == 0) policy_state(e, 1, 1);
    size_t off = 0;
    for (int i = 0; i < e->nl; i++) if (e->L[i].kind == DQN_LAYER_LSTM) {
        const size_t m = (size_t)e->L[i].H * e->pol_n; if (off + 2 * m > n) FAIL("set_hidden: buffer too small");
        memcpy(e->pol_h[i], hc + off, m * 4); off += m; memcpy(e->pol_c[i], hc + off, m * 4); off += m;
    }
    return 0;
}
int ref_forward(ref_engine* e, int which, const float* obs, int n, float* q_out) {
    const float* P = which == DQN_NET_TARGET ? e->p_tg : e->p_on; int E = e->obs_elems;
    float* x = (float*)malloc((size_t)E * n * 4); float* act[MAXL];
    for (int f = 0; f < E; f++) for (int b = 0; b < n; b++) x[(size_t)f * n + b] = obs[(size_t)b * E + f];
    for (int i = 0; i < e->nl; i++) act[i] = (float*)malloc((size_t)e->L[i].out_feat * n * 4);
    if (!e->hp.recurrence) net_forward(e, P, act, x, n, 0, n);
    else {      /* one Recur step per call: the hidden state persists between calls (policy.jl:38-46) */
        policy_state(e, n, 0);
        for (int i = 0; i < e->nl; i++) {
            const RLayer* L = &e->L[i]; const float* X = L->src < 0 ? x : act[L->src];
            if (L->kind != DQN_LAYER_LSTM) { layer_forward(L, P, X, n, 0, n, act[i]); continue; }
            const int H = L->H; float* gx = (float*)malloc((size_t)L->N * n * 4);
            lstm_input_proj(L, P, X, n, 0, n, gx);
            const float *Wh = P + L->wh_off, *bias = P + L->b_off;
            float* hn = (float*)malloc((size_t)H * n * 4); float* cn = (float*)malloc((size_t)H * n * 4);
            for (int u = 0; u < H; u++) for (int b = 0; b < n; b++) {
                float g[4];
                for (int q = 0; q < 4; q++) {
                    const int nn = q * H + u; float ch = 0.0f;
                    for (int j = 0; j < H; j++) ch = fmaf(e->pol_h[i][j * n + b], Wh[(size_t)j * L->N + nn], ch);
                    g[q] = (gx[(size_t)nn * n + b] + ch) + bias[nn];
                }
                const float ig = sigm_f(g[0]), fg = sigm_f(g[1]), gg = tanh_f(g[2]), og = sigm_f(g[3]);
                const float t1 = fg * e->pol_c[i][u * n + b]; const float t2 = ig * gg; const float c = t1 + t2; const float tc = tanh_f(c);
                cn[u * n + b] = c; hn[u * n + b] = og * tc;
            }
            memcpy(e->pol_h[i], hn, (size_t)H * n * 4); memcpy(e->pol_c[i], cn, (size_t)H * n * 4); memcpy(act[i], hn, (size_t)H * n * 4);
            free(hn); free(cn); free(gx);
        }
    }
    for (int b = 0; b < n; b++) q_column(e, act, n, b, q_out + (size_t)b * e->nA);
    for (int i = 0; i < e->nl; i++) free(act[i]); free(x); return 0;
}
int ref_greedy_action(ref_engine* e, const float* obs, int n, int32_t* a_out) {
    float* q = (float*)malloc((size_t)n * e->nA * 4); ref_forward(e, DQN_NET_ONLINE, obs, n, q);
    for (int b = 0; b < n; b++) a_out[b] = argmax_first(q + (size_t)b * e->nA, e->nA);
    free(q); return 0;
}

/* ================================================================= DRQN
 * EpisodeReplayBuffer  src/episode_replay.jl:3-95      batch_train! (recurrent)  src/solver.jl:239-287
 * Flux LSTM (third-party; recalled): g = Wi*x .+ Wh*h .+ b, gates input/forget/cell/output,
 * c' = sigm(f).*c .+ sigm(i).*tanh(g), h' = sigm(o).*tanh(c'), trainable state0 = (h0, c0).
 * Column of timestep t, sample b: t*B + b.  Canonical order: gate pre-activation
 * = ((chain_k Wi x) + (chain_j Wh h)) + b; sigm/tanh evaluated in double and rounded once. */
static inline float sigm_f(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }
static inline float tanh_f(float x) { return (float)tanh((double)x); }

/* checkpoint seam of the episode replay (the twin of dqn_episode_export / dqn_episode_import): episodes as [n][T] slots + their true lengths */
int ref_episode_export(ref_engine* e, int64_t first, int64_t n, float* s, float* sp, int32_t* a, float* r, uint8_t* done, int32_t* len) {
    if (!e->hp.recurrence) FAIL("this engine was created with recurrence = false");
    if (first < 0 || n < 0 || first + n > e->ep_size) FAIL("BoundsError: episodes %lld..%lld outside 0..%lld", (long long)first, (long long)(first + n - 1), (long long)e->ep_size - 1);
    const size_t T = (size_t)e->T, E = (size_t)e->obs_elems, off = (size_t)first * T, cnt = (size_t)n * T;
    if (s) memcpy(s, e->ep_s + off * E, cnt * E * 4);
    if (sp) memcpy(sp, e->ep_sp + off * E, cnt * E * 4);
    if (a) memcpy(a, e->ep_a + off, cnt * 4);
    if (r) memcpy(r, e->ep_r + off, cnt * 4);
    if (done) memcpy(done, e->ep_done + off, cnt);
    if (len) memcpy(len, e->ep_len + first, (size_t)n * 4);
    return 0;
}
int ref_episode_import(ref_engine* e, int64_t n, const float* s, const float* sp, const int32_t* a, const float* r, const uint8_t* done, const int32_t* len) {
    if (!e->hp.recurrence) FAIL("this engine was created with recurrence = false");
    if (n < 0 || n > e->ep_cap) FAIL("import of %lld episodes into an episode replay of capacity %lld", (long long)n, (long long)e->ep_cap);
    const size_t T = (size_t)e->T, E = (size_t)e->obs_elems, cnt = (size_t)n * T;
    for (int64_t i = 0; i < n; i++) {
        if (len[i] < 1) FAIL("episode %lld: length %d < 1", (long long)i, len[i]);
        const int m = len[i] < e->T ? len[i] : e->T;
        for (int t = 0; t < m; t++) if (a[(size_t)i * T + t] < 0 || a[(size_t)i * T + t] >= e->nA) FAIL("action index %d out of range 0..%d", a[(size_t)i * T + t], e->nA - 1);
    }
    memcpy(e->ep_s, s, cnt * E * 4); memcpy(e->ep_sp, sp, cnt * E * 4); memcpy(e->ep_a, a, cnt * 4); memcpy(e->ep_r, r, cnt * 4); memcpy(e->ep_done, done, cnt);
    memcpy(e->ep_len, len, (size_t)n * 4);
    e->ep_size = n; e->ep_widx = n % e->ep_cap; e->ep_cur_len = 0;
    return 0;
}
int ref_episode_count(ref_engine* e, int64_t* cur, int64_t* cap) { if (cur) *cur = e->ep_size; if (cap) *cap = e->ep_cap; return 0; }
int ref_episode_commit(ref_engine* e) {          /* add_episode! (:54-60) */
    if (!e->hp.recurrence) FAIL("engine was created with recurrence = false");
    e->ep_len[e->ep_widx] = (int32_t)e->ep_cur_len;
    e->ep_widx = (e->ep_widx + 1) % e->ep_cap; if (e->ep_size < e->ep_cap) e->ep_size++;
    e->ep_cur_len = 0; return 0;
}
int ref_episode_add(ref_engine* e, const void* s, const int32_t* a, const float* r, const void* sp, const uint8_t* done, int n) {
    if (!e->hp.recurrence) FAIL("engine was created with recurrence = false");
    const int E = e->obs_elems, T = e->T;
    for (int i = 0; i < n; i++) {                 /* add_exp! (:46-52): push, store the episode when done */
        if (a[i] < 0 || a[i] >= e->nA) FAIL("action index %d out of range", a[i]);
        const int64_t t = e->ep_cur_len;
        if (t < T) {
            const size_t slot = (size_t)e->ep_widx * T + t;
            memcpy(e->ep_s + slot * E, (const float*)s + (size_t)i * E, (size_t)E * 4); memcpy(e->ep_sp + slot * E, (const float*)sp + (size_t)i * E, (size_t)E * 4);
            e->ep_a[slot] = a[i]; e->ep_r[slot] = r[i]; e->ep_done[slot] = done[i] ? 1 : 0;
        }
        e->ep_cur_len++;
        if (done[i]) ref_episode_commit(e);
    }
    return 0;
}
static int drqn_check(ref_engine* e, const int64_t* ep_idx, const int32_t* ep_start) {
    if (e->ep_size < e->B) FAIL("AssertionError: r._curr_size >= r.batch_size");
    for (int b = 0; b < e->B; b++) {
        if (ep_idx[b] < 0 || ep_idx[b] >= e->ep_size) FAIL("BoundsError: episode index %lld outside 0..%lld", (long long)ep_idx[b], (long long)e->ep_size - 1);
        const int len = e->ep_len[ep_idx[b]];
        if (len > 0 && (ep_start[b] < 0 || ep_start[b] >= len)) FAIL("episode start %d outside 0..%d", ep_start[b], len - 1);
    }
    return 0;
}
/* number of transitions copied for sample b: the reference's `for j = ep_start:min(len,T)` with t counting from 1 copies
 * the episode PREFIX of length max(0, min(len,T) - ep_start) (0-based start), episode_replay.jl:82-92 */
static inline int prefix_len(const ref_engine* e, int64_t ep, int start) { int len = e->ep_len[ep]; int m = len < e->T ? len : e->T; int n = m - start; return n < 0 ? 0 : n; }
int ref_episode_get_batch(ref_engine* e, const int64_t* ep_idx, const int32_t* ep_start, float* s, int32_t* a, float* r, float* sp, float* done, int32_t* mask) {
    if (drqn_check(e, ep_idx, ep_start)) return -1;
    const int T = e->T, B = e->B, E = e->obs_elems;
    for (int t = 0; t < T; t++) for (int b = 0; b < B; b++) {
        const int ok = t < prefix_len(e, ep_idx[b], ep_start[b]); const size_t slot = (size_t)ep_idx[b] * T + t; const size_t o = (size_t)t * B + b;
        if (s) for (int f = 0; f < E; f++) s[o * E + f] = ok ? e->ep_s[slot * E + f] : 0.0f;
        if (sp) for (int f = 0; f < E; f++) sp[o * E + f] = ok ? e->ep_sp[slot * E + f] : 0.0f;
        if (a) a[o] = ok ? e->ep_a[slot] : 0;      /* CartesianIndex(1,1) for masked rows (:29,:64): harmless, mask multiplies inside huber */
        if (r) r[o] = ok ? e->ep_r[slot] : 0.0f; if (done) done[o] = ok ? (float)e->ep_done[slot] : 0.0f; if (mask) mask[o] = ok;
    }
    return 0;
}
/* a dense view of (part of) an LSTM parameter block, so that the feed-forward routines above can be reused */
static RLayer dense_view(const RLayer* L, int K, int N, size_t w_off, size_t b_off) {
    RLayer v; memset(&v, 0, sizeof v); v.kind = DQN_LAYER_DENSE; v.act = DQN_ACT_IDENTITY; v.K = K; v.N = N; v.in_feat = K; v.out_feat = N; v.oh = v.ow = v.ih = v.iw = 1;
    v.w_off = w_off; v.b_off = b_off; v.plan = L->plan; v.src = L->src; return v;
}
/* Gx[n][col] = chain_k X[k][col] Wi[k][n]  (no bias: it is added inside the recurrence, after the Wh*h chain) */
static void lstm_input_proj(const RLayer* L, const float* P, const float* X, int ldx, int col0, int ncols, float* Gx) {
    RLayer v = dense_view(L, L->K, L->N, L->w_off, L->b_off);
    float* zero = (float*)calloc(L->N, 4);
    /* layer_forward adds P[b_off + n]: run it against a parameter image whose bias is zero */
    const size_t span = (size_t)L->K * L->N;
    float* img = (float*)malloc((span + L->N) * 4); memcpy(img, P + L->w_off, span * 4); memcpy(img + span, zero, (size_t)L->N * 4);
    v.w_off = 0; v.b_off = span; layer_forward(&v, img, X, ldx, col0, ncols, Gx);
    free(img); free(zero);
}
/* one sequence of T steps on B columns starting at column c0 of arrays with leading dimension ld.
 * keep != 0: store gates (i,f,g,o post-activation), c, tanh(c), h_{t-1}, c_{t-1} for BPTT (online s-sequence). */
static void lstm_recurrence(ref_engine* e, int li, const float* P, const float* Gx, int ld, int c0, float* Hout, int keep) {
    const RLayer* L = &e->L[li]; const int H = L->H, B = e->B, T = e->T, TB = T * B;
    const float *Wh = P + L->wh_off, *bias = P + L->b_off, *h0 = P + L->h0_off, *c0p = P + L->c0_off;
    float* hp = (float*)malloc((size_t)H * B * 4); float* cp = (float*)malloc((size_t)H * B * 4); float* hn = (float*)malloc((size_t)H * B * 4); float* cn = (float*)malloc((size_t)H * B * 4);
    for (int u = 0; u < H; u++) for (int b = 0; b < B; b++) { hp[u * B + b] = h0[u]; cp[u * B + b] = c0p[u]; }
    for (int t = 0; t < T; t++) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int u = 0; u < H; u++) for (int b = 0; b < B; b++) {
            const int col = c0 + t * B + b; float g[4];
            for (int q = 0; q < 4; q++) {
                const int n = q * H + u; float ch = 0.0f;
                for (int j = 0; j < H; j++) ch = fmaf(hp[j * B + b], Wh[(size_t)j * L->N + n], ch);
                g[q] = (Gx[(size_t)n * ld + col] + ch) + bias[n];
            }
            const float ig = sigm_f(g[0]), fg = sigm_f(g[1]), gg = tanh_f(g[2]), og = sigm_f(g[3]);
            const float t1 = fg * cp[u * B + b]; const float t2 = ig * gg; const float c = t1 + t2; const float tc = tanh_f(c); const float h = og * tc;
            cn[u * B + b] = c; hn[u * B + b] = h; Hout[(size_t)u * ld + col] = h;
            if (keep) {
                const int k = t * B + b;
                e->gates[li][(size_t)(0 * H + u) * TB + k] = ig; e->gates[li][(size_t)(1 * H + u) * TB + k] = fg;
                e->gates[li][(size_t)(2 * H + u) * TB + k] = gg; e->gates[li][(size_t)(3 * H + u) * TB + k] = og;
                e->cst[li][(size_t)u * TB + k] = c; e->cst[li][(size_t)(H + u) * TB + k] = tc;
                e->hprev[li][(size_t)u * TB + k] = hp[u * B + b]; e->hprev[li][(size_t)(H + u) * TB + k] = cp[u * B + b];
            }
        }
        float* x = hp; hp = hn; hn = x; x = cp; cp = cn; cn = x;
    }
    free(hp); free(cp); free(hn); free(cn);
}
static void seq_forward(ref_engine* e, const float* P, float** act, float** gx, const float* X0, int ld0, int col0, int ncols, int nseq) {
    /* ncols = nseq * T * B columns; each group of T*B columns is an independent sequence from the reset state */
    for (int i = 0; i < e->nl; i++) {
        const RLayer* L = &e->L[i];
        const float* X = L->src < 0 ? X0 : act[L->src]; const int ldx = L->src < 0 ? ld0 : ncols; const int c0 = L->src < 0 ? col0 : 0;
        if (L->kind == DQN_LAYER_LSTM) {
            lstm_input_proj(L, P, X, ldx, c0, ncols, gx[i]);
            for (int q = 0; q < nseq; q++) lstm_recurrence(e, i, P, gx[i], ncols, q * e->T * e->B, act[i], (act == e->racc_on && q == 0));
        } else layer_forward(L, P, X, ldx, c0, ncols, act[i]);
    }
}
int ref_train_step_drqn(ref_engine* e, const int64_t* ep_idx_in, const int32_t* ep_start_in, float* loss_out, float* gnorm_out) {
    if (!e->hp.recurrence) FAIL("engine was created with recurrence = false");
    const int B = e->B, T = e->T, TB = T * B, nA = e->nA, E = e->obs_elems, ld0 = 2 * TB;
    int64_t ep_idx[1024]; int32_t ep_start[1024];
    if (!ep_idx_in) {      /* sample(rng, 1:n, B, replace=false); ep_start = rand(rng, 1:length(ep)) (src/episode_replay.jl:75,81): the engine's SplitMix64 draws */
        if (e->ep_size < B) FAIL("AssertionError: r._curr_size >= r.batch_size");
        int64_t* perm = (int64_t*)malloc((size_t)e->ep_size * 8);
        for (int64_t i = 0; i < e->ep_size; i++) perm[i] = i;
        for (int b = 0; b < B; b++) {
            uint64_t z = (e->drqn_ctr += 0x9E3779B97F4A7C15ull) ^ e->hp.seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
            const int64_t j = b + (int64_t)(z % (uint64_t)(e->ep_size - b)); const int64_t tmp = perm[b]; perm[b] = perm[j]; perm[j] = tmp;
        }
        for (int b = 0; b < B; b++) ep_idx[b] = perm[b];
        free(perm);
        for (int b = 0; b < B; b++) {
            uint64_t z = (e->drqn_ctr += 0x9E3779B97F4A7C15ull) ^ e->hp.seed; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z = z ^ (z >> 31);
            const int len = e->ep_len[ep_idx[b]]; ep_start[b] = len > 0 ? (int32_t)(z % (uint64_t)len) : 0;
        }
    } else { memcpy(ep_idx, ep_idx_in, (size_t)B * 8); memcpy(ep_start, ep_start_in, (size_t)B * 4); }
    if (drqn_check(e, ep_idx, ep_start)) return -1;
    /* sample(r) (:71-95) into the batch-innermost arena: columns t*B+b = s, TB + t*B+b = sp */
    for (int t = 0; t < T; t++) for (int b = 0; b < B; b++) {
        const int ok = t < prefix_len(e, ep_idx[b], ep_start[b]); const size_t slot = (size_t)ep_idx[b] * T + t; const int k = t * B + b;
        for (int f = 0; f < E; f++) { e->rx0[(size_t)f * ld0 + k] = ok ? e->ep_s[slot * E + f] : 0.0f; e->rx0[(size_t)f * ld0 + TB + k] = ok ? e->ep_sp[slot * E + f] : 0.0f; }
        e->r_a[k] = ok ? e->ep_a[slot] : 0; e->r_r[k] = ok ? e->ep_r[slot] : 0.0f; e->r_done[k] = ok ? (float)e->ep_done[slot] : 0.0f; e->r_mask[k] = (float)ok;
    }
    /* online net: the s sequence (loss) and, for double-Q, the sp sequence; target net: the sp sequence (:249-271) */
    const int nseq_on = e->hp.double_q ? 2 : 1;
    seq_forward(e, e->p_on, e->racc_on, e->gx_on, e->rx0, ld0, 0, nseq_on * TB, nseq_on);
    seq_forward(e, e->p_tg, e->racc_tg, e->gx_tg, e->rx0, ld0, TB, TB, 1);
    const int ncon = nseq_on * TB;
    const float gamma = e->hp.gamma, invT = 1.0f / (float)T; float q[64], qt[64];
    const int lastv = e->last_val, lasta = e->hp.dueling ? e->last_adv : e->last_base;
    float loss = 0.0f;
    for (int t = 0; t < T; t++) {
        float lsum = 0.0f;
        for (int b = 0; b < B; b++) {
            const int k = t * B + b;
            q_column(e, e->racc_tg, TB, k, qt);
            int best; float qsp;
            if (e->hp.double_q) { q_column(e, e->racc_on, ncon, TB + k, q); best = argmax_first(q, nA); qsp = qt[best]; } else { best = argmax_first(qt, nA); qsp = qt[best]; }
            const float t1 = 1.0f - e->r_done[k]; const float t2 = t1 * gamma; const float t3 = t2 * qsp; const float y = e->r_r[k] + t3;    /* :268 */
            q_column(e, e->racc_on, ncon, k, q);
            const float td = q[e->r_a[k]] - y; const float m = e->r_mask[k];
            const float x = m * td; const float ab = fabsf(x); const float qd = ab < 1.0f ? ab : 1.0f; const float lin = ab - qd;
            lsum = lsum + ((0.5f * qd) * qd + lin);
            const float cl = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);
            const float g = ((invT / (float)B) * cl) * m;
            if (e->hp.dueling) {
                e->rdact[lastv][k] = g; const float gm = g / (float)nA;
                for (int a = 0; a < nA; a++) e->rdact[lasta][(size_t)a * TB + k] = (a == e->r_a[k] ? g : 0.0f) - gm;
            } else for (int a = 0; a < nA; a++) e->rdact[lasta][(size_t)a * TB + k] = (a == e->r_a[k] ? g : 0.0f);
        }
        loss = loss + lsum / (float)B;                                                       /* :279 */
    }
    e->loss = loss / (float)T;                                                               /* :281 */
    /* backward over the s-sequence columns 0..TB-1 */
    memset(e->grad, 0, e->P * 4);
    int joined = 0;
    const int cgm = e->L[0].plan.dw_kc < 0 ? -e->L[0].plan.dw_kc : 0;      /* column-group dW chunks (validated at creation: all layers alike, B % cg == 0) */
    for (int i = e->nl - 1; i >= 0; i--) {
        const RLayer* L = &e->L[i];
        float* d = e->rdact[i]; const float* y = e->racc_on[i];
        const float* X = L->src < 0 ? e->rx0 : e->racc_on[L->src]; const int ldx = L->src < 0 ? ld0 : ncon;
        float* dxout = NULL; RLayer xv; const float* dpre_for_dx = d;
        if (L->kind == DQN_LAYER_LSTM) {
            const int H = L->H; const float* Wh = e->p_on + L->wh_off;
            float* dG = e->dgates[i]; const float* G = e->gates[i]; const float* C = e->cst[i]; const float* HP = e->hprev[i];
            float* dhn = (float*)calloc((size_t)H * B, 4); float* dcn = (float*)calloc((size_t)H * B, 4);
            for (int t = T - 1; t >= 0; t--) {
                for (int u = 0; u < H; u++) for (int b = 0; b < B; b++) {
                    const int k = t * B + b;
                    const float ig = G[(size_t)u * TB + k], fg = G[(size_t)(H + u) * TB + k], gg = G[(size_t)(2 * H + u) * TB + k], og = G[(size_t)(3 * H + u) * TB + k];
                    const float tc = C[(size_t)(H + u) * TB + k], cprev = HP[(size_t)(H + u) * TB + k];
                    const float dh = d[(size_t)u * TB + k] + dhn[u * B + b];
                    const float dov = dh * tc; const float t1 = dh * og; const float t2 = tc * tc; const float t3 = 1.0f - t2; const float t4 = t1 * t3; const float dc = dcn[u * B + b] + t4;
                    const float di = dc * gg, df = dc * cprev, dgc = dc * ig; dcn[u * B + b] = dc * fg;
                    const float a1 = di * ig, a2 = 1.0f - ig; dG[(size_t)u * TB + k] = a1 * a2;
                    const float b1 = df * fg, b2 = 1.0f - fg; dG[(size_t)(H + u) * TB + k] = b1 * b2;
                    const float c1 = gg * gg, c2 = 1.0f - c1; dG[(size_t)(2 * H + u) * TB + k] = dgc * c2;
                    const float d1 = dov * og, d2 = 1.0f - og; dG[(size_t)(3 * H + u) * TB + k] = d1 * d2;
                }
                for (int j = 0; j < H; j++) for (int b = 0; b < B; b++) {       /* dh_{t-1} = Wh * dG_t : n ascending over 4H */
                    float acc = 0.0f; const int k = t * B + b;
                    for (int n = 0; n < L->N; n++) acc = fmaf(dG[(size_t)n * TB + k], Wh[(size_t)j * L->N + n], acc);
                    dhn[j * B + b] = acc;
                }
            }
            for (int u = 0; u < H; u++) {                                          /* trainable initial state: sum over the batch, ascending b */
                float sh = 0.0f, sc = 0.0f;
                if (cgm) {                                                           /* column-group plan: per-group sums (from +0, ascending b), groups added ascending */
                    for (int g = 0; g < B / cgm; g++) {
                        float ah = 0.0f, ac = 0.0f; for (int b = g * cgm; b < (g + 1) * cgm; b++) { ah = ah + dhn[u * B + b]; ac = ac + dcn[u * B + b]; }
                        sh = g == 0 ? ah : sh + ah; sc = g == 0 ? ac : sc + ac;
                    }
                } else for (int b = 0; b < B; b++) { sh = sh + dhn[u * B + b]; sc = sc + dcn[u * B + b]; }
                e->grad[L->h0_off + u] = sh; e->grad[L->c0_off + u] = sc;
            }
            free(dhn); free(dcn);
            RLayer wi = dense_view(L, L->K, L->N, L->w_off, L->b_off);             /* dWi and db over all T*B columns (t-major, b-minor) */
            /* layer_backward_w writes db right after dW: stage into a scratch (K+1) x N block */
            float* scratch = (float*)calloc(((size_t)(L->K > H ? L->K : H) + 1) * L->N, 4);
            wi.w_off = 0; wi.b_off = (size_t)L->K * L->N; if (cgm) layer_backward_w_cg(&wi, X, ldx, dG, T, B, cgm, scratch); else layer_backward_w(&wi, X, ldx, dG, TB, scratch);
            memcpy(e->grad + L->w_off, scratch, (size_t)L->K * L->N * 4); memcpy(e->grad + L->b_off, scratch + (size_t)L->K * L->N, (size_t)L->N * 4);
            RLayer wh = dense_view(L, H, L->N, 0, (size_t)H * L->N);               /* dWh: X = h_{t-1} (h0 broadcast at t = 0) */
            if (cgm) layer_backward_w_cg(&wh, HP, TB, dG, T, B, cgm, scratch); else layer_backward_w(&wh, HP, TB, dG, TB, scratch);
            memcpy(e->grad + L->wh_off, scratch, (size_t)H * L->N * 4);
            free(scratch);
            xv = dense_view(L, L->K, L->N, L->w_off, L->b_off); dpre_for_dx = dG;
        } else {
            for (size_t t = 0; t < (size_t)L->out_feat; t++) for (int k = 0; k < TB; k++) d[t * TB + k] = dact_f(d[t * TB + k], y[t * ncon + k], L->act);
            if (cgm) layer_backward_w_cg(L, X, ldx, d, T, B, cgm, e->grad); else layer_backward_w(L, X, ldx, d, TB, e->grad);
            xv = *L;
        }
        if (L->src >= 0) {
            const int src = L->src; const size_t n = (size_t)e->L[src].out_feat * TB;
            const int is_join = e->hp.dueling && src == e->last_base && L->stream != DQN_STREAM_BASE;
            if (!is_join) layer_backward_x(&xv, e->p_on, dpre_for_dx, TB, e->rdact[src]);
            else {
                float* tmp = (float*)malloc(n * 4); layer_backward_x(&xv, e->p_on, dpre_for_dx, TB, tmp);
                if (!joined) { memcpy(e->rdact[src], tmp, n * 4); joined = 1; } else for (size_t t = 0; t < n; t++) e->rdact[src][t] = tmp[t] + e->rdact[src][t];
                free(tmp);
            }
        }
        (void)dxout;
    }
    float gn = 0.0f; for (size_t i = 0; i < e->P; i++) { float a = fabsf(e->grad[i]); if (a > gn) gn = a; }
    e->gnorm = gn;
    {   /* Flux Adam, Float64-scalar form (same arithmetic as the feed-forward step) */
        const double b1 = e->hp.adam_beta1, b2 = e->hp.adam_beta2, eps = e->hp.adam_eps, eta = (double)e->hp.learning_rate;
        const double omb1 = 1.0 - b1, omb2 = 1.0 - b2, c1 = 1.0 - e->bp1, c2 = 1.0 - e->bp2;
        for (size_t i = 0; i < e->P; i++) {
            if (e->hp.adam_f64_scalars) {
                double g = (double)e->grad[i]; double t1 = b1 * (double)e->m[i]; double t2 = omb1 * g; float mn = (float)(t1 + t2);
                double u1 = b2 * (double)e->v[i]; double u2 = omb2 * g; double u3 = u2 * g; float vn = (float)(u1 + u3);
                double mh = (double)mn / c1; double vh = (double)vn / c2; double den = sqrt(vh) + eps; double q1 = mh / den; float dl = (float)(q1 * eta);
                e->m[i] = mn; e->v[i] = vn; e->p_on[i] = e->p_on[i] - dl;
            } else {
                const float fb1 = (float)b1, fb2 = (float)b2; float g = e->grad[i];
                float t1 = fb1 * e->m[i]; float t2 = (1.0f - fb1) * g; float mn = t1 + t2; float u1 = fb2 * e->v[i]; float u2 = (1.0f - fb2) * g; float u3 = u2 * g; float vn = u1 + u3;
                float mh = mn / (1.0f - (float)e->bp1); float vh = vn / (1.0f - (float)e->bp2); float den = sqrtf(vh) + (float)eps; float q1 = mh / den; float dl = q1 * e->hp.learning_rate;
                e->m[i] = mn; e->v[i] = vn; e->p_on[i] = e->p_on[i] - dl;
            }
        }
        e->bp1 *= b1; e->bp2 *= b2;
    }
    if (loss_out) *loss_out = e->loss; if (gnorm_out) *gnorm_out = e->gnorm;
    return 0;
}

/* ================================================================= vectorised environments
 * The env loop of dqn_train! (src/solver.jl:82-145) for n lock-stepped copies of TestMDP (test/test_env.jl:10-87) or
 * SimpleGridWorld (POMDPModels defaults; third-party, recalled).  Randomness: Philox4x32-10, key = seed, counter =
 * (vector step lo, hi, env, purpose) -- the same draws as deepqlearning.jl_amd/csrc/envs.hip. */
typedef struct ref_envs {
    dqn_env_spec sp; int n, E, H, W; uint8_t* images;
    int8_t* tm_s; int32_t* tm_t; int32_t* gw_pos;
    int32_t* actions; float* rewards; uint8_t* dones; float* ep_reward; int32_t* ep_step; int64_t* fin_eps; double* fin_reward;
} ref_envs;
static uint32_t env_rand(uint64_t seed, uint64_t t, int env, uint32_t purpose) {
    uint32_t c[4] = {(uint32_t)t, (uint32_t)(t >> 32), (uint32_t)env, purpose};
    philox((uint32_t)seed, (uint32_t)(seed >> 32), c); return c[0];
}
static float u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
static void envs_free(struct ref_envs* v) {
    if (!v) return;
    free(v->images); free(v->tm_s); free(v->tm_t); free(v->gw_pos); free(v->actions); free(v->rewards); free(v->dones); free(v->ep_reward); free(v->ep_step);
    free(v->fin_eps); free(v->fin_reward); free(v);
}
static void envs_reset_one(ref_envs* v, int i, uint64_t t) {
    v->ep_reward[i] = 0.0f; v->ep_step[i] = 0;      /* dones[] keeps the flags of the last act! (inspection) */
    if (v->sp.kind == DQN_ENV_TESTMDP) { for (int k = 0; k < 4; k++) v->tm_s[i * 4 + k] = 1; v->tm_t[i] = 1; }                 /* initialstate, :46-52 */
    else { v->gw_pos[i * 2] = 1 + (int)(env_rand(v->sp.seed, t, i, 5u) % (uint32_t)v->sp.size_x); v->gw_pos[i * 2 + 1] = 1 + (int)(env_rand(v->sp.seed, t, i, 6u) % (uint32_t)v->sp.size_y); }
}
/* observation of env i as float and (TestMDP) as the raw byte */
static void envs_observe(ref_envs* v, int i, float* of, uint8_t* ob) {
    if (v->sp.kind == DQN_ENV_TESTMDP) {
        int hw = v->H * v->W;
        for (int f = 0; f < v->E; f++) {                                   /* obs[.., c] = observations[s[end - c]], :56-58 */
            int c = f / hw, px = f % hw; uint8_t b = v->images[(v->tm_s[i * 4 + (3 - c)] - 1) * hw + px];
            if (of) of[f] = (float)b / 255.0f; if (ob) ob[f] = b;
        }
    } else for (int f = 0; f < 2; f++) { if (of) of[f] = (float)v->gw_pos[i * 2 + f]; }
}
int ref_envs_reset(ref_engine* e) {
    if (!e->envs) FAIL("no environments");
    for (int i = 0; i < e->envs->n; i++) envs_reset_one(e->envs, i, 0);
    return 0;
}
int ref_envs_create(ref_engine* e, const dqn_env_spec* sp) {
    if (e->hp.recurrence) FAIL("feed-forward only");
    envs_free(e->envs); e->envs = NULL;
    ref_envs* v = (ref_envs*)calloc(1, sizeof *v); v->sp = *sp; v->sp.images = NULL; v->n = sp->n_envs; v->E = e->obs_elems; v->H = e->hp.obs_h; v->W = e->hp.obs_w;
    int n = v->n;
    if (sp->kind == DQN_ENV_TESTMDP) {
        if (!sp->images || sp->o_stack != e->hp.obs_c || e->nA != 4) { envs_free(v); FAIL("TestMDP spec does not match the network"); }
        size_t ib = (size_t)3 * v->H * v->W; v->images = (uint8_t*)malloc(ib); memcpy(v->images, sp->images, ib);
        v->tm_s = (int8_t*)calloc((size_t)n * 4, 1); v->tm_t = (int32_t*)calloc(n, 4);
    } else if (sp->kind == DQN_ENV_GRIDWORLD) {
        if (e->obs_elems != 2 || e->nA != 4 || e->hp.obs_dtype == DQN_OBS_U8) { envs_free(v); FAIL("SimpleGridWorld spec does not match the network"); }
        v->gw_pos = (int32_t*)calloc((size_t)n * 2, 4);
    } else { envs_free(v); FAIL("unknown environment kind"); }
    v->actions = (int32_t*)calloc(n, 4); v->rewards = (float*)calloc(n, 4); v->dones = (uint8_t*)calloc(n, 1); v->ep_reward = (float*)calloc(n, 4);
    v->ep_step = (int32_t*)calloc(n, 4); v->fin_eps = (int64_t*)calloc(n, 8); v->fin_reward = (double*)calloc(n, 8);
    e->envs = v;
    return ref_envs_reset(e);
}
static void envs_act(ref_envs* v, int i, int a, uint64_t t) {
    float r; uint8_t done;
    if (v->sp.kind == DQN_ENV_TESTMDP) {
        int8_t* s = v->tm_s + i * 4;
        int was_second = s[3] == 2;                                       /* :62-64 */
        int8_t s0 = s[1], s1 = s[2], s2 = s[3], last = a < 3 ? (int8_t)(a + 1) : s2;    /* circshift(s, -1); a < 4 ? a : s_new[end-1], :69-74 */
        s[0] = s0; s[1] = s1; s[2] = s2; s[3] = last;
        r = (last == 1 ? -0.1f : (last == 2 ? 0.0f : 0.1f));
        if (was_second) r = r * -10.0f;                                   /* :77-83 */
        v->tm_t[i] += 1; done = v->tm_t[i] >= v->sp.max_time;             /* :85-87 */
    } else {
        int32_t* p = v->gw_pos + i * 2; float rv = 0.0f;
        for (int k = 0; k < v->sp.n_reward_cells; k++) if (p[0] == v->sp.reward_xy[k][0] && p[1] == v->sp.reward_xy[k][1]) rv = v->sp.reward_val[k];
        int at_reward = rv != 0.0f;
        int intended = u01(env_rand(v->sp.seed, t, i, 3u)) < v->sp.tprob;
        int other = (int)(env_rand(v->sp.seed, t, i, 4u) % 3u);
        int eff = intended ? a : (a + 1 + other) % 4;
        int dx = eff == 2 ? -1 : (eff == 3 ? 1 : 0), dy = eff == 0 ? 1 : (eff == 1 ? -1 : 0);
        int nx = p[0] + dx, ny = p[1] + dy;
        if (!at_reward && nx >= 1 && nx <= v->sp.size_x && ny >= 1 && ny <= v->sp.size_y) { p[0] = nx; p[1] = ny; }
        r = rv; done = (uint8_t)at_reward;
    }
    v->actions[i] = a; v->rewards[i] = r; v->dones[i] = done; v->ep_reward[i] += r; v->ep_step[i] += 1;
}
int ref_rollout(ref_engine* e, int n_steps, const dqn_rollout_cfg* cfg, dqn_rollout_stats* out) {
    ref_envs* v = e->envs; if (!v) FAIL("no environments");
    if (cfg->t0 < 1) FAIL("t0 counts from 1");
    int n = v->n, E = v->E, u8 = e->hp.obs_dtype == DQN_OBS_U8; int64_t trained = 0; float loss = 0.0f, gn = 0.0f;
    float* obs = (float*)malloc((size_t)n * E * 4); int32_t* greedy = (int32_t*)malloc((size_t)n * 4); float* td0 = (float*)malloc((size_t)n * 4);
    size_t rowb = (size_t)E * (u8 ? 1 : 4); uint8_t* srow = (uint8_t*)malloc(rowb * n); uint8_t* sprow = (uint8_t*)malloc(rowb * n);
    for (int k = 0; k < n_steps; k++) {
        int64_t t = cfg->t0 + k;
        float eps = cfg->eps_start - (float)t * ((cfg->eps_start - cfg->eps_stop) / cfg->eps_steps);
        if (!(cfg->eps_steps > 0.0f) || eps < cfg->eps_stop) eps = cfg->eps_stop;
        for (int i = 0; i < n; i++) envs_observe(v, i, obs + (size_t)i * E, u8 ? srow + i * rowb : NULL);
        if (!u8) memcpy(srow, obs, rowb * n);
        ref_greedy_action(e, obs, n, greedy);
        for (int i = 0; i < n; i++) {
            int a = greedy[i];
            if (u01(env_rand(v->sp.seed, (uint64_t)t, i, 1u)) < eps) a = (int)(env_rand(v->sp.seed, (uint64_t)t, i, 2u) % (uint32_t)e->nA);
            envs_act(v, i, a, (uint64_t)t);
            td0[i] = e->hp.prioritized_replay ? fabsf(v->rewards[i]) : 0.0f;
            if (u8) envs_observe(v, i, NULL, sprow + i * rowb); else envs_observe(v, i, (float*)(sprow + i * rowb), NULL);
        }
        if (ref_replay_add(e, srow, v->actions, v->rewards, sprow, v->dones, td0, n)) return -1;
        for (int i = 0; i < n; i++) if (v->dones[i] || v->ep_step[i] >= v->sp.max_episode_length) {
            v->fin_eps[i] += 1; v->fin_reward[i] += (double)v->ep_reward[i]; envs_reset_one(v, i, (uint64_t)t);
        }
        if (cfg->cadence_env_steps) {      /* train_freq / target_update_freq count ENV steps (src/solver.jl:136-145); the n transitions of the vector step are already in the replay */
            int64_t due = cfg->train_freq > 0 ? (t * n) / cfg->train_freq - ((t - 1) * n) / cfg->train_freq : 0;
            if (e->size >= e->B) for (int64_t q = 0; q < due; q++) { if (ref_train_step(e, NULL, &loss, &gn, NULL)) return -1; trained++; }
            if (cfg->target_update_freq > 0 && (t * n) / cfg->target_update_freq != ((t - 1) * n) / cfg->target_update_freq) ref_sync_target(e);
            continue;
        }
        if (cfg->train_freq > 0 && t % cfg->train_freq == 0 && e->size >= e->B) { if (ref_train_step(e, NULL, &loss, &gn, NULL)) return -1; trained++; }
        if (cfg->target_update_freq > 0 && t % cfg->target_update_freq == 0) ref_sync_target(e);
    }
    if (out) { out->episodes = 0; out->reward_sum = 0.0; out->train_steps = trained; out->last_loss = loss; out->last_grad_norm = gn;
               for (int i = 0; i < n; i++) { out->episodes += v->fin_eps[i]; out->reward_sum += v->fin_reward[i]; } }
    free(obs); free(greedy); free(td0); free(srow); free(sprow); return 0;
}
int ref_envs_peek(ref_engine* e, float* obs, int32_t* actions, float* rewards, uint8_t* dones) {
    ref_envs* v = e->envs; if (!v) FAIL("no environments");
    if (obs) for (int i = 0; i < v->n; i++) envs_observe(v, i, obs + (size_t)i * v->E, NULL);
    if (actions) memcpy(actions, v->actions, (size_t)v->n * 4);
    if (rewards) memcpy(rewards, v->rewards, (size_t)v->n * 4);
    if (dones) memcpy(dones, v->dones, (size_t)v->n);
    return 0;
}

/* basic_evaluation (src/evaluation_policy.jl:17-42): n_eval fresh copies of the MDP of ref_envs_create, one greedy episode each,
 * while !done && step <= max_episode_length; r_tot accumulates in Float64.  Same Philox draws as dqn_evaluate (seed, t from 1). */
int ref_evaluate(ref_engine* e, int n_eval, int max_episode_length, uint64_t seed, double* avg_reward, double* avg_steps) {
    ref_envs* tr = e->envs; if (!tr) FAIL("no environments");
    ref_envs v = *tr; int n = n_eval, E = tr->E;
    v.n = n; v.sp.seed = seed;
    v.tm_s = (int8_t*)calloc((size_t)n * 4, 1); v.tm_t = (int32_t*)calloc(n, 4); v.gw_pos = (int32_t*)calloc((size_t)n * 2, 4);
    v.actions = (int32_t*)calloc(n, 4); v.rewards = (float*)calloc(n, 4); v.dones = (uint8_t*)calloc(n, 1); v.ep_reward = (float*)calloc(n, 4);
    v.ep_step = (int32_t*)calloc(n, 4); v.fin_eps = NULL; v.fin_reward = NULL;
    float* obs = (float*)malloc((size_t)n * E * 4); int32_t* greedy = (int32_t*)malloc((size_t)n * 4);
    uint8_t* over = (uint8_t*)calloc(n, 1); double* rtot = (double*)calloc(n, 8);
    for (int i = 0; i < n; i++) envs_reset_one(&v, i, 0);
    for (int k = 0; k <= max_episode_length; k++) {
        uint64_t t = (uint64_t)k + 1;
        for (int i = 0; i < n; i++) envs_observe(&v, i, obs + (size_t)i * E, NULL);
        ref_greedy_action(e, obs, n, greedy);
        int alive = 0;
        for (int i = 0; i < n; i++) {
            if (over[i]) continue;
            envs_act(&v, i, greedy[i], t);
            rtot[i] += (double)v.rewards[i];
            over[i] = v.dones[i] || v.ep_step[i] > max_episode_length;
            alive |= !over[i];
        }
        if (!alive) break;
    }
    double r = 0.0, st = 0.0;
    for (int i = 0; i < n; i++) { r += rtot[i]; st += (double)v.ep_step[i]; }
    if (avg_reward) *avg_reward = r / n; if (avg_steps) *avg_steps = st / n;
    free(v.tm_s); free(v.tm_t); free(v.gw_pos); free(v.actions); free(v.rewards); free(v.dones); free(v.ep_reward); free(v.ep_step);
    free(obs); free(greedy); free(over); free(rtot); return 0;
}
