"""
ctypes view of include/dqn_mi355x.h: struct layouts, function prototypes and a
thin handle class over libdqn_mi355x.so.  Standalone (no package-relative
imports); `bind` and `Handle` are generic in the symbol prefix and in how a
handle is created, so other libraries with the same struct layouts can reuse
them (the test infrastructure under oracle/ does; nothing about it lives here).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

LAYER_DENSE, LAYER_CONV, LAYER_LSTM = 0, 1, 2
ACT_IDENTITY, ACT_RELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
STREAM_BASE, STREAM_VAL, STREAM_ADV = 0, 1, 2
OBS_F32, OBS_U8 = 0, 1
NET_ONLINE, NET_TARGET = 0, 1


class LayerDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("act", C.c_int32), ("stream", C.c_int32),
                ("n_in", C.c_int32), ("n_out", C.c_int32),
                ("cin", C.c_int32), ("cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("sh", C.c_int32), ("sw", C.c_int32)]


class LayerPlan(C.Structure):
    _fields_ = [("fwd_kc", C.c_int32), ("dx_kc", C.c_int32), ("dw_kc", C.c_int32)]

    def astuple(self):
        return (self.fwd_kc, self.dx_kc, self.dw_kc)


class HParams(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("n_actions", C.c_int32),
                ("obs_c", C.c_int32), ("obs_h", C.c_int32), ("obs_w", C.c_int32), ("obs_dtype", C.c_int32),
                ("learning_rate", C.c_float),
                ("adam_beta1", C.c_double), ("adam_beta2", C.c_double), ("adam_eps", C.c_double),
                ("adam_f64_scalars", C.c_int32),
                ("gamma", C.c_float),
                ("double_q", C.c_int32), ("dueling", C.c_int32), ("prioritized_replay", C.c_int32),
                ("buffer_size", C.c_int64),
                ("prio_alpha", C.c_float), ("prio_beta", C.c_float), ("prio_eps", C.c_float),
                ("seed", C.c_uint64),
                ("use_graph", C.c_int32), ("use_mfma", C.c_int32),
                ("recurrence", C.c_int32), ("trace_length", C.c_int32),
                ("sample_distinct", C.c_int32), ("reserved", C.c_int32 * 3)]


def default_hparams(**kw) -> HParams:
    """Reference defaults: src/solver.jl:3-27, src/prioritized_experience_replay.jl:42-45."""
    hp = HParams()
    hp.batch_size, hp.n_actions = 32, 0
    hp.obs_c, hp.obs_h, hp.obs_w, hp.obs_dtype = 0, 1, 1, OBS_F32
    hp.learning_rate = 1e-4
    hp.adam_beta1, hp.adam_beta2, hp.adam_eps, hp.adam_f64_scalars = 0.9, 0.999, 1e-8, 1
    hp.gamma = 1.0
    hp.double_q, hp.dueling, hp.prioritized_replay = 1, 1, 1
    hp.buffer_size = 1000
    hp.prio_alpha, hp.prio_beta, hp.prio_eps = 0.6, 0.4, 1e-3
    hp.seed = 0
    hp.use_graph, hp.use_mfma = 1, 1
    hp.recurrence, hp.trace_length = 0, 40
    for k, v in kw.items():
        if not hasattr(hp, k):
            raise AttributeError(k)
        setattr(hp, k, v)
    return hp


class EnvSpec(C.Structure):
    """dqn_env_spec (include/dqn_mi355x.h): n lock-stepped device environments."""
    _fields_ = [("kind", C.c_int32), ("n_envs", C.c_int32), ("max_episode_length", C.c_int32), ("seed", C.c_uint64),
                ("o_stack", C.c_int32), ("max_time", C.c_int32), ("images", C.c_void_p),
                ("size_x", C.c_int32), ("size_y", C.c_int32), ("tprob", C.c_float), ("n_reward_cells", C.c_int32),
                ("reward_xy", (C.c_int32 * 2) * 8), ("reward_val", C.c_float * 8)]


class RolloutCfg(C.Structure):
    _fields_ = [("train_freq", C.c_int32), ("target_update_freq", C.c_int32),
                ("eps_start", C.c_float), ("eps_stop", C.c_float), ("eps_steps", C.c_float), ("cadence_env_steps", C.c_int32), ("t0", C.c_int64)]


class RolloutStats(C.Structure):
    _fields_ = [("episodes", C.c_int64), ("reward_sum", C.c_double), ("train_steps", C.c_int64),
                ("last_loss", C.c_float), ("last_grad_norm", C.c_float)]


class Counters(C.Structure):
    _fields_ = [("size", C.c_int64), ("widx", C.c_int64), ("sample_ctr", C.c_uint64), ("train_steps", C.c_uint64)]


ENV_TESTMDP, ENV_GRIDWORLD = 0, 1

_P = C.POINTER
_f32p, _i32p, _i64p, _u8p, _f64p = _P(C.c_float), _P(C.c_int32), _P(C.c_int64), _P(C.c_uint8), _P(C.c_double)
_vp, _sz = C.c_void_p, C.c_size_t

class CommInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rccl_nranks", "rccl_rank", "rccl_device", "engine_world", "engine_rank", "sim_world", "exchange", "dp_overlap")]


# name -> argtypes (without prefix).  All return int unless listed in _RESTYPE.
PROTOS = {
    "plan_version": [],
    "plan_default": [_P(LayerDesc), C.c_int, _P(HParams), _P(LayerPlan)],
    "engine_create": [_P(LayerDesc), C.c_int, _P(HParams), _P(LayerPlan), C.c_int, _P(_vp)],
    "engine_destroy": [_vp],
    "engine_get_plan": [_vp, _P(LayerPlan)],
    "n_params": [_vp, _P(_sz)],
    "batch_arena_elem_bytes": [_vp, _P(C.c_int)],
    "set_params": [_vp, C.c_int, _f32p, _sz],
    "get_params": [_vp, C.c_int, _f32p, _sz],
    "sync_target": [_vp],
    "get_adam_state": [_vp, _f32p, _f32p, _f64p, _sz],
    "set_adam_state": [_vp, _f32p, _f32p, _f64p, _sz],
    "replay_add": [_vp, _vp, _i32p, _f32p, _vp, _u8p, _f32p, C.c_int],
    "replay_size": [_vp, _i64p, _i64p],
    "replay_get_priorities": [_vp, _f32p, C.c_int64],
    "replay_sample": [_vp, _i64p],
    "replay_get_batch": [_vp, _i64p, _f32p, _i32p, _f32p, _f32p, _f32p, _f32p],
    "update_priorities": [_vp, _i64p, _f32p, C.c_int],
    "train_step": [_vp, _i64p, _f32p, _f32p, _f32p],
    "train_steps": [_vp, C.c_int, _f32p, _f32p],
    "train_step_async": [_vp, _i64p, _P(C.c_uint64)],
    "step_scalars": [_vp, C.c_uint64, C.c_int, _f32p, _f32p, _P(C.c_uint64)],
    "get_last_q": [_vp, _f32p, _f32p, _f32p, _i32p, _f32p],
    "get_last_indices": [_vp, _i64p],
    "get_grads": [_vp, _f32p, _sz],
    "forward": [_vp, C.c_int, _f32p, C.c_int, _f32p],
    "greedy_action": [_vp, _f32p, C.c_int, _i32p],
    "comm_unique_id": [_vp],
    "comm_init": [_vp, _vp, C.c_int, C.c_int],
    "comm_info": [_vp, _P(CommInfo)],
    "comm_exchange_bytes": [_vp, _P(C.c_int64)],
    "sim_ranks_step": [_vp, _i64p, _f32p, _f32p, _f32p],
    "debug_ktrace": [_vp, _P(C.c_uint64), _sz],
    "stream_sync": [_vp],
    "stream_handle": [_vp, _P(_vp)],
    "profile_step": [_vp, C.c_int, _P(C.c_char_p), _f32p, _P(C.c_int)],
    "profile_steady_step": [_vp, C.c_int, _P(C.c_char_p), _f32p, _P(C.c_int)],
    "hparams_default": [_P(HParams)],
    "episode_add": [_vp, _vp, _i32p, _f32p, _vp, _u8p, C.c_int],
    "episode_commit": [_vp],
    "episode_count": [_vp, _i64p, _i64p],
    "episode_get_batch": [_vp, _i64p, _i32p, _f32p, _i32p, _f32p, _f32p, _f32p, _i32p],
    "train_step_drqn": [_vp, _i64p, _i32p, _f32p, _f32p],
    "reset_state": [_vp],
    "get_hidden": [_vp, _f32p, _sz],
    "set_hidden": [_vp, _f32p, _sz],
    "envs_create": [_vp, _P(EnvSpec)],
    "envs_reset": [_vp],
    "rollout": [_vp, C.c_int, _P(RolloutCfg), _P(RolloutStats)],
    "envs_peek": [_vp, _f32p, _i32p, _f32p, _u8p],
    "envs_info": [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "evaluate": [_vp, C.c_int, C.c_int, C.c_uint64, _f64p, _f64p],
    "replay_export": [_vp, C.c_int64, C.c_int64, _vp, _vp, _i32p, _f32p, _u8p, _f32p],
    "replay_import": [_vp, C.c_int64, _vp, _vp, _i32p, _f32p, _u8p, _f32p],
    "episode_export": [_vp, C.c_int64, C.c_int64, _f32p, _f32p, _i32p, _f32p, _u8p, _i32p],
    "episode_import": [_vp, C.c_int64, _f32p, _f32p, _i32p, _f32p, _u8p, _i32p],
    "get_counters": [_vp, _P(Counters)],
    "set_counters": [_vp, _P(Counters)],
}

class DQNError(RuntimeError):
    """Mirrors the reference's thrown Strings / AssertionErrors."""


def bind(lib: C.CDLL, prefix: str, aliases=None, argtypes=None):
    """Attach prototypes; returns {name: callable} for every symbol the library exports.
    aliases: {name: spelling} for symbols spelled differently; argtypes: {name: [ctypes]} overriding PROTOS."""
    fns = {}
    for name, args in PROTOS.items():
        f = getattr(lib, prefix + (aliases or {}).get(name, name), None)
        if f is None:
            continue
        f.argtypes = (argtypes or {}).get(name, args)
        f.restype = C.c_int
        fns[name] = f
    le = getattr(lib, prefix + "last_error")
    le.restype = C.c_char_p
    fns["last_error"] = le
    return fns


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _as(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


class Handle:
    """NumPy-friendly wrapper over one engine handle."""

    def __init__(self, fns, layers, hp: HParams, plan=None, device=0):
        self.f, self.hp = fns, hp
        self.layers = (LayerDesc * len(layers))(*layers)
        self.n_layers = len(layers)
        self.B, self.nA = hp.batch_size, hp.n_actions
        self.obs_elems = hp.obs_c * hp.obs_h * hp.obs_w
        self.obs_shape = (hp.obs_c, hp.obs_h, hp.obs_w)
        self.obs_np = np.uint8 if hp.obs_dtype == OBS_U8 else np.float32
        parr = None
        if plan is not None:
            parr = (LayerPlan * len(layers))(*[LayerPlan(*p) for p in plan])
        h = C.c_void_p()
        rc = self._create(parr, device, h)
        self._handle = h
        self._check(rc)
        n = C.c_size_t()
        self._check(fns["n_params"](h, C.byref(n)))
        self.P = n.value

    def _create(self, parr, device, h):
        """dqn_engine_create; subclasses binding another library override this"""
        return self.f["engine_create"](self.layers, self.n_layers, C.byref(self.hp), parr, device, C.byref(h))

    def _check(self, rc):
        if rc != 0:
            raise DQNError(self.f["last_error"]().decode())

    @property
    def _h(self):
        h = getattr(self, "_handle", None)
        if h is None or not h.value:
            raise DQNError("the engine has been closed (or was never created)")
        return h

    def close(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self.f["engine_destroy"](h)
            self._handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- plan / params
    def plan(self):
        p = (LayerPlan * self.n_layers)()
        self._check(self.f["engine_get_plan"](self._h, p))
        return [x.astuple() for x in p]

    def batch_arena_elem_bytes(self):
        n = C.c_int()
        self._check(self.f["batch_arena_elem_bytes"](self._h, C.byref(n)))
        return n.value

    def set_params(self, flat, which=NET_ONLINE):
        flat = _as(flat, np.float32).reshape(-1)
        self._check(self.f["set_params"](self._h, which, _ptr(flat, _f32p), flat.size))

    def get_params(self, which=NET_ONLINE):
        out = np.empty(self.P, np.float32)
        self._check(self.f["get_params"](self._h, which, _ptr(out, _f32p), out.size))
        return out

    def get_grads(self):
        out = np.empty(self.P, np.float32)
        self._check(self.f["get_grads"](self._h, _ptr(out, _f32p), out.size))
        return out

    def get_adam_state(self):
        m, v, bp = np.empty(self.P, np.float32), np.empty(self.P, np.float32), np.empty(2, np.float64)
        self._check(self.f["get_adam_state"](self._h, _ptr(m, _f32p), _ptr(v, _f32p), _ptr(bp, _f64p), self.P))
        return m, v, bp

    def set_adam_state(self, m, v, bp):
        m, v, bp = _as(m, np.float32), _as(v, np.float32), _as(bp, np.float64)
        self._check(self.f["set_adam_state"](self._h, _ptr(m, _f32p), _ptr(v, _f32p), _ptr(bp, _f64p), self.P))

    def sync_target(self):
        self._check(self.f["sync_target"](self._h))

    def _obs_rows(self, x, what):
        """observation rows in the replay's storage dtype.  A u8 replay stores BYTES that training reads back as byte / 255f0: a float
        array would be silently truncated to 0/1 by a cast, so only uint8 input is accepted there."""
        if self.obs_np is np.uint8 and np.asarray(x).dtype != np.uint8:
            raise DQNError(f"{what}: this replay stores uint8 observations (obs_dtype = OBS_U8); got {np.asarray(x).dtype} -- pass the raw bytes")
        return _as(x, self.obs_np).reshape(-1, self.obs_elems)

    # ---- replay
    def replay_add(self, s, a, r, sp, done, td_err=None):
        s = self._obs_rows(s, "replay_add")
        sp = self._obs_rows(sp, "replay_add")
        n = s.shape[0]
        a, r, done = _as(np.atleast_1d(a), np.int32), _as(np.atleast_1d(r), np.float32), _as(np.atleast_1d(done), np.uint8)
        td = None if td_err is None else _as(np.atleast_1d(td_err), np.float32)
        assert sp.shape[0] == n and a.size == n and r.size == n and done.size == n
        self._check(self.f["replay_add"](self._h, s.ctypes.data_as(_vp), _ptr(a, _i32p), _ptr(r, _f32p),
                                         sp.ctypes.data_as(_vp), _ptr(done, _u8p), _ptr(td, _f32p), n))

    def replay_size(self):
        cur, cap = C.c_int64(), C.c_int64()
        self._check(self.f["replay_size"](self._h, C.byref(cur), C.byref(cap)))
        return cur.value, cap.value

    def replay_priorities(self):
        n = self.replay_size()[0]
        out = np.empty(n, np.float32)
        self._check(self.f["replay_get_priorities"](self._h, _ptr(out, _f32p), n))
        return out

    def replay_sample(self):
        idx = np.empty(self.B, np.int64)
        self._check(self.f["replay_sample"](self._h, _ptr(idx, _i64p)))
        return idx

    def get_batch(self, idx):
        idx = _as(idx, np.int64)
        B = self.B
        s = np.empty((B,) + self.obs_shape, np.float32)
        sp = np.empty_like(s)
        a, r, done, w = np.empty(B, np.int32), np.empty(B, np.float32), np.empty(B, np.float32), np.empty(B, np.float32)
        self._check(self.f["replay_get_batch"](self._h, _ptr(idx, _i64p), _ptr(s, _f32p), _ptr(a, _i32p), _ptr(r, _f32p),
                                               _ptr(sp, _f32p), _ptr(done, _f32p), _ptr(w, _f32p)))
        return s, a, r, sp, done, w

    def update_priorities(self, idx, td):
        idx, td = _as(idx, np.int64), _as(td, np.float32)
        self._check(self.f["update_priorities"](self._h, _ptr(idx, _i64p), _ptr(td, _f32p), idx.size))

    # ---- train
    def train_step(self, idx=None, want_td=True, sync=True):
        """batch_train! -> (loss_val, grad_norm[, td]) (src/solver.jl:235)."""
        idx = _as(idx, np.int64)
        if not sync:
            self._check(self.f["train_step"](self._h, _ptr(idx, _i64p), None, None, None))
            return None
        loss, gn = C.c_float(), C.c_float()
        td = np.empty(self.B, np.float32) if want_td else None
        self._check(self.f["train_step"](self._h, _ptr(idx, _i64p), C.byref(loss), C.byref(gn), _ptr(td, _f32p)))
        return (loss.value, gn.value, td) if want_td else (loss.value, gn.value)

    def comm_info(self):
        """what the engine's RCCL communicator itself reports (ncclCommCount / ncclCommUserRank / ncclCommCuDevice) beside what the engine was told"""
        ci = CommInfo()
        self._check(self.f["comm_info"](self._h, C.byref(ci)))
        return {n: int(getattr(ci, n)) for n, _ in ci._fields_}

    def comm_exchange_bytes(self):
        n = C.c_int64()
        self._check(self.f["comm_exchange_bytes"](self._h, C.byref(n)))
        return n.value

    def train_step_async(self, idx=None):
        """enqueue one batch_train! and return at once; the ticket names the (loss, grad_norm) record the step's last launch publishes to the host mailbox"""
        idx = _as(idx, np.int64)
        t = C.c_uint64()
        self._check(self.f["train_step_async"](self._h, _ptr(idx, _i64p), C.byref(t)))
        return t.value

    def step_scalars(self, ticket, wait=True):
        """(loss, grad_norm) of the step that returned `ticket`; wait=False returns None while the step is still running"""
        loss, gn, pub = C.c_float(), C.c_float(), C.c_uint64()
        self._check(self.f["step_scalars"](self._h, int(ticket), 1 if wait else 0, C.byref(loss), C.byref(gn), C.byref(pub)))
        return (loss.value, gn.value) if pub.value == ticket else None

    def train_steps(self, n):
        loss, gn = C.c_float(), C.c_float()
        self._check(self.f["train_steps"](self._h, n, C.byref(loss), C.byref(gn)))
        return loss.value, gn.value

    def last_q(self):
        B, nA = self.B, self.nA
        qs, qsp, qt = (np.empty((B, nA), np.float32) for _ in range(3))
        best, y = np.empty(B, np.int32), np.empty(B, np.float32)
        self._check(self.f["get_last_q"](self._h, _ptr(qs, _f32p), _ptr(qsp, _f32p), _ptr(qt, _f32p), _ptr(best, _i32p), _ptr(y, _f32p)))
        return dict(q_on_s=qs, q_on_sp=qsp, q_tg_sp=qt, best_a=best, y=y)

    def last_indices(self):
        idx = np.empty(self.B, np.int64)
        self._check(self.f["get_last_indices"](self._h, _ptr(idx, _i64p)))
        return idx

    # ---- policy
    def forward(self, obs, which=NET_ONLINE):
        obs = _as(obs, np.float32).reshape(-1, self.obs_elems)
        q = np.empty((obs.shape[0], self.nA), np.float32)
        self._check(self.f["forward"](self._h, which, _ptr(obs, _f32p), obs.shape[0], _ptr(q, _f32p)))
        return q

    def greedy_action(self, obs):
        obs = _as(obs, np.float32).reshape(-1, self.obs_elems)
        a = np.empty(obs.shape[0], np.int32)
        self._check(self.f["greedy_action"](self._h, _ptr(obs, _f32p), obs.shape[0], _ptr(a, _i32p)))
        return a

    # ---- DRQN
    def episode_add(self, s, a, r, sp, done):
        # the episode replay always stores Float32 rows (dqn_episode_add copies E*4 bytes per observation; recurrence with a u8 replay is refused at creation)
        s = _as(s, np.float32).reshape(-1, self.obs_elems)
        sp = _as(sp, np.float32).reshape(-1, self.obs_elems)
        n = s.shape[0]
        a, r, done = _as(np.atleast_1d(a), np.int32), _as(np.atleast_1d(r), np.float32), _as(np.atleast_1d(done), np.uint8)
        self._check(self.f["episode_add"](self._h, s.ctypes.data_as(_vp), _ptr(a, _i32p), _ptr(r, _f32p), sp.ctypes.data_as(_vp), _ptr(done, _u8p), n))

    def episode_commit(self):
        self._check(self.f["episode_commit"](self._h))

    def episode_count(self):
        cur, cap = C.c_int64(), C.c_int64()
        self._check(self.f["episode_count"](self._h, C.byref(cur), C.byref(cap)))
        return cur.value, cap.value

    def episode_get_batch(self, ep_idx, ep_start):
        T, B = self.hp.trace_length, self.B
        ep_idx, ep_start = _as(ep_idx, np.int64), _as(ep_start, np.int32)
        s = np.empty((T, B) + self.obs_shape, np.float32); sp = np.empty_like(s)
        a, m = np.empty((T, B), np.int32), np.empty((T, B), np.int32)
        r, d = np.empty((T, B), np.float32), np.empty((T, B), np.float32)
        self._check(self.f["episode_get_batch"](self._h, _ptr(ep_idx, _i64p), _ptr(ep_start, _i32p), _ptr(s, _f32p), _ptr(a, _i32p), _ptr(r, _f32p),
                                                _ptr(sp, _f32p), _ptr(d, _f32p), _ptr(m, _i32p)))
        return s, a, r, sp, d, m

    def train_step_drqn(self, ep_idx=None, ep_start=None):
        ep_idx, ep_start = _as(ep_idx, np.int64), _as(ep_start, np.int32)
        loss, gn = C.c_float(), C.c_float()
        self._check(self.f["train_step_drqn"](self._h, _ptr(ep_idx, _i64p), _ptr(ep_start, _i32p), C.byref(loss), C.byref(gn)))
        return loss.value, gn.value

    def reset_state(self):
        self._check(self.f["reset_state"](self._h))

    def hidden_size(self, streams=1):
        """floats in the flat Recur state: per LSTM layer h then c, each [out][streams] (src/helpers.jl:61-63)"""
        return sum(2 * int(l.n_out) * streams for l in self.layers[:self.n_layers] if l.kind == LAYER_LSTM)

    def get_hidden(self, streams=1):
        """hiddenstates(m) (src/helpers.jl:61-63): list of (h, c) per LSTM layer, each [out, streams]"""
        buf = np.empty(self.hidden_size(streams), np.float32)
        self._check(self.f["get_hidden"](self._h, _ptr(buf, _f32p), buf.size))
        out, off = [], 0
        for l in self.layers[:self.n_layers]:
            if l.kind == LAYER_LSTM:
                m = int(l.n_out) * streams
                out.append((buf[off:off + m].reshape(int(l.n_out), streams).copy(), buf[off + m:off + 2 * m].reshape(int(l.n_out), streams).copy()))
                off += 2 * m
        return out

    def set_hidden(self, hs):
        """sethiddenstates!(m, hs) (src/helpers.jl:71-79)"""
        buf = np.concatenate([np.concatenate([_as(h, np.float32).ravel(), _as(c, np.float32).ravel()]) for h, c in hs]) if hs else np.empty(0, np.float32)
        buf = np.ascontiguousarray(buf, np.float32)
        self._check(self.f["set_hidden"](self._h, _ptr(buf, _f32p), buf.size))

    # ---- vectorised environments on the device
    def envs_create(self, env, n_envs=None, max_episode_length=100, seed=0):
        """`env` is an envs.TestMDP / envs.SimpleGridWorld instance used as the SPEC (images, sizes, rewards); its own state is not used."""
        sp = EnvSpec()
        sp.n_envs = int(n_envs if n_envs is not None else env.n)
        sp.max_episode_length, sp.seed = int(max_episode_length), int(seed)
        if hasattr(env, "images"):
            sp.kind, sp.o_stack, sp.max_time = ENV_TESTMDP, env.o_stack, env.max_time
            self._env_images = np.ascontiguousarray(env.images, np.uint8)
            sp.images = self._env_images.ctypes.data
        else:
            sp.kind, sp.size_x, sp.size_y, sp.tprob = ENV_GRIDWORLD, env.size[0], env.size[1], env.tprob
            sp.n_reward_cells = len(env.reward_cells)
            for k, ((x, y), v) in enumerate(env.reward_cells.items()):
                sp.reward_xy[k][0], sp.reward_xy[k][1], sp.reward_val[k] = x, y, v
        self._check(self.f["envs_create"](self._h, C.byref(sp)))
        self.n_envs = sp.n_envs

    def envs_reset(self):
        self._check(self.f["envs_reset"](self._h))

    def rollout(self, n_steps, t0=1, train_freq=4, target_update_freq=500, eps=(1.0, 0.01, 5000.0), stats=True, env_step_cadence=False):
        """env_step_cadence: train_freq / target_update_freq count ENV steps as in the reference's loop (src/solver.jl:136-145) instead of vector steps"""
        cfg = RolloutCfg(int(train_freq), int(target_update_freq), float(eps[0]), float(eps[1]), float(eps[2]), 1 if env_step_cadence else 0, int(t0))
        st = RolloutStats()
        self._check(self.f["rollout"](self._h, int(n_steps), C.byref(cfg), C.byref(st) if stats else None))
        return dict(episodes=st.episodes, reward_sum=st.reward_sum, train_steps=st.train_steps, loss=st.last_loss, grad_norm=st.last_grad_norm) if stats else None

    def evaluate(self, n_eval, max_episode_length=100, seed=0):
        """basic_evaluation on the device: (average return, average steps) of n_eval greedy episodes."""
        r, st = C.c_double(), C.c_double()
        self._check(self.f["evaluate"](self._h, int(n_eval), int(max_episode_length), int(seed), C.byref(r), C.byref(st)))
        return r.value, st.value

    def envs_info(self):
        """(n_envs, fused_tail): fused_tail = the acting step's tail runs as one launch (act_head.hip)"""
        n, f = C.c_int32(), C.c_int32()
        self._check(self.f["envs_info"](self._h, C.byref(n), C.byref(f)))
        return n.value, bool(f.value)

    def envs_peek(self):
        n = self.n_envs
        obs = np.empty((n,) + self.obs_shape, np.float32)
        a, r, d = np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.uint8)
        self._check(self.f["envs_peek"](self._h, _ptr(obs, _f32p), _ptr(a, _i32p), _ptr(r, _f32p), _ptr(d, _u8p)))
        return obs, a, r, d

    # ---- checkpoint / resume (product only)
    def replay_export(self, first=0, n=None):
        n = self.replay_size()[0] - first if n is None else n
        s = np.empty((n,) + self.obs_shape, self.obs_np); sp = np.empty_like(s)
        a, r, d, pr = np.empty(n, np.int32), np.empty(n, np.float32), np.empty(n, np.uint8), np.empty(n, np.float32)
        self._check(self.f["replay_export"](self._h, first, n, s.ctypes.data_as(_vp), sp.ctypes.data_as(_vp), _ptr(a, _i32p), _ptr(r, _f32p), _ptr(d, _u8p), _ptr(pr, _f32p)))
        return s, sp, a, r, d, pr

    def replay_import(self, s, sp, a, r, done, priorities):
        s = self._obs_rows(s, "replay_import"); sp = self._obs_rows(sp, "replay_import")
        a, r, done, priorities = _as(a, np.int32), _as(r, np.float32), _as(done, np.uint8), _as(priorities, np.float32)
        self._check(self.f["replay_import"](self._h, s.shape[0], s.ctypes.data_as(_vp), sp.ctypes.data_as(_vp), _ptr(a, _i32p), _ptr(r, _f32p), _ptr(done, _u8p), _ptr(priorities, _f32p)))

    def get_counters(self):
        c = Counters()
        self._check(self.f["get_counters"](self._h, C.byref(c)))
        return dict(size=c.size, widx=c.widx, sample_ctr=c.sample_ctr, train_steps=c.train_steps)

    def set_counters(self, size, widx, sample_ctr, train_steps):
        c = Counters(int(size), int(widx), int(sample_ctr), int(train_steps))
        self._check(self.f["set_counters"](self._h, C.byref(c)))

    def episode_export(self, first=0, n=None):
        n = self.episode_count()[0] - first if n is None else n
        T = self.hp.trace_length
        s = np.empty((n, T) + self.obs_shape, np.float32); sp = np.empty_like(s)
        a, r, d, ln = np.empty((n, T), np.int32), np.empty((n, T), np.float32), np.empty((n, T), np.uint8), np.empty(n, np.int32)
        self._check(self.f["episode_export"](self._h, first, n, _ptr(s, _f32p), _ptr(sp, _f32p), _ptr(a, _i32p), _ptr(r, _f32p), _ptr(d, _u8p), _ptr(ln, _i32p)))
        return s, sp, a, r, d, ln

    def episode_import(self, s, sp, a, r, done, ep_len):
        s, sp = _as(s, np.float32), _as(sp, np.float32)
        a, r, done, ep_len = _as(a, np.int32), _as(r, np.float32), _as(done, np.uint8), _as(ep_len, np.int32)
        self._check(self.f["episode_import"](self._h, ep_len.size, _ptr(s, _f32p), _ptr(sp, _f32p), _ptr(a, _i32p), _ptr(r, _f32p), _ptr(done, _u8p), _ptr(ep_len, _i32p)))

    def checkpoint(self):
        """everything a bit-exact resume of the train loop needs, as a dict of NumPy arrays (np.savez-able)."""
        if self.hp.recurrence:
            s, sp, a, r, d, ln = self.episode_export()
            m, v, bp = self.get_adam_state()
            c = self.get_counters()
            return dict(p_on=self.get_params(NET_ONLINE), p_tg=self.get_params(NET_TARGET), adam_m=m, adam_v=v, adam_bp=np.asarray(bp, np.float64),
                        s=s, sp=sp, a=a, r=r, done=d, ep_len=ln, counters=np.array([c["size"], c["widx"], c["sample_ctr"], c["train_steps"]], np.uint64),      # the draw counter uses all 64 bits
                        plan=np.array(self.plan(), np.int32), plan_version=np.int32(self.f["plan_version"]()))
        s, sp, a, r, d, pr = self.replay_export()
        m, v, bp = self.get_adam_state()
        c = self.get_counters()
        return dict(p_on=self.get_params(NET_ONLINE), p_tg=self.get_params(NET_TARGET), adam_m=m, adam_v=v, adam_bp=np.asarray(bp, np.float64),
                    s=s, sp=sp, a=a, r=r, done=d, priorities=pr, counters=np.array([c["size"], c["widx"], c["sample_ctr"], c["train_steps"]], np.int64),
                    plan=np.array(self.plan(), np.int32), plan_version=np.int32(self.f["plan_version"]()))

    def restore(self, ck):
        # a run continues bit for bit only under the plan (rounding order) and the plan semantics it was saved with
        if "plan_version" in ck and int(ck["plan_version"]) != self.f["plan_version"]():
            raise DQNError(f"checkpoint was written under plan version {int(ck['plan_version'])}, this library is version {self.f['plan_version']()}: the rounding order differs, a bit-exact resume is impossible")
        if "plan" in ck and [tuple(int(x) for x in r) for r in np.asarray(ck["plan"]).reshape(-1, 3)] != [tuple(p) for p in self.plan()]:
            raise DQNError("checkpoint was written under a different summation plan: create the engine with plan=checkpoint['plan']")
        self.set_params(ck["p_on"], NET_ONLINE); self.set_params(ck["p_tg"], NET_TARGET)
        if self.hp.recurrence:
            self.episode_import(ck["s"], ck["sp"], ck["a"], ck["r"], ck["done"], ck["ep_len"])
        else:
            self.replay_import(ck["s"], ck["sp"], ck["a"], ck["r"], ck["done"], ck["priorities"])
        size, widx, sctr, steps = (int(x) for x in ck["counters"])
        self.set_counters(size, widx, sctr, steps)
        self.set_adam_state(ck["adam_m"], ck["adam_v"], ck["adam_bp"])

    # ---- misc (product only)
    def sync(self):
        self._check(self.f["stream_sync"](self._h))

    def stream_handle(self):
        p = C.c_void_p()
        self._check(self.f["stream_handle"](self._h, C.byref(p)))
        return p.value

    def profile_step(self, max_entries=128, steady=False):
        """[(launch name, ms)] of one train step, eager launches timed with HIP events.  steady: the timed step is a MIDDLE step of
        train_steps(n) (two steps run; see dqn_profile_steady_step)."""
        names = (C.c_char_p * max_entries)()
        ms = np.zeros(max_entries, np.float32)
        n = C.c_int()
        self._check(self.f["profile_steady_step" if steady else "profile_step"](self._h, max_entries, names, _ptr(ms, _f32p), C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]

    def sim_ranks_step(self, idx):
        """test hook (engine created under DQN_SIM_WORLD=k): one data-parallel step with k distinct batches idx[k][B] -> (loss[k], grad_norm, td[k][B])."""
        idx = _as(idx, np.int64)
        k = idx.size // self.B
        loss, td, gn = np.empty(k, np.float32), np.empty((k, self.B), np.float32), C.c_float()
        self._check(self.f["sim_ranks_step"](self._h, _ptr(idx, _i64p), _ptr(loss, _f32p), C.byref(gn), _ptr(td, _f32p)))
        return loss, gn.value, td

    def comm_init(self, id128: bytes, rank: int, world: int):
        buf = C.create_string_buffer(id128, 128)
        self._check(self.f["comm_init"](self._h, C.cast(buf, _vp), rank, world))
