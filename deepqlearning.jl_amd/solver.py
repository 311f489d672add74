"""
Python mirror of the reference's operator interface for the hot path -- same names, argument meaning and error
behaviour as src/DeepQLearning.jl:19-33 exports -- driving the HIP engine through the C ABI:

    DeepQLearningSolver          src/solver.jl:1-28   (all 26 keyword fields, same defaults)
    solve(solver, env)           src/solver.jl:40-57
    dqn_train(...)               src/solver.jl:59-178 (env loop and cadence; host control, as in the reference)
    batch_train(...)             src/solver.jl:191-236 -> ONE dqn_train_step call
    HIPReplayBuffer              src/prioritized_experience_replay.jl:19-134 protocol: add_exp, sample, get_batch,
                                 update_priorities, populate_replay_buffer, is_full, max_size
    NNPolicy(AbstractNNPolicy)   src/policy.jl:1-76: getnetwork, resetstate, actionmap, action, actionvalues, value
    basic_evaluation             src/evaluation_policy.jl:17-42
    EpsGreedyPolicy / LinearDecaySchedule   POMDPTools (third-party; the exploration policy the reference's tests use)

Environments follow envs.py (reset / observe / act / terminated on n lock-stepped copies; the reference steps n=1).
Errors are DQNError carrying the reference's strings.  There is no CPU fallback.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Any, Callable, Optional

import numpy as np

from . import _abi, bson, nn
from ._abi import DQNError


# ------------------------------------------------------------------ exploration (POMDPTools)
@dataclass
class LinearDecaySchedule:
    start: float
    stop: float
    steps: float

    def __call__(self, t):
        rate = (self.start - self.stop) / self.steps
        return max(self.stop, self.start - t * rate)


class EpsGreedyPolicy:
    def __init__(self, env, eps, rng=None):
        self.n_actions = env.n_actions
        self.schedule = eps
        self.eps = eps if callable(eps) else (lambda t, e=eps: e)
        self.rng = rng if rng is not None else np.random.default_rng(1)

    def action(self, policy, t, obs):
        """action(exploration_policy, policy, t, obs) (src/solver.jl:83) for a batch of n observations."""
        n = obs.shape[0]
        greedy = policy.action(obs)
        rnd = self.rng.random(n) < self.eps(t)
        return np.where(rnd, self.rng.integers(0, self.n_actions, n), greedy)

    def loginfo(self, t):
        return {"eps": self.eps(t)}


# ------------------------------------------------------------------ replay protocol
class HIPReplayBuffer:
    """PrioritizedReplayBuffer whose storage, sum-tree and batch arena live in HBM (engine-owned)."""

    def __init__(self, engine):
        self.e = engine

    @property
    def batch_size(self):
        return self.e.B

    def max_size(self):
        return self.e.replay_size()[1]

    def is_full(self):
        cur, cap = self.e.replay_size()
        return cur == cap

    def add_exp(self, s, a, r, sp, done, td_err=None):
        """add_exp!(r, DQExperience(s,a,r,sp,done), td_err=abs(r)) (...replay.jl:65-74); vectorised over n."""
        self.e.replay_add(s, a, r, sp, done, td_err)

    def update_priorities(self, indices, td_errors):
        self.e.update_priorities(indices, td_errors)

    def get_batch(self, sample_indices):
        return self.e.get_batch(sample_indices) + (np.asarray(sample_indices),)

    def sample(self):
        """StatsBase.sample(r) (...replay.jl:82-87): sum-tree indices then get_batch."""
        idx = self.e.replay_sample()
        s, a, r, sp, done, w = self.e.get_batch(idx)
        return s, a, r, sp, done, idx, w


class HIPEpisodeReplayBuffer:
    """EpisodeReplayBuffer (src/episode_replay.jl:3-95) whose episode storage lives in HBM (engine-owned)."""

    def __init__(self, engine):
        self.e = engine

    @property
    def batch_size(self):
        return self.e.B

    def max_size(self):
        return self.e.episode_count()[1]

    def is_full(self):
        cur, cap = self.e.episode_count()
        return cur == cap

    def add_exp(self, s, a, r, sp, done, td_err=None):
        """add_exp!(r::EpisodeReplayBuffer, exp) (:46-52): the episode is stored when done (truncated ones keep growing)."""
        self.e.episode_add(s, a, r, sp, done)

    def add_episode(self):
        self.e.episode_commit()


def generate_episode(replay, env, max_steps=100, rng=None):
    """src/episode_replay.jl:109-130: one random rollout (n=1 stream), stored whether or not it terminated."""
    rng = rng if rng is not None else np.random.default_rng(0)
    env.reset()
    o = env.observe()
    done, step = False, 1
    while not done and step < max_steps:
        a = rng.integers(0, env.n_actions, env.n)
        rew = env.act(a)
        op = env.observe()
        done = bool(env.terminated()[0])
        replay.e.episode_add(o[:1], a[:1].astype(np.int32), rew[:1], op[:1], np.array([done], np.uint8))   # done => the engine stores the episode (add_exp!)
        o = op
        step += 1
    if not done:
        replay.add_episode()       # add_episode!(r, ep) for a rollout cut at max_steps (:100-101)


def populate_episode_replay(replay, env, max_pop=None, max_steps=100, rng=None):
    """src/episode_replay.jl:97-107."""
    max_pop = replay.max_size() if max_pop is None else max_pop
    for _ in range(max_pop - replay.e.episode_count()[0]):
        generate_episode(replay, env, max_steps=max_steps, rng=rng)
    if replay.e.episode_count()[0] < replay.batch_size:
        raise DQNError("AssertionError: r._curr_size >= r.batch_size")


def populate_replay_buffer(replay: HIPReplayBuffer, env, max_pop=None, max_steps=100, rng=None):
    """...replay.jl:106-134: random policy, priority = |r|, episodes cut at max_steps."""
    rng = rng if rng is not None else np.random.default_rng(0)
    max_pop = replay.max_size() if max_pop is None else max_pop
    env.reset()
    o = env.observe()
    step = np.zeros(env.n, np.int64)
    todo = max_pop - replay.e.replay_size()[0]
    while todo > 0:
        a = rng.integers(0, env.n_actions, env.n)
        rew = env.act(a)
        op = env.observe()
        done = env.terminated()
        k = min(todo, env.n)
        replay.add_exp(o[:k], a[:k].astype(np.int32), rew[:k], op[:k], done[:k].astype(np.uint8), np.abs(rew[:k]))
        todo -= k
        step += 1
        rs = done | (step >= max_steps)
        env.reset(rs)
        step[rs] = 0
        o = env.observe()
    if replay.e.replay_size()[0] < replay.batch_size:
        raise DQNError("AssertionError: replay._curr_size >= replay.batch_size")


# ------------------------------------------------------------------ policy
class AbstractNNPolicy:
    pass


class NNPolicy(AbstractNNPolicy):
    """src/policy.jl:17-76.  The Q-network lives in the engine; getnetwork returns the flat Flux.params vector."""

    def __init__(self, env, engine, action_map, n_input_dims, qnetwork=None):
        self.problem, self.engine = env, engine
        self.action_map = list(action_map)
        self.n_input_dims = n_input_dims
        self.qnetwork = qnetwork           # the nn.Chain / nn.DuelingNetwork description (array shapes for qnetwork.bson)

    def getnetwork(self):
        return self.engine.get_params(_abi.NET_ONLINE)

    def resetstate(self):
        self.engine.reset_state()      # Flux.reset!(qnetwork): Recur state <- state0 (no-op for feed-forward networks)

    def actionmap(self):
        return self.action_map

    def _check(self, o):
        o = np.asarray(o)
        if self.engine.obs_np is np.uint8 and o.dtype == np.uint8:
            # u8 replay storage: training consumes byte / 255f0 (test/test_env.jl:59), so the policy must see the same scale
            o = o.astype(np.float32) / np.float32(255.0)
        else:
            o = np.asarray(o, np.float32)
        if o.ndim == self.n_input_dims:
            return o[None], True
        if o.ndim == self.n_input_dims + 1:
            return o, False
        raise DQNError(f"NNPolicyError: was expecting an array with {self.n_input_dims} dimensions, got {o.ndim}")

    def action(self, o):
        ob, single = self._check(o)
        a = self.engine.greedy_action(ob)
        return self.action_map[int(a[0])] if single else a

    def actionvalues(self, o):
        ob, single = self._check(o)
        q = self.engine.forward(ob)
        return q[0] if single else q

    def value(self, o):
        ob, single = self._check(o)
        q = self.engine.forward(ob).max(axis=1)
        return float(q[0]) if single else q


def basic_evaluation(policy, env, n_eval, max_episode_length, verbose=False):
    """src/evaluation_policy.jl:17-42 on the vectorised env (episodes are run env.n at a time)."""
    tot_r, tot_steps, done_eps = 0.0, 0.0, 0
    while done_eps < n_eval:
        env.reset()
        policy.resetstate()                    # resetstate!(policy) before every evaluation episode (src/evaluation_policy.jl:26)
        obs = env.observe()
        alive = np.ones(env.n, bool)
        r_ep = np.zeros(env.n)
        steps = np.zeros(env.n)
        step = 0
        while alive.any() and step <= max_episode_length:
            act = policy.action(obs)
            rew = env.act(act)
            obs = env.observe()
            r_ep += np.where(alive, rew, 0.0)
            steps += alive
            alive &= ~env.terminated()
            step += 1
        k = min(env.n, n_eval - done_eps)
        tot_r += r_ep[:k].sum()
        tot_steps += steps[:k].sum()
        done_eps += k
    if verbose:
        print(f"Evaluation ... Avg Reward {tot_r / n_eval:2.2f} | Avg Step {tot_steps / n_eval:2.2f}")
    return tot_r / n_eval, tot_steps / n_eval, {}


# ------------------------------------------------------------------ solver
@dataclass
class DeepQLearningSolver:
    qnetwork: Any = None
    learning_rate: float = 1e-4
    max_steps: int = 1000
    batch_size: int = 32
    train_freq: int = 4
    eval_freq: int = 500
    target_update_freq: int = 500
    num_ep_eval: int = 100
    double_q: bool = True
    dueling: bool = True
    recurrence: bool = False
    evaluation_policy: Callable = basic_evaluation
    exploration_policy: Any = None
    trace_length: int = 40
    prioritized_replay: bool = True
    prioritized_replay_alpha: float = 0.6      # dead config in the reference (src/solver.jl:185 never forwards it)
    prioritized_replay_epsilon: float = 1e-6   # dead config (the buffer default 1e-3 is what runs)
    prioritized_replay_beta: float = 0.4       # dead config
    buffer_size: int = 1000
    max_episode_length: int = 100
    train_start: int = 200
    rng: Any = field(default_factory=lambda: np.random.default_rng(0))
    logdir: Optional[str] = "log/"
    save_freq: int = 3000
    log_freq: int = 100
    verbose: bool = True
    # engine knobs (no reference equivalent)
    device: int = 0
    obs_dtype: int = _abi.OBS_F32
    seed: int = 0
    device_envs: bool = False                  # step env.n copies of the MDP on the GPU (dqn_envs_create / dqn_rollout)


def initialize_replay_buffer(solver, env, engine):
    """src/solver.jl:180-189: buffer defaults alpha=0.6 beta=0.4 eps=1e-3 are used whatever the solver says."""
    if solver.recurrence:
        replay = HIPEpisodeReplayBuffer(engine)                      # EpisodeReplayBuffer(env, buffer_size, batch_size, trace_length), :183
        populate_episode_replay(replay, env, max_pop=solver.train_start, rng=solver.rng)
        return replay
    replay = HIPReplayBuffer(engine)
    populate_replay_buffer(replay, env, max_pop=solver.train_start, rng=solver.rng)
    return replay


def make_engine(pkg_engine_cls, solver, env, net, discount):
    layers, dueling = nn.lower(net)
    shp = env.obs_shape
    c, h, w = shp if len(shp) == 3 else (int(np.prod(shp)), 1, 1)
    hp = _abi.default_hparams(batch_size=solver.batch_size, n_actions=env.n_actions, obs_c=c, obs_h=h, obs_w=w,
                              obs_dtype=solver.obs_dtype, learning_rate=solver.learning_rate, gamma=float(discount),
                              double_q=int(solver.double_q), dueling=int(dueling), prioritized_replay=int(solver.prioritized_replay),
                              buffer_size=solver.buffer_size, seed=solver.seed, recurrence=int(solver.recurrence), trace_length=solver.trace_length)
    return pkg_engine_cls(layers, hp, device=solver.device)


def solve(solver: DeepQLearningSolver, env, engine_cls=None, init_seed=1):
    """POMDPs.solve(solver, env) (src/solver.jl:40-57)."""
    if engine_cls is None:
        from . import Engine as engine_cls
    if nn.isrecurrent(solver.qnetwork) and not solver.recurrence:
        raise DQNError("DeepQLearningError: you passed in a recurrent model but recurrence is set to false")   # src/solver.jl:45-47
    action_map = list(range(env.n_actions))
    net = nn.create_dueling_network(solver.qnetwork) if solver.dueling else solver.qnetwork
    engine = make_engine(engine_cls, solver, env, net, getattr(env, "discount", 1.0))
    params = nn.glorot_params(net, seed=init_seed)
    engine.set_params(params, _abi.NET_ONLINE)
    replay = initialize_replay_buffer(solver, env, engine)
    policy = NNPolicy(env, engine, action_map, len(env.obs_shape), qnetwork=net)
    return dqn_train(solver, env, policy, replay)


def batch_train(solver, env, policy, optimizer, target_q, replay, discount=None):
    """batch_train!(solver, env, policy, optimizer, target_q, replay) -> (loss_val, grad_norm) (src/solver.jl:191-236).
    optimizer / target_q live inside the engine; the arguments are kept for signature parity."""
    if isinstance(replay, HIPEpisodeReplayBuffer):                    # dispatch on the replay type, src/solver.jl:239-246
        return policy.engine.train_step_drqn()
    loss, gn = policy.engine.train_step(want_td=False)
    return loss, gn


def save_model(solver, policy, scores_eval, saved_mean_reward, model_saved):
    """src/solver.jl:290-300: bson(joinpath(logdir, "qnetwork.bson"), qnetwork=[w for w in Flux.params(active_q)]) when the evaluation
    score did not get worse.  The file is written in BSON.jl's array lowering (bson.py; unverified against BSON.jl -- no Julia here)."""
    if scores_eval >= saved_mean_reward:
        os.makedirs(solver.logdir, exist_ok=True)
        if policy.qnetwork is None:
            raise DQNError("save_model: the policy carries no network description (NNPolicy(..., qnetwork=net)); qnetwork.bson needs the array shapes")
        bson.save_qnetwork(os.path.join(solver.logdir, "qnetwork.bson"), policy.getnetwork(), bson.julia_param_shapes(policy.qnetwork))
        if solver.verbose:
            print(f"Saving new model with eval reward {scores_eval:1.3f}")
        return True, scores_eval
    return model_saved, saved_mean_reward


def restore_best_model(solver, policy):
    """src/solver.jl:302-318: weights = BSON.load(logdir * "qnetwork.bson")[:qnetwork]; Flux.loadparams!(getnetwork(policy), weights)."""
    w, sizes = bson.load_qnetwork(os.path.join(solver.logdir, "qnetwork.bson"))
    if policy.qnetwork is not None:
        want = [s for s, _ in bson.julia_param_shapes(policy.qnetwork)]
        if sizes != want:
            raise DQNError(f"restore_best_model: qnetwork.bson holds arrays of sizes {sizes}, the network expects {want}")   # loadparams! dimension check
    policy.engine.set_params(w, _abi.NET_ONLINE)
    return policy


def dqn_train_device(solver, env, policy, replay):
    """dqn_train! (src/solver.jl:59-178) with the env loop on the device: `env` is only the SPEC (images, grid, rewards) of the
    env.n copies that dqn_envs_create builds in HBM; exploration uses the engine's Philox eps-greedy with the solver's
    LinearDecaySchedule.  Differences from the host loop: evaluation runs right at t % eval_freq == 0 (not at the next
    episode end); the default basic_evaluation runs on the device too (dqn_evaluate), a user-supplied one on the host copy."""
    e = policy.engine
    e.sync_target()
    sch = getattr(solver.exploration_policy, "schedule", None)
    if isinstance(sch, LinearDecaySchedule):
        eps = (sch.start, sch.stop, sch.steps)
    else:
        v = float(solver.exploration_policy.eps(1))
        eps = (v, v, 1.0)
    e.envs_create(env, n_envs=env.n, max_episode_length=solver.max_episode_length, seed=solver.seed)
    saved_mean_reward, scores_eval, model_saved = -np.inf, -np.inf, False
    marks = sorted({solver.eval_freq, solver.log_freq, solver.save_freq})
    t, episodes, reward_sum = 1, 0, 0.0
    save_next = False                                  # set at t % save_freq == 0, consumed at the next evaluation (src/solver.jl:109-113,150-152)
    while t <= solver.max_steps:
        nxt = min(min((t + m - 1) // m * m for m in marks), solver.max_steps)      # run up to the next eval/log/save boundary
        st = e.rollout(nxt - t + 1, t0=t, train_freq=solver.train_freq, target_update_freq=solver.target_update_freq, eps=eps)
        t = nxt + 1
        d_eps, d_rew = st["episodes"] - episodes, st["reward_sum"] - reward_sum
        episodes, reward_sum = st["episodes"], st["reward_sum"]
        if nxt % solver.save_freq == 0:
            save_next = True
        if nxt % solver.eval_freq == 0:
            if solver.evaluation_policy is basic_evaluation:        # the default rollout evaluation also runs on the device
                scores_eval, steps_eval = e.evaluate(min(solver.num_ep_eval, 1024), solver.max_episode_length, seed=solver.seed + nxt)
                if solver.verbose:
                    print(f"Evaluation ... Avg Reward {scores_eval:2.2f} | Avg Step {steps_eval:2.2f}")
            else:
                scores_eval, _, _ = solver.evaluation_policy(policy, env, solver.num_ep_eval, solver.max_episode_length, solver.verbose)
            if save_next and solver.logdir is not None:
                model_saved, saved_mean_reward = save_model(solver, policy, scores_eval, saved_mean_reward, model_saved)
                save_next = False
        if nxt % solver.log_freq == 0 and solver.verbose:
            avg = d_rew / d_eps if d_eps else float("nan")
            print(f"{nxt:5d} / {solver.max_steps:5d} eps {max(eps[1], eps[0] - nxt * (eps[0] - eps[1]) / eps[2]):0.3f} |  avgR {avg:1.3f} | "
                  f"Loss {st['loss']:2.3e} | Grad {st['grad_norm']:2.3e} | EvalR {scores_eval:1.3f}")
    if model_saved and solver.verbose:
        restore_best_model(solver, policy)
    return policy


def dqn_train(solver, env, policy, replay):
    """src/solver.jl:59-178.  `env` holds env.n lock-stepped copies (the reference: 1); t counts vector steps."""
    if solver.device_envs:
        if solver.recurrence:
            raise DQNError("device_envs drives the feed-forward path (recurrence = false)")
        return dqn_train_device(solver, env, policy, replay)
    e = policy.engine
    if solver.recurrence and env.n != 1:
        # one EpisodeReplayBuffer episode and one Recur state are open at a time, as in the reference's single-env loop: n > 1 streams
        # would interleave transitions of different environments in one episode
        raise DQNError("recurrence = true drives ONE environment stream on the host loop (env.n == 1), like the reference's dqn_train!")
    e.sync_target()                                   # target_q = deepcopy(active_q), :65
    policy.resetstate()
    env.reset()
    obs = env.observe()
    step = np.zeros(env.n, np.int64)
    episode_rewards = [0.0]
    cur = np.zeros(env.n)
    saved_mean_reward, scores_eval, model_saved = -np.inf, -np.inf, False
    eval_next = save_next = False
    loss_val = grad_val = float("nan")
    for t in range(1, solver.max_steps + 1):
        act = solver.exploration_policy.action(policy, t, obs)
        rew = env.act(act)
        op = env.observe()
        done = env.terminated()
        if solver.recurrence:
            replay.add_exp(obs, act.astype(np.int32), rew, op, done.astype(np.uint8))     # :89-90
        else:
            td0 = np.abs(rew) if solver.prioritized_replay else np.zeros_like(rew)      # :91-94
            replay.add_exp(obs, act.astype(np.int32), rew, op, done.astype(np.uint8), td0)
        obs = op
        step += 1
        cur += rew
        ended = done | (step >= solver.max_episode_length)
        if ended.any():
            if eval_next:                                                                # :101-122
                scores_eval, steps_eval, _ = solver.evaluation_policy(policy, env, solver.num_ep_eval, solver.max_episode_length, solver.verbose)
                eval_next = False
                if save_next and solver.logdir is not None:
                    model_saved, saved_mean_reward = save_model(solver, policy, scores_eval, saved_mean_reward, model_saved)
                    save_next = False
                env.reset()
                ended[:] = True
            episode_rewards.extend(cur[ended].tolist())
            cur[ended] = 0
            env.reset(ended)
            step[ended] = 0
            obs = env.observe()
            policy.resetstate()
        if t % solver.train_freq == 0:
            loss_val, grad_val = batch_train(solver, env, policy, None, None, replay)    # :136-140
        if t % solver.target_update_freq == 0:
            e.sync_target()                                                              # :142-145
        if t % solver.eval_freq == 0:
            eval_next = True
        if t % solver.save_freq == 0:
            save_next = True
        if t % solver.log_freq == 0 and solver.verbose:
            avg = float(np.mean(episode_rewards[-101:]))
            info = solver.exploration_policy.loginfo(t)
            print(f"{t:5d} / {solver.max_steps:5d} eps {list(info.values())[0]:0.3f} |  avgR {avg:1.3f} | Loss {loss_val:2.3e} | Grad {grad_val:2.3e} | EvalR {scores_eval:1.3f}")
    if model_saved and solver.verbose:                                                   # quirk kept: only if verbose, :170-176
        print(f"Restore model with eval reward {saved_mean_reward:1.3f}")
        restore_best_model(solver, policy)
    return policy
