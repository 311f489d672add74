# DeepQLearningMI355X.jl -- thin `ccall` shim that plugs libdqn_mi355x.so (include/dqn_mi355x.h) into
# DeepQLearning.jl v0.7.1 WITHOUT touching the package: it adds one replay type and one policy type and lets Julia's
# multiple dispatch route the hot path to the GPU:
#
#   batch_train!(solver, env, policy, optimizer, target_q, replay::HIPReplayBuffer)   <- src/solver.jl:191-198 seam
#   add_exp!, sample, get_batch, update_priorities!, populate_replay_buffer!            <- src/prioritized_experience_replay.jl:61-134
#   AbstractNNPolicy interface: getnetwork, resetstate!, actionmap, action, actionvalues, value  <- src/policy.jl:1-76
#
#   solve(MI355XSolver(solver), mdp)  /  initialize_replay_buffer(solver, env, action_indices, engine)  /  dqn_train!(solver, env, ::HIPNNPolicy, replay)
#                                                                                        <- src/solver.jl:30-57, :59-178, :180-189
#
# NOT EXECUTED IN THIS BUILD: neither the build container nor the GPU box has a `julia` binary (SURVEY.md), so this file
# ships as reviewed source.  Every call below goes through exactly the C entry points that tests/ exercise from Python
# (ctypes) with the same buffers, so its behaviour is pinned by the same fixtures.
module DeepQLearningMI355X

using DeepQLearning, Flux, POMDPs, POMDPTools, Random, Printf, Statistics, BSON
import DeepQLearning: batch_train!, add_exp!, update_priorities!, get_batch, populate_replay_buffer!, is_full, max_size,
                      getnetwork, resetstate!, actionmap, initialize_replay_buffer, dqn_train!,
                      AbstractNNPolicy, DQExperience, DeepQLearningSolver
import CommonRLInterface: AbstractEnv, observe, actions, act!, reset!, terminated
import TensorBoardLogger: TBLogger, log_value
import StatsBase
export MI355XSolver, HIPReplayBuffer, HIPEpisodeReplayBuffer, HIPNNPolicy, train_steps!

const LIB = get(ENV, "DQN_MI355X_LIB", "libdqn_mi355x.so")

# ---- C structs (must match include/dqn_mi355x.h field for field)
struct LayerDesc
    kind::Int32; act::Int32; stream::Int32
    n_in::Int32; n_out::Int32
    cin::Int32; cout::Int32; kh::Int32; kw::Int32; sh::Int32; sw::Int32
end
mutable struct HParams
    batch_size::Int32; n_actions::Int32
    obs_c::Int32; obs_h::Int32; obs_w::Int32; obs_dtype::Int32
    learning_rate::Float32
    adam_beta1::Float64; adam_beta2::Float64; adam_eps::Float64
    adam_f64_scalars::Int32
    gamma::Float32
    double_q::Int32; dueling::Int32; prioritized_replay::Int32
    buffer_size::Int64
    prio_alpha::Float32; prio_beta::Float32; prio_eps::Float32
    seed::UInt64
    use_graph::Int32; use_mfma::Int32
    recurrence::Int32; trace_length::Int32
    sample_distinct::Int32
    reserved::NTuple{3,Int32}
    HParams() = new()
end

check(rc) = rc == 0 ? nothing : throw(unsafe_string(ccall((:dqn_last_error, LIB), Cstring, ())))  # reference errors are thrown Strings

const ACT = Dict(identity => 0, relu => 1, tanh => 2, sigmoid => 3)

function lower_layer(l, stream)::LayerDesc
    if l isa Dense
        return LayerDesc(0, ACT[l.σ], stream, size(l.weight, 2), size(l.weight, 1), 0, 0, 0, 0, 0, 0)
    elseif l isa Conv
        kw, kh, cin, cout = size(l.weight)
        all(==(0), l.pad) || throw("DeepQLearningError: the MI355X engine supports Conv with pad=0 only")
        return LayerDesc(1, ACT[l.σ], stream, 0, 0, cin, cout, kh, kw, l.stride[2], l.stride[1])
    elseif l isa Flux.Recur && l.cell isa Flux.LSTMCell     # Flux.params order Wi, Wh, b, state0 (h0, c0) == the ABI's LSTM block
        return LayerDesc(2, 0, stream, size(l.cell.Wi, 2), size(l.cell.Wh, 2), 0, 0, 0, 0, 0, 0)
    end
    throw("DeepQLearningError: unsupported layer $(typeof(l)) (Conv / Dense / LSTM / flattenbatch only)")
end
is_glue(l) = l === identity || l === flattenbatch || l isa Function
function lower(q)
    descs = LayerDesc[]
    if q isa DeepQLearning.DuelingNetwork
        for (chain, s) in ((q.base, 0), (q.val, 1), (q.adv, 2)), l in chain.layers
            is_glue(l) || push!(descs, lower_layer(l, Int32(s)))
        end
    else
        for l in q.layers; is_glue(l) || push!(descs, lower_layer(l, Int32(0))); end
    end
    descs
end
flatparams(q) = reduce(vcat, [vec(Float32.(w)) for w in Flux.params(q)])   # Flux.params order, Julia memory order: exactly the ABI layout

# ---- engine handle
mutable struct Engine
    h::Ptr{Cvoid}
    B::Int; nA::Int; obs_dims::Tuple
    obs_u8::Bool      # replay rows are BYTES (hp.obs_dtype = DQN_OBS_U8): training reads byte / 255f0 (test/test_env.jl:59)
end
# observation rows in the replay's storage type.  dqn_replay_add reads the void* as bytes for a u8 replay: a Float32 buffer there
# would be cut to its first E bytes, so only UInt8 arrays are accepted (the Python mirror's _obs_rows does the same).  The EPISODE replay
# always stores Float32 rows (dqn_episode_add copies E*4 bytes per observation): dqn_engine_create refuses recurrence with obs_u8, so
# obs_rows never returns bytes for a HIPEpisodeReplayBuffer.
function obs_rows(e::Engine, x, what)
    e.obs_u8 || return Float32.(vec(x))
    eltype(x) == UInt8 || error("$what: this replay stores UInt8 observations (obs_u8 = true); got $(eltype(x)) -- pass the raw bytes")
    return collect(vec(x))
end
# what the POLICY must see for such a replay: the scale training uses
policy_obs(e::Engine, o) = (e.obs_u8 && eltype(o) == UInt8) ? Float32.(vec(o)) ./ 255f0 : Float32.(vec(o))
function Engine(solver::DeepQLearningSolver, env::AbstractEnv, q; device=0, obs_u8=false)
    o = observe(env); dims = size(o)
    c, h, w = length(dims) == 3 ? (dims[3], dims[2], dims[1]) : (prod(dims), 1, 1)   # Julia (W,H,C) -> C [C][H][W]
    hp = HParams(); check(ccall((:dqn_hparams_default, LIB), Cint, (Ref{HParams},), hp))
    hp.batch_size = solver.batch_size; hp.n_actions = length(actions(env))
    hp.obs_c, hp.obs_h, hp.obs_w, hp.obs_dtype = c, h, w, obs_u8 ? 1 : 0
    hp.learning_rate = solver.learning_rate; hp.gamma = Float32(DeepQLearning.default_discount(env))
    hp.double_q = solver.double_q; hp.dueling = q isa DeepQLearning.DuelingNetwork; hp.prioritized_replay = solver.prioritized_replay
    hp.buffer_size = solver.buffer_size
    hp.recurrence = solver.recurrence; hp.trace_length = solver.trace_length
    descs = lower(q); out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:dqn_engine_create, LIB), Cint, (Ptr{LayerDesc}, Cint, Ref{HParams}, Ptr{Cvoid}, Cint, Ref{Ptr{Cvoid}}),
                descs, length(descs), hp, C_NULL, device, out))
    e = Engine(out[], solver.batch_size, length(actions(env)), dims, obs_u8)
    finalizer(x -> ccall((:dqn_engine_destroy, LIB), Cint, (Ptr{Cvoid},), x.h), e)
    p = flatparams(q)
    check(ccall((:dqn_set_params, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Csize_t), e.h, 0, p, length(p)))
    check(ccall((:dqn_sync_target, LIB), Cint, (Ptr{Cvoid},), e.h))
    e
end

# ---- replay protocol (src/prioritized_experience_replay.jl)
mutable struct HIPReplayBuffer
    e::Engine
    rng::AbstractRNG
end
max_size(r::HIPReplayBuffer) = (cap = Ref{Int64}(); ccall((:dqn_replay_size, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), r.e.h, C_NULL, cap); cap[])
cur_size(r::HIPReplayBuffer) = (cur = Ref{Int64}(); ccall((:dqn_replay_size, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}), r.e.h, cur, C_NULL); cur[])
is_full(r::HIPReplayBuffer) = cur_size(r) == max_size(r)

function add_exp!(r::HIPReplayBuffer, expe::DQExperience, td_err=abs(expe.r))      # :65-74
    s = obs_rows(r.e, expe.s, "add_exp!"); sp = obs_rows(r.e, expe.sp, "add_exp!")
    check(ccall((:dqn_replay_add, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}, Ref{Float32}, Ptr{Cvoid}, Ref{UInt8}, Ref{Float32}, Cint),
                r.e.h, s, Int32(expe.a - 1), Float32(expe.r), sp, UInt8(expe.done), Float32(td_err), 1))   # 1-based -> 0-based action
end
function update_priorities!(r::HIPReplayBuffer, indices::Vector{Int64}, td_errors)  # :76-80
    check(ccall((:dqn_update_priorities, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float32}, Cint), r.e.h, indices .- 1, Float32.(td_errors), length(indices)))
end
function get_batch(r::HIPReplayBuffer, idx::Vector{Int64})                           # :89-104
    B = r.e.B
    s = zeros(Float32, r.e.obs_dims..., B); sp = similar(s)
    a = zeros(Int32, B); rew = zeros(Float32, B); done = zeros(Float32, B); w = zeros(Float32, B)
    check(ccall((:dqn_replay_get_batch, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Float32}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                r.e.h, idx .- 1, s, a, rew, sp, done, w))
    return s, [CartesianIndex(Int(a[i]) + 1, i) for i in 1:B], rew, sp, done, idx, w
end
function StatsBase.sample(r::HIPReplayBuffer)                                        # :82-87 (sum-tree on the device)
    idx = zeros(Int64, r.e.B)
    check(ccall((:dqn_replay_sample, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), r.e.h, idx))
    get_batch(r, idx .+ 1)
end

# populate_replay_buffer!(replay, env, action_indices; max_pop, max_steps, policy) (:106-134): the package's method is typed on
# PrioritizedReplayBuffer and reads replay._curr_size, so the HIP replay gets its own: a random (or given) policy fills the ring with
# priority |r|, episodes are cut at max_steps
function populate_replay_buffer!(replay::HIPReplayBuffer, env::AbstractEnv, action_indices;
                                 max_pop::Int64 = max_size(replay), max_steps::Int64 = 100,
                                 policy::Policy = FunctionPolicy(o -> rand(actions(env))))
    reset!(env); o = observe(env); step = 0
    for _ in 1:(max_pop - cur_size(replay))
        a = action(policy, o)
        rew = Float32(act!(env, a)); op = observe(env); done = terminated(env)
        add_exp!(replay, DQExperience(o, action_indices[a], rew, op, done), abs(rew))    # "assume initial td error is r" (:122)
        o = op; step += 1
        if done || step >= max_steps
            reset!(env); o = observe(env); step = 0
        end
    end
    cur_size(replay) >= replay.e.B || throw(AssertionError("replay._curr_size >= replay.batch_size"))    # :133
    replay
end

# ---- the hot path: ONE ccall per batch_train!  (src/solver.jl:191-236)
function batch_train!(solver::DeepQLearningSolver, env::AbstractEnv, policy::AbstractNNPolicy, optimizer, target_q,
                      replay::HIPReplayBuffer; discount=DeepQLearning.default_discount(env))
    loss = Ref{Float32}(0); gn = Ref{Float32}(0)
    check(ccall((:dqn_train_step, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{Float32}, Ref{Float32}, Ptr{Float32}), replay.e.h, C_NULL, loss, gn, C_NULL))
    return loss[], gn[]
end

# The same step WITHOUT waiting for it (feed-forward engines): returns once the step is enqueued; the ticket names the (loss, grad_norm) record the
# step's last launch publishes into a mapped host ring.  The reference looks at batch_train!'s return values only every log_freq env steps
# (src/solver.jl:154-167), so the shim's dqn_train! loop below trains with this call and fetches the scalars of the newest step when it logs.
function batch_train_async!(replay::HIPReplayBuffer)
    ticket = Ref{UInt64}(0)
    check(ccall((:dqn_train_step_async, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{UInt64}), replay.e.h, C_NULL, ticket))
    return ticket[]
end
function step_scalars(e::Engine, ticket::UInt64; wait::Bool = true)
    loss = Ref{Float32}(0); gn = Ref{Float32}(0); pub = Ref{UInt64}(0)
    check(ccall((:dqn_step_scalars, LIB), Cint, (Ptr{Cvoid}, UInt64, Cint, Ref{Float32}, Ref{Float32}, Ref{UInt64}), e.h, ticket, wait ? 1 : 0, loss, gn, pub))
    return pub[] == ticket ? (loss[], gn[]) : nothing
end

# n sampled batch_train! steps back to back with nothing in between (no reference equivalent: offline / catch-up training on a filled replay).
# Bit-identical to n batch_train! calls; inside the call step i's last launch already gathers step i+1's batch (dqn_train_steps).
function train_steps!(e::Engine, n::Integer)
    loss = Ref{Float32}(0); gn = Ref{Float32}(0)
    check(ccall((:dqn_train_steps, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{Float32}, Ref{Float32}), e.h, n, loss, gn))
    return loss[], gn[]
end

# ---- DRQN: EpisodeReplayBuffer protocol (src/episode_replay.jl) and the recurrent batch_train! (src/solver.jl:239-287)
mutable struct HIPEpisodeReplayBuffer
    e::Engine
    rng::AbstractRNG
end
function add_exp!(r::HIPEpisodeReplayBuffer, expe::DQExperience)                      # :46-52 (the engine stores the episode when done)
    check(ccall((:dqn_episode_add, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int32}, Ref{Float32}, Ptr{Cvoid}, Ref{UInt8}, Cint),
                r.e.h, obs_rows(r.e, expe.s, "add_exp!"), Int32(expe.a - 1), Float32(expe.r), obs_rows(r.e, expe.sp, "add_exp!"), UInt8(expe.done), 1))
end
add_episode!(r::HIPEpisodeReplayBuffer) = check(ccall((:dqn_episode_commit, LIB), Cint, (Ptr{Cvoid},), r.e.h))   # :54-60, after generate_episode
ep_count(r::HIPEpisodeReplayBuffer) = (cur = Ref{Int64}(); ccall((:dqn_episode_count, LIB), Cint, (Ptr{Cvoid}, Ref{Int64}, Ptr{Int64}), r.e.h, cur, C_NULL); cur[])
max_size(r::HIPEpisodeReplayBuffer) = (cap = Ref{Int64}(); ccall((:dqn_episode_count, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ref{Int64}), r.e.h, C_NULL, cap); cap[])
is_full(r::HIPEpisodeReplayBuffer) = ep_count(r) == max_size(r)
# populate_replay_buffer!(r::EpisodeReplayBuffer, env, action_indices; max_pop, max_steps) (src/episode_replay.jl:97-130): random rollouts of
# fewer than max_steps steps, each stored as one episode whether or not it terminated
function populate_replay_buffer!(r::HIPEpisodeReplayBuffer, env::AbstractEnv, action_indices; max_pop::Int64 = max_size(r), max_steps::Int64 = 100)
    for _ in 1:(max_pop - ep_count(r))
        reset!(env); o = observe(env); done = false; step = 1
        while !done && step < max_steps
            a = rand(actions(env)); rew = Float32(act!(env, a)); op = observe(env); done = terminated(env)
            add_exp!(r, DQExperience(o, action_indices[a], rew, op, done))     # a terminal transition stores the episode (add_exp!, :46-52)
            o = op; step += 1
        end
        done || add_episode!(r)                                                 # cut at max_steps: add_episode!(r, ep), :100-101
    end
    ep_count(r) >= r.e.B || throw(AssertionError("r._curr_size >= r.batch_size"))
    r
end
function batch_train!(solver::DeepQLearningSolver, env::AbstractEnv, policy::AbstractNNPolicy, optimizer, target_q,
                      replay::HIPEpisodeReplayBuffer; discount=DeepQLearning.default_discount(env))
    loss = Ref{Float32}(0); gn = Ref{Float32}(0)
    check(ccall((:dqn_train_step_drqn, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}, Ptr{Int32}, Ref{Float32}, Ref{Float32}), replay.e.h, C_NULL, C_NULL, loss, gn))
    return loss[], gn[]
end

# ---- policy (src/policy.jl)
struct HIPNNPolicy{P,A} <: AbstractNNPolicy
    problem::P
    e::Engine
    qnetwork::Any            # the Flux model, refreshed on demand by getnetwork
    action_map::Vector{A}
    n_input_dims::Int64
end
function getnetwork(p::HIPNNPolicy)                    # Flux.params(active_q) <- engine (BSON save path keeps working, src/solver.jl:292)
    ps = Flux.params(p.qnetwork); n = sum(length, ps); flat = zeros(Float32, n)
    check(ccall((:dqn_get_params, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Csize_t), p.e.h, 0, flat, n))
    off = 0; for w in ps; copyto!(w, reshape(flat[off+1:off+length(w)], size(w))); off += length(w); end
    p.qnetwork
end
resetstate!(p::HIPNNPolicy) = check(ccall((:dqn_reset_state, LIB), Cint, (Ptr{Cvoid},), p.e.h))   # Flux.reset!: Recur state <- state0
# hiddenstates(m) / sethiddenstates!(m, hs) (src/helpers.jl:61-79) for the policy's Recur state, which lives in the engine: flat Float32 vector, per LSTM
# layer h then c.  The engine keeps it apart from the train step's sequences, so the save / restore the reference does around batch_train! (:137-139) is a no-op
# here; these exist for callers that checkpoint or transplant the state themselves.
function hiddenstates(p::HIPNNPolicy)
    n = sum(Int[2 * size(l.cell.Wh, 2) for l in filter(l -> l isa Flux.Recur, collect(p.qnetwork))])      # Wh: (4h, h)
    hc = zeros(Float32, n)
    check(ccall((:dqn_get_hidden, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), p.e.h, hc, n))
    hc
end
sethiddenstates!(p::HIPNNPolicy, hc::Vector{Float32}) = check(ccall((:dqn_set_hidden, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Csize_t), p.e.h, hc, length(hc)))
actionmap(p::HIPNNPolicy) = p.action_map
function _q(p::HIPNNPolicy, o)
    ndims(o) == p.n_input_dims || throw("NNPolicyError: was expecting an array with $(p.n_input_dims) dimensions, got $(ndims(o))")
    q = zeros(Float32, p.e.nA)
    check(ccall((:dqn_forward, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Cint, Ptr{Float32}), p.e.h, 0, policy_obs(p.e, o), 1, q))
    q
end
POMDPs.action(p::HIPNNPolicy, o::AbstractArray) = p.action_map[argmax(_q(p, o))]
POMDPTools.actionvalues(p::HIPNNPolicy, o::AbstractArray) = _q(p, o)
POMDPs.value(p::HIPNNPolicy, o::AbstractArray) = maximum(_q(p, o))
sync_target!(p::HIPNNPolicy) = check(ccall((:dqn_sync_target, LIB), Cint, (Ptr{Cvoid},), p.e.h))   # replaces Flux.loadparams! at src/solver.jl:142-145
function setnetwork!(p::HIPNNPolicy, weights)          # Flux.loadparams!(getnetwork(policy), weights) -> engine (restore_best_model, src/solver.jl:172-174,314-315)
    Flux.loadparams!(p.qnetwork, weights)
    flat = flatparams(p.qnetwork)
    check(ccall((:dqn_set_params, LIB), Cint, (Ptr{Cvoid}, Cint, Ptr{Float32}, Csize_t), p.e.h, 0, flat, length(flat)))
    p
end

# ---- solve / initialize_replay_buffer / dqn_train! routes (src/solver.jl:30-57, :180-189, :59-178): ZERO edits to the package.
# `solve(MI355XSolver(solver), mdp)` is the one-line change in user code; everything behind it dispatches on the HIP types.
struct MI355XSolver <: POMDPs.Solver
    solver::DeepQLearningSolver
    device::Int
    obs_u8::Bool
end
MI355XSolver(solver::DeepQLearningSolver; device = 0, obs_u8 = false) = MI355XSolver(solver, device, obs_u8)
POMDPs.solve(s::MI355XSolver, problem::MDP) = solve(s, MDPCommonRLEnv{AbstractArray{Float32}}(problem))       # :30-33
POMDPs.solve(s::MI355XSolver, problem::POMDP) = solve(s, POMDPCommonRLEnv{AbstractArray{Float32}}(problem))   # :35-38
function POMDPs.solve(s::MI355XSolver, env::AbstractEnv)                                                       # :40-57
    solver = s.solver
    action_map = collect(actions(env))
    action_indices = Dict(a => i for (i, a) in enumerate(action_map))
    DeepQLearning.isrecurrent(solver.qnetwork) && !solver.recurrence &&
        throw("DeepQLearningError: you passed in a recurrent model but recurrence is set to false")
    active_q = solver.dueling ? DeepQLearning.create_dueling_network(solver.qnetwork) : solver.qnetwork
    engine = Engine(solver, env, active_q; device = s.device, obs_u8 = s.obs_u8)
    replay = initialize_replay_buffer(solver, env, action_indices, engine)
    policy = HIPNNPolicy(env, engine, active_q, action_map, length(DeepQLearning.obs_dimensions(env)))
    dqn_train!(solver, env, policy, replay)
end
HIPNNPolicy(env::MDPCommonRLEnv, e::Engine, q, action_map::Vector, n::Int) = HIPNNPolicy(convert(MDP, env), e, q, action_map, n)       # src/policy.jl:24
HIPNNPolicy(env::POMDPCommonRLEnv, e::Engine, q, action_map::Vector, n::Int) = HIPNNPolicy(convert(POMDP, env), e, q, action_map, n)   # :25

function initialize_replay_buffer(solver::DeepQLearningSolver, env::AbstractEnv, action_indices, engine::Engine)   # :180-189
    replay = solver.recurrence ? HIPEpisodeReplayBuffer(engine, solver.rng) : HIPReplayBuffer(engine, solver.rng)
    populate_replay_buffer!(replay, env, action_indices, max_pop = solver.train_start)
    replay
end

# dqn_train! is dispatchable on the policy type (src/solver.jl:59).  The generic method would keep a Flux deepcopy as target network and
# refresh THAT every target_update_freq steps (:65, :142-145) -- the engine's own target net would never be synced -- so the HIP policy
# gets its own driver: same cadence (train / target / eval / save / log), same logged scalars, the optimizer, the target network and the
# hidden-state save/restore around batch_train! (:137-139: the engine keeps the policy's Recur state apart from the train step) inside the engine.
function dqn_train!(solver::DeepQLearningSolver, env::AbstractEnv, policy::HIPNNPolicy, replay)
    logger = nothing
    if solver.logdir !== nothing
        logger = TBLogger(solver.logdir); solver.logdir = logger.logdir
    end
    sync_target!(policy)                                   # target_q = deepcopy(active_q), :65
    resetstate!(policy); reset!(env); obs = observe(env)
    action_indices = Dict(a => i for (i, a) in enumerate(actionmap(policy)))
    ep_rewards = Float64[0.0]; ep_steps = Int64[]; step = 0
    best_eval = -Inf; scores_eval = -Inf; model_saved = false; eval_next = false; save_next = false
    loss_val = NaN32; grad_val = NaN32; ticket = UInt64(0)
    for t in 1:solver.max_steps
        act = action(solver.exploration_policy, policy, t, obs)
        rew = act!(env, act); op = observe(env); done = terminated(env)
        expe = DQExperience(obs, action_indices[act], Float32(rew), op, done)
        if solver.recurrence
            add_exp!(replay, expe)
        else
            add_exp!(replay, expe, solver.prioritized_replay ? abs(expe.r) : 0f0)     # :91-94
        end
        obs = op; step += 1; ep_rewards[end] += rew
        if done || step >= solver.max_episode_length
            if eval_next                                   # evaluation waits for the episode to end, :101-122
                scores_eval, steps_eval, info_eval = DeepQLearning.evaluation(solver.evaluation_policy, policy, env, solver.num_ep_eval,
                                                                              solver.max_episode_length, solver.verbose)
                eval_next = false
                if save_next                               # only right after an evaluation
                    model_saved, best_eval = DeepQLearning.save_model(solver, getnetwork(policy), scores_eval, best_eval, model_saved)   # qnetwork.bson, :290-300
                    save_next = false
                end
                if logger !== nothing
                    log_value(logger, "eval_reward", scores_eval, step = t); log_value(logger, "eval_steps", steps_eval, step = t)
                    for (k, v) in info_eval; log_value(logger, k, v, step = t); end
                end
            end
            reset!(env); obs = observe(env); resetstate!(policy)
            push!(ep_steps, step); push!(ep_rewards, 0.0); step = 0
        end
        if t % solver.train_freq == 0                                                            # ONE ccall, :136-140
            if replay isa HIPReplayBuffer
                ticket = batch_train_async!(replay)          # returns once enqueued: the env loop runs on while the GPU trains
            else
                loss_val, grad_val = batch_train!(solver, env, policy, nothing, nothing, replay)
            end
        end
        t % solver.target_update_freq == 0 && sync_target!(policy)                             # :142-145 inside the engine
        t % solver.eval_freq == 0 && (eval_next = true)
        t % solver.save_freq == 0 && (save_next = true)
        if t % solver.log_freq == 0 && logger !== nothing
            nt = POMDPTools.loginfo(solver.exploration_policy, t)
            for (k, v) in pairs(nt); log_value(logger, String(k), v, step = t); end
            avg100 = mean(ep_rewards[max(1, length(ep_rewards) - 101):end])
            ticket != 0 && ((loss_val, grad_val) = step_scalars(policy.e, ticket))      # (loss, grad_norm) of the newest train step: the values :154-167 print
            solver.verbose && @printf("%5d / %5d eps %0.3f |  avgR %1.3f | Loss %2.3e | Grad %2.3e | EvalR %1.3f \n",
                                      t, solver.max_steps, nt[1], avg100, loss_val, grad_val, scores_eval)
            log_value(logger, "avg_reward", avg100, step = t); log_value(logger, "loss", loss_val, step = t); log_value(logger, "grad_val", grad_val, step = t)
        end
    end
    if model_saved && solver.verbose                       # quirk kept: the best weights come back only if verbose, :170-176
        @printf("Restore model with eval reward %1.3f \n", best_eval)
        setnetwork!(policy, BSON.load(joinpath(solver.logdir, "qnetwork.bson"))[:qnetwork])
    else
        getnetwork(policy)                                 # leave the Flux model in step with the engine for the caller
    end
    policy
end

# ---- device-resident vectorised environments (dqn_env_spec / dqn_rollout_cfg / dqn_rollout_stats, include/dqn_mi355x.h)
# the env loop of dqn_train! (src/solver.jl:82-145) for n copies of a built-in MDP without observations crossing PCIe
struct EnvSpec
    kind::Int32; n_envs::Int32; max_episode_length::Int32; seed::UInt64
    o_stack::Int32; max_time::Int32; images::Ptr{UInt8}                       # TestMDP: UInt8[H*W, 3] (bad, normal, good)
    size_x::Int32; size_y::Int32; tprob::Float32; n_reward_cells::Int32
    reward_xy::NTuple{16,Int32}; reward_val::NTuple{8,Float32}                # SimpleGridWorld reward cells (x1,y1,x2,y2,...)
end
struct RolloutCfg
    train_freq::Int32; target_update_freq::Int32; eps_start::Float32; eps_stop::Float32; eps_steps::Float32; cadence_env_steps::Int32; t0::Int64
end
mutable struct RolloutStats
    episodes::Int64; reward_sum::Float64; train_steps::Int64; last_loss::Float32; last_grad_norm::Float32
    RolloutStats() = new(0, 0.0, 0, 0f0, 0f0)
end
function envs_create_gridworld!(e::Engine, mdp; n_envs, max_episode_length = 100, seed = 0)      # mdp::POMDPModels.SimpleGridWorld
    cells = collect(mdp.rewards); xy = zeros(Int32, 16); rv = zeros(Float32, 8)
    for (k, (pos, r)) in enumerate(cells); xy[2k-1] = pos[1]; xy[2k] = pos[2]; rv[k] = r; end
    spec = EnvSpec(1, n_envs, max_episode_length, seed, 0, 0, C_NULL, mdp.size[1], mdp.size[2], mdp.tprob, length(cells), Tuple(xy), Tuple(rv))
    check(ccall((:dqn_envs_create, LIB), Cint, (Ptr{Cvoid}, Ref{EnvSpec}), e.h, spec))
end
function envs_create_testmdp!(e::Engine, images::Matrix{UInt8}, o_stack, max_time; n_envs, max_episode_length = 100, seed = 0)
    GC.@preserve images begin
        spec = EnvSpec(0, n_envs, max_episode_length, seed, o_stack, max_time, pointer(images), 0, 0, 0f0, 0, ntuple(_ -> Int32(0), 16), ntuple(_ -> 0f0, 8))
        check(ccall((:dqn_envs_create, LIB), Cint, (Ptr{Cvoid}, Ref{EnvSpec}), e.h, spec))
    end
end
function evaluate(e::Engine, n_eval, max_episode_length; seed = 0)                   # basic_evaluation (src/evaluation_policy.jl:17-42) on the device
    r = Ref{Float64}(0); st = Ref{Float64}(0)
    check(ccall((:dqn_evaluate, LIB), Cint, (Ptr{Cvoid}, Cint, Cint, UInt64, Ref{Float64}, Ref{Float64}), e.h, n_eval, max_episode_length, seed, r, st))
    r[], st[]
end
# env_step_cadence = true: train_freq / target_update_freq count ENV steps as dqn_train! does (src/solver.jl:136-145) -- n / train_freq train steps per vector step
function rollout!(e::Engine, n_steps; t0 = 1, train_freq = 4, target_update_freq = 500, eps = (1f0, 0.01f0, 5000f0), env_step_cadence = false)
    st = RolloutStats()
    check(ccall((:dqn_rollout, LIB), Cint, (Ptr{Cvoid}, Cint, Ref{RolloutCfg}, Ref{RolloutStats}), e.h, n_steps,
                RolloutCfg(train_freq, target_update_freq, eps[1], eps[2], eps[3], env_step_cadence ? 1 : 0, t0), st))
    st
end

end # module
