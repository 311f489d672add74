# oracle/make_golden.jl -- TEST INFRASTRUCTURE; the ONLY route by which parity can be pinned by the reference itself.
#
#   julia --project=<env with DeepQLearning.jl v0.7 + Flux 0.14 + BSON> oracle/make_golden.jl [outdir = tests/golden]
#
# NEVER EXECUTED in the build image (no `julia` binary there or on the GPU box; SURVEY.md 8c): written against the reference's sources
# (src/solver.jl:36-57,191-287, src/prioritized_experience_replay.jl:19-135, src/episode_replay.jl:3-95, src/dueling.jl:36-58, test/test_env.jl)
# and reviewed, not run.  A maintainer with Julia runs it ONCE; it writes
#     tests/golden/julia_<case>.dqnvec      inputs and outputs of the REFERENCE's own batch_train! on a seeded replay
#     tests/golden/julia_qnetwork.bson      what save_model writes (src/solver.jl:290-300) for the dueling case, + its parameters in the .dqnvec
# and `pytest tests/test_julia_golden_cpu.py tests/test_julia_golden_gpu.py` then compares the fp64 oracle, the C twin and the HIP engine with
# those files (the tests skip while the files are absent).  The container is deliberately trivial so that no Julia package beyond the reference's
# own dependencies is needed:   for every array   "<name> <eltype> <ndims> <dim1> ... <dimN>\n"  + raw little-endian column-major bytes + "\n".
#
# What is dumped per feed-forward case (all BEFORE the step unless marked):
#   meta            Float64[B, gamma, lr, double_q, dueling, n, alpha, beta, eps, prioritized]
#   p_on, p_tg      flat Flux.params(active_q) / Flux.params(target_q) (target perturbed so that it differs from the online net)
#   rs, ra, rr, rsp, rdone, rprio     the whole replay (n transitions; ra is 1-based as stored, src/solver.jl:84-88), priorities as held by the buffer
#   idx             the indices sample(replay) drew (1-based), bs/ba/br/bsp/bdone/bw = get_batch(replay, idx) (ba as (action, column) pairs -> action only)
#   q_on_s, q_on_sp, q_tg_sp           forwards of the reference's networks on that batch (nA x B)
#   loss, grad_norm                    what batch_train! RETURNED
#   td, grads                          from a verbatim re-evaluation of the closure of src/solver.jl:219-225 on the same batch (batch_train! does not return them)
#   p_new, rprio_new                   parameters and priorities AFTER batch_train!
using DeepQLearning, POMDPs, POMDPTools, Flux, Random, StatsBase, BSON, CommonRLInterface
using DeepQLearning: flattenbatch, create_dueling_network, NNPolicy, PrioritizedReplayBuffer, EpisodeReplayBuffer, initialize_replay_buffer,
                     batch_train!, get_batch, huber_loss, globalnorm, getnetwork, DQExperience
const RL = CommonRLInterface
include(joinpath(pkgdir(DeepQLearning), "test", "test_env.jl"))        # TestMDP (the reference's own image MDP)

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "tests", "golden")
mkpath(outdir)

function put(io, name, x::AbstractArray{T}) where T
    println(io, name, " ", T, " ", ndims(x), " ", join(size(x), " "))
    write(io, Array(x)); println(io)
end
put(io, name, x::Number) = put(io, name, [x])
flat(ps) = reduce(vcat, [vec(Float32.(w)) for w in ps])

function feedforward_case(name, mdp, model; B, double_q, dueling, prioritized, lr=1f-3, n_pop=200, seed=1)
    Random.seed!(seed)
    solver = DeepQLearningSolver(qnetwork=model, learning_rate=lr, batch_size=B, double_q=double_q, dueling=dueling, prioritized_replay=prioritized,
                                 buffer_size=n_pop, train_start=n_pop, logdir=nothing,
                                 exploration_policy=EpsGreedyPolicy(mdp, 0.1))
    env = MDPCommonRLEnv{AbstractArray{Float32}}(mdp)                                   # src/solver.jl:30-33
    action_map = collect(RL.actions(env)); action_indices = Dict(a => i for (i, a) in enumerate(action_map))
    replay = initialize_replay_buffer(solver, env, action_indices)                     # :180-189 (populate with |r| priorities)
    active_q = dueling ? create_dueling_network(solver.qnetwork) : solver.qnetwork     # :48-52
    policy = NNPolicy(env, active_q, action_map, length(DeepQLearning.obs_dimensions(env)))
    target_q = deepcopy(active_q)
    for w in Flux.params(target_q); w .*= 0.9f0; end                                    # a target that differs from the online network
    optimizer = Adam(solver.learning_rate)                                             # :66
    n = replay._curr_size
    open(joinpath(outdir, "julia_" * name * ".dqnvec"), "w") do io
        put(io, "meta", Float64[B, discount(mdp), lr, double_q, dueling, n, replay.α, replay.β, replay.ϵ, prioritized])
        put(io, "p_on", flat(Flux.params(active_q))); put(io, "p_tg", flat(Flux.params(target_q)))
        put(io, "rs", cat([Float32.(replay._experience[i].s) for i in 1:n]...; dims=ndims(replay._experience[1].s) + 1))
        put(io, "rsp", cat([Float32.(replay._experience[i].sp) for i in 1:n]...; dims=ndims(replay._experience[1].sp) + 1))
        put(io, "ra", Int32[replay._experience[i].a for i in 1:n]); put(io, "rr", Float32[replay._experience[i].r for i in 1:n])
        put(io, "rdone", UInt8[replay._experience[i].done for i in 1:n]); put(io, "rprio", replay._priorities[1:n])
        # the draw batch_train! is about to make, on a copy of the buffer (same rng state -> same indices)
        rc = deepcopy(replay)
        idx = sample(rc.rng, 1:rc._curr_size, Weights(rc._priorities[1:rc._curr_size]), rc.batch_size, replace=false)      # ...replay.jl:85
        s, a, r, sp, done, indices, w = get_batch(rc, idx)
        put(io, "idx", Int64.(idx)); put(io, "bs", copy(s)); put(io, "ba", Int32[ci[1] for ci in a]); put(io, "br", copy(r)); put(io, "bsp", copy(sp))
        put(io, "bdone", copy(done)); put(io, "bw", copy(w))
        γ = convert(Float32, discount(mdp))
        qp = active_q(sp); qt = target_q(sp); qs = active_q(s)
        put(io, "q_on_s", qs); put(io, "q_on_sp", qp); put(io, "q_tg_sp", qt)
        # the closure of src/solver.jl:209-225, re-evaluated verbatim for td and the gradients
        if double_q
            best_a = [CartesianIndex(argmax(qp[:, i]), i) for i = 1:B]; q_sp_max = qt[best_a]
        else
            q_sp_max = dropdims(maximum(qt, dims=1), dims=1)
        end
        q_targets = r .+ (1f0 .- done) .* γ .* q_sp_max
        p = Flux.params(active_q); td_vals = nothing
        gs = Flux.gradient(p) do
            q_values = active_q(s); q_sa = q_values[a]; td_vals = q_sa .- q_targets
            sum(huber_loss, w .* td_vals) / B
        end
        put(io, "td", Float32.(td_vals)); put(io, "grads", flat([gs[x] === nothing ? zero(x) : gs[x] for x in p]))
        put(io, "grad_norm_closure", Float32(globalnorm(p, gs)))
        # THE REFERENCE CALL
        loss_val, grad_norm = batch_train!(solver, env, policy, optimizer, target_q, replay)
        put(io, "loss", Float32(loss_val)); put(io, "grad_norm", Float32(grad_norm))
        put(io, "p_new", flat(Flux.params(active_q))); put(io, "rprio_new", replay._priorities[1:n])
        if name == "mlp_dueling_ddqn_per"      # what save_model writes (src/solver.jl:290-300)
            bson(joinpath(outdir, "julia_qnetwork.bson"), qnetwork=[w for w in Flux.params(active_q)])
        end
    end
    println("wrote julia_", name, ".dqnvec  (n = ", n, " transitions)")
end

function drqn_case(name, mdp, model; B, T, double_q, lr=1f-3, n_ep=24, seed=2)
    Random.seed!(seed)
    solver = DeepQLearningSolver(qnetwork=model, learning_rate=lr, batch_size=B, double_q=double_q, dueling=false, prioritized_replay=false, recurrence=true,
                                 trace_length=T, buffer_size=n_ep, train_start=n_ep, logdir=nothing, exploration_policy=EpsGreedyPolicy(mdp, 0.1))
    env = MDPCommonRLEnv{AbstractArray{Float32}}(mdp)
    action_map = collect(RL.actions(env)); action_indices = Dict(a => i for (i, a) in enumerate(action_map))
    replay = initialize_replay_buffer(solver, env, action_indices)
    active_q = solver.qnetwork
    policy = NNPolicy(env, active_q, action_map, length(DeepQLearning.obs_dimensions(env)))
    target_q = deepcopy(active_q)
    for w in Flux.params(target_q); w .*= 0.9f0; end
    optimizer = Adam(solver.learning_rate)
    n = replay._curr_size; E = length(replay._experience[1][1].s)
    open(joinpath(outdir, "julia_" * name * ".dqnvec"), "w") do io
        put(io, "meta", Float64[B, discount(mdp), lr, double_q, 0, n, T])
        put(io, "p_on", flat(Flux.params(active_q))); put(io, "p_tg", flat(Flux.params(target_q)))      # LSTM: Wi, Wh, b, state0 h, state0 c (Flux 0.14 Recur(LSTMCell))
        lens = Int32[length(replay._experience[i]) for i in 1:n]; put(io, "ep_len", lens)
        es = zeros(Float32, E, T, n); esp = zeros(Float32, E, T, n); ea = ones(Int32, T, n); er = zeros(Float32, T, n); ed = zeros(UInt8, T, n)
        for i in 1:n, t in 1:min(lens[i], T)        # the sampler can only ever read the first trace_length transitions of an episode (episode_replay.jl:82-92)
            x = replay._experience[i][t]; es[:, t, i] = vec(x.s); esp[:, t, i] = vec(x.sp); ea[t, i] = x.a; er[t, i] = x.r; ed[t, i] = x.done
        end
        put(io, "es", es); put(io, "esp", esp); put(io, "ea", ea); put(io, "er", er); put(io, "edone", ed)
        # the draws sample(replay) is about to make (episode_replay.jl:75,81), replicated on a copy of the rng
        rng2 = copy(replay.rng)
        sidx = sample(rng2, 1:n, B, replace=false); starts = Int32[rand(rng2, 1:lens[i]) for i in sidx]
        put(io, "ep_idx", Int64.(sidx)); put(io, "ep_start", starts)
        rc = deepcopy(replay); s, a, r, sp, done, mask = StatsBase.sample(rc)
        put(io, "bmask", hcat(mask...)); put(io, "br", hcat(r...)); put(io, "ba", Int32[a[t][i][1] for i in 1:B, t in 1:T])
        put(io, "bs", cat([reshape(s[t], E, B) for t in 1:T]...; dims=3)); put(io, "bsp", cat([reshape(sp[t], E, B) for t in 1:T]...; dims=3)); put(io, "bdone", hcat(done...))
        loss_val, grad_norm = batch_train!(solver, env, policy, optimizer, target_q, replay)           # THE REFERENCE CALL (src/solver.jl:239-287)
        put(io, "loss", Float32(loss_val)); put(io, "grad_norm", Float32(grad_norm)); put(io, "p_new", flat(Flux.params(active_q)))
    end
    println("wrote julia_", name, ".dqnvec")
end

mdp5 = TestMDP((5, 5), 4, 6)
feedforward_case("mlp_tanh_plain", mdp5, Chain(x -> flattenbatch(x), Dense(100, 8, tanh), Dense(8, 4)); B=32, double_q=false, dueling=false, prioritized=false)   # test/runtests.jl:45-61
feedforward_case("mlp_dueling_ddqn_per", mdp5, Chain(x -> flattenbatch(x), Dense(100, 8, tanh), Dense(8, 4)); B=32, double_q=true, dueling=true, prioritized=true)  # :97-111
mdp10 = TestMDP((10, 10), 4, 6)
feedforward_case("conv_dueling_ddqn_per", mdp10, Chain(Conv((3, 3), 4 => 8, relu; stride=2), Conv((2, 2), 8 => 16, relu), x -> flattenbatch(x), Dense(144, 32, relu), Dense(32, 4));
                 B=16, double_q=true, dueling=true, prioritized=true)
drqn_case("drqn_lstm", TestMDP((5, 5), 1, 6), Chain(x -> flattenbatch(x), LSTM(25, 8), Dense(8, 4)); B=8, T=6, double_q=true)                                       # test/runtests.jl:114-128
println("done: ", abspath(outdir))
