"""CPU: static checks of the Julia ccall shim (deepqlearning.jl_amd/julia/DeepQLearningMI355X.jl), which cannot be executed here (no julia binary):
  * the isbits structs it passes by reference have the C layout of the header's structs (field order, sizes, natural alignment -- Julia lays
    out isbits structs exactly like C), checked against ctypes mirrors that test_abi_cpu.py ties to the header with a compiled C program;
  * every C entry point it ccalls exists in the header, with the same number of arguments;
  * METHOD COVERAGE: every generic function the shim imports from DeepQLearning to extend has at least one method typed on a HIP type
    (an imported-but-never-extended generic -- round 1's populate_replay_buffer! -- would fall through to the package's CPU method or
    die with a MethodError), and the solve / dqn_train! / initialize_replay_buffer routes exist."""
import ctypes
import os
import re

import __graft_entry__ as ge

pkg = ge.load_package()
abi = pkg._abi
SHIM = os.path.join(ge.ROOT, "deepqlearning.jl_amd", "julia", "DeepQLearningMI355X.jl")
HEADER = os.path.join(ge.ROOT, "include", "dqn_mi355x.h")

SCALARS = {"Int32": (4, 4), "UInt32": (4, 4), "Cint": (4, 4), "Float32": (4, 4), "Int64": (8, 8), "UInt64": (8, 8), "Float64": (8, 8), "Csize_t": (8, 8),
           "UInt8": (1, 1), "Bool": (1, 1)}


def type_layout(t):
    t = t.strip()
    if t in SCALARS:
        return SCALARS[t]
    if t.startswith("Ptr{"):
        return 8, 8
    m = re.fullmatch(r"NTuple\{(\d+),\s*(\w+)\}", t)
    if m:
        sz, al = SCALARS[m.group(2)]
        return int(m.group(1)) * sz, al
    raise KeyError(t)


def julia_struct_layout(src, name):
    m = re.search(r"(?:mutable\s+)?struct\s+" + name + r"\b(.*?)\nend", src, re.S)
    assert m, name
    body = re.sub(r"#.*", "", m.group(1))
    fields = re.findall(r"(\w+)::([\w{}, ]+?)(?=;|\n|$)", body)
    off, maxal, offsets = 0, 1, {}
    for fname, ftype in fields:
        sz, al = type_layout(ftype)
        off = (off + al - 1) // al * al
        offsets[fname] = off
        off += sz
        maxal = max(maxal, al)
    return (off + maxal - 1) // maxal * maxal, offsets


def test_shim_structs_have_the_c_layout():
    src = open(SHIM).read()
    for jl, ct in (("LayerDesc", pkg.LayerDesc), ("HParams", pkg.HParams), ("EnvSpec", abi.EnvSpec), ("RolloutCfg", abi.RolloutCfg), ("RolloutStats", abi.RolloutStats)):
        size, offsets = julia_struct_layout(src, jl)
        assert size == ctypes.sizeof(ct), (jl, size, ctypes.sizeof(ct))
        cfields = [f[0] for f in ct._fields_]
        assert list(offsets) == cfields, (jl, list(offsets), cfields)             # same names in the same order
        for f in cfields:
            assert offsets[f] == getattr(ct, f).offset, (jl, f)


def test_every_ccall_names_a_declared_entry_point_with_the_right_arity():
    src = open(SHIM).read()
    header = open(HEADER).read()
    protos = {m.group(1): m.group(2) for m in re.finditer(r"\b(?:int|const char\*)\s+(dqn_\w+)\s*\(([^;{]*?)\)\s*;", header, re.S)}
    def ccalls(text):
        """(name, [argument types]) of every ccall((:name, LIB), ret, (types...), args...) -- a small balanced-parenthesis scan"""
        out = []
        for m in re.finditer(r"ccall\(\(:(dqn_\w+),\s*LIB\),\s*\w+,\s*\(", text):
            i, depth, cur, items = m.end(), 1, "", []
            while depth:
                ch = text[i]; i += 1
                if ch in "({":
                    depth += 1
                elif ch in ")}":
                    depth -= 1
                    if depth == 0:
                        break
                if ch == "," and depth == 1:
                    items.append(cur.strip()); cur = ""
                else:
                    cur += ch
            if cur.strip():
                items.append(cur.strip())
            out.append((m.group(1), items))
        return out

    calls = ccalls(src)
    assert len(calls) >= 20 and ("dqn_last_error", []) in calls
    for name, argtypes in calls:
        assert name in protos, name
        want = 0 if protos[name].strip() in ("", "void") else protos[name].count(",") + 1
        assert len(argtypes) == want, (name, argtypes, protos[name])
    # argument categories: pointer vs 32-bit int vs 64-bit int vs float, position by position
    def c_cat(param):
        param = param.strip()
        if "*" in param:
            return "ptr"
        t = param.rsplit(" ", 1)[0].replace("const ", "").strip()
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "u64", "size_t": "u64", "float": "f32", "double": "f64"}[t]

    def jl_cat(t):
        if t.startswith(("Ptr{", "Ref{")) or t == "Cstring":
            return "ptr"
        return {"Cint": "i32", "Int32": "i32", "Int64": "i64", "UInt64": "u64", "Csize_t": "u64", "Float32": "f32", "Float64": "f64"}[t]

    for name, argtypes in calls:
        cparams = [x for x in protos[name].split(",") if x.strip() and x.strip() != "void"]
        assert [jl_cat(t) for t in argtypes] == [c_cat(x) for x in cparams], (name, argtypes, cparams)
    # and the data-path entry points are all bound
    bound = {n for n, _ in calls}
    for must in ("dqn_engine_create", "dqn_train_step", "dqn_replay_add", "dqn_replay_get_batch", "dqn_update_priorities", "dqn_forward", "dqn_sync_target",
                 "dqn_get_params", "dqn_set_params", "dqn_train_step_drqn", "dqn_episode_add", "dqn_reset_state", "dqn_envs_create", "dqn_rollout", "dqn_evaluate"):
        assert must in bound, must


def _imported_generics(src):
    m = re.search(r"import DeepQLearning:(.*?)\n(?=import|export|using|\n)", src, re.S)
    assert m
    names = [x.strip() for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    return [n for n in names if n[0].islower()]          # types (AbstractNNPolicy, DQExperience, DeepQLearningSolver) start upper-case


def test_every_imported_generic_has_a_hip_typed_method():
    src = re.sub(r"#.*", "", open(SHIM).read())
    generics = _imported_generics(src)
    assert {"batch_train!", "add_exp!", "update_priorities!", "get_batch", "populate_replay_buffer!", "is_full", "max_size", "getnetwork", "resetstate!",
            "actionmap", "initialize_replay_buffer", "dqn_train!"} <= set(generics)
    hip_types = ("HIPReplayBuffer", "HIPEpisodeReplayBuffer", "HIPNNPolicy", "Engine")
    def signatures(g):
        """argument lists of the method DEFINITIONS of g (`function g(...)` or `g(...) = ...` at the start of a line; may span lines)"""
        out = []
        for m in re.finditer(r"(?:^|\n)[ \t]*(?:function[ \t]+)?(?:\w+\.)?" + re.escape(g) + r"\(", src):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(src[i], 0); i += 1
            if m.group(0).lstrip().startswith("function") or re.match(r"\s*(?:where[^=\n]*)?=(?!=)", src[i:]):
                out.append(src[m.end():i - 1])
        return out

    for g in generics:
        sigs = signatures(g)
        assert any(any("::" + t in sig for t in hip_types) for sig in sigs), (g, sigs)
    # replay protocol: BOTH replay types serve the whole protocol the driver loop uses (src/solver.jl:89-95,187)
    for g in ("add_exp!", "populate_replay_buffer!", "is_full", "max_size", "batch_train!"):
        for t in ("HIPReplayBuffer", "HIPEpisodeReplayBuffer"):
            assert any("::" + t in sig for sig in signatures(g)), (g, t)
    # policy protocol (src/policy.jl): action / actionvalues / value on the HIP policy
    for g in ("POMDPs.action", "POMDPTools.actionvalues", "POMDPs.value"):
        assert re.search(re.escape(g) + r"\(p::HIPNNPolicy", src), g
    # routes: solve on the wrapper solver for MDP, POMDP and raw AbstractEnv (src/solver.jl:30-57); dqn_train! on the HIP policy (:59)
    for t in ("MDP", "POMDP", "AbstractEnv"):
        assert re.search(r"POMDPs\.solve\(s::MI355XSolver,\s*\w+::" + t + r"\)", src), t
    assert re.search(r"function dqn_train!\(solver::DeepQLearningSolver, env::AbstractEnv, policy::HIPNNPolicy, replay\)", src)
    assert "sync_target!(policy)" in src and "Flux.loadparams!(target_q" not in src     # the engine's target net is the one that is synced
    assert re.search(r"function initialize_replay_buffer\(solver::DeepQLearningSolver, env::AbstractEnv, action_indices, engine::Engine\)", src)


def test_julia_delimiters_balance():
    """the shim cannot be parsed here; at least its (), [], {} and function/struct/for/if/while/begin ... end blocks balance"""
    src = re.sub(r'"(?:[^"\\]|\\.)*"', '""', open(SHIM).read())
    src = re.sub(r"#.*", "", src)
    for a, b in ("()", "[]", "{}"):
        assert src.count(a) == src.count(b), (a, src.count(a), src.count(b))
    opens = len(re.findall(r"(?<![\w!.:])(?:function|struct|if|while|begin|module|do|let|try)\b(?!\s*=)", src))
    opens += len(re.findall(r"(?:^|;|\n)[ \t]*for\b", src))       # statement-level `for` only: comprehensions / generators carry no `end`
    ends = len(re.findall(r"(?<![\w!.:\[])end\b", src))
    assert opens == ends, (opens, ends)


def test_u8_replay_is_fed_bytes_and_policy_sees_training_scale():
    """with MI355XSolver(...; obs_u8 = true) the engine reads the void* of dqn_replay_add / dqn_episode_add as BYTES; the shim
    must hand over UInt8 rows (never Float32.(...)) and the policy forward must see byte / 255f0, the scale training uses."""
    src = open(SHIM).read()
    assert re.search(r"obs_u8::Bool", src) and "Engine(out[], solver.batch_size, length(actions(env)), dims, obs_u8)" in src
    for call in ("dqn_replay_add", "dqn_episode_add"):
        stmt = src[src.index(f"(:{call}, LIB)") - 200:src.index(f"(:{call}, LIB)") + 400]
        assert "obs_rows(r.e, expe.s" in stmt and "obs_rows(r.e, expe.sp" in stmt, call
        assert "Float32.(vec(expe" not in stmt, call
    fwd = src[src.index("(:dqn_forward, LIB)"):src.index("(:dqn_forward, LIB)") + 200]
    assert "policy_obs(p.e, o)" in fwd
    assert "./ 255f0" in src and 'eltype(x) == UInt8 || error(' in src
