"""
deepqlearning.jl_amd -- host side of the MI355X-native DQN training engine that drops in behind
DeepQLearning.jl's DeepQLearningSolver / solve / AbstractNNPolicy surface (src/DeepQLearning.jl:19-33).

The compute lives in libdqn_mi355x.so (hand-written HIP for gfx950, C ABI in include/dqn_mi355x.h).
This package is the Python mirror of the reference's operator interface for the hot path; it has NO CPU
fallback: importing works anywhere, but creating an engine without a HIP device raises DQNError.

The directory name contains a dot, so load it with `__graft_entry__.load_package()` (importlib by path).
"""
import ctypes as _C
import os as _os

from . import _abi
from ._abi import (ACT_IDENTITY, ACT_RELU, ACT_SIGMOID, ACT_TANH, NET_ONLINE, NET_TARGET, OBS_F32, OBS_U8, DQNError, HParams,  # noqa: F401
                   LayerDesc, LayerPlan, default_hparams)

_HERE = _os.path.dirname(_os.path.abspath(__file__))
LIB_PATH = _os.environ.get("DQN_MI355X_LIB") or _os.path.join(_HERE, "libdqn_mi355x.so")   # same variable the Julia shim reads
_lib = None
_fns = None


def lib():
    """dlopen libdqn_mi355x.so (built in-tree by __graft_entry__.build()); fails loudly if missing."""
    global _lib, _fns
    if _lib is None:
        if not _os.path.exists(LIB_PATH):
            raise DQNError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _lib = _C.CDLL(LIB_PATH)
        _fns = _abi.bind(_lib, "dqn_")
    return _lib


def fns():
    lib()
    return _fns


class Engine(_abi.Handle):
    """One engine per GPU: replay + sum-tree + networks + Adam state + a HIP stream."""

    def __init__(self, layers, hp, plan=None, device=0):
        super().__init__(fns(), layers, hp, plan=plan, device=device)


def default_plan(layers, hp):
    """Host-only: the summation-order plan the engine will use (dqn_plan_default)."""
    arr = (LayerDesc * len(layers))(*layers)
    plan = (LayerPlan * len(layers))()
    if fns()["plan_default"](arr, len(layers), _C.byref(hp), plan) != 0:
        raise DQNError(fns()["last_error"]().decode())
    return [p.astuple() for p in plan]


def comm_unique_id() -> bytes:
    buf = _C.create_string_buffer(128)
    if fns()["comm_unique_id"](_C.cast(buf, _C.c_void_p)) != 0:
        raise DQNError(fns()["last_error"]().decode())
    return buf.raw
