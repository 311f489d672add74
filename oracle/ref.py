"""
oracle/ref.py -- TEST INFRASTRUCTURE.  ctypes loader for the CPU twin
(oracle/_ref/libdqn_ref.so, built by oracle/Makefile from oracle/dqn_ref.c) and
glue between the NumPy oracle's network description and the C-ABI structs.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
# struct layouts come from the product's ABI module (ONE definition of the C structs); importing the package does
# not load the HIP library.
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as _ge  # noqa: E402

abi = _ge.load_package()._abi

LIB_PATH = os.path.join(HERE, "_ref", "libdqn_ref.so")
_fns = None
# how the twin's C interface differs from the product's (this knowledge is test infrastructure and lives HERE, not in the product
# package): symbols carry the prefix "ref_", three are spelled shorter, and ref_create takes no device argument
PREFIX = "ref_"
ALIASES = {"engine_create": "create", "engine_destroy": "destroy", "engine_get_plan": "get_plan"}
ARGTYPES = {"engine_create": [C.POINTER(abi.LayerDesc), C.c_int, C.POINTER(abi.HParams), C.POINTER(abi.LayerPlan), C.POINTER(C.c_void_p)]}


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "dqn_ref.c")):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB_PATH


def fns():
    global _fns
    if _fns is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        _fns = abi.bind(lib, PREFIX, aliases=ALIASES, argtypes=ARGTYPES)
        lib.ref_set_threads.argtypes = [C.c_void_p, C.c_int]
        _fns["set_threads"] = lib.ref_set_threads
    return _fns


def layers_from_network(net):
    """oracle.dqn_oracle.Network -> list[LayerDesc] (base, val, adv order)."""
    out = []

    def add(layers, stream):
        for l in layers:
            d = abi.LayerDesc()
            d.act, d.stream = l.act, stream
            if l.kind == "dense":
                d.kind, d.n_in, d.n_out = abi.LAYER_DENSE, l.n_in, l.n_out
            elif l.kind == "lstm":
                d.kind, d.n_in, d.n_out = abi.LAYER_LSTM, l.n_in, l.n_out
            else:
                d.kind = abi.LAYER_CONV
                d.cin, d.cout, d.kh, d.kw, d.sh, d.sw = l.cin, l.cout, l.kh, l.kw, l.sh, l.sw
            out.append(d)

    add(net.base, abi.STREAM_BASE)
    if net.dueling:
        add(net.val, abi.STREAM_VAL)
        add(net.adv, abi.STREAM_ADV)
    return out


def hparams_for(net, **kw):
    shp = net.obs_shape
    c, h, w = (shp if len(shp) == 3 else (int(np.prod(shp)), 1, 1))
    base = dict(n_actions=net.n_actions, obs_c=c, obs_h=h, obs_w=w, dueling=int(net.dueling))
    base.update(kw)
    return abi.default_hparams(**base)


def default_plan(layers, hp):
    f = fns()
    arr = (abi.LayerDesc * len(layers))(*layers)
    plan = (abi.LayerPlan * len(layers))()
    f["plan_default"](arr, len(layers), C.byref(hp), plan)
    return [p.astuple() for p in plan]


class Twin(abi.Handle):
    is_twin = True

    def __init__(self, layers, hp, plan=None, threads=1):
        super().__init__(fns(), layers, hp, plan=plan)
        self.f["set_threads"](self._h, threads)

    def _create(self, parr, device, h):
        return self.f["engine_create"](self.layers, self.n_layers, C.byref(self.hp), parr, C.byref(h))

    def set_threads(self, n):
        self.f["set_threads"](self._h, n)
