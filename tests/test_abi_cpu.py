"""CPU: the C-ABI library loads, exports every symbol include/dqn_mi355x.h declares, and its host-only entry points
work without a GPU.  No compute call is made here (there is no CPU fallback to call)."""
import ctypes
import numpy as np
import os
import re

import pytest

import __graft_entry__ as ge
import ref
from nets import GOLDEN_CASES, nature_dueling


@pytest.fixture(scope="module")
def pkg():
    return ge.build()


def test_every_declared_symbol_is_exported(pkg):
    hdr = open(os.path.join(ge.ROOT, "include", "dqn_mi355x.h")).read()
    names = set(re.findall(r"\b(dqn_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 30
    lib = ctypes.CDLL(pkg.LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_structs_match_the_header_sizes(pkg):
    # sizeof and the offset of one late field per struct, via a tiny C program compiled against the header
    import subprocess, tempfile
    abi = pkg._abi
    src = ('#include <stdio.h>\n#include <stddef.h>\n#include "dqn_mi355x.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu  %zu %zu %zu %zu\\n", '
           'sizeof(dqn_layer_desc), sizeof(dqn_layer_plan), sizeof(dqn_hparams), sizeof(dqn_env_spec), sizeof(dqn_rollout_cfg), sizeof(dqn_rollout_stats), sizeof(dqn_counters), '
           'offsetof(dqn_hparams, trace_length), offsetof(dqn_env_spec, images), offsetof(dqn_env_spec, reward_val), offsetof(dqn_rollout_cfg, t0));}')
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-I", os.path.join(ge.ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")], check=True)
        got = list(map(int, subprocess.run([os.path.join(d, "t")], capture_output=True, text=True, check=True).stdout.split()))
    want = [ctypes.sizeof(x) for x in (pkg.LayerDesc, pkg.LayerPlan, pkg.HParams, abi.EnvSpec, abi.RolloutCfg, abi.RolloutStats, abi.Counters)]
    want += [pkg.HParams.trace_length.offset, abi.EnvSpec.images.offset, abi.EnvSpec.reward_val.offset, abi.RolloutCfg.t0.offset]
    assert got == want, (got, want)


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
@pytest.mark.parametrize("B", [4, 32, 512])
def test_default_plan_product_equals_twin(pkg, name, B):
    net = GOLDEN_CASES[name]()
    hp = ref.hparams_for(net, batch_size=B, buffer_size=1024)
    layers = ref.layers_from_network(net)
    assert pkg.default_plan(layers, hp) == ref.default_plan(layers, hp)


@pytest.mark.parametrize("B,T,dq", [(32, 8, 1), (4, 3, 0), (16, 10, 1), (32, 40, 1), (6, 5, 1), (33, 8, 1)])
def test_default_plan_recurrent_column_groups_product_equals_twin(pkg, B, T, dq):
    """recurrent networks: the default plan of the networks the fused column-parallel step covers cuts dW / db by batch-column groups (dw_kc = -cg);
    the product's rule (common.h drqn_fused_cg) and the twin's restatement must agree, covered or not"""
    from drqn_common import drqn_nets
    seen = set()
    for name, (net, _, _, kw) in drqn_nets().items():
        hp = ref.hparams_for(net, batch_size=B, buffer_size=64, recurrence=1, trace_length=T, prioritized_replay=0, gamma=0.9, double_q=dq)
        layers = ref.layers_from_network(net)
        a, b = pkg.default_plan(layers, hp), ref.default_plan(layers, hp)
        assert a == b, (name, a, b)
        seen.add((name, a[0][2]))
    if (B, T, dq) == (32, 8, 1):
        assert ("cfg4_lstm_plain", -2) in seen and ("lstm16_dueling_b16", -2) in seen      # BASELINE config 4: groups of 2 columns, 16 workgroups
        assert ("dense_lstm_dueling", 0) in seen                                            # a Dense layer in front of the LSTM: the multi-launch program


def test_hparams_default_matches_reference_defaults(pkg):
    hp = pkg.HParams()
    pkg.fns()["hparams_default"](ctypes.byref(hp))
    # src/solver.jl:3-20, src/prioritized_experience_replay.jl:43-45
    assert hp.batch_size == 32 and abs(hp.learning_rate - 1e-4) < 1e-10 and hp.double_q == 1 and hp.dueling == 1
    assert hp.prioritized_replay == 1 and hp.buffer_size == 1000
    assert abs(hp.prio_alpha - 0.6) < 1e-7 and abs(hp.prio_beta - 0.4) < 1e-7 and abs(hp.prio_eps - 1e-3) < 1e-9


def test_dueling_incompatible_network_is_rejected(pkg):
    net = nature_dueling()
    hp = ref.hparams_for(net, batch_size=32)
    layers = ref.layers_from_network(net)[:-1]  # drop the adv head -> incompatible
    arr = (pkg.LayerDesc * len(layers))(*layers)
    plan = (pkg.LayerPlan * len(layers))()
    assert pkg.fns()["plan_default"](arr, len(layers), ctypes.byref(hp), plan) != 0
    assert b"incompatible with dueling" in pkg.fns()["last_error"]()  # src/dueling.jl:47


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    net = nature_dueling()
    with pytest.raises(pkg.DQNError, match="no CPU fallback"):
        pkg.Engine(ref.layers_from_network(net), ref.hparams_for(net, batch_size=32))


def test_null_handle_is_an_error_not_a_crash():
    """every entry point taking an engine checks the handle (C callers): rc = -1 and a message, no dereference"""
    import ctypes as C
    pkg = ge.load_package()
    lib = pkg.lib()
    lib.dqn_last_error.restype = C.c_char_p
    for name in ("dqn_sync_target", "dqn_stream_sync", "dqn_envs_reset", "dqn_reset_state", "dqn_episode_commit"):
        f = getattr(lib, name); f.argtypes = [C.c_void_p]; f.restype = C.c_int
        assert f(None) == -1 and b"null engine handle" in lib.dqn_last_error()
    lib.dqn_engine_destroy.argtypes = [C.c_void_p]; lib.dqn_engine_destroy.restype = C.c_int
    assert lib.dqn_engine_destroy(None) == 0


def test_u8_unit_two_operation_form_equals_the_division_for_every_byte():
    """common.h u8_unit: fma(b, r_lo, b * r_hi) == (float)b / 255f0 (test/test_env.jl:59) for all 256 bytes; b * r_hi is exact."""
    b = np.arange(256, dtype=np.float64)
    ref = (b.astype(np.float32) / np.float32(255.0)).astype(np.float32)
    r_hi, r_lo = float.fromhex("0x1.01p-8"), float.fromhex("0x1.010102p-24")
    assert np.float32(r_hi) == r_hi and np.float32(r_lo) == r_lo
    q = b * r_hi
    assert np.array_equal(q.astype(np.float32).astype(np.float64), q)          # the product is exact in fp32
    y = (b * r_lo + q).astype(np.float32)                                      # exact in fp64, rounded once == fmaf
    assert np.array_equal(y, ref)


def test_reciprocal_division_of_the_prologues_is_exact_below_2_to_21():
    """common.h fdiv_q: floor((x + 0.5) * rcp(d)) == x // d for 0 <= x < 2^21 and the divisors the kernels use (tile counts, kernel extents, strides, widths),
    with the reciprocal off by up to one ulp either way (v_rcp_f32 is a 1-ulp instruction).  Checked at the risky points: x = k d - 1, k d, k d + 1."""
    f32 = np.float32
    for d in list(range(1, 130)) + [144, 196, 256, 400, 441, 512, 784, 1000, 1536, 2401, 3136, 4096, 7056, 65535]:
        k = np.arange(0, (1 << 21) // d + 1, dtype=np.int64)
        x = np.unique(np.concatenate([k * d - 1, k * d, k * d + 1]))
        x = x[(x >= 0) & (x < (1 << 21))]
        r0 = f32(1.0) / f32(d)
        for r in (np.nextafter(r0, f32(0)), r0, np.nextafter(r0, f32(2))):
            q = ((x.astype(np.float32) + f32(0.5)) * r).astype(np.int64)      # float32 arithmetic, truncation == floor for non-negative values
            assert np.array_equal(q, x // d), d
