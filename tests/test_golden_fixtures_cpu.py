"""CPU: every field of every committed golden fixture is consumed.
 (1) the NumPy fp64 oracle, given each fixture's own inputs (IS weights / masks included), reproduces EVERY stored torch-autograd output to 1e-12
     (the cross-check oracle/make_golden.py asserts once at generation time, re-verified on every run);
 (2) the canonical-order C twin, fed through the replay protocol, reproduces the fixture's IS weights and its loss / td / Q / gradients / Adam step
     DIRECTLY (fp32 round-off), feed-forward and DRQN;
 (3) fixtures a maintainer generates with Julia + the reference (oracle/make_golden.jl -> tests/golden/julia_*) are consumed when present."""
import glob
import os

import numpy as np
import pytest

import ref
from golden_common import DRQN_GOLDEN, engine_vs_ff_fixture, load, oracle_reproduces_drqn, oracle_reproduces_ff, run_drqn_fixture
from nets import GOLDEN_CASES
from test_twin_vs_oracle import run_case


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_oracle_reproduces_every_field_of_feedforward_fixture(name, golden_dir):
    errs = oracle_reproduces_ff(name, load(golden_dir, name))
    assert max(errs.values()) < 1e-12, errs


@pytest.mark.parametrize("name", list(DRQN_GOLDEN))
def test_oracle_reproduces_every_field_of_drqn_fixture(name, golden_dir):
    errs = oracle_reproduces_drqn(name, load(golden_dir, name))
    assert max(errs.values()) < 1e-12, errs


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_twin_reproduces_fixture_outputs_directly(name, golden_dir):
    out = run_case(name, golden_dir, ref.Twin, threads=8)
    engine_vs_ff_fixture(None, name, load(golden_dir, name), out)


@pytest.mark.parametrize("name", list(DRQN_GOLDEN))
def test_twin_reproduces_drqn_fixture(name, golden_dir):
    run_drqn_fixture(ref.Twin, name, load(golden_dir, name), threads=4)


def test_every_golden_file_has_a_consumer(golden_dir):
    have = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(golden_dir, "*.npz"))}
    known = set(GOLDEN_CASES) | set(DRQN_GOLDEN)
    assert {n for n in have if not n.startswith("julia_")} == known, have ^ known
