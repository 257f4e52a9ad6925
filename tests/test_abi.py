"""The C-ABI library loads and exports every symbol include/pgf_hip.h declares (CPU)."""

import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "pgf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pgf_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_bound_and_exported():
    from pygradflow_amd import _lib

    declared = _declared_symbols()
    assert len(declared) >= 30
    assert set(declared) == set(_lib.SIGNATURES), (
        set(declared) ^ set(_lib.SIGNATURES)
    )
    lib = _lib.load()  # raises if a symbol is missing
    for name in declared:
        assert hasattr(lib, name)
    assert lib.pgf_version() >= 1


def test_no_cpu_fallback_without_gpu(gpu_available):
    """The product path must fail loudly when no GPU is present."""
    if gpu_available:
        pytest.skip("GPU present")
    import numpy as np

    from pygradflow_amd import Iterate, Params, problems
    from pygradflow_amd.step_solver import HipStepSolver

    prob = problems.dense_qp(8, 2, seed=0)
    params = Params()
    it = Iterate(prob, params, np.zeros(8), np.zeros(2))
    with pytest.raises(RuntimeError, match="no CPU fallback|GPU"):
        HipStepSolver(prob, params, it, 1.0, 1.0)


def test_product_never_imports_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(REPO, "pygradflow_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f


def test_single_precision_is_refused_before_touching_the_gpu():
    """Precision.Single is outside the 1e-10 contract: the step solver must say
    so instead of computing in the wrong precision."""
    import numpy as np
    import pytest

    from pygradflow_amd import problems
    from pygradflow_amd.iterate import Iterate
    from pygradflow_amd.params import Params
    from pygradflow_amd.step_solver import HipStepSolver

    prob = problems.dense_qp(8, 2, seed=0)
    par = Params(precision="Single")
    it = Iterate(prob, par, np.zeros(8, dtype=np.float32), np.zeros(2, dtype=np.float32))
    with pytest.raises(ValueError, match="float64 only"):
        HipStepSolver(prob, par, it, 1.0, 1.0)
    with pytest.raises(ValueError, match="positive"):
        HipStepSolver(prob, Params(), it, -1.0, 1.0)


def test_shard_ranges_cover_every_instance_once():
    from pygradflow_amd.batched import shard_range

    for B in (0, 1, 7, 256, 257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = shard_range(B, world, r)
                assert 0 <= lo <= hi <= B
                seen.extend(range(lo, hi))
            assert seen == list(range(B))


def test_float32_params_are_refused_by_every_formulation():
    """The HIP solvers compute in float64 only (Precision.Single is outside the 1e-10 parity
    contract): all four formulations say so instead of silently widening."""
    import numpy as np
    import pytest

    from pygradflow_amd import problems
    from pygradflow_amd.iterate import Iterate
    from pygradflow_amd.params import Params
    from pygradflow_amd.unsym_step_solvers import StandardStepSolver

    prob = problems.dense_qp(8, 2, seed=0)
    par = Params(precision="Single")
    it = Iterate(prob, par, np.zeros(8, dtype=np.float32), np.zeros(2, dtype=np.float32))
    with pytest.raises(ValueError, match="float64"):
        StandardStepSolver(prob, par, it, 1.0, 1.0)
