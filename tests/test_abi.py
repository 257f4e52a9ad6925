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
