"""Randomised condition estimate of the reduced KKT matrix (``Params.report_rcond``).

Restates Dixon's estimator as the reference uses it
(``pygradflow/step/cond_estimate.py:13-114``; called from
``step/solver/step_solver.py:100-113`` and ``symmetric_step_solver.py:123-125``): two
random unit vectors from ``default_rng(42)``, ``k`` power iterations with ``A'A`` and with
``A^-1 A^-T`` (the latter through the device factor: this is why ``LinearSolver.solve`` has
a ``trans`` argument), ``cond ~ (x'(A'A)^k x)^(1/2k) (y'(A'A)^-k y)^(1/2k)``.
The products with the small host copy of ``K`` stay on the host; the solves run on the GPU.
Diagnostic only, off the hot path.
"""

from __future__ import annotations

import math

import numpy as np

from .errors import LinearSolverError

SEED = 42  # cond_estimate.py:10


def _iterations(size, min_prob=0.99, factor=10.0):
    f = (1.0 - min_prob) / 1.6 * math.pow(size, -0.5)
    return -2 * math.ceil(math.log(f, factor))


def _unit_vector(rng, size, dtype):
    vec = rng.normal(size=size)
    while not (vec != 0.0).any():
        vec = rng.normal(size=size)
    return (vec / np.linalg.norm(vec)).astype(dtype)


def estimate_rcond(mat, solver, params):
    size = mat.shape[0]
    if size == 0:
        return None
    rng = np.random.default_rng(seed=SEED)
    its = _iterations(size)
    x = _unit_vector(rng, size, params.dtype)
    y = _unit_vector(rng, size, params.dtype)
    xp, yp = np.copy(x), np.copy(y)
    xfac = yfac = 1.0
    try:
        for _ in range(its):
            xp = mat.T @ (mat @ xp)
            yp = solver.solve(solver.solve(yp, trans=True))
            xn, yn = float(np.linalg.norm(xp)), float(np.linalg.norm(yp))
            xfac *= xn
            xp /= xn
            yfac *= yn
            yp /= yn
    except LinearSolverError:
        return None
    power = 1.0 / (2.0 * its)
    xd = math.pow(x.dot(xp) * xfac, power)
    yd = math.pow(y.dot(yp) * yfac, power)
    if np.isinf(xd) or np.isinf(yd) or np.isinf(xd * yd):
        return 0.0
    return 1.0 / (xd * yd)
