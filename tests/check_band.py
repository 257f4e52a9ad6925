#!/usr/bin/env python3
"""Banded path under the cyclic-reduction switches (GPU): a sparse optimal-control problem
(N = 9000) and a tridiagonal box QP with a churning mask against
the CPU oracle.  Run in a child process by tests/test_gpu_schedules.py (the switches are read once per process)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf  # noqa: E402
from pygradflow_amd import problems  # noqa: E402
from oracle import newton_oracle as O  # noqa: E402  (test infrastructure: the checker)


def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b)))) if a.size else 0.0


worst = 0.0
for name, prob in (("ocp", problems.sparse_ocp(3000, seed=1)), ("box", problems.box_qp(4099, seed=2))):
    prob.pgf_force_band = True
    n, m = prob.num_vars, prob.num_cons
    x0, y0 = np.zeros(n), np.zeros(m)
    for pol, steps in (("Full", 4), ("Simplified", 3)):
        recs = O.NewtonOracle(prob, pol, x0, y0, 1.0, 1.0).run(x0, y0, steps)
        dn = pgf.DeviceNewton(prob, pol, x0, y0, 1.0, 1.0)
        assert dn.sparse
        for k, rec in enumerate(recs):
            diff, n_neg = dn.step()
            x, y = dn.point()
            assert np.array_equal(dn.mask(), rec["mask"]), (name, pol, k)
            e = max(rel(x, rec["xn"]), rel(y, rec["yn"]))
            worst = max(worst, e)
            assert e <= 1e-10, (name, pol, k, e)
            assert n_neg == m, (name, pol, k, n_neg)
        dn.close()
    print(f"{name}: ok", flush=True)
print("band ok, worst", worst, flush=True)
