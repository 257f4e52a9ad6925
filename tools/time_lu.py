#!/usr/bin/env python3
"""Time the dense LU with partial pivoting (HipLinearSolver(symmetric=False)) against the
LDL^T path on the same quasi-definite matrix; rocprofv3-friendly."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf  # noqa: E402

for N in [int(a) for a in sys.argv[1:]] or [1024, 2560, 5120]:
    rng = np.random.default_rng(N)
    n1 = (4 * N) // 5
    G = rng.standard_normal((n1, n1)) / np.sqrt(n1)
    A = G @ G.T + np.eye(n1)
    B = rng.standard_normal((N - n1, n1)) / np.sqrt(n1)
    K = np.block([[A, B.T], [B, -0.5 * np.eye(N - n1)]])
    rhs = rng.standard_normal(N)
    for sym in (True, False):
        sv = pgf.HipLinearSolver(K, symmetric=sym)  # warm-up (allocations)
        sv.close()
        t0 = time.perf_counter()
        sv = pgf.HipLinearSolver(K, symmetric=sym)
        t1 = time.perf_counter()
        x = sv.solve(rhs)
        t2 = time.perf_counter()
        err = np.max(np.abs(K @ x - rhs))
        print(f"N={N} {'LDLt' if sym else 'LU  '}: create+factor {1e3 * (t1 - t0):8.2f} ms (incl. {K.nbytes / 1e6:.0f} MB upload)  solve {1e3 * (t2 - t1):6.2f} ms  resid {err:.1e}", flush=True)
        sv.close()
