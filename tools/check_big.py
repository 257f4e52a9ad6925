#!/usr/bin/env python3
"""Dense factorisation beyond 96 column blocks (N > 24576: adjacent column blocks share update
jobs): HipLinearSolver on a quasi-definite matrix, checked through the residual."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24900
m = N // 10
rng = np.random.default_rng(1)
K = rng.standard_normal((N, N))
K += K.T
K *= 0.5
K[np.diag_indices(N)] += 4.0 * np.sqrt(N)
K[N - m:, N - m:] *= -1.0
rhs = rng.standard_normal(N)
t0 = time.perf_counter()
sv = pgf.HipLinearSolver(K, symmetric=True)
x = sv.solve(rhs)
t1 = time.perf_counter()
r = np.max(np.abs(K @ x - rhs)) / np.max(np.abs(rhs))
print(f"N={N}: n_neg={sv.num_neg_eigvals()} (want {m})  rel. residual {r:.2e}  ({t1 - t0:.2f} s incl. upload)", flush=True)
assert sv.num_neg_eigvals() == m and r < 1e-10
sv.close()
