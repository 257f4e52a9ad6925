#!/usr/bin/env python3
"""GPU check of the dense factorisation schedule selected by PGF_FACTOR / PGF_CHAIN_WAVES:
HipLinearSolver against numpy on quasi-definite matrices of awkward sizes, then the time of
Full Newton steps of BASELINE config 2 (device-resident)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf  # noqa: E402
from pygradflow_amd import problems  # noqa: E402


def sqd(rng, n1, n2):
    G1 = rng.standard_normal((n1, n1)) / np.sqrt(max(n1, 1))
    A = G1 @ G1.T + np.eye(n1)
    B = rng.standard_normal((n2, n1)) / np.sqrt(max(n1, 1))
    return np.block([[A, B.T], [B, -0.5 * np.eye(n2)]])


def main():
    worst = 0.0
    for n1, n2 in [(1, 0), (3, 2), (63, 0), (64, 0), (65, 0), (64, 64), (100, 30), (130, 61), (200, 56),
                   (255, 0), (256, 0), (257, 128), (300, 213), (511, 1), (700, 189), (1500, 500),
                   (2049, 512)]:
        rng = np.random.default_rng(n1 * 1000 + n2)
        K = sqd(rng, n1, n2)
        rhs = rng.standard_normal(n1 + n2)
        sv = pgf.HipLinearSolver(K, symmetric=True)
        sol = sv.solve(rhs)
        ref = np.linalg.solve(K, rhs)
        err = np.max(np.abs(sol - ref)) / max(1.0, np.max(np.abs(ref)))
        nn = sv.num_neg_eigvals()
        F = sv.factor_matrix()
        L = np.tril(F, -1) + np.eye(n1 + n2)
        rec = np.max(np.abs((L * np.diag(F)) @ L.T - K))
        print(f"N={n1 + n2:5d} err={err:.2e} n_neg={nn} (want {n2}) |LDL'-K|={rec:.2e}", flush=True)
        worst = max(worst, err)
        assert nn == n2 and err < 1e-11, (n1, n2, err, nn)
        sv.close()
    print("linear solver ok, worst", worst, flush=True)
    if "--no-time" in sys.argv:
        return
    n, m = 4096, 1024
    prob = problems.dense_qp(n, m, seed=0)
    dn = pgf.DeviceNewton(prob, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
    for i in range(3):
        dn.step()
    x1, y1 = dn.point()
    t0 = time.perf_counter()
    K = 20
    for i in range(K):
        if i % 2 == 0:
            dn.advance_outer(1.0, 1.0)
        dn.step()
    el = time.perf_counter() - t0
    print(f"config 2: {1e3 * el / K:.3f} ms/step  ({K / el:.1f} steps/s)", flush=True)
    r = dn.residual_norm()
    print("residual norm after steps", r)
    dn.close()


if __name__ == "__main__":
    main()
