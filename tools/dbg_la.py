import numpy as np, sys
sys.path.insert(0, ".")
import pygradflow_amd as pgf
rng = np.random.default_rng(0)
for n1, n2 in [(4096, 1024)]:
    G1 = rng.standard_normal((n1, n1)) / np.sqrt(n1)
    A = G1 @ G1.T + np.eye(n1)
    B = rng.standard_normal((n2, n1)) / np.sqrt(n1)
    K = np.block([[A, B.T], [B, -0.5 * np.eye(n2)]])
    rhs = rng.standard_normal(n1 + n2)
    ref = np.linalg.solve(K, rhs)
    for rep in range(2):
        sv = pgf.HipLinearSolver(K, symmetric=True)
        sol = sv.solve(rhs)
        print(n1 + n2, rep, "relerr", np.max(np.abs(sol - ref)) / np.max(np.abs(ref)), "nneg", sv.num_neg_eigvals(), flush=True)
        sv.close()
