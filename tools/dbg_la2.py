import numpy as np, sys, os
sys.path.insert(0, ".")
import pygradflow_amd as pgf
rng = np.random.default_rng(0)
n1, n2 = 4096, 1024
G1 = rng.standard_normal((n1, n1)) / np.sqrt(n1)
A = G1 @ G1.T + np.eye(n1)
B = rng.standard_normal((n2, n1)) / np.sqrt(n1)
K = np.block([[A, B.T], [B, -0.5 * np.eye(n2)]])
sv = pgf.HipLinearSolver(K, symmetric=True)
F = sv.factor_matrix()
np.save("gpurun_out/factor_%s.npy" % os.environ.get("TAG", "x"), F)
