"""cProfile of Full Newton steps of BASELINE config 2 through the plug-in boundary (HipStepSolver
under newton_method): where the host time of the drop-in path goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf
from pygradflow_amd import problems
n, m = 4096, 1024
prob = problems.dense_qp(n, m, seed=0)
x0, y0 = np.zeros(n), np.zeros(m)
params = pgf.Params(newton_type="Full", step_solver=pgf.HipStepSolver)
orig = pgf.Iterate(prob, params, x0, y0)
method = pgf.newton_method(prob, params, orig, 1.0, 1.0)
it = method.step(orig).iterate
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(6):
    it = method.step(it).iterate
print("ms/step", 1e3 * (time.perf_counter() - t0) / 6)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
