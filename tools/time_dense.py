#!/usr/bin/env python3
"""Time Full Newton steps of BASELINE config 2 (device-resident); used under rocprofv3."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pygradflow_amd as pgf  # noqa: E402
from pygradflow_amd import problems  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob = problems.dense_qp(n, m, seed=0)
dn = pgf.DeviceNewton(prob, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
for i in range(3):
    dn.step()
t0 = time.perf_counter()
for i in range(K):
    if i % 2 == 0:
        dn.advance_outer(1.0, 1.0)
    dn.step()
el = time.perf_counter() - t0
print(f"n={n} m={m}: {1e3 * el / K:.3f} ms/step  ({K / el:.1f} steps/s)", flush=True)
dn.close()
