#!/usr/bin/env python3
"""Device batch against the same instances driven one by one (GPU): B instances of a boxed
dense QP whose reduced sizes differ, Full / Simplified / ActiveSet steps.  Used by
tests/test_gpu_schedules.py under the batched schedule switches."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygradflow_amd import problems  # noqa: E402
from pygradflow_amd.batched import BatchedDeviceNewton  # noqa: E402

n, m, B = 600, 150, 11  # reduced sizes around 700: three outer blocks, ragged last one


def make(i):
    return problems.dense_qp(n, m, seed=100 + i, boxed_frac=0.05 * (i % 5), box=0.05)


worst = 0.0
for kind in ("Full", "Simplified", "ActiveSet"):
    a = BatchedDeviceNewton(make, B, kind, 1.0, 1.0)
    b = BatchedDeviceNewton(make, B, kind, 1.0, 1.0, sequential=True)
    for k in range(3):
        ra, rb = a.step().cpu().numpy(), b.step().cpu().numpy()
        assert np.allclose(ra, rb, rtol=1e-9, atol=1e-12), (kind, k)
    xa, ya = a.points()
    xb, yb = b.points()
    err = max(np.max(np.abs(xa - xb)) / max(1.0, np.max(np.abs(xb))),
              np.max(np.abs(ya - yb)) / max(1.0, np.max(np.abs(yb))))
    worst = max(worst, err)
    print(f"{kind}: batch vs one-by-one {err:.2e}", flush=True)
    assert err <= 1e-11, (kind, err)
    a.close()
    b.close()
print("batch ok, worst", worst, flush=True)
