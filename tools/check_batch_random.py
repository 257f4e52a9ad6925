#!/usr/bin/env python3
"""Randomised check of the device batch (GPU): random (n, m, B), boxed shares, per-instance dt and
rho (pgf_batch_advance_outer_each: every instance its own lambda and delta -- the condensed
order's panel scaling is per instance), rejected instances, all three policies; the batch against
the same instances driven one by one through the single-instance path (sequential=True), points
to 1e-11, step lengths and statuses equal.  argv[1]: seed (default 0).  Used by
tests/test_gpu_schedules.py under PGF_CONDENSED = 2 / 0 / default."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pygradflow_amd import problems  # noqa: E402
from pygradflow_amd.batched import BatchedDeviceNewton  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(500 + seed)
worst = 0.0
for case in range(6):
    n = int(rng.choice([40, 130, 257, 300, 520]))
    m = int(rng.integers(1, min(n, 280) + 1))
    B = int(rng.choice([1, 3, 7, 9, 17]))
    kind = str(rng.choice(["Full", "Simplified", "ActiveSet"]))
    fracs = rng.choice([0.0, 0.1, 0.4], size=B)
    seeds = rng.integers(1 << 30, size=B)

    def make(i):
        return problems.dense_qp(n, m, seed=int(seeds[i]), boxed_frac=float(fracs[i]), box=0.05)

    a = BatchedDeviceNewton(make, B, kind, 1.0, 1.0)
    b = BatchedDeviceNewton(make, B, kind, 1.0, 1.0, sequential=True)
    for outer in range(3):
        dt = rng.choice([0.2, 1.0, 5.0, 50.0], size=B).astype(np.float64)
        rho = rng.choice([0.5, 1.0, 4.0], size=B).astype(np.float64)
        acc = None
        if outer > 0:
            acc = rng.random(B) > 0.3  # some instances go back to their outer point
        a.advance_outer_each(dt, rho, acc)
        b.advance_outer_each(dt, rho, acc)
        for k in range(2):
            sa, na, da = a.step_local()
            sb, nb_, db = b.step_local()
            assert np.array_equal(sa, sb), (case, outer, k, sa, sb)
            assert np.array_equal(na, nb_), (case, outer, k, na, nb_)
            assert np.allclose(da, db, rtol=1e-9, atol=1e-12), (case, outer, k)
        xa, ya = a.points()
        xb, yb = b.points()
        err = max(np.max(np.abs(xa - xb)) / max(1.0, np.max(np.abs(xb))),
                  np.max(np.abs(ya - yb)) / max(1.0, np.max(np.abs(yb))))
        worst = max(worst, err)
        assert err <= 1e-11, (case, outer, kind, n, m, B, err)
    a.close()
    b.close()
    print(f"case {case}: n={n} m={m} B={B} {kind}: ok (worst so far {worst:.1e})", flush=True)
print("batch random ok, worst", worst, flush=True)
