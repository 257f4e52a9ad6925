#!/usr/bin/env python3
"""Randomised parity sweep (GPU): boxed dense QPs of random shapes -- n, m, share of boxed
variables, box width, dt, rho, policy -- through DeviceNewton against the CPU oracle, masks bit for
bit and iterates to 1e-10, inertia m, no refinement; the pivot order is whatever the library picks
(PGF_CONDENSED=2 forces the condensed order wherever its bounds allow, =0 the natural one).  The
seed comes from argv[1] (default 0), the number of cases from argv[2] (default 24).  Used by
tests/test_gpu_schedules.py with two seeds."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import newton_oracle as O  # noqa: E402  (checker)
from pygradflow_amd import problems  # noqa: E402
from pygradflow_amd.newton import DeviceNewton  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(1000 + seed)
worst, kinds_seen = 0.0, set()
for c in range(cases):
    n = int(rng.choice([17, 40, 64, 100, 255, 256, 257, 300, 513, 700, 1030]))
    m = int(rng.integers(0, max(1, min(n, 600)) + 1)) if rng.random() < 0.9 else 0
    if rng.random() < 0.3:
        m = min(m, 8)
    frac = float(rng.choice([0.0, 0.05, 0.3, 0.7]))
    box = float(rng.choice([0.01, 0.05, 0.5]))
    dt = float(rng.choice([0.1, 1.0, 1.0, 10.0, 1000.0]))
    rho = float(rng.choice([0.1, 1.0, 10.0]))
    kind = str(rng.choice(["Full", "Simplified", "ActiveSet"]))
    steps = 3
    prob = problems.dense_qp(n, m, seed=int(rng.integers(1 << 30)), boxed_frac=frac, box=box)
    x0 = 0.1 * rng.standard_normal(n)
    y0 = 0.1 * rng.standard_normal(m)
    recs = O.NewtonOracle(prob, kind, x0, y0, dt, rho).run(x0, y0, steps)
    dn = DeviceNewton(prob, kind, x0, y0, dt, rho)
    for k, rec in enumerate(recs):
        diff, n_neg = dn.step()
        kinds_seen.add(dn.factor_kind())
        x, y = dn.point()
        tag = (c, n, m, frac, box, dt, rho, kind, k)
        assert np.array_equal(dn.mask(), rec["mask"]), tag
        ex = np.max(np.abs(x - rec["xn"])) / max(1.0, np.max(np.abs(rec["xn"])))
        ey = np.max(np.abs(y - rec["yn"])) / max(1.0, np.max(np.abs(rec["yn"]))) if m else 0.0
        worst = max(worst, ex, ey)
        assert ex <= 1e-10 and ey <= 1e-10, tag + (ex, ey)
        assert n_neg == m, tag + (n_neg,)
    refined, lu, rel = dn.refinement_stats()
    assert lu == 0, (c, n, m, kind, refined, lu, rel)
    dn.close()
    print(f"case {c}: n={n} m={m} boxed={frac} box={box} dt={dt} rho={rho} {kind}: ok (worst so far {worst:.1e})",
          flush=True)
print("random ok, worst", worst, "factor kinds", sorted(kinds_seen), flush=True)
