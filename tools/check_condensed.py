#!/usr/bin/env python3
"""The condensed KKT factorisation (constraint block eliminated first, csrc/pgf_api.hip
``condensed_wanted``) against the CPU oracle (GPU): boxed dense QPs whose reduced sizes put the
constraint block across zero, one and several 256-column blocks, Full and Simplified steps
(factorisation with the right-hand side riding along, then back-solve steps with the same
factor), the linear-solver view with an arbitrary right-hand side and the inertia.

Run under PGF_CONDENSED=2 (condensed whenever the growth bound allows -- the sizes here are below
what the default mode would pick) and, for the same numbers from the natural pivot order,
PGF_CONDENSED=0.  Used by tests/test_gpu_schedules.py."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import newton_oracle as O  # noqa: E402  (checker)
from pygradflow_amd import problems  # noqa: E402
from pygradflow_amd.newton import DeviceNewton  # noqa: E402

CASES = [
    # n, m, boxed share, policy, steps
    (96, 24, 0.25, "Full", 3),
    (300, 70, 0.10, "Full", 3),
    (300, 70, 0.10, "Simplified", 4),
    (520, 260, 0.05, "Full", 2),       # one whole virtual block + a few columns
    (700, 300, 0.00, "Simplified", 3),
    (1100, 530, 0.02, "Full", 2),      # three virtual blocks, ragged; five real ones
    (1100, 530, 0.02, "ActiveSet", 3),
    (900, 544, 0.02, "Simplified", 3),  # depth 544 without padding
]

# which factorisation every step must have used (pgf_debug_factor_kind): the condensed one under
# PGF_CONDENSED=2, the natural order under 0; the default mode picks by size
WANT = {"2": 2, "0": 1}.get(os.environ.get("PGF_CONDENSED", ""), None)
if len(sys.argv) == 2:  # only the cases with this n
    CASES = [c for c in CASES if c[0] == int(sys.argv[1])]
elif len(sys.argv) >= 4:  # n m boxed-share [policy]: one case of one's own
    CASES = [(int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4] if len(sys.argv) > 4 else "Full", 2)]
worst = 0.0
for n, m, frac, kind, steps in CASES:
    prob = problems.dense_qp(n, m, seed=7 + n, boxed_frac=frac, box=0.05)
    x0, y0 = np.zeros(n), np.zeros(m)
    recs = O.NewtonOracle(prob, kind, x0, y0, 1.0, 1.0).run(x0, y0, steps)
    dn = DeviceNewton(prob, kind, x0, y0, 1.0, 1.0)
    kinds = []
    for k, rec in enumerate(recs):
        diff, n_neg = dn.step()
        kinds.append(dn.factor_kind())
        x, y = dn.point()
        assert np.array_equal(dn.mask(), rec["mask"]), (n, m, kind, k, "mask")
        ex = np.max(np.abs(x - rec["xn"])) / max(1.0, np.max(np.abs(rec["xn"])))
        ey = np.max(np.abs(y - rec["yn"])) / max(1.0, np.max(np.abs(rec["yn"]))) if m else 0.0
        worst = max(worst, ex, ey)
        assert ex <= 1e-10 and ey <= 1e-10, (n, m, kind, k, ex, ey)
        assert n_neg == m, (n, m, kind, k, n_neg)
    refined, lu, rel = dn.refinement_stats()
    print(f"   factor kinds {kinds}, refined {refined}, LU fallbacks {lu}, last residual {rel:.1e}", flush=True)
    if WANT is not None:
        assert all(kd == WANT for kd in kinds), (n, m, kind, kinds)
    assert refined == 0 and lu == 0, (n, m, kind, refined, lu, rel)
    dn.close()
    print(f"n={n} m={m} boxed={frac} {kind}: {steps} steps, worst so far {worst:.2e}", flush=True)
if len(sys.argv) == 1:
    # The condensed pivot order can meet an exactly zero pivot where the natural order does not:
    # A = diag(-2, 3, 2, 5), J = (1, 1, 0, 0), delta = 1/2 gives S[0][0] = -2 + 1 / delta = 0.  The
    # library then repeats the factorisation in the natural order inside the same call
    # (finish_factor_state, csrc/pgf_api.hip) -- the caller sees the reference's behaviour: the
    # step goes through with inertia m + 1.
    Q = np.diag([-3.0, 2.0, 1.0, 4.0])
    A = np.array([[1.0, 1.0, 0.0, 0.0]])
    prob = problems.LinearQuadraticProblem(Q, np.ones(4), A, np.zeros(1), np.full(4, -np.inf), np.full(4, np.inf))
    rec = O.NewtonOracle(prob, "Full", np.zeros(4), np.zeros(1), 1.0, 1.0).run(np.zeros(4), np.zeros(1), 1)[0]
    dn = DeviceNewton(prob, "Full", np.zeros(4), np.zeros(1), 1.0, 1.0)
    diff, n_neg = dn.step()
    x, y = dn.point()
    assert n_neg == 2, n_neg
    assert dn.factor_kind() == 1, dn.factor_kind()  # natural order, whatever PGF_CONDENSED says
    assert np.max(np.abs(x - rec["xn"])) <= 1e-12 and np.max(np.abs(y - rec["yn"])) <= 1e-12
    dn.close()
    print("zero pivot of the condensed order: repeated in the natural order, inertia", n_neg, flush=True)
print("condensed ok, worst", worst, flush=True)
