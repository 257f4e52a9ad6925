"""TEST-ONLY adapter: the CPU oracle behind the StepSolver plugin surface.

Lets the host-side policy code (Newton methods, step controllers) be exercised without a
GPU by passing ``Params(step_solver=OracleStepSolver)`` -- the same hook the HIP solver is
installed through.  Never imported by the product.
"""

import numpy as np

from oracle import newton_oracle as O
from pygradflow_amd.errors import StepSolverError


class _Result:
    def __init__(self, orig, iterate, xn, yn, dx, dy, diff, mask):
        self.orig_iterate = orig
        self.iterate = type(iterate)(iterate.problem, iterate.params, xn, yn)
        self.xn, self.dx, self.dy, self.diff = xn, dx, dy, diff
        self.active_set = mask
        self.rcond = None


class _Func:
    def __init__(self, owner):
        self.owner = owner

    def compute_active_set(self, iterate, rho, tau=None):
        o = self.owner
        return o.sv.compute_active_set(O.PointData(o.problem, iterate.x, iterate.y), tau)


class OracleStepSolver:
    def __init__(self, problem, params, orig_iterate, dt, rho):
        self.problem, self.params = problem, params
        self.n, self.m = problem.num_vars, problem.num_cons
        self.orig_iterate = orig_iterate
        self.sv = O.SymmetricStep(problem, orig_iterate.x, orig_iterate.y, dt, rho)
        self.func = _Func(self)

    def update_active_set(self, mask):
        self.sv.update_active_set(np.asarray(mask, dtype=bool))

    def update_derivs(self, iterate):
        self.sv.update_derivs(O.PointData(self.problem, iterate.x, iterate.y))

    def solve(self, iterate):
        try:
            xn, yn, diff = self.sv.solve(O.PointData(self.problem, iterate.x, iterate.y))
        except O.FactorError as e:
            raise StepSolverError(str(e)) from e
        rec = self.sv.record
        return _Result(self.orig_iterate, iterate, xn, yn, rec["dx"], rec["dy"], diff, rec["mask"])
