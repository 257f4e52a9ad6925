"""The reference's three UNSYMMETRIC step-solver formulations over the GPU LU.

``StandardStepSolver`` (``pygradflow/step/solver/standard_step_solver.py:15-92``),
``ExtendedStepSolver`` (``extended_step_solver.py:13-112``) and ``AsymmetricStepSolver``
(``asymmetric_step_solver.py:15-173``) take the SAME Newton step as the default Symmetric
formulation, through an unsymmetric ``(n + m) x (n + m)`` matrix instead of the reduced
symmetric KKT matrix.  They exist in the reference as alternative linear-algebra routes and
its tests parametrise every Newton policy over all four (``tests/pygradflow/test_newton.py:142-214``,
``test_solver.py:191-215``); SURVEY.md section 8(f) ranks them "next" after the hot path.

Division of labour here: the matrices are put together on the host with scipy, exactly as the
reference does it (a handful of block operations per factorisation, never the hot path: the
production formulation is ``HipStepSolver``); the factorisation and the solves -- the part that
costs -- run on the GPU through ``HipLinearSolver(symmetric=False)``: dense LU with partial
pivoting (``csrc/pgf_lu.hip``), which is what the reference's ``LUSolver`` does for every
matrix (``linear_solver/lu_solver.py:9-21``).  The scaled residual / active set of the Extended
and Asymmetric formulations come from the device kernels (``HipStepFunc``); the Standard
formulation works with the UNSCALED residual (``implicit_func.py:102-199``), whose O(n)
elementwise arithmetic is restated below in the reference's operation order (its active-set
thresholds ``lb - 1e-8`` differ from the scaled ``lambda lb - 1e-8``, so it cannot borrow the
scaled kernels bit for bit).
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from .errors import LinearSolverError, StepSolverError
from .linear_solver import HipLinearSolver
from .params import enum_name
from .step_solver import HipStepSolver, StepResult


def _keep_rows(mat, row_filter):
    """Rows with ``row_filter`` False are cleared (reference ``util.keep_rows``, util.py:27-55)."""
    if row_filter.all():
        return mat
    mat = sps.coo_matrix(mat)
    keep = row_filter[mat.row]
    return sps.coo_matrix((mat.data[keep], (mat.row[keep], mat.col[keep])), shape=mat.shape)


class UnscaledStepFunc:
    """``F(x, y) = [x - P(x^ - dt g); y - (y^ + dt c)]`` (reference ``ImplicitFunc``,
    implicit_func.py:102-199) -- the residual the Standard formulation linearises."""

    def __init__(self, problem, orig_iterate, dt):
        self.problem = problem
        self.orig_iterate = orig_iterate
        self.dt = dt
        self.n, self.m = problem.num_vars, problem.num_cons

    def projection_initial(self, iterate, rho, tau=None):
        x_0, dt = self.orig_iterate.x, self.dt
        if tau is not None:
            lamb = 1.0 / dt
            return ((1.0 - tau * lamb) * iterate.x + (tau * lamb) * x_0
                    - tau * iterate.aug_lag_deriv_x(rho))
        return x_0 - dt * iterate.aug_lag_deriv_x(rho)

    def compute_active_set(self, iterate, rho, tau=None):
        p = self.projection_initial(iterate, rho, tau)
        lb, ub = self.problem.var_lb, self.problem.var_ub
        return np.logical_or(p < lb - 1e-8, p > ub + 1e-8)  # implicit_func.py:44

    def project(self, p, active_set):
        lb, ub = self.problem.var_lb, self.problem.var_ub
        proj = np.copy(p)
        proj[active_set] = np.clip(p[active_set], lb[active_set], ub[active_set])
        return proj

    def value_at(self, iterate, rho, active_set=None):
        p = self.projection_initial(iterate, rho)
        if active_set is None:
            active_set = self.compute_active_set(iterate, rho)
        xval = iterate.x - self.project(p, active_set)
        yval = iterate.y - (self.orig_iterate.y + self.dt * iterate.aug_lag_deriv_y())
        return np.concatenate([xval, yval])

    def deriv(self, jac, hess, active_set):
        n, m, dt = self.n, self.m, self.dt
        inactive = np.logical_not(active_set)
        F_11 = sps.eye(n) + _keep_rows(dt * sps.csr_matrix(hess), inactive)
        F_12 = _keep_rows(dt * sps.csr_matrix(jac).T, inactive)
        F_21 = -dt * sps.csr_matrix(jac)
        F_22 = sps.eye(m)
        return sps.bmat([[F_11, F_12], [F_21, F_22]], format="csc")

    def deriv_at(self, iterate, rho, active_set=None):
        if active_set is None:
            active_set = self.compute_active_set(iterate, rho)
        return self.deriv(iterate.aug_lag_deriv_xy(), iterate.aug_lag_deriv_xx(rho), active_set)


class _UnsymmetricStepSolver:
    """``StepSolver`` surface (step/solver/step_solver.py:66-130) shared by the three
    formulations: stash-and-invalidate setters, ``linear_solver(mat)`` override point."""

    def __init__(self, problem, params, orig_iterate, dt, rho, device=0):
        if not (dt > 0.0 and rho > 0.0):
            raise ValueError("dt and rho must be positive")
        if np.dtype(params.dtype) != np.float64:
            raise ValueError("the HIP solvers compute in float64 only")
        self.problem, self.params = problem, params
        self.n, self.m = problem.num_vars, problem.num_cons
        self.orig_iterate = orig_iterate
        self.dt, self.rho = dt, rho
        self.device = device
        self._active_set = self._jac = self._hess = None
        self._deriv = None
        self.solver = None

    @property
    def active_set(self):
        assert self._active_set is not None
        return self._active_set

    @property
    def jac(self):
        assert self._jac is not None
        return self._jac

    @property
    def hess(self):
        assert self._hess is not None
        return self._hess

    @property
    def deriv(self):
        assert self._deriv is not None
        return self._deriv

    def linear_solver(self, mat):
        return HipLinearSolver(mat, symmetric=False, device=self.device)

    def estimate_rcond(self, mat, solver):
        from .cond_estimate import estimate_rcond

        return estimate_rcond(sps.csr_matrix(mat), solver, self.params)

    def reset_deriv(self):
        self._deriv = None
        self.solver = None

    def update_active_set(self, active_set):
        self._active_set = np.array(active_set, dtype=np.bool_, copy=True)
        self.reset_deriv()

    def _factor_and_solve(self, rhs, initial_sol=None):
        try:
            if self.solver is None:
                self.solver = self.linear_solver(self.deriv)
            return self.solver.solve(rhs) if initial_sol is None else self.solver.solve(
                rhs, initial_sol=initial_sol)
        except LinearSolverError as e:
            raise StepSolverError(str(e)) from e

    def _rcond(self):
        if not getattr(self.params, "report_rcond", False):
            return None
        try:
            return self.estimate_rcond(self.deriv, self.solver)
        except LinearSolverError:
            return None

    def close(self):
        pass


class StandardStepSolver(_UnsymmetricStepSolver):
    """Newton step on the unscaled residual: ``F'(z) s = F(z)`` with ``F'`` from
    ``ImplicitFunc.deriv`` (standard_step_solver.py:40-92); the Hessian carries the
    ``rho J'J`` term (``aug_lag_deriv_xx(rho)``, :52)."""

    def __init__(self, problem, params, orig_iterate, dt, rho, device=0):
        super().__init__(problem, params, orig_iterate, dt, rho, device)
        self._func = UnscaledStepFunc(problem, orig_iterate, dt)

    @property
    def func(self):
        return self._func

    def update_derivs(self, iterate):
        self._jac = iterate.aug_lag_deriv_xy()
        self._hess = iterate.aug_lag_deriv_xx(self.rho)
        self.reset_deriv()

    def solve(self, iterate):
        if self._deriv is None:
            self._deriv = self.func.deriv(self.jac, self.hess, self.active_set)
        rhs = self.func.value_at(iterate, self.rho, self.active_set)
        sol = self._factor_and_solve(rhs)
        n = self.n
        return StepResult(iterate, sol[:n], sol[n:], self.active_set, self._rcond())


class _ScaledUnsymmetricStepSolver(_UnsymmetricStepSolver):
    """``ScaledStepSolver`` (scaled_step_solver.py:15-107): scaled residual ``lambda F`` from
    the device kernels, right-hand side split ``b0 = dt F_x[A]``, ``b1 = F_x[I]``,
    ``b2 = F_y``, ``dy = fact (sy - rho b2)``."""

    def __init__(self, problem, params, orig_iterate, dt, rho, device=0):
        super().__init__(problem, params, orig_iterate, dt, rho, device)
        self._dev = None
        self._func = self._make_func()

    def _make_func(self):
        # the device handle behind the scaled residual / active-set kernels
        self._dev = HipStepSolver(self.problem, self.params, self.orig_iterate, self.dt, self.rho,
                                  device=self.device)
        return self._dev.func

    @property
    def func(self):
        return self._func

    def update_derivs(self, iterate):
        self._jac = iterate.aug_lag_deriv_xy()
        self._hess = iterate.aug_lag_deriv_xx(rho=0.0)
        self.reset_deriv()

    def initial_rhs(self, iterate):
        rhs = self.func.value_at(iterate, self.rho, self.active_set)
        rx, ry = rhs[: self.n], rhs[self.n:]
        act = np.where(self.active_set)[0]
        ina = np.where(np.logical_not(self.active_set))[0]
        return self.dt * rx[act], rx[ina], ry

    def solve_scaled(self, b0, b1, b2t):
        raise NotImplementedError

    def solve(self, iterate):
        b0, b1, b2 = self.initial_rhs(iterate)
        lamb = 1.0 / self.dt
        fact = 1.0 / (1.0 + lamb * self.rho)
        sx, sy, rcond = self.solve_scaled(b0, b1, fact * b2)
        return StepResult(iterate, sx, fact * (sy - self.rho * b2), self.active_set, rcond)

    def close(self):
        if getattr(self, "_dev", None) is not None:
            self._dev.close()
            self._dev = None


class ExtendedStepSolver(_ScaledUnsymmetricStepSolver):
    """Rows of the active variables replaced by unit rows, gathered on top
    (extended_step_solver.py:39-112): ``[[E_A, 0], [H_lambda[I, :], J[:, I]'], [J, -delta I]]``."""

    def _compute_deriv(self):
        act = np.where(self.active_set)[0]
        ina = np.where(np.logical_not(self.active_set))[0]
        n, m = self.n, self.m
        lamb, rho = 1.0 / self.dt, self.rho
        top = sps.coo_matrix((np.ones(act.size), (np.arange(act.size), act)), shape=(act.size, n))
        lower = sps.diags([-lamb / (1.0 + lamb * rho)], shape=(m, m))
        hess = sps.csc_matrix(self.hess) + sps.diags([lamb], shape=(n, n))
        jac = sps.csc_matrix(self.jac)
        blocks = [[top, None], [sps.csc_matrix(hess)[ina, :], sps.csc_matrix(jac.T)[ina, :]],
                  [jac, lower]]
        if m == 0:
            blocks = [[top], [sps.csc_matrix(hess)[ina, :]]]
        self._deriv = sps.bmat(blocks, format="csc")
        assert self._deriv.shape == (n + m, n + m)

    def solve_scaled(self, b0, b1, b2t):
        if self._deriv is None:
            self._compute_deriv()
        sol = self._factor_and_solve(np.concatenate((b0, b1, b2t)))
        return sol[: self.n], sol[self.n:], self._rcond()


class AsymmetricStepSolver(_ScaledUnsymmetricStepSolver):
    """Full KKT matrix with the rows of the active variables overwritten in place by unit rows
    (asymmetric_step_solver.py:38-173)."""

    def _compute_deriv(self):
        n, m = self.n, self.m
        lamb, rho = 1.0 / self.dt, self.rho
        hess = sps.csr_matrix(self.hess) + sps.diags([lamb], shape=(n, n))
        jac = sps.csr_matrix(self.jac)
        lower = sps.diags([-lamb / (1.0 + lamb * rho)], shape=(m, m))
        full = sps.bmat([[hess, jac.T], [jac, lower]], format="lil") if m else sps.lil_matrix(hess)
        for j in np.where(self.active_set)[0]:
            full.rows[j] = [int(j)]
            full.data[j] = [1.0]
        self._deriv = sps.csr_matrix(full)

    def compute_rhs(self, b0, b1, b2t):
        rhs = np.empty(self.n + self.m)
        rhs[self.n:] = b2t
        rhs[: self.n][self.active_set] = b0
        rhs[: self.n][np.logical_not(self.active_set)] = b1
        return rhs

    def solve_scaled(self, b0, b1, b2t):
        if self._deriv is None:
            self._compute_deriv()
        rhs = self.compute_rhs(b0, b1, b2t)

        def initial_sol():  # starting point of the Krylov back-ends (:125-136); LU ignores it
            sol = np.zeros(self.n + self.m)
            sol[: self.n][self.active_set] = b0
            return sol

        sol = self._factor_and_solve(rhs, initial_sol=initial_sol)
        return sol[: self.n], sol[self.n:], self._rcond()


_BY_TYPE = {
    "Standard": StandardStepSolver,
    "Extended": ExtendedStepSolver,
    "Asymmetric": AsymmetricStepSolver,
    "Symmetric": HipStepSolver,
}


def step_solver(problem, params, iterate, dt, rho):
    """Reference factory (step/solver/__init__.py:12-31): the ``params.step_solver`` hook
    first, then ``params.step_solver_type``."""
    assert dt > 0.0 and rho > 0.0
    hook = getattr(params, "step_solver", None)
    if hook is not None:
        return hook(problem, params, iterate, dt, rho)
    return _BY_TYPE[enum_name(getattr(params, "step_solver_type", "Symmetric"))](
        problem, params, iterate, dt, rho)
