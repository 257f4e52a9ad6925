"""CPU ORACLE for the semi-smooth Newton / KKT hot path -- TEST INFRASTRUCTURE.

This file is a numpy/scipy *restatement* of the reference algorithm
(chrhansk/pygradflow @ v0.5.24).  It is NOT part of the product: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the product path (``pygradflow_amd``) never does
and fails loudly when the HIP library is missing.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the real reference
(pure Python, importable in the build container) and dumps per-step inputs
and outputs into ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors (masks bit-exact, floats
to <= 1e-13 relative).

The factor/solve arithmetic of the reference lives in a third-party
dependency: SuperLU via ``scipy.sparse.linalg.splu`` (``scipy>=1.14`` in the
reference's ``pyproject.toml:13``, no lock file; 1.15.3 installed here),
called at ``pygradflow/linear_solver/lu_solver.py:14,21``.  ``factor_kkt``
below makes the same call on the same ``bmat``-assembled CSC matrix, so the
``"port"`` CPU baseline costs what the reference costs.

Every function cites the reference lines it follows (paths relative to the
reference checkout).
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

ACTIVE_EPS = 1e-8  # implicit_func.py:44


# ---------------------------------------------------------------- a1
def aug_lag_deriv_x(obj_grad, jac, cons, y, rho):
    """g = grad f + J'(rho c + y)   (iterate.py:91-94)."""
    lhs = rho * cons + y
    return obj_grad + jac.T.dot(lhs)


# ---------------------------------------------------------------- a2
def scaled_bounds(lamb, var_lb, var_ub, dtype=np.float64):
    """lamb*lb, lamb*ub kept for one outer step (implicit_func.py:211-216)."""
    return (lamb * var_lb).astype(dtype), (lamb * var_ub).astype(dtype)


# ---------------------------------------------------------------- a3
def projection_initial(dt, x_hat, x, g, tau=None):
    """p (implicit_func.py:233-246); ``tau`` only ever feeds the mask."""
    lamb = 1.0 / dt
    if tau is not None:
        f_x = lamb * (1 - tau * lamb)
        f_x0 = tau * lamb * lamb
        f_d = tau * lamb
        return f_x * x + f_x0 * x_hat - f_d * g
    return lamb * x_hat - g


# ---------------------------------------------------------------- a4
def active_set_box(p, slb, sub):
    """mask = p < lb-1e-8  or  p > ub+1e-8   (implicit_func.py:21-44)."""
    return np.logical_or(p < slb - ACTIVE_EPS, p > sub + ACTIVE_EPS)


# ---------------------------------------------------------------- a5
def project_box(p, slb, sub, mask):
    """clip only the masked entries (implicit_func.py:46-60)."""
    out = np.copy(p)
    out[mask] = np.clip(p[mask], slb[mask], sub[mask])
    return out


# ---------------------------------------------------------------- a6
def residual(dt, x_hat, y_hat, x, y, g, cons, slb, sub, mask):
    """Scaled residual F = [lamb x - P(p) ; lamb y_hat + c - lamb y]
    (implicit_func.py:219-231).  ``mask`` is given, never recomputed here."""
    lamb = 1.0 / dt
    p = projection_initial(dt, x_hat, x, g)
    xval = lamb * x - project_box(p, slb, sub, mask)
    yval = -(lamb * y - (lamb * y_hat + cons))
    return np.concatenate([xval, yval])


def unscaled_residual(dt, x_hat, y_hat, x, y, g, cons, var_lb, var_ub):
    """ImplicitFunc.value_at with its own mask (implicit_func.py:131-161); the
    controllers' convergence measure ||F||."""
    p = x_hat - dt * g
    mask = active_set_box(p, var_lb, var_ub)
    xval = x - project_box(p, var_lb, var_ub, mask)
    yval = y - (y_hat + dt * cons)
    return np.concatenate([xval, yval])


# ---------------------------------------------------------------- a8
def initial_rhs(dt, F, mask, n):
    """b0 = dt F_x[A], b1 = F_x[I], b2 = F_y  (scaled_step_solver.py:38-60)."""
    rx, ry = F[:n], F[n:]
    act = np.where(mask)[0]
    inact = np.where(np.logical_not(mask))[0]
    return dt * rx[act], rx[inact], ry


# ---------------------------------------------------------------- a10-a12
def shifted_hess_rows(hess, lamb, mask):
    """(H + lamb I)[I, :] as CSC (symmetric_step_solver.py:27-39)."""
    n = hess.shape[0]
    inact = np.where(np.logical_not(mask))[0]
    hl = hess + sps.diags([lamb], shape=(n, n), dtype=np.float64)
    return hl.tocsr()[inact, :].tocsc()


def reduced_rhs(hess_rows, jac_csc, mask, b0, b1, b2t):
    """rhs = [b1 - Hl[I,A] b0 ; b2t - J[:,A] b0]  (symmetric_step_solver.py:79-94)."""
    act = np.where(mask)[0]
    return np.concatenate((b1 - hess_rows[:, act] @ b0, b2t - jac_csc[:, act] @ b0))


def kkt_matrix(hess_rows, jac_csc, mask, lamb, rho):
    """K = [[Hl[I,I], J[:,I]'],[J[:,I], -lamb/(1+lamb rho) I]], CSC
    (symmetric_step_solver.py:49-77)."""
    m = jac_csc.shape[0]
    inact = np.where(np.logical_not(mask))[0]
    lower = sps.diags([-lamb / (1.0 + lamb * rho)], shape=(m, m), dtype=np.float64)
    ij = jac_csc[:, inact]
    return sps.bmat([[hess_rows[:, inact], ij.T], [ij, lower]], format="csc")


# ---------------------------------------------------------------- a13-a14
class FactorError(Exception):
    """Stands for LinearSolverError (linear_solver.py:8-15)."""


def factor_kkt(K):
    """SuperLU, COLAMD + partial pivoting, scipy defaults (lu_solver.py:9-17)."""
    try:
        return spla.splu(K)
    except RuntimeError as err:  # singular
        raise FactorError(str(err))


def num_neg_eigvals_dense(K):
    """Inertia by dense symmetric eigenvalues (oracle only; the reference's LU
    back-end reports None, linear_solver.py:27-28)."""
    if K.shape[0] == 0:
        return 0
    return int((np.linalg.eigvalsh(K.toarray() if sps.issparse(K) else K) < 0).sum())


# ---------------------------------------------------------------- a9, a15
def scatter_step(n, mask, s, b0):
    """dx[I] = s[:|I|], dx[A] = b0, dy' = s[|I|:]  (symmetric_step_solver.py:115-121)."""
    act = np.where(mask)[0]
    inact = np.where(np.logical_not(mask))[0]
    dx = np.zeros((n,), dtype=np.float64)
    dx[inact] = s[: inact.size]
    dx[act] = b0
    return dx, s[inact.size :]


# ---------------------------------------------------------------- a16
def clip_step(x, y, dx, dy, var_lb, var_ub):
    """xn = clip(x-dx), dx rewritten where clipped, yn = y-dy, diff
    (step_solver.py:16-63, util.py:19-24)."""
    xn = x - dx
    dx = np.copy(dx)
    at_lb = xn < var_lb
    xn[at_lb] = var_lb[at_lb]
    dx[at_lb] = x[at_lb] - var_lb[at_lb]
    at_ub = xn > var_ub
    xn[at_ub] = var_ub[at_ub]
    dx[at_ub] = x[at_ub] - var_ub[at_ub]
    yn = y - dy
    diff = np.sqrt(np.dot(dx, dx) + np.dot(dy, dy))
    return xn, yn, dx, diff


# ---------------------------------------------------------------- point data
class PointData:
    """What the path consumes at a point: g inputs, c, J, H(x, y) (rho=0)."""

    def __init__(self, problem, x, y):
        self.x = np.asarray(x, dtype=np.float64)
        self.y = np.asarray(y, dtype=np.float64)
        n, m = problem.num_vars, problem.num_cons
        self.obj_grad = np.asarray(problem.obj_grad(self.x), dtype=np.float64)
        if m > 0:
            self.cons = np.asarray(problem.cons(self.x), dtype=np.float64)
            self.jac = sps.csr_matrix(problem.cons_jac(self.x))
        else:
            self.cons = np.zeros((0,))
            self.jac = sps.csr_matrix((0, n), dtype=np.float64)
        self._problem = problem
        self._hess = None

    def g(self, rho):
        return aug_lag_deriv_x(self.obj_grad, self.jac, self.cons, self.y, rho)

    @property
    def hess(self):
        # aug_lag_deriv_xx(rho=0.0): lag_hess(x, y), no rho J'J term
        # (iterate.py:102-110; scaled_step_solver.py:78)
        if self._hess is None:
            self._hess = sps.csr_matrix(self._problem.lag_hess(self.x, self.y))
        return self._hess


# ---------------------------------------------------------------- a7, a9, a15
class SymmetricStep:
    """State of one ``SymmetricStepSolver`` (symmetric_step_solver.py:13-164,
    scaled_step_solver.py:15-107) without the redundant recomputation."""

    def __init__(self, problem, x_hat, y_hat, dt, rho):
        self.problem = problem
        self.n, self.m = problem.num_vars, problem.num_cons
        self.x_hat = np.asarray(x_hat, dtype=np.float64)
        self.y_hat = np.asarray(y_hat, dtype=np.float64)
        self.dt, self.rho = dt, rho
        self.lamb = 1.0 / dt
        self.slb, self.sub = scaled_bounds(self.lamb, problem.var_lb, problem.var_ub)
        self.mask = None
        self.hess = None
        self.jac = None
        self._hess_rows = None
        self._K = None
        self._lu = None
        self.record = {}

    def compute_active_set(self, point: PointData, tau=None):
        p = projection_initial(self.dt, self.x_hat, point.x, point.g(self.rho), tau)
        return active_set_box(p, self.slb, self.sub)

    def update_active_set(self, mask):
        self.mask = np.copy(mask)
        self._hess_rows = self._K = self._lu = None

    def update_derivs(self, point: PointData):
        self.jac = point.jac.tocsc()
        self.hess = point.hess
        self._hess_rows = self._K = self._lu = None

    def solve(self, point: PointData):
        n, mask = self.n, self.mask
        lamb, rho, dt = self.lamb, self.rho, self.dt
        g = point.g(rho)
        F = residual(dt, self.x_hat, self.y_hat, point.x, point.y, g, point.cons,
                     self.slb, self.sub, mask)
        b0, b1, b2 = initial_rhs(dt, F, mask, n)
        fact = 1.0 / (1.0 + lamb * rho)
        b2t = fact * b2
        if self._hess_rows is None:
            self._hess_rows = shifted_hess_rows(self.hess, lamb, mask)
        rhs = reduced_rhs(self._hess_rows, self.jac, mask, b0, b1, b2t)
        if self._K is None:
            self._K = kkt_matrix(self._hess_rows, self.jac, mask, lamb, rho)
        if self._lu is None:
            self._lu = factor_kkt(self._K)
        s = self._lu.solve(rhs)
        dx, dyp = scatter_step(n, mask, s, b0)
        dy = fact * (dyp - rho * b2)
        xn, yn, dxc, diff = clip_step(point.x, point.y, dx, dy,
                                      self.problem.var_lb, self.problem.var_ub)
        self.record = dict(g=g, F=F, rhs=rhs, s=s, dx=dxc, dy=dy, xn=xn, yn=yn,
                           diff=diff, mask=np.copy(mask), K=self._K)
        return xn, yn, diff


# ---------------------------------------------------------------- a17
class NewtonOracle:
    """Policy state machine of ``newton_method`` (newton.py:35-60 Simplified,
    :63-89 Full, :181-215 ActiveSet, factory :307-323) driving a
    ``SymmetricStep``; equals ``NewtonController.newton_steps``
    (newton_control.py:22-38) when ``step`` is called repeatedly on its own
    output."""

    def __init__(self, problem, newton_type, x_hat, y_hat, dt, rho, tau=None):
        self.problem = problem
        self.kind = newton_type
        self.rho, self.tau = rho, tau
        self.solver = SymmetricStep(problem, x_hat, y_hat, dt, rho)
        self._curr_mask = None
        orig = PointData(problem, x_hat, y_hat)
        if newton_type == "Simplified":
            self.solver.update_active_set(self.solver.compute_active_set(orig, tau))
            self.solver.update_derivs(orig)
        elif newton_type == "ActiveSet":
            self.solver.update_derivs(orig)
        elif newton_type != "Full":
            raise ValueError(newton_type)

    def step(self, x, y):
        pt = PointData(self.problem, x, y)
        sv = self.solver
        if self.kind == "Full":
            sv.update_active_set(sv.compute_active_set(pt, self.tau))
            sv.update_derivs(pt)
        elif self.kind == "ActiveSet":
            mask = sv.compute_active_set(pt, self.tau)
            if self._curr_mask is None or (self._curr_mask != mask).any():
                sv.update_active_set(mask)
            self._curr_mask = mask
        return sv.solve(pt)

    def run(self, x0, y0, k):
        """k successive steps; returns the list of per-step records."""
        out = []
        x, y = np.asarray(x0, dtype=np.float64), np.asarray(y0, dtype=np.float64)
        for _ in range(k):
            x, y, _ = self.step(x, y)
            out.append(dict(self.solver.record))
        return out
