"""Semi-smooth Newton policies over a ``StepSolver``-shaped object, and the
device-resident driver for linear-quadratic problems.

``newton_method`` / ``SimplifiedNewtonMethod`` / ``FullNewtonMethod`` /
``ActiveSetNewtonMethod`` follow the reference state machine
(``pygradflow/newton.py:35-60, 63-89, 181-215``, factory ``:307-323``): which
iterate supplies (mask, derivatives) and when the KKT matrix is re-factorised.
They are host-side policy only; every number comes from the step solver.

``DeviceNewton`` runs the same policies with the point, H, J, q, b resident in
HBM: one ``pgf_qp_step`` per Newton step, no vectors crossing
PCIe.  This is what ``bench.py`` times and what the batched mode shards.
"""

from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from .errors import LinearSolverError, StepSolverError
from .params import enum_name
from .sparse import MAX_BANDWIDTH, BandPlan
from .step_solver import DENSE_LIMIT, POOL, HipStepSolver, residency_key, same_key


# --------------------------------------------------------------------------- policies
class NewtonMethod:
    def __init__(self, problem, orig_iterate, dt, rho, step_solver, tau=None):
        self.problem = problem
        self.orig_iterate = orig_iterate
        self.dt = dt
        self.rho = rho
        self.tau = tau
        self.step_solver = step_solver
        self.func = step_solver.func

    @property
    def params(self):
        return self.orig_iterate.params

    def step(self, iterate):
        raise NotImplementedError()


class SimplifiedNewtonMethod(NewtonMethod):
    """Mask and derivatives frozen at the outer iterate: one factorisation, then
    back-solves (reference newton.py:35-60)."""

    def __init__(self, problem, orig_iterate, dt, rho, step_solver, tau=None):
        super().__init__(problem, orig_iterate, dt, rho, step_solver, tau)
        mask = self.func.compute_active_set(orig_iterate, rho, tau)
        step_solver.update_active_set(mask)
        step_solver.update_derivs(orig_iterate)

    def step(self, iterate):
        return self.step_solver.solve(iterate)


class FullNewtonMethod(NewtonMethod):
    """Mask, derivatives and factorisation from the current iterate, every step
    (reference newton.py:63-89)."""

    def step(self, iterate):
        mask = self.func.compute_active_set(iterate, self.rho, self.tau)
        self.step_solver.update_active_set(mask)
        self.step_solver.update_derivs(iterate)
        return self.step_solver.solve(iterate)


class ActiveSetNewtonMethod(NewtonMethod):
    """Derivatives frozen at the outer iterate, mask from the current one; refactor
    only when the mask differs elementwise (reference newton.py:181-215)."""

    def __init__(self, problem, orig_iterate, dt, rho, step_solver, tau=None):
        super().__init__(problem, orig_iterate, dt, rho, step_solver, tau)
        step_solver.update_derivs(orig_iterate)
        self._curr_active_set = None

    def step(self, iterate):
        mask = self.func.compute_active_set(iterate, self.rho, self.tau)
        if self._curr_active_set is None or (self._curr_active_set != mask).any():
            self.step_solver.update_active_set(mask)
        self._curr_active_set = mask
        return self.step_solver.solve(iterate)


class GlobalizedNewtonMethod(NewtonMethod):
    """Newton step followed by an Armijo line search on 1/2 ||F||^2 (reference
    newton.py:218-304).  Host-side policy: residuals, masks and the linear solve come from
    the step solver; the generalised Jacobian for the directional derivative is assembled on
    the host (off the hot path, as in the reference's own TODO at :250-252)."""

    MAX_BACKTRACKS = 30

    def _set_iterate(self, iterate):
        self.step_solver.update_derivs(iterate)
        mask = self.func.compute_active_set(iterate, self.rho, self.tau)
        self.step_solver.update_active_set(mask)

    def step(self, iterate):
        from .step_solver import StepResult

        params = iterate.params
        n = self.problem.num_vars
        self._set_iterate(iterate)
        # the reference solves at the OUTER iterate here (newton.py:248); kept as is
        full = self.step_solver.solve(self.orig_iterate)
        value = self.func.value_at(iterate, self.rho)
        merit = 0.5 * np.dot(value, value)
        if merit <= params.newton_tol:
            return full
        grad = self.func.deriv_at(iterate, self.rho).T @ value
        slope = np.dot(grad[:n], full.dx) + np.dot(grad[n:], full.dy)
        alpha, dx, dy = 1.0, full.dx, full.dy
        for _ in range(self.MAX_BACKTRACKS):
            trial = type(iterate)(self.problem, params, iterate.x - dx, iterate.y - dy, iterate.eval)
            tv = self.func.value_at(trial, self.rho)
            tm = 0.5 * np.dot(tv, tv)
            if tm <= params.newton_tol or tm <= merit + 1e-4 * alpha * slope:
                break
            alpha *= 0.5
            dx, dy = alpha * full.dx, alpha * full.dy
        else:
            raise Exception("Line search failed to converge")
        result = StepResult(self.orig_iterate, dx, dy, active_set=None, rcond=None)
        result.active_set = self.func.compute_active_set(result.iterate, self.rho, self.tau)
        return result


def make_step_solver(problem, params, iterate, dt, rho):
    """Reference factory ``step_solver`` (step/solver/__init__.py:12-31): honours the
    ``params.step_solver`` hook, then ``params.step_solver_type``: Symmetric (default) is the
    HIP hot path, Standard / Extended / Asymmetric go through the GPU LU."""
    from .unsym_step_solvers import step_solver

    return step_solver(problem, params, iterate, dt, rho)


def newton_method(problem, params, iterate, dt, rho, tau=None):
    if not (dt > 0.0 and rho > 0.0):
        raise ValueError("dt and rho must be positive")
    solver = make_step_solver(problem, params, iterate, dt, rho)
    kind = enum_name(params.newton_type)
    if kind == "Simplified":
        return SimplifiedNewtonMethod(problem, iterate, dt, rho, solver, tau)
    if kind == "Full":
        return FullNewtonMethod(problem, iterate, dt, rho, solver, tau)
    if kind == "ActiveSet":
        return ActiveSetNewtonMethod(problem, iterate, dt, rho, solver, tau)
    if kind == "Globalized":
        return GlobalizedNewtonMethod(problem, iterate, dt, rho, solver, tau)
    raise ValueError(f"unknown newton_type {kind}")


def newton_steps(problem, params, orig_iterate, dt, rho, tau=None):
    """Generator of successive ``StepResult``s (``NewtonController.newton_steps``,
    reference step/newton_control.py:22-38, for a given tau)."""
    method = newton_method(problem, params, orig_iterate, dt, rho, tau)
    curr = orig_iterate
    while True:
        step = method.step(curr)
        yield step
        curr = step.iterate


# --------------------------------------------------------------------------- device-resident
_POLICY_BITS = {
    "Simplified": 0,
    "Full": _lib.STEP_RECOMPUTE_MASK | _lib.STEP_REFACTOR,
    "ActiveSet": _lib.STEP_RECOMPUTE_MASK | _lib.STEP_REFACTOR_ON_CHANGE,
}


class DeviceNewton:
    """Newton policies for a ``LinearQuadraticProblem`` held entirely in HBM.

    ``DeviceNewton(problem, newton_type, x_hat, y_hat, dt, rho, tau)`` mirrors
    ``newton_method(...)``; ``step()`` advances the device-resident point and returns
    ``(diff, n_neg)``; ``point()`` / ``mask()`` copy results out for checking.
    """

    def __init__(self, problem, newton_type, x_hat, y_hat, dt, rho, tau=None, device=0,
                 start=None):
        _lib.require_gpu()
        self._lib = _lib.load()
        self.problem = problem
        self.kind = enum_name(newton_type) if not isinstance(newton_type, str) else newton_type
        if self.kind not in _POLICY_BITS:
            raise NotImplementedError(self.kind)
        self.n, self.m = problem.num_vars, problem.num_cons
        self.dt, self.rho = float(dt), float(rho)
        self.tau = math.nan if tau is None else float(tau)
        self.sparse = (not problem.is_dense) and (
            bool(getattr(problem, "pgf_force_band", False)) or self.n + self.m > DENSE_LIMIT)
        self._hd = POOL.acquire(self.n, self.m, device, sparse=self.sparse)
        h = self._hd.h
        lib = self._lib
        lb, ub = _lib.as_f64(problem.var_lb), _lib.as_f64(problem.var_ub)
        _lib.check(lib.pgf_set_bounds(h, _lib.dptr(lb), _lib.dptr(ub)), h, "pgf_set_bounds")
        key = residency_key(problem)
        stale = not same_key(key, self._hd.derivs_key)
        if self.sparse and (stale or not getattr(self._hd, "qp_loaded", False)):
            plan = BandPlan(problem.hess_sparse(), problem.jac_sparse(), self.n, self.m)
            if not plan.supported:
                raise NotImplementedError(
                    f"banded path: half-bandwidth {plan.bw} > {MAX_BANDWIDTH}; use HipStepSolver "
                    "(plugin path), which falls back to the dense factorisation")
            plan.upload(lib, h)
            hv, jv = plan.values(problem.hess_sparse(), problem.jac_sparse())
            _lib.check(lib.pgf_sparse_set_values(h, _lib.dptr(hv), _lib.dptr(jv)), h,
                       "pgf_sparse_set_values")
            q, b = _lib.as_f64(problem.q), _lib.as_f64(problem.b)
            _lib.check(lib.pgf_qp_set_vectors(h, _lib.dptr(q), _lib.dptr(b)), h, "pgf_qp_set_vectors")
            self._hd.plan = plan
            self._hd.derivs_key = key
            self._hd.qp_loaded = True
        elif stale or not getattr(self._hd, "qp_loaded", False):
            Q = np.ascontiguousarray(problem.hess_dense(), dtype=np.float64)
            A = np.ascontiguousarray(problem.jac_dense(), dtype=np.float64).reshape(self.m, self.n)
            q, b = _lib.as_f64(problem.q), _lib.as_f64(problem.b)
            rc = lib.pgf_qp_set_problem(
                h, Q.ctypes.data_as(C.c_void_p), max(self.n, 1), q.ctypes.data_as(C.c_void_p),
                A.ctypes.data_as(C.c_void_p), max(self.n, 1), b.ctypes.data_as(C.c_void_p),
                _lib.PGF_HOST)
            _lib.check(rc, h, "pgf_qp_set_problem")
            self._hd.derivs_key = key
            self._hd.qp_loaded = True
        self.set_outer(x_hat, y_hat, dt, rho, start=start)

    def set_outer(self, x_hat, y_hat, dt, rho, start=None):
        """Begin a new outer step at (x_hat, y_hat) (a new NewtonMethod in the reference)."""
        h, lib = self._hd.h, self._lib
        self.dt, self.rho = float(dt), float(rho)
        xh, yh = _lib.as_f64(x_hat), _lib.as_f64(y_hat)
        _lib.check(lib.pgf_set_outer(h, _lib.dptr(xh), _lib.dptr(yh), self.dt, self.rho), h,
                   "pgf_set_outer")
        _lib.check(lib.pgf_qp_set_point(h, _lib.dptr(xh), _lib.dptr(yh)), h, "pgf_qp_set_point")
        if self.kind == "Simplified":
            ch = C.c_int(0)
            _lib.check(lib.pgf_qp_update_active_set(h, self.tau, C.byref(ch)), h,
                       "pgf_qp_update_active_set")
        if start is not None:
            xs, ys = _lib.as_f64(start[0]), _lib.as_f64(start[1])
            _lib.check(lib.pgf_qp_set_point(h, _lib.dptr(xs), _lib.dptr(ys)), h, "pgf_qp_set_point")

    def advance_outer(self, dt=None, rho=None):
        """(x_hat, y_hat) <- current device point; starts the next outer step on device."""
        h, lib = self._hd.h, self._lib
        self.dt = self.dt if dt is None else float(dt)
        self.rho = self.rho if rho is None else float(rho)
        _lib.check(lib.pgf_qp_advance_outer(h, self.dt, self.rho), h, "pgf_qp_advance_outer")
        if self.kind == "Simplified":
            ch = C.c_int(0)
            _lib.check(lib.pgf_qp_update_active_set(h, self.tau, C.byref(ch)), h,
                       "pgf_qp_update_active_set")

    def step(self, inertia_check=False):
        n_neg, diff = C.c_int(0), C.c_double(0.0)
        try:
            rc = self._lib.pgf_qp_step(self._hd.h, _POLICY_BITS[self.kind], self.tau,
                                       int(inertia_check), C.byref(n_neg), C.byref(diff))
            _lib.check(rc, self._hd.h, "pgf_qp_step")
        except LinearSolverError as e:
            raise StepSolverError(str(e)) from e
        return diff.value, n_neg.value

    def step_async(self):
        _lib.check(self._lib.pgf_qp_step_async(self._hd.h, _POLICY_BITS[self.kind], self.tau),
                   self._hd.h, "pgf_qp_step_async")

    def sync(self):
        n_neg, diff = C.c_int(0), C.c_double(0.0)
        _lib.check(self._lib.pgf_qp_sync(self._hd.h, C.byref(n_neg), C.byref(diff)), self._hd.h,
                   "pgf_qp_sync")
        return diff.value, n_neg.value

    def point(self):
        x, y = np.empty(self.n), np.empty(self.m)
        _lib.check(self._lib.pgf_qp_get_point(self._hd.h, _lib.dptr(x), _lib.dptr(y)), self._hd.h)
        return x, y

    def set_point(self, x, y):
        xs, ys = _lib.as_f64(x), _lib.as_f64(y)
        _lib.check(self._lib.pgf_qp_set_point(self._hd.h, _lib.dptr(xs), _lib.dptr(ys)), self._hd.h)

    def mask(self):
        mk = np.empty(self.n, dtype=np.bool_)
        _lib.check(self._lib.pgf_qp_get_mask(self._hd.h, _lib.u8ptr(mk)), self._hd.h)
        return mk

    def residual_norm(self, out_dev_ptr=None):
        """||F(z)||_2 of the unscaled residual at the device point; optionally also
        stored to a device address (the rank-local slot of the RCCL all-gather)."""
        out = C.c_double(0.0)
        _lib.check(self._lib.pgf_qp_residual_norm(self._hd.h, C.byref(out),
                                                  C.c_void_p(out_dev_ptr) if out_dev_ptr else None),
                   self._hd.h, "pgf_qp_residual_norm")
        return out.value

    def measures(self, active_tol=1e-8):
        """Termination measures of the device point: dict(stat_res, cons_violation,
        bound_violation, y_inf) -- ``Iterate.stat_res`` etc. without moving the point."""
        out = np.empty(4)
        _lib.check(self._lib.pgf_qp_measures(self._hd.h, float(active_tol), _lib.dptr(out)),
                   self._hd.h, "pgf_qp_measures")
        return dict(stat_res=out[0], cons_violation=out[1], bound_violation=out[2], y_inf=out[3])

    def factor_kind(self):
        """0 none, 1 LDL^T in the natural order, 2 LDL^T of the condensed system, 3 pivoted LU
        (``pgf_debug_factor_kind``)."""
        return int(self._lib.pgf_debug_factor_kind(self._hd.h))

    def refinement_stats(self):
        """(refinement steps, LU fallbacks, relative residual of the last checked solve) of the
        residual guard (``pgf_refinement_stats``)."""
        a, b, r = C.c_int(0), C.c_int(0), C.c_double(0.0)
        _lib.check(self._lib.pgf_refinement_stats(self._hd.h, C.byref(a), C.byref(b), C.byref(r)),
                   self._hd.h)
        return a.value, b.value, r.value

    def profile(self, on=True):
        """on = True / 1: the factorisation's kernels as separate launches with a HIP-event span
        each (per-kernel figures); 2: spans around the PRODUCTION launches (``fused_*``:
        k_chain_update, ``trsmud_ms``: k_trsm_ud); False: off."""
        _lib.check(self._lib.pgf_profile_enable(self._hd.h, int(on)), self._hd.h)

    def profile_read(self):
        out = np.zeros(14)
        _lib.check(self._lib.pgf_profile_read_ex(self._hd.h, _lib.dptr(out), out.size), self._hd.h)
        keys = ("update_ms", "update_launches", "update_flops", "update_bytes", "factor_ms",
                "chain_ms", "chain_launches", "trsm_ms", "udiag_ms", "fused_ms", "fused_launches",
                "fused_flops", "fused_bytes", "trsmud_ms")
        rec = dict(zip(keys, (float(v) for v in out)))
        for k in ("update_launches", "chain_launches", "fused_launches"):
            rec[k] = int(rec[k])
        return rec

    def close(self):
        if getattr(self, "_hd", None) is not None:
            POOL.release(self._hd)
            self._hd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
