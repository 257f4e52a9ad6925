"""Point ``(x, y)`` with lazily evaluated problem data.

A stand-in for the reference's ``Iterate`` (``pygradflow/iterate.py:19-110``)
restricted to what the Newton/KKT path consumes: ``x, y, problem, params,
eval`` and the cached ``obj_grad / cons / cons_jac / lag_hess`` plus the
derived ``aug_lag_deriv_*``.  The step solver never *requires* this class: it
accepts any object with this surface (in particular the reference's own
``Iterate``) and builds the next point with ``type(iterate)(...)``.
"""

from __future__ import annotations

import functools

import numpy as np
import scipy.sparse as sps


def _frozen(a: np.ndarray) -> np.ndarray:
    a.flags.writeable = False
    return a


class DirectEvaluator:
    """Calls the problem callbacks, casting to ``params.dtype``
    (cf. ``SimpleEvaluator``, reference ``pygradflow/eval.py:107-127``)."""

    def __init__(self, problem, params):
        self.problem = problem
        self.dtype = params.dtype

    def obj(self, x):
        return self.problem.obj(x)

    def obj_grad(self, x):
        return np.asarray(self.problem.obj_grad(x), dtype=self.dtype)

    def cons(self, x):
        if self.problem.num_cons == 0:
            return np.zeros((0,), dtype=self.dtype)
        return np.asarray(self.problem.cons(x), dtype=self.dtype)

    def cons_jac(self, x):
        if self.problem.num_cons == 0:
            return sps.csr_matrix((0, self.problem.num_vars), dtype=self.dtype)
        return self.problem.cons_jac(x)

    def lag_hess(self, x, y):
        return self.problem.lag_hess(x, y)


class Iterate:
    def __init__(self, problem, params, x, y, eval=None):
        if x.shape != (problem.num_vars,) or y.shape != (problem.num_cons,):
            raise ValueError("iterate shape mismatch")
        self.problem = problem
        self.params = params
        self.x = _frozen(np.array(x, copy=True))
        self.y = _frozen(np.array(y, copy=True))
        self.eval = eval if eval is not None else DirectEvaluator(problem, params)

    @property
    def z(self):
        return np.concatenate((self.x, self.y))

    @functools.cached_property
    def obj(self):
        return self.eval.obj(self.x)

    @functools.cached_property
    def obj_grad(self):
        return _frozen(np.array(self.eval.obj_grad(self.x)))

    @functools.cached_property
    def cons(self):
        return _frozen(np.array(self.eval.cons(self.x)))

    @functools.cached_property
    def cons_jac(self):
        return self.eval.cons_jac(self.x)

    def lag_hess(self, y):
        return self.eval.lag_hess(self.x, y)

    # derived quantities, reference iterate.py:91-110 ----------------------
    def aug_lag_deriv_x(self, rho):
        return self.obj_grad + self.cons_jac.T.dot(rho * self.cons + self.y)

    def aug_lag_deriv_y(self):
        return self.cons

    def aug_lag_deriv_xy(self):
        return self.cons_jac

    def aug_lag_deriv_xx(self, rho):
        mult = self.y + rho * self.cons
        hess = self.lag_hess(mult)
        if rho == 0.0:
            return hess
        jac = self.cons_jac
        return hess + rho * (jac.T @ jac)

    def dist(self, other):
        return float(np.sqrt(np.sum((self.x - other.x) ** 2) + np.sum((self.y - other.y) ** 2)))

    # termination measures, reference iterate.py:136-181 -------------------
    @functools.cached_property
    def active_set(self):
        return ActiveSet(self)

    @functools.cached_property
    def bounds_dual(self):
        """Multiplier estimate of the bounds: the part of -(grad f + J'y) that a bound can
        carry (non-negative at an upper bound, non-positive at a lower one, all of it where
        both coincide)."""
        r = -(self.obj_grad + self.cons_jac.T.dot(self.y))
        act = self.active_set
        d = np.zeros_like(self.x)
        d[act.at_upper] = np.maximum(r[act.at_upper], 0.0)
        d[act.at_lower] = np.minimum(r[act.at_lower], 0.0)
        d[act.at_both] = r[act.at_both]
        return d

    @functools.cached_property
    def bound_violation(self):
        lb, ub = self.problem.var_lb, self.problem.var_ub
        below = float(np.linalg.norm(np.maximum(lb - self.x, 0.0), np.inf))
        above = float(np.linalg.norm(np.maximum(self.x - ub, 0.0), np.inf))
        return max(below, above)

    @functools.cached_property
    def cons_violation(self):
        c = self.cons
        return float(np.linalg.norm(c, np.inf)) if c.size else 0.0

    @functools.cached_property
    def stat_res(self):
        r = self.obj_grad + self.cons_jac.T.dot(self.y) + self.bounds_dual
        return float(np.linalg.norm(r, np.inf))

    def is_feasible(self, tol):
        return self.cons_violation <= tol and self.bound_violation <= tol

    @property
    def total_res(self):
        return max(self.cons_violation, self.bound_violation, self.stat_res)

    def check_eval(self):
        self.obj
        self.obj_grad
        if self.problem.num_cons > 0:
            self.cons
            self.cons_jac


class ActiveSet:
    """Which variables sit at (or beyond) a bound, to ``params.active_tol``
    (reference active_set.py:4-29).  ``at_lower`` / ``at_upper`` exclude ``at_both``."""

    def __init__(self, iterate):
        tol = getattr(iterate.params, "active_tol", 1e-8)
        lb, ub, x = iterate.problem.var_lb, iterate.problem.var_ub, iterate.x
        near_lb = np.absolute(x - lb) <= tol
        near_ub = np.absolute(ub - x) <= tol
        self.violated = np.logical_or(lb - x > tol, x - ub > tol)
        self.at_either = np.logical_or(near_lb, near_ub)
        self.at_both = np.logical_and(near_lb, near_ub)
        self.at_lower = np.logical_and(near_lb, np.logical_not(self.at_both))
        self.at_upper = np.logical_and(near_ub, np.logical_not(self.at_both))

    @property
    def satisfied(self):
        return np.logical_not(self.violated)
