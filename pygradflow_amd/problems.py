"""Problem shapes and synthetic generators for the Newton/KKT hot path.

The classes here duck-type the reference's ``Problem`` surface
(``pygradflow/problem.py:8-192``: ``var_lb, var_ub, num_vars, num_cons,
cons_lb, cons_ub, obj, obj_grad, cons, cons_jac, lag_hess``) so that one
object can be handed both to the reference (when it is importable, to
generate golden vectors) and to this package.  Nothing is imported from the
reference.

Generators follow SURVEY.md section 8(d) (BASELINE.json ``configs``).

``LinearQuadraticProblem`` additionally advertises *constant derivatives*
(``pgf_constant_derivs``): its Hessian and Jacobian do not depend on the point,
so a device step solver may upload them once and keep them resident in HBM.
"""

from __future__ import annotations

import numpy as np
import scipy.sparse as sps


class _ProblemShape:
    """Common bound bookkeeping (mirrors ``Problem.__init__``,
    reference ``pygradflow/problem.py:32-96``, equality constraints only)."""

    def __init__(self, var_lb, var_ub, num_cons):
        var_lb = np.asarray(var_lb, dtype=np.float64)
        var_ub = np.asarray(var_ub, dtype=np.float64)
        if var_lb.shape != var_ub.shape or var_lb.ndim != 1:
            raise ValueError("bounds must be 1-d arrays of equal shape")
        if not (var_lb <= var_ub).all():
            raise ValueError("var_lb <= var_ub violated")
        if not (var_lb < np.inf).all() or not (var_ub > -np.inf).all():
            raise ValueError("var_lb must be < inf and var_ub > -inf")
        self.var_lb = var_lb.copy()
        self.var_ub = var_ub.copy()
        self.num_cons = int(num_cons)
        self.cons_lb = np.zeros((self.num_cons,))
        self.cons_ub = np.zeros((self.num_cons,))

    @property
    def num_vars(self) -> int:
        return self.var_lb.shape[0]

    @property
    def var_bounded(self) -> bool:
        return bool(np.isfinite(self.var_lb).any() or np.isfinite(self.var_ub).any())


class LinearQuadraticProblem(_ProblemShape):
    """``min 1/2 x'Qx + q'x  s.t.  Ax - b = 0,  lb <= x <= ub``.

    ``Q`` (n x n, symmetric) and ``A`` (m x n) may be dense ``ndarray`` or
    scipy sparse; callbacks return scipy sparse matrices because the reference
    step solver calls ``.tocsc()`` on them
    (``symmetric_step_solver.py:41-43``).
    """

    pgf_constant_derivs = True

    def __init__(self, Q, q, A, b, lb, ub):
        n = q.shape[0]
        m = b.shape[0]
        super().__init__(lb, ub, m)
        assert Q.shape == (n, n)
        assert A.shape == (m, n)
        self._pgf_version = 0
        self.Q = Q
        self.q = np.asarray(q, dtype=np.float64)
        self.A = A
        self.b = np.asarray(b, dtype=np.float64)

    # Q, A, q, b stay resident in HBM across Newton and outer steps (``pgf_constant_derivs``),
    # keyed on (problem object, version): ASSIGNING a new array bumps the version and the data are
    # uploaded again; the arrays themselves are frozen (``writeable = False``), so that an
    # in-place edit -- which no key could notice short of hashing 168 MB per step -- raises
    # instead of silently leaving a stale copy on the device.
    @staticmethod
    def _freeze(arr):
        if sps.issparse(arr):
            for part in ("data", "indices", "indptr"):
                a = getattr(arr, part, None)
                if isinstance(a, np.ndarray):
                    a.flags.writeable = False
        elif isinstance(arr, np.ndarray):
            arr.flags.writeable = False
        return arr

    def _set(self, name, value):
        setattr(self, "_" + name, self._freeze(value))
        self._pgf_version += 1
        self._Qs = self._As = None

    Q = property(lambda self: self._Q, lambda self, v: self._set("Q", v))
    A = property(lambda self: self._A, lambda self, v: self._set("A", v))
    q = property(lambda self: self._q, lambda self, v: self._set("q", np.asarray(v, dtype=np.float64)))
    b = property(lambda self: self._b, lambda self, v: self._set("b", np.asarray(v, dtype=np.float64)))

    # dense / sparse agnostic helpers -------------------------------------
    @property
    def is_dense(self) -> bool:
        return isinstance(self.Q, np.ndarray)

    def hess_dense(self) -> np.ndarray:
        return self.Q if isinstance(self.Q, np.ndarray) else self.Q.toarray()

    def jac_dense(self) -> np.ndarray:
        return self.A if isinstance(self.A, np.ndarray) else self.A.toarray()

    def hess_sparse(self):
        if self._Qs is None:
            self._Qs = sps.csr_matrix(self.Q)
        return self._Qs

    def jac_sparse(self):
        if self._As is None:
            self._As = sps.csr_matrix(self.A) if self.num_cons > 0 else sps.csr_matrix(
                (0, self.num_vars), dtype=np.float64
            )
        return self._As

    # Problem surface -----------------------------------------------------
    def obj(self, x):
        return float(0.5 * x @ (self.Q @ x) + self.q @ x)

    def obj_grad(self, x):
        return np.asarray(self.Q @ x).ravel() + self.q

    def cons(self, x):
        if self.num_cons == 0:
            return np.zeros((0,))
        return np.asarray(self.A @ x).ravel() - self.b

    def cons_jac(self, x):
        return self.jac_sparse()

    def lag_hess(self, x, y):
        return self.hess_sparse()


class QuarticProblem(_ProblemShape):
    """A small smooth *nonlinear, nonconvex* NLP used for policy parity:

    ``f(x) = 1/2 x'Qx + q'x + sum_j a_j x_j^4 / 4``
    ``c(x) = A x + B (x * x) - b``

    so that the Hessian of the Lagrangian depends on both ``x`` and ``y``:
    ``H(x, y) = Q + diag(3 a x^2) + 2 diag(B' y)``.
    """

    pgf_constant_derivs = False

    def __init__(self, Q, q, a, A, B, b, lb, ub):
        super().__init__(lb, ub, b.shape[0])
        self.Q, self.q, self.a, self.A, self.B, self.b = Q, q, a, A, B, b

    def obj(self, x):
        return float(0.5 * x @ self.Q @ x + self.q @ x + 0.25 * np.sum(self.a * x**4))

    def obj_grad(self, x):
        return self.Q @ x + self.q + self.a * x**3

    def cons(self, x):
        return self.A @ x + self.B @ (x * x) - self.b

    def cons_jac(self, x):
        return sps.csr_matrix(self.A + 2.0 * self.B * x[None, :])

    def lag_hess(self, x, y):
        d = 3.0 * self.a * x * x + 2.0 * (self.B.T @ y)
        return sps.csr_matrix(self.Q + np.diag(d))


# --------------------------------------------------------------------------
# Synthetic generators, SURVEY.md 8(d)
# --------------------------------------------------------------------------


class RosenbrockProblem(_ProblemShape):
    """f(x) = (a - x_0)^2 + b (x_1 - x_0^2)^2, no constraints, no bounds: BASELINE config 1's
    problem (the reference's docs/rosenbrock.py), written out from the formula."""

    pgf_constant_derivs = False

    def __init__(self, a=1.0, b=100.0):
        super().__init__(np.full(2, -np.inf), np.full(2, np.inf), 0)
        self.a, self.b = float(a), float(b)

    def obj(self, x):
        return (self.a - x[0]) ** 2 + self.b * (x[1] - x[0] ** 2) ** 2

    def obj_grad(self, x):
        r = x[1] - x[0] ** 2
        return np.array([-2.0 * (self.a - x[0]) - 4.0 * self.b * x[0] * r, 2.0 * self.b * r])

    def cons(self, x):
        return np.zeros(0)

    def cons_jac(self, x):
        return sps.csr_matrix((0, 2))

    def lag_hess(self, x, y):
        r = x[1] - x[0] ** 2
        h00 = 2.0 - 4.0 * self.b * r + 8.0 * self.b * x[0] ** 2
        h01 = -4.0 * self.b * x[0]
        return sps.csr_matrix(np.array([[h00, h01], [h01, 2.0 * self.b]]))


def dense_qp(n=4096, m=1024, seed=0, boxed_frac=0.0, box=0.01) -> LinearQuadraticProblem:
    """BASELINE config 2 (and, with seeds 0..255 and n=1024/m=256, config 4):
    ``G~N(0,1)/sqrt(n)``, ``Q=GG'+I``, ``A~N(0,1)/sqrt(n)``; bounds +-inf.
    ``boxed_frac>0`` gives variant 2b (that share of variables boxed at +-box)."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n)) / np.sqrt(n)
    Q = G @ G.T + np.eye(n)
    Q = 0.5 * (Q + Q.T)
    q = rng.standard_normal(n)
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    b = rng.standard_normal(m)
    lb = np.full(n, -np.inf)
    ub = np.full(n, np.inf)
    if boxed_frac > 0.0:
        k = int(round(boxed_frac * n))
        idx = rng.permutation(n)[:k]
        lb[idx] = -box
        ub[idx] = box
    return LinearQuadraticProblem(Q, q, A, b, lb, ub)


def illcond_qp(n=200, m=56, seed=4, lo=-5.0, hi=4.0, boxed_frac=0.25,
               box=0.05) -> LinearQuadraticProblem:
    """Dense QP whose Hessian has eigenvalues ``logspace(hi, lo)`` on a random orthogonal
    basis: with a small ``lambda = 1/dt`` the reduced KKT matrix is ill-conditioned and an
    unpivoted LDL^T shows element growth (the regime VERDICT r1 asked to pin)."""
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.logspace(hi, lo, n)
    Q = (U * ev) @ U.T
    Q = 0.5 * (Q + Q.T)
    q = rng.standard_normal(n)
    A = rng.standard_normal((m, n)) / np.sqrt(n)
    b = rng.standard_normal(m)
    lb = np.full(n, -np.inf)
    ub = np.full(n, np.inf)
    k = int(round(boxed_frac * n))
    idx = rng.permutation(n)[:k]
    lb[idx] = -box
    ub[idx] = box
    return LinearQuadraticProblem(Q, q, A, b, lb, ub)


def sparse_ocp(m=50_000, seed=0) -> LinearQuadraticProblem:
    """BASELINE config 3: x=[s_1..s_m, u_1..u_m], H=blkdiag(tridiag(-1,3,-1),
    0.1 I), c_i = s_i - s_{i-1} - h u_i, h=1/m."""
    n = 2 * m
    rng = np.random.default_rng(seed)
    e = np.ones(m)
    T = sps.diags([-e[:-1], 3.0 * e, -e[:-1]], [-1, 0, 1], format="csr")
    H = sps.block_diag([T, 0.1 * sps.identity(m, format="csr")], format="csr")
    B = sps.diags([e, -e[:-1]], [0, -1], format="csr")
    h = 1.0 / m
    J = sps.hstack([B, -h * sps.identity(m, format="csr")], format="csr")
    q = rng.standard_normal(n)
    return LinearQuadraticProblem(
        H, q, J, np.zeros(m), np.full(n, -np.inf), np.full(n, np.inf)
    )


def box_qp(n=16_384, seed=0, qscale=0.7413, bound=0.5, dense=False) -> LinearQuadraticProblem:
    """BASELINE config 5: H=tridiag(-1,2.5,-1), q=qscale*N(0,1), |x_j|<=bound,
    m=0.  ``dense=True`` is variant 5b (H stored dense)."""
    rng = np.random.default_rng(seed)
    e = np.ones(n)
    H = sps.diags([-e[:-1], 2.5 * e, -e[:-1]], [-1, 0, 1], format="csr")
    if dense:
        H = H.toarray()
    q = qscale * rng.standard_normal(n)
    A = sps.csr_matrix((0, n), dtype=np.float64)
    if dense:
        A = np.zeros((0, n))
    return LinearQuadraticProblem(H, q, A, np.zeros(0), np.full(n, -bound), np.full(n, bound))


def quartic_nlp(n=12, m=4, seed=0, bounded=True) -> QuarticProblem:
    """Random bounded non-convex NLP (SURVEY.md 8(a) verification case)."""
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n))
    Q = 0.5 * (G + G.T) / np.sqrt(n)
    q = rng.standard_normal(n)
    a = rng.uniform(0.5, 1.5, n)
    A = rng.standard_normal((m, n))
    B = 0.3 * rng.standard_normal((m, n))
    b = rng.standard_normal(m)
    if bounded:
        lb = np.where(rng.uniform(size=n) < 0.6, -rng.uniform(0.05, 0.6, n), -np.inf)
        ub = np.where(rng.uniform(size=n) < 0.6, rng.uniform(0.05, 0.6, n), np.inf)
    else:
        lb = np.full(n, -np.inf)
        ub = np.full(n, np.inf)
    return QuarticProblem(Q, q, a, A, B, b, lb, ub)
