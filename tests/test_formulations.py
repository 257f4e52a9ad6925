"""the reference's Standard / Extended / Asymmetric step-solver
formulations (step/solver/*.py) and the LU they need, against trajectories and matrices
recorded from the reference (tools/gen_golden.py, formul_*.npz, linear_solver_lu.npz).

CPU tests run the host-side assembly logic with scipy's LU and the oracle's scaled residual
standing in for the GPU pieces (test-only subclasses below); the GPU tests run the product
classes: assembly on the host as in the reference, LU factorisation + solves and the scaled
residual / mask kernels on the device."""

import numpy as np
import pytest
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from oracle import newton_oracle as O
from tests import golden_util as G

from pygradflow_amd import unsym_step_solvers as U
from pygradflow_amd.iterate import Iterate
from pygradflow_amd.newton import newton_steps
from pygradflow_amd.params import Params

TOL = 1e-10
KINDS = ("Standard", "Extended", "Asymmetric")


# ---- test-only stand-ins for the GPU pieces (never imported by the product) ---------------
class _ScipyLU:
    def __init__(self, mat):
        self.lu = spla.splu(sps.csc_matrix(mat))

    def solve(self, rhs, trans=False, initial_sol=None):
        return self.lu.solve(rhs, trans="T" if trans else "N")

    def num_neg_eigvals(self):
        return None


class _OracleScaledFunc:
    """Scaled residual / mask through the CPU restatement (oracle/newton_oracle.py)."""

    def __init__(self, owner):
        self.o = owner

    def _bounds(self):
        return O.scaled_bounds(1.0 / self.o.dt, self.o.problem.var_lb, self.o.problem.var_ub)

    def compute_active_set(self, iterate, rho, tau=None):
        slb, sub = self._bounds()
        p = O.projection_initial(self.o.dt, self.o.orig_iterate.x, iterate.x,
                                 iterate.aug_lag_deriv_x(rho), tau)
        return O.active_set_box(p, slb, sub)

    def value_at(self, iterate, rho, active_set=None):
        slb, sub = self._bounds()
        if active_set is None:
            active_set = self.compute_active_set(iterate, rho)
        o = self.o
        return O.residual(o.dt, o.orig_iterate.x, o.orig_iterate.y, iterate.x, iterate.y,
                          iterate.aug_lag_deriv_x(rho), iterate.aug_lag_deriv_y(), slb, sub,
                          np.asarray(active_set, dtype=bool))


def _cpu_class(kind):
    base = {"Standard": U.StandardStepSolver, "Extended": U.ExtendedStepSolver,
            "Asymmetric": U.AsymmetricStepSolver}[kind]

    class Cpu(base):
        def linear_solver(self, mat):
            return _ScipyLU(mat)

        def _make_func(self):
            return _OracleScaledFunc(self)

    return Cpu


def _replay(case, make_params):
    problem = G.rebuild_problem(case)
    dt, rho, tau = float(case["dt"]), float(case["rho"]), G.case_tau(case)
    for kind in KINDS:
        for pol in case["policies"]:
            params = make_params(kind, str(pol))
            orig = Iterate(problem, params, case["x0"], case["y0"])
            gen = newton_steps(problem, params, orig, dt, rho, tau)
            for k in range(int(case["steps"])):
                step = next(gen)
                pre = f"{kind}/{pol}/{k}/"
                assert np.array_equal(step.active_set, case[pre + "mask"]), (kind, pol, k)
                assert G.rel_err(step.dx, case[pre + "dx"]) <= TOL, (kind, pol, k)
                assert G.rel_err(step.dy, case[pre + "dy"]) <= TOL, (kind, pol, k)
                assert G.rel_err(step.iterate.x, case[pre + "xn"]) <= TOL, (kind, pol, k)
                assert G.rel_err(step.iterate.y, case[pre + "yn"]) <= TOL, (kind, pol, k)
                assert abs(step.diff - float(case[pre + "diff"])) <= TOL * max(1.0, step.diff)


@pytest.mark.parametrize("name", G.formulation_case_names())
def test_formulations_host_logic_replays_reference(name):
    """Assembly of the three Newton matrices and the right-hand-side splits, LU by scipy."""
    case = G.load_case(name)
    _replay(case, lambda kind, pol: Params(newton_type=pol, step_solver=_cpu_class(kind)))


@pytest.mark.parametrize("name", G.formulation_case_names())
def test_formulation_matrices_equal_reference(name):
    """The assembled (n + m)^2 matrix and the residual of the first step, entry for entry."""
    case = G.load_case(name)
    problem = G.rebuild_problem(case)
    dt, rho, tau = float(case["dt"]), float(case["rho"]), G.case_tau(case)
    for kind in KINDS:
        pre = f"{kind}/Full/0/"
        params = Params(newton_type="Full")
        orig = Iterate(problem, params, case["x0"], case["y0"])
        sv = _cpu_class(kind)(problem, params, orig, dt, rho)
        mask = sv.func.compute_active_set(orig, rho, tau)
        assert np.array_equal(mask, case[pre + "mask"])
        sv.update_active_set(mask)
        sv.update_derivs(orig)
        sv.solve(orig)
        assert G.rel_err(sv.deriv.toarray(), case[pre + "deriv"]) <= 1e-14, kind
        assert G.rel_err(sv.func.value_at(orig, rho, mask), case[pre + "F"]) <= 1e-13, kind


def test_factory_dispatches_on_step_solver_type():
    problem = G.rebuild_problem(G.load_case("formul_quartic_n12_m4"))
    for kind, cls in (("Standard", U.StandardStepSolver), ("Extended", U.ExtendedStepSolver),
                      ("Asymmetric", U.AsymmetricStepSolver)):
        assert U._BY_TYPE[kind] is cls
    marker = object()
    params = Params(step_solver=lambda *a: marker)
    assert U.step_solver(problem, params, None, 1.0, 1.0) is marker  # the hook wins


# ---- GPU ---------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", G.formulation_case_names())
def test_formulations_on_gpu_replay_reference(pgf, name):
    case = G.load_case(name)
    _replay(case, lambda kind, pol: pgf.Params(newton_type=pol, step_solver_type=kind))


@pytest.mark.gpu
def test_lu_linear_solver_golden(pgf):
    ls = np.load(G.GOLDEN + "/linear_solver_lu.npz")
    for nm in ("n7", "n40", "n150"):
        mat, rhs = ls[nm + "/mat"], ls[nm + "/rhs"]
        sv = pgf.HipLinearSolver(sps.csc_matrix(mat), symmetric=False)
        assert G.rel_err(sv.solve(rhs), ls[nm + "/sol"]) <= 1e-10
        assert G.rel_err(sv.solve(rhs, trans=True), ls[nm + "/sol_trans"]) <= 1e-10
        assert sv.num_neg_eigvals() is None
        sv.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 31, 32, 33, 64, 65, 100, 257, 700, 1300])
def test_lu_linear_solver_random(pgf, n):
    rng = np.random.default_rng(900 + n)
    A = rng.standard_normal((n, n))
    A[0, 0] = 0.0 if n > 1 else 2.0  # a zero in the first pivot position: pivoting must move it
    rhs = rng.standard_normal(n)
    sv = pgf.HipLinearSolver(A, symmetric=False)
    ref, refT = np.linalg.solve(A, rhs), np.linalg.solve(A.T, rhs)
    cond = np.linalg.cond(A)
    tol = max(1e-11, 1e-15 * cond)
    assert G.rel_err(sv.solve(rhs), ref) <= tol, (n, cond)
    assert G.rel_err(sv.solve(rhs, trans=True), refT) <= tol, (n, cond)
    # P A = L U with |L| <= 1 (partial pivoting)
    F = sv.factor_matrix()
    L = np.tril(F, -1) + np.eye(n)
    assert np.abs(np.tril(F, -1)).max(initial=0.0) <= 1.0 + 1e-12
    PA = L @ np.triu(F)  # a row permutation of A: same rows in another order
    key = np.array([1.0, np.pi, np.e])[np.arange(n) % 3]
    assert np.allclose(np.sort(PA @ key), np.sort(A @ key), rtol=0, atol=1e-9 * max(1.0, np.abs(A).max() * n))
    assert abs(np.linalg.norm(PA) - np.linalg.norm(A)) <= 1e-10 * np.linalg.norm(A)
    sv.close()


@pytest.mark.gpu
def test_lu_singular_raises(pgf):
    A = np.ones((5, 5))
    with pytest.raises(pgf.LinearSolverError):
        pgf.HipLinearSolver(A, symmetric=False)
    Z = np.zeros((3, 3))
    with pytest.raises(pgf.LinearSolverError):
        pgf.HipLinearSolver(Z, symmetric=False)
    e = pgf.HipLinearSolver(np.zeros((0, 0)), symmetric=False)
    assert e.solve(np.zeros(0)).shape == (0,)
