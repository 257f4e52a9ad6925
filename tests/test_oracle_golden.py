"""Pin the CPU oracle to the reference's golden vectors (CPU, no GPU)."""

import numpy as np
import pytest

from oracle import newton_oracle as O
from tests import golden_util as G

TOL = 1e-13


@pytest.mark.parametrize("name", G.case_names())
def test_oracle_replays_every_recorded_step(name):
    case = G.load_case(name)
    shape = G.shape_only_problem(case)
    dt, rho, tau = float(case["dt"]), float(case["rho"]), G.case_tau(case)
    for pol in case["policies"]:
        for k in range(int(case["steps"])):
            pre = f"{pol}/{k}/"
            pt = G.RecordedPoint(case, pol, k)
            sv = O.SymmetricStep(shape, case["x0"], case["y0"], dt, rho)
            if pol != "Simplified":
                p = O.projection_initial(dt, case["x0"], pt.x, pt.g(rho), tau)
                assert G.rel_err(p, case[pre + "p"]) <= TOL
                assert np.array_equal(sv.compute_active_set(pt, tau), case[pre + "mask"])
            sv.update_active_set(case[pre + "mask"])
            H, J = G.step_derivs(case, pol, k)
            pt_derivs = G.RecordedPoint(case, pol, k)
            sv.jac = pt_derivs.jac.__class__(J).tocsc()
            sv.hess = pt_derivs.hess.__class__(H)
            xn, yn, diff = sv.solve(pt)
            rec = sv.record
            assert G.rel_err(rec["g"], case[pre + "g"]) <= TOL
            assert G.rel_err(rec["F"], case[pre + "F"]) <= TOL
            assert G.rel_err(rec["rhs"], case[pre + "rhs"]) <= TOL
            Kd = rec["K"].toarray()
            assert Kd.shape == case[pre + "K"].shape
            assert G.rel_err(Kd, case[pre + "K"]) <= TOL
            assert G.rel_err(rec["s"], case[pre + "s"]) <= 1e-12
            assert G.rel_err(rec["dx"], case[pre + "dx"]) <= 1e-12
            assert G.rel_err(rec["dy"], case[pre + "dy"]) <= 1e-12
            assert G.rel_err(xn, case[pre + "xn"]) <= 1e-12
            assert G.rel_err(yn, case[pre + "yn"]) <= 1e-12
            assert abs(diff - float(case[pre + "diff"])) <= 1e-12 * max(1.0, diff)
            assert O.num_neg_eigvals_dense(0.5 * (Kd + Kd.T)) == int(case[pre + "n_neg"])


@pytest.mark.parametrize(
    "name", [n for n in G.case_names() if G.has_problem(G.load_case(n))]
)
def test_oracle_policies_step_for_step(name):
    """Free-running policy state machine == reference trajectory."""
    case = G.load_case(name)
    problem = G.rebuild_problem(case)
    dt, rho, tau = float(case["dt"]), float(case["rho"]), G.case_tau(case)
    for pol in case["policies"]:
        orc = O.NewtonOracle(problem, str(pol), case["x0"], case["y0"], dt, rho, tau)
        recs = orc.run(case["x0"], case["y0"], int(case["steps"]))
        for k, rec in enumerate(recs):
            pre = f"{pol}/{k}/"
            assert np.array_equal(rec["mask"], case[pre + "mask"]), (pol, k)
            assert G.rel_err(rec["xn"], case[pre + "xn"]) <= 1e-12, (pol, k)
            assert G.rel_err(rec["yn"], case[pre + "yn"]) <= 1e-12, (pol, k)


@pytest.mark.parametrize("name", G.illcond_case_names())
def test_oracle_illconditioned_cases(name):
    """cond(K) 2e7 ... 4e9 (illcond_*.npz): the restatement makes the reference's own bmat + splu
    calls, so it reproduces the reference's solution far inside the reference's own forward
    error (stored per step against an extended-precision solve), masks bit for bit."""
    case = G.load_case(name)
    problem = G.rebuild_problem(case)
    dt, rho = float(case["dt"]), float(case["rho"])
    orc = O.NewtonOracle(problem, "Full", case["x0"], case["y0"], dt, rho, None)
    recs = orc.run(case["x0"], case["y0"], int(case["steps"]))
    for k, rec in enumerate(recs):
        pre = f"Full/{k}/"
        assert float(case[pre + "cond"]) > 1e7
        tol = max(1e-12, 0.1 * float(case[pre + "ref_err"]))
        assert np.array_equal(rec["mask"], case[pre + "mask"]), k
        assert G.rel_err(rec["s"], case[pre + "s"]) <= tol, k
        assert G.rel_err(rec["xn"], case[pre + "xn"]) <= tol, k
        assert G.rel_err(rec["yn"], case[pre + "yn"]) <= tol, k


def test_oracle_linear_solver_vectors():
    import scipy.sparse as sps

    ls = np.load(G.GOLDEN + "/linear_solver_5x5.npz")
    for nm in ("indef", "posdef", "negdef"):
        mat = ls[nm + "/mat"]
        lu = O.factor_kkt(sps.csc_matrix(mat))
        assert G.rel_err(lu.solve(ls["rhs"]), ls[nm + "/sol"]) <= 1e-14
        assert O.num_neg_eigvals_dense(mat) == int(ls[nm + "/n_neg"])
        assert np.allclose(mat @ ls[nm + "/sol"], ls["rhs"])
