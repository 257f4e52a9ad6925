"""Termination measures and penalty update against values recorded from
the reference's Iterate / ActiveSet / DualNormUpdate (tools/gen_golden.py, measures_*.npz)."""

import numpy as np
import pytest

from tests import golden_util as G

from pygradflow_amd.iterate import Iterate
from pygradflow_amd.params import Params
from pygradflow_amd.penalty import ConstantPenalty, DualNormUpdate

CASES = ["measures_lq", "measures_quartic", "measures_box"]


@pytest.mark.parametrize("name", CASES)
def test_iterate_measures_match_reference(name):
    case = G.load_case(name)
    prob = G.rebuild_problem(case)
    par = Params()
    for k in range(case["x"].shape[0]):
        it = Iterate(prob, par, case["x"][k], case["y"][k].reshape(prob.num_cons))
        act = it.active_set
        for key in ("at_lower", "at_upper", "at_both", "violated"):
            assert np.array_equal(getattr(act, key), case[key][k]), (k, key)
        assert np.array_equal(act.satisfied, ~case["violated"][k])
        assert np.allclose(it.bounds_dual, case["bounds_dual"][k], rtol=1e-13, atol=0)
        for key in ("stat_res", "cons_violation", "bound_violation"):
            assert getattr(it, key) == pytest.approx(float(case[key][k]), rel=1e-13), (k, key)
        assert it.total_res == max(it.stat_res, it.cons_violation, it.bound_violation)
        assert it.is_feasible(1e300) and (it.is_feasible(0.0) == (it.cons_violation == 0.0 and
                                                                 it.bound_violation == 0.0))


@pytest.mark.parametrize("name", CASES)
def test_dual_norm_update_matches_reference(name):
    case = G.load_case(name)
    prob = G.rebuild_problem(case)
    par = Params()
    pen = DualNormUpdate(prob, par)
    trace = [pen.initial(None)]
    for k in range(case["x"].shape[0]):
        it = Iterate(prob, par, case["x"][k], case["y"][k].reshape(prob.num_cons))
        res = pen.update(None, it)
        assert res.accept
        trace.append(res.next_rho)
    assert np.array_equal(np.array(trace), case["rho_trace"])
    assert ConstantPenalty(prob, par).update(None, None).next_rho == par.rho


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["measures_lq", "measures_box"])
def test_device_measures(pgf, name):
    """pgf_qp_measures on the device-resident point (dense and banded storage)."""
    import scipy.sparse as sps

    from pygradflow_amd import problems

    case = G.load_case(name)
    prob = G.rebuild_problem(case)
    variants = [prob]
    if name == "measures_box":  # same problem through the CSR / banded path
        sp = problems.LinearQuadraticProblem(
            sps.csr_matrix(prob.hess_dense()), prob.q,
            sps.csr_matrix(prob.jac_dense().reshape(prob.num_cons, prob.num_vars)), prob.b,
            prob.var_lb, prob.var_ub)
        sp.pgf_force_band = True
        variants.append(sp)
    for pv in variants:
        n, m = pv.num_vars, pv.num_cons
        dn = pgf.DeviceNewton(pv, "Full", np.zeros(n), np.zeros(m), 1.0, 1.0)
        for k in range(case["x"].shape[0]):
            dn.set_point(case["x"][k], case["y"][k].reshape(m))
            ms = dn.measures()
            assert ms["stat_res"] == pytest.approx(float(case["stat_res"][k]), rel=1e-12)
            assert ms["cons_violation"] == pytest.approx(float(case["cons_violation"][k]), rel=1e-12)
            assert ms["bound_violation"] == float(case["bound_violation"][k])
            assert ms["y_inf"] == (np.max(np.abs(case["y"][k])) if m else 0.0)
        dn.close()


@pytest.mark.gpu
def test_batch_measures(pgf):
    from pygradflow_amd import problems
    from pygradflow_amd.batched import BatchedDeviceNewton

    B, n, m = 4, 96, 24

    def make(i):
        return problems.dense_qp(n, m, seed=20 + i, boxed_frac=0.3, box=0.02)

    bd = BatchedDeviceNewton(make, B, "Full", 1.0, 1.0)
    bd.step_local()
    got = bd.measures()
    x, y = bd.points()
    par = Params()
    for i in range(B):
        it = Iterate(make(i), par, x[i], y[i])
        want = [it.stat_res, it.cons_violation, it.bound_violation, np.max(np.abs(y[i]))]
        assert np.allclose(got[i], want, rtol=1e-11, atol=1e-13), i
    # the measures leave the point and the next step untouched
    st, nn, _ = bd.step_local()
    assert not st.any() and (nn == m).all()
    bd.close()
