"""Step controllers against trajectories recorded from the reference's
DistanceRatioController (tools/gen_golden.py, ctl_*.npz) and the reference's own controller
tests (tests/pygradflow/test_controller.py).  CPU tests run the host logic over the oracle
step solver; the GPU tests run the same logic over the HIP step solver and the
device-resident driver."""

import math

import numpy as np
import pytest

from tests import golden_util as G
from tests.oracle_step_solver import OracleStepSolver

from pygradflow_amd.controller import Controller, ControllerSettings, LogController
from pygradflow_amd.iterate import Iterate
from pygradflow_amd.params import Params
from pygradflow_amd import step_control as SC


# ---- reference tests/pygradflow/test_controller.py, restated ------------------------------
@pytest.fixture
def settings():
    return ControllerSettings(K_P=1e-1, K_I=0.0, lamb_init=0.0, lamb_red=0.5)


@pytest.mark.parametrize("val", [0.0, 1.0, 2.0])
def test_controller_sign_and_convergence(settings, val):
    ctl = Controller(settings, 1.0)
    u = ctl.update(val)
    assert (u < 0.0) if val > 1.0 else (u == 0.0 if val == 1.0 else u > 0.0)
    ctl = Controller(settings, 1.0)
    for _ in range(100):
        val = val + ctl.update(val)  # control x' = u
    assert np.allclose(val, 1.0, atol=1e-2)


@pytest.mark.parametrize("val", [1e-1, 1e0, 1e1])
def test_log_controller_sign_and_convergence(settings, val):
    u = math.log(LogController(settings, 1.0).update(val))
    assert (u < 0.0) if val > 1.0 else (u == 0.0 if val == 1.0 else u > 0.0)
    ctl = LogController(settings, 1.0)
    for _ in range(200):
        val = val * ctl.update(val)
    assert np.allclose(val, 1.0, atol=1e-2)


def test_pi_law_values():
    ctl = Controller(ControllerSettings(K_P=0.2, K_I=0.005, lamb_init=1.0, lamb_red=0.5), 0.5)
    assert ctl.value == 1.0
    assert ctl.update(0.3) == pytest.approx(0.2 * 0.2 + 0.005 * 0.2)
    assert ctl.update(0.9) == pytest.approx(0.2 * -0.4 + 0.005 * (0.2 - 0.4))
    ctl.reset()
    assert ctl.error_sum == 0.0


# ---- tau selection (newton_control.py:40-88) ----------------------------------------------
def _tiny_problem():
    from pygradflow_amd import problems as P

    Q = np.eye(3)
    q = np.array([1.0, -2.0, 0.0])
    return P.LinearQuadraticProblem(Q, q, np.zeros((0, 3)), np.zeros(0),
                                    np.array([-1.0, -1.0, -1.0]), np.array([1.0, 3.0, 1.0]))


def test_tau_vals_and_compute_tau():
    prob = _tiny_problem()
    it = Iterate(prob, Params(), np.zeros(3), np.zeros(0))
    ctl = SC.DistanceRatioController(prob, Params())
    # g = q at x = 0: variable 0 moves down to lb (distance 1 / 1), variable 1 up to ub
    # (distance 3 / 2), variable 2 does not move
    assert np.array_equal(ctl.tau_vals(it, 1.0), np.array([1.0, 1.5, -1.0]))
    assert ctl.compute_tau(it, 1.0) is None
    small = SC.DistanceRatioController(prob, Params(active_set_type="SmallestActiveSet"))
    assert small.compute_tau(it, 1.0) == 0.5
    large = SC.DistanceRatioController(prob, Params(active_set_type="LargestActiveSet"))
    assert large.compute_tau(it, 1.0) == 1.5
    expl = SC.DistanceRatioController(prob, Params(active_set_type="Explicit", active_set_tau=0.25))
    assert expl.compute_tau(it, 1.0) == 0.25
    hook = SC.DistanceRatioController(prob, Params(active_set_method=lambda it, lamb, rho: 7.0))
    assert hook.compute_tau(it, 1.0) == 7.0


def test_implicit_residual_matches_oracle():
    from oracle import newton_oracle as O
    from pygradflow_amd import problems as P

    prob = P.quartic_nlp(12, 4, seed=5)
    rng = np.random.default_rng(0)
    par = Params()
    xh, yh = np.clip(rng.standard_normal(12), prob.var_lb, prob.var_ub), rng.standard_normal(4)
    x, y = np.clip(xh + 0.3 * rng.standard_normal(12), prob.var_lb, prob.var_ub), yh + 0.1
    orig, it = Iterate(prob, par, xh, yh), Iterate(prob, par, x, y)
    mine = SC.implicit_residual(prob, orig, 0.4, it, 0.7)
    ref = O.unscaled_residual(0.4, xh, yh, x, y, it.aug_lag_deriv_x(0.7), it.cons, prob.var_lb,
                              prob.var_ub)
    assert np.array_equal(mine, ref)


def test_failed_step_doubles_lambda():
    from pygradflow_amd.errors import StepSolverError

    prob = _tiny_problem()

    class Boom(SC.NewtonController):
        def step(self, iterate, rho, dt, display=False, timer=None):
            raise StepSolverError("singular")

    it = Iterate(prob, Params(), np.zeros(3), np.zeros(0))
    res = Boom(prob, Params()).compute_step(it, 1.0, 0.25)
    assert not res.accepted and res.lamb == 8.0 and res.iterate is it and res.active_set is None


# ---- trajectories recorded from the reference ---------------------------------------------
def _run_host(case, name, step_solver):
    prob = G.rebuild_problem(case)
    nt = name.rsplit("_", 1)[1]
    par = Params(newton_type=nt, step_solver=step_solver, lamb_init=float(case["lamb_init"]))
    ctl = SC.DistanceRatioController(prob, par)
    return SC.gradient_flow(ctl, lambda x, y: Iterate(prob, par, x, y), case["x0"], case["y0"],
                            float(case["rho"]), int(case["iterations"]))


def _check(recs, case, tol, lamb_tol=None):
    lamb_tol = tol if lamb_tol is None else lamb_tol
    for k, r in enumerate(recs):
        assert r["accepted"] == bool(case["accepted"][k]), k
        assert r["lamb"] == pytest.approx(float(case["lamb"][k]), rel=lamb_tol), k
        assert r["lamb_next"] == pytest.approx(float(case["lamb_next"][k]), rel=lamb_tol), k
        assert G.rel_err(r["x"], case["x"][k]) <= tol, k
        assert G.rel_err(r["y"], case["y"][k]) <= tol, k


@pytest.mark.parametrize("name", G.controller_case_names())
def test_distance_ratio_controller_host_logic(name):
    case = G.load_case(name)
    _check(_run_host(case, name, OracleStepSolver), case, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", G.controller_case_names())
def test_distance_ratio_controller_hip(pgf, name):
    """Same trajectories with the HIP step solver behind the plugin hook.  Iterates to 1e-8.
    lambda only to 1e-5: once Newton has converged the second step length is a few ulps of
    the iterate, theta = ||d_2|| / ||d_1|| is then known to ~1e-6 relative for ANY linear
    solver other than the recording one, and d(lambda)/lambda = K_P d(theta)/theta."""
    case = G.load_case(name)
    _check(_run_host(case, name, pgf.HipStepSolver), case, 1e-8, lamb_tol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", [n for n in G.controller_case_names() if "dense_qp" in n])
def test_device_distance_ratio_controller(pgf, name):
    """Device-resident outer loop: point, H, J in HBM; the host sees two step lengths and one
    residual norm per outer iteration."""
    case = G.load_case(name)
    prob = G.rebuild_problem(case)
    nt = name.rsplit("_", 1)[1]
    par = Params(newton_type=nt, lamb_init=float(case["lamb_init"]))
    dn = pgf.DeviceNewton(prob, nt, case["x0"], case["y0"], 1.0 / par.lamb_init, float(case["rho"]))
    ctl = SC.DeviceDistanceRatioController(dn, par)
    lamb = par.lamb_init
    for k in range(int(case["iterations"])):
        assert lamb == pytest.approx(float(case["lamb"][k]), rel=1e-5), k
        res = ctl.step(float(case["rho"]), 1.0 / lamb)
        assert res.accepted == bool(case["accepted"][k]), k
        assert res.lamb == pytest.approx(float(case["lamb_next"][k]), rel=1e-5), k
        x, y = dn.point()
        assert G.rel_err(x, case["x"][k]) <= 1e-8 and G.rel_err(y, case["y"][k]) <= 1e-8, k
        lamb = res.lamb
    dn.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["Full", "Simplified"])
def test_batched_distance_ratio_controller(pgf, kind):
    """One controller object driving a whole device batch (per-instance lambda, per-instance
    accept / reject with restore, early exits as frozen instances) against one
    DeviceDistanceRatioController per instance: same decisions, same lambdas, same points."""
    from pygradflow_amd import problems
    from pygradflow_amd.batched import BatchedDeviceNewton

    B, n, m = 4, 96, 24

    def make(i):
        return problems.dense_qp(n, m, seed=30 + i, boxed_frac=0.2 + 0.1 * i, box=0.05)

    par = Params(newton_type=kind, lamb_init=1.0 + 0.0)
    bd = BatchedDeviceNewton(make, B, kind, 1.0, 1.0)
    bc = SC.BatchedDistanceRatioController(bd, par, rho=1.0)
    singles = []
    for i in range(B):
        dn = pgf.DeviceNewton(make(i), kind, np.zeros(n), np.zeros(m), 1.0, 1.0)
        singles.append((dn, SC.DeviceDistanceRatioController(dn, par), [par.lamb_init]))
    rejected_seen = False
    for it in range(9):
        lamb_used, lamb_next, acc = bc.step()
        x, y = bd.points()
        for i, (dn, ctl, lam) in enumerate(singles):
            assert lamb_used[i] == pytest.approx(lam[0], rel=1e-6), (it, i)
            res = ctl.step(1.0, 1.0 / lam[0])
            assert res.accepted == bool(acc[i]), (it, i)
            assert res.lamb == pytest.approx(lamb_next[i], rel=1e-6), (it, i)
            lam[0] = res.lamb
            if res.accepted:
                xs, ys = dn.point()
                assert G.rel_err(x[i], xs) <= 1e-9 and G.rel_err(y[i], ys) <= 1e-9, (it, i)
            else:
                rejected_seen = True
    # one more advance puts rejected instances back on their outer points, as the single
    # controllers did at once
    bd.advance_outer_each(1.0 / bc.lamb, bc.rho, bc.accepted)
    x, y = bd.points()
    for i, (dn, _, _) in enumerate(singles):
        xs, ys = dn.point()
        assert G.rel_err(x[i], xs) <= 1e-9 and G.rel_err(y[i], ys) <= 1e-9, i
        dn.close()
    bd.close()
    assert rejected_seen or kind == "Full"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["Full", "Simplified", "ActiveSet"])
def test_device_resident_controller_matches_host_controller(pgf, kind):
    """pgf_batch_ctl_*: nine outer iterations enqueued back to back with every decision taken
    on the device, against the host-driven BatchedDistanceRatioController on a twin batch:
    the same accept / reject flags, lambdas equal to rounding (device log / exp vs libm), the
    same points; then against the reference's own controller trajectory for one instance."""
    from pygradflow_amd import problems
    from pygradflow_amd.batched import BatchedDeviceNewton

    B, n, m, iters = 5, 96, 24, 9

    def make(i):
        return problems.dense_qp(n, m, seed=30 + i, boxed_frac=0.2 + 0.1 * i, box=0.05)

    par = Params(newton_type=kind, lamb_init=1.0)
    host_b = BatchedDeviceNewton(make, B, kind, 1.0, 1.0)
    host = SC.BatchedDistanceRatioController(host_b, par, rho=1.0)
    dev_b = BatchedDeviceNewton(make, B, kind, 1.0, 1.0)
    dev = SC.DeviceResidentDistanceRatioController(dev_b, par, rho=1.0, max_iterations=iters)
    rec = [host.step() for _ in range(iters)]
    dev.run(iters)  # ONE host synchronisation for all nine iterations
    rejected = False
    for it, (lamb_used, lamb_next, acc) in enumerate(rec):
        assert np.array_equal(dev.history[it, :, 2] != 0.0, acc), it
        assert np.allclose(dev.history[it, :, 0], lamb_used, rtol=1e-12, atol=0), it
        assert np.allclose(dev.history[it, :, 1], lamb_next, rtol=1e-12, atol=0), it
        rejected |= not acc.all()
    assert np.allclose(dev.lamb, host.lamb, rtol=1e-12)
    # put rejected instances back (the next outer step would) and compare the points
    host_b.advance_outer_each(1.0 / host.lamb, host.rho, host.accepted)
    dev_b.advance_outer_each(1.0 / dev.lamb, np.full(B, 1.0), dev.accepted)
    xh, yh = host_b.points()
    xd, yd = dev_b.points()
    assert G.rel_err(xd, xh) <= 1e-10 and G.rel_err(yd, yh) <= 1e-10
    host_b.close()
    dev_b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("newton_type", ["Simplified", "Full"])
def test_device_resident_controller_against_reference_trajectory(pgf, newton_type):
    """The reference's DistanceRatioController trajectory (ctl_dense_qp_boxed_n96_m24_*.npz:
    lambda, accept flag and iterate per outer iteration) with all decisions on the device."""
    from pygradflow_amd.batched import BatchedDeviceNewton

    case = G.load_case(f"ctl_dense_qp_boxed_n96_m24_{newton_type}")
    problem = G.rebuild_problem(case)
    iters = int(case["iterations"])
    par = Params(newton_type=newton_type, lamb_init=float(case["lamb_init"]))
    bd = BatchedDeviceNewton(lambda i: problem, 1, newton_type, 1.0, float(case["rho"]))
    ctl = SC.DeviceResidentDistanceRatioController(bd, par, rho=float(case["rho"]),
                                                   max_iterations=iters)
    ctl.run(iters)
    assert np.array_equal(ctl.history[:, 0, 2] != 0.0, case["accepted"])
    assert np.allclose(ctl.history[:, 0, 0], case["lamb"], rtol=1e-9)
    assert np.allclose(ctl.history[:, 0, 1], case["lamb_next"], rtol=1e-9)
    bd.advance_outer_each(1.0 / ctl.lamb, np.array([float(case["rho"])]), ctl.accepted)
    x, y = bd.points()
    assert G.rel_err(x[0], case["x"][-1]) <= 1e-9 and G.rel_err(y[0], case["y"][-1]) <= 1e-9
    bd.close()


@pytest.mark.gpu
def test_device_resident_controller_notices_a_failed_helper_handover(pgf):
    """inside pgf_batch_ctl_iterate a failed chain-helper hand-over looks like a
    singular matrix to the device-resident controller (reject, 2 lambda) and its flag is
    overwritten by the next step; pgf_batch_ctl_read must still learn of it (sticky word) and
    switch the helpers off, so that the failure cannot repeat iteration after iteration.  The
    run then continues to the same accept / reject pattern as an undisturbed batch from the
    iteration after the injected failure on."""
    from pygradflow_amd import _lib, problems
    from pygradflow_amd.batched import BatchedDeviceNewton

    lib = _lib.load()
    B, n, m = 3, 400, 100
    make = lambda i: problems.dense_qp(n, m, seed=60 + i, boxed_frac=0.1, box=0.05)
    par = Params(newton_type="Full", lamb_init=1.0)
    lib.pgf_debug_chain_helpers(1)
    bd = BatchedDeviceNewton(make, B, "Full", 1.0, 1.0)
    try:
        ctl = SC.DeviceResidentDistanceRatioController(bd, par, rho=1.0, max_iterations=6)
        _lib.check(lib.pgf_batch_debug_fail_next_helper(bd._b))
        ctl.run(2)
        # instance 0's first iteration was rejected (its factorisation reported failed helpers)
        assert ctl.history[0, 0, 2] == 0.0
        assert ctl.history[0, 1:, 2].all()
        assert lib.pgf_debug_chain_helpers(-1) == 0  # switched off by pgf_batch_ctl_read
        ctl.run(3)
        assert ctl.history[2:5, :, 2].all()  # no further failures: lambda does not run away
        assert np.all(ctl.lamb < 10.0)
    finally:
        lib.pgf_debug_chain_helpers(1)
        lib.pgf_debug_chain_enable(1)
        bd.close()
