"""Helpers to read the golden vectors written by tools/gen_golden.py."""

import glob
import os
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def case_names():
    return sorted(
        os.path.basename(p)[:-4]
        for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
        if not p.endswith("linear_solver_5x5.npz")
        and not os.path.basename(p).startswith(("extras_", "ctl_", "measures_", "formul_",
                                                "linear_solver_", "illcond_"))
    )


def illcond_case_names():
    """cond(K) 2e7 ... 4e9: beyond what can be replayed at 1e-10 (the reference's own forward
    error there is 6e-11 ... 3e-9); held to the stored extended-precision solutions instead."""
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "illcond_*.npz")))


def formulation_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "formul_*.npz")))


def controller_case_names():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "ctl_*.npz")))


def load_case(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def case_tau(case):
    tau = float(case["tau"])
    return None if np.isnan(tau) else tau


def has_problem(case):
    return "problem/kind" in case.files


def rebuild_problem(case):
    """Problem object from the stored definition (LQ / quartic cases)."""
    from pygradflow_amd import problems as P

    kind = str(case["problem/kind"])
    lb, ub = case["var_lb"], case["var_ub"]
    if kind == "lq":
        return P.LinearQuadraticProblem(
            case["problem/Q"], case["problem/q"], case["problem/A"], case["problem/b"], lb, ub
        )
    if kind == "rosenbrock":
        return P.RosenbrockProblem(float(case["problem/a"]), float(case["problem/b"]))
    assert kind == "quartic"
    return P.QuarticProblem(
        case["problem/Q"], case["problem/q"], case["problem/a"], case["problem/A"],
        case["problem/B"], case["problem/b"], lb, ub,
    )


def shape_only_problem(case):
    return SimpleNamespace(
        var_lb=case["var_lb"], var_ub=case["var_ub"],
        num_vars=int(case["n"]), num_cons=int(case["m"]),
    )


def step_derivs(case, pol, k):
    """(H, J) the solver had frozen at step k (dense)."""
    if has_problem(case) and str(case["problem/kind"]) == "lq":
        return case["problem/Q"], case["problem/A"]
    return case[f"{pol}/{k}/H"], case[f"{pol}/{k}/J"]


class RecordedPoint:
    """An iterate whose evaluations are the recorded ones (oracle PointData
    surface *and* the Iterate surface the step solver reads)."""

    def __init__(self, case, pol, k, problem=None, params=None):
        pre = f"{pol}/{k}/"
        self.x = case[pre + "x"]
        self.y = case[pre + "y"]
        self.obj_grad = case[pre + "obj_grad"]
        self.cons = case[pre + "cons"]
        H, J = step_derivs(case, pol, k)
        jx = case[pre + "jac_at_x"] if (pre + "jac_at_x") in case.files else J
        self.jac = sps.csr_matrix(jx)
        self.cons_jac = self.jac
        self.hess = sps.csr_matrix(H)
        self.problem = problem
        self.params = params
        self.eval = None
        self._g = case[pre + "g"]

    def g(self, rho):
        return self.aug_lag_deriv_x(rho)

    # Iterate surface
    def aug_lag_deriv_x(self, rho):
        return self.obj_grad + self.jac.T.dot(rho * self.cons + self.y)

    def aug_lag_deriv_y(self):
        return self.cons

    def aug_lag_deriv_xy(self):
        return self.jac

    def aug_lag_deriv_xx(self, rho):
        assert rho == 0.0
        return self.hess


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1.0, np.max(np.abs(b))))
