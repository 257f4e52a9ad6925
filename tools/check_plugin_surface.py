#!/usr/bin/env python3
"""Pin the drop-in surface against the REFERENCE (build container only: imports /root/reference;
nothing of it is copied, nothing here travels to the GPU box as a test dependency).

1. ``inspect``: HipStepSolver / StepResult / HipStepFunc / HipLinearSolver expose every public
   method, property and attribute the reference's plug-in classes define
   (``step/solver/step_solver.py:16-130``, ``linear_solver/linear_solver.py:18-31``,
   ``implicit_func.py`` StepFunc surface used by ``newton.py``), with the same leading parameter
   names, and the hook call ``params.step_solver(problem, params, iterate, dt, rho)``
   (``step/solver/__init__.py:18-19``) binds.
2. The reference's OWN driver -- ``Solver(problem, params).solve()`` with its ``newton_method`` and
   ``DistanceRatioController`` -- runs over a test-only object that has EXACTLY HipStepSolver's
   public surface (checked name by name) but computes with the CPU oracle, on the reference's
   Rosenbrock example: 30 iterations, 25 accepted steps, x = (1, 1)
   (``docs/solve_rosenbrock.output:5-14``).  What the reference calls on a step solver is
   therefore what HipStepSolver offers.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/check_plugin_surface.py
"""

import importlib.util
import inspect
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PGF_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, REPO)
sys.path.append(os.path.join(REPO, "tools", "_stubs"))  # cosmetic termcolor stand-in

from pygradflow.step.solver.step_solver import StepResult as RefStepResult  # noqa: E402
from pygradflow.step.solver.step_solver import StepSolver as RefStepSolver  # noqa: E402
from pygradflow.step.solver.symmetric_step_solver import SymmetricStepSolver as RefSymmetric  # noqa: E402
from pygradflow.linear_solver.linear_solver import LinearSolver as RefLinearSolver  # noqa: E402
from pygradflow.implicit_func import ScaledImplicitFunc as RefFunc  # noqa: E402
from pygradflow.params import Params as RefParams  # noqa: E402
from pygradflow.solver import Solver as RefSolver  # noqa: E402

from oracle import newton_oracle as O  # noqa: E402
from pygradflow_amd import step_solver as ours  # noqa: E402
from pygradflow_amd.linear_solver import HipLinearSolver  # noqa: E402

problems = []


def public(cls):
    return {n for n in dir(cls) if not n.startswith("_")}


def params_of(fn):
    return [p for p in inspect.signature(fn).parameters if p != "self"]


def check_methods(ref_cls, our_cls, names, what):
    for nm in names:
        if not hasattr(our_cls, nm):
            problems.append(f"{what}: missing {nm}")
            continue
        r, o = getattr(ref_cls, nm), getattr(our_cls, nm)
        if isinstance(r, property) or not callable(r):
            if not isinstance(o, property) and callable(o):
                problems.append(f"{what}.{nm}: property in the reference, method here")
            continue
        rp, op = params_of(r), params_of(o)
        if op[: len(rp)] != rp:
            problems.append(f"{what}.{nm}: parameters {op} do not start with the reference's {rp}")
        else:
            print(f"  {what}.{nm}({', '.join(op)})  ~  reference ({', '.join(rp)})")


print("StepSolver surface (step/solver/step_solver.py:66-130):")
check_methods(RefStepSolver, ours.HipStepSolver,
              ["update_active_set", "update_derivs", "solve", "func", "active_set", "jac", "hess",
               "linear_solver"], "HipStepSolver")
# attributes the reference sets in __init__ (:67-77) and its callers read
src = inspect.getsource(ours.HipStepSolver.__init__)
for attr in ("problem", "params", "n", "m", "solver"):
    if f"self.{attr} =" not in src:
        problems.append(f"HipStepSolver.__init__ does not set self.{attr}")
# the hook: params.step_solver(problem, params, iterate, dt, rho)
hp = params_of(ours.HipStepSolver.__init__)
if len(hp) < 5:
    problems.append(f"HipStepSolver.__init__ takes {hp}: the hook passes five positional arguments")
print(f"  hook: HipStepSolver({', '.join(hp)})")
rp = params_of(RefSymmetric.__init__)
print(f"        reference SymmetricStepSolver({', '.join(rp)})")

print("StepResult surface (step/solver/step_solver.py:16-63):")
rr, orr = params_of(RefStepResult.__init__), params_of(ours.StepResult.__init__)
if orr[: len(rr)] != rr:
    problems.append(f"StepResult.__init__ {orr} does not start with the reference's {rr}")
for nm in ("iterate", "diff"):
    if not hasattr(ours.StepResult, nm):
        problems.append(f"StepResult: missing {nm}")
rsrc = inspect.getsource(ours.StepResult)
for attr in ("orig_iterate", "dx", "dy", "active_set", "rcond", "xn"):
    if f"self.{attr}" not in rsrc:
        problems.append(f"StepResult never sets self.{attr}")
print(f"  StepResult({', '.join(orr)})  ~  reference ({', '.join(rr)}); iterate, diff present")

print("StepFunc surface used by newton.py (:54, :84, :159-160, :206, :239):")
check_methods(RefFunc, ours.HipStepFunc, ["compute_active_set", "value_at", "deriv_at"], "HipStepFunc")

print("LinearSolver surface (linear_solver/linear_solver.py:18-31):")
check_methods(RefLinearSolver, HipLinearSolver, ["solve", "num_neg_eigvals", "rcond"], "HipLinearSolver")
lp, rlp = params_of(HipLinearSolver.__init__), params_of(RefLinearSolver.__init__)
if lp[: len(rlp)] != rlp:
    problems.append(f"HipLinearSolver.__init__ {lp} does not start with the reference's {rlp}")


# ---------------------------------------------------------------------------------------------
# 2. the reference's own driver over an oracle-backed object with HipStepSolver's surface
class _Func:
    """compute_active_set / value_at / deriv_at, as HipStepFunc (the reference's controllers and
    Newton policies call nothing else on ``step_solver.func``)."""

    def __init__(self, owner):
        self.owner = owner

    def compute_active_set(self, iterate, rho, tau=None):
        o = self.owner
        return o.sv.compute_active_set(O.PointData(o.problem, iterate.x, iterate.y), tau)

    def value_at(self, iterate, rho, active_set=None):  # pragma: no cover - Globalized only
        raise NotImplementedError()

    def deriv(self, jac, hess, active_set):  # pragma: no cover - Globalized only
        raise NotImplementedError()

    def deriv_at(self, iterate, rho, active_set=None):  # pragma: no cover - Globalized only
        raise NotImplementedError()


class SurfaceOracleStepSolver:
    """Public surface = HipStepSolver's, name for name; arithmetic = the CPU oracle."""

    def __init__(self, problem, params, orig_iterate, dt, rho, device: int = 0):
        self.problem, self.params = problem, params
        self.n, self.m = problem.num_vars, problem.num_cons
        self.orig_iterate, self.dt, self.rho = orig_iterate, dt, rho
        self.solver = None
        self.sparse = False
        self.last_n_neg = None
        self.sv = O.SymmetricStep(problem, orig_iterate.x, orig_iterate.y, dt, rho)
        self._func = _Func(self)
        self._active_set = self._jac = self._hess = None

    func = property(lambda self: self._func)
    active_set = property(lambda self: self._active_set)
    jac = property(lambda self: self._jac)
    hess = property(lambda self: self._hess)

    def close(self):
        pass

    def linear_solver(self, mat):
        return O.factor_kkt(mat)

    def update_active_set(self, active_set):
        self._active_set = np.array(active_set, dtype=bool)
        self.sv.update_active_set(self._active_set)

    def update_derivs(self, iterate):
        self.sv.update_derivs(O.PointData(self.problem, iterate.x, iterate.y))
        self._jac, self._hess = self.sv.jac, self.sv.hess

    def reset_deriv(self):
        pass

    def reduced_dims(self):
        raise NotImplementedError()

    def kkt_matrix(self):
        raise NotImplementedError()

    def refinement_stats(self):
        return 0, 0, 0.0

    def solver_for_tests(self):
        raise NotImplementedError()

    def solve(self, iterate):
        xn, yn, diff = self.sv.solve(O.PointData(self.problem, iterate.x, iterate.y))
        rec = self.sv.record
        return RefStepResult(iterate, rec["dx"], rec["dy"], rec["mask"])


missing = public(ours.HipStepSolver) - public(SurfaceOracleStepSolver)
extra = public(SurfaceOracleStepSolver) - public(ours.HipStepSolver) - {"sv", "dt", "rho", "orig_iterate"}
if missing or extra:
    problems.append(f"stand-in surface differs from HipStepSolver: missing {sorted(missing)}, extra {sorted(extra)}")

spec = importlib.util.spec_from_file_location("ref_rosenbrock", os.path.join(REF, "docs", "rosenbrock.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
prob = mod.Rosenbrock()
import logging  # noqa: E402

logging.disable(logging.CRITICAL)
result = RefSolver(prob, RefParams(step_solver=SurfaceOracleStepSolver)).solve(np.array([0.0, 0.0]), np.array([]))
print(f"reference Solver over the stand-in, Rosenbrock: iterations {result.iterations}, accepted "
      f"{result.num_accepted_steps}, x = {result.x}")
if result.iterations != 30 or result.num_accepted_steps != 25:
    problems.append(f"Rosenbrock: {result.iterations} iterations / {result.num_accepted_steps} accepted, "
                    "docs/solve_rosenbrock.output says 30 / 25")
if not np.allclose(result.x, [1.0, 1.0], atol=1e-5):
    problems.append(f"Rosenbrock: x = {result.x}")

if problems:
    print("SURFACE MISMATCH:")
    for p_ in problems:
        print("  -", p_)
    sys.exit(1)
print("plugin surface ok")
