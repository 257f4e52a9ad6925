"""MI355X-native semi-smooth Newton / KKT step for pygradflow's plugin surface.

``HipStepSolver`` / ``HipLinearSolver`` are the StepSolver / LinearSolver pair
(``Params(step_solver=HipStepSolver)``); ``newton_method`` mirrors the reference's
policy factory; ``DeviceNewton`` is the HBM-resident driver for linear-quadratic
problems, ``BatchedDeviceNewton`` its many-instance form; ``DistanceRatioController`` is the
reference's default step controller on top of either.  All arithmetic is in ``libpgf_hip.so`` (C ABI: ``include/pgf_hip.h``);
there is no CPU fallback.
"""

from .errors import EvalError, LinearSolverError, StepSolverError  # noqa: F401
from .params import (  # noqa: F401
    ActiveSetType,
    LinearSolverType,
    NewtonType,
    Params,
    Precision,
    StepSolverType,
)
from .iterate import Iterate  # noqa: F401


def __getattr__(name):
    # the HIP-backed classes load the shared library on first use
    if name in ("HipStepSolver", "StepResult", "HipStepFunc"):
        from . import step_solver

        return getattr(step_solver, name)
    if name == "HipLinearSolver":
        from .linear_solver import HipLinearSolver

        return HipLinearSolver
    if name in ("newton_method", "newton_steps", "DeviceNewton", "SimplifiedNewtonMethod",
                "FullNewtonMethod", "ActiveSetNewtonMethod", "GlobalizedNewtonMethod"):
        from . import newton

        return getattr(newton, name)
    if name in ("DistanceRatioController", "DeviceDistanceRatioController", "NewtonController",
                "BatchedDistanceRatioController", "DeviceResidentDistanceRatioController",
                "StepController", "StepControlResult", "gradient_flow"):
        from . import step_control

        return getattr(step_control, name)
    if name in ("StandardStepSolver", "ExtendedStepSolver", "AsymmetricStepSolver", "step_solver",
                "UnscaledStepFunc"):
        from . import unsym_step_solvers

        return getattr(unsym_step_solvers, name)
    if name == "BatchedDeviceNewton":
        from .batched import BatchedDeviceNewton

        return BatchedDeviceNewton
    raise AttributeError(name)
