"""Host-side parameter shapes for the hot path.

Only the knobs the Newton/KKT path reads are present; names and enum member
names are the reference's (``pygradflow/params.py:14-303``) so that a
reference ``Params`` object can be passed wherever this one is accepted and
vice versa (the code below always compares enums by ``.name``).
"""

from __future__ import annotations

import enum
from dataclasses import dataclass
from typing import Any, Callable, Optional

import numpy as np


class NewtonType(enum.Enum):
    Simplified = enum.auto()
    Full = enum.auto()
    ActiveSet = enum.auto()
    Globalized = enum.auto()


class StepSolverType(enum.Enum):
    Standard = enum.auto()
    Extended = enum.auto()
    Symmetric = enum.auto()
    Asymmetric = enum.auto()


class LinearSolverType(enum.Enum):
    LU = enum.auto()
    MINRES = enum.auto()
    GMRES = enum.auto()
    Cholesky = enum.auto()
    MA57 = enum.auto()
    MUMPS = enum.auto()
    SSIDS = enum.auto()


class Precision(enum.Enum):
    Single = enum.auto()
    Double = enum.auto()


class ActiveSetType(enum.Enum):
    Standard = enum.auto()
    SmallestActiveSet = enum.auto()
    LargestActiveSet = enum.auto()
    Explicit = enum.auto()


def enum_name(value) -> str:
    """Name of an enum member of either this module or the reference's."""
    return value.name if hasattr(value, "name") else str(value)


@dataclass
class Params:
    rho: float = 1e-8
    newton_type: NewtonType = NewtonType.Simplified
    newton_tol: float = 1e-8
    step_solver: Optional[Callable[..., Any]] = None
    step_solver_type: StepSolverType = StepSolverType.Symmetric
    linear_solver_type: LinearSolverType = LinearSolverType.LU
    precision: Precision = Precision.Double
    active_set_type: ActiveSetType = ActiveSetType.Standard
    active_set_method: Optional[Callable[..., float]] = None
    active_set_tau: Optional[float] = None
    report_rcond: bool = False
    inertia_correction: bool = False
    active_tol: float = 1e-8
    # step-size control (reference params.py:206-217)
    theta_max: float = 0.9
    theta_ref: float = 0.5
    lamb_init: float = 1.0
    lamb_min: float = 1e-12
    lamb_max: float = 1e12
    lamb_inc: float = 2.0
    lamb_red: float = 0.5
    K_P: float = 0.2
    K_I: float = 0.005

    def __post_init__(self):
        for key, typ in (
            ("newton_type", NewtonType),
            ("step_solver_type", StepSolverType),
            ("linear_solver_type", LinearSolverType),
            ("precision", Precision),
            ("active_set_type", ActiveSetType),
        ):
            val = getattr(self, key)
            if isinstance(val, str):
                setattr(self, key, typ[val])

    @property
    def dtype(self):
        return np.float32 if enum_name(self.precision) == "Single" else np.float64
