"""Exception convention of the hot path.

``LinearSolverError`` (reference ``pygradflow/linear_solver/linear_solver.py:8-15``)
signals a failed factorisation; the step solver re-raises it as
``StepSolverError`` (``pygradflow/step/step_solver_error.py:1-7``,
``symmetric_step_solver.py:155-156``), which the reference's step controllers turn
into "reject, lambda <- 2 lambda" (``step/step_control.py:80-107``).  When the
reference is importable the classes below derive from its own, so its ``except``
clauses keep catching them; otherwise they stand alone.
"""

try:  # drop-in inside a pygradflow installation
    from pygradflow.linear_solver.linear_solver import LinearSolverError as _RefLinearSolverError
    from pygradflow.step.step_solver_error import StepSolverError as _RefStepSolverError
except Exception:  # stand-alone (e.g. on the GPU box)
    _RefLinearSolverError = Exception
    _RefStepSolverError = Exception
try:
    from pygradflow.eval import EvalError as _RefEvalError
except Exception:
    _RefEvalError = ValueError


class LinearSolverError(_RefLinearSolverError):
    """The linear solver failed, e.g. because the matrix is (near) singular."""


class StepSolverError(_RefStepSolverError):
    """The step solver failed, e.g. because the Newton matrix is (near) singular."""


class EvalError(_RefEvalError):
    """A problem callback failed or returned non-finite values at ``x`` (reference
    ``pygradflow/eval.py:18-21``); step controllers reject the step and halve ``dt``
    (``step/step_control.py:103-107``)."""

    def __init__(self, msg, x=None):
        self.x = x
        ValueError.__init__(self, msg)


# what StepController.compute_step turns into "reject, lambda <- 2 lambda": our classes and,
# when the reference is importable, its own (an evaluator of the reference raises those)
STEP_FAILURES = tuple({StepSolverError, EvalError, _RefEvalError} - {ValueError, Exception})
