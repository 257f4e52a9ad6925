"""Penalty-parameter updates between accepted outer steps (reference
``pygradflow/penalty.py:12-74``): ``ConstantPenalty`` and the default ``DualNormUpdate``,
which keeps ``rho`` within a factor 10 of ``||y||_inf``.  ``update_from_norm`` takes the norm
directly, e.g. ``measures()[3]`` of a device-resident point.
"""

from __future__ import annotations

import numpy as np


class PenaltyResult:
    def __init__(self, next_rho, accept):
        self.next_rho = next_rho
        self.accept = accept

    @staticmethod
    def accept_with_penalty(next_rho):
        return PenaltyResult(next_rho, True)

    @staticmethod
    def reject_with_penalty(next_rho):
        return PenaltyResult(next_rho, False)


class PenaltyStrategy:
    def __init__(self, problem, params):
        self.problem = problem
        self.params = params

    def initial(self, iterate):
        return self.params.rho

    def update(self, prev_iterate, next_iterate):
        raise NotImplementedError()


class ConstantPenalty(PenaltyStrategy):
    def update(self, prev_iterate, next_iterate):
        return PenaltyResult.accept_with_penalty(self.params.rho)


class DualNormUpdate(PenaltyStrategy):
    def initial(self, iterate):
        self.rho = self.params.rho
        return self.rho

    def update_from_norm(self, ynorm):
        if self.problem.num_cons == 0:
            return PenaltyResult.accept_with_penalty(self.rho)
        if ynorm >= 10.0 * self.rho:
            self.rho = min(ynorm, 10.0 * self.rho)
        return PenaltyResult.accept_with_penalty(self.rho)

    def update(self, prev_iterate, next_iterate):
        y = next_iterate.y
        return self.update_from_norm(float(np.linalg.norm(y, ord=np.inf)) if y.size else 0.0)
