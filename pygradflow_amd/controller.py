"""PI controllers for the step size, host side.

Mirrors the reference's ``pygradflow/controller.py:7-77`` (``ControllerSettings``,
``Controller``, ``LogController``): the distance-ratio step controller feeds the measured
contraction ``theta`` into a PI law on the log scale and divides ``lambda = 1/dt`` by the
result.
"""

from __future__ import annotations

import math
from dataclasses import dataclass


@dataclass
class ControllerSettings:
    K_P: float = 0.0
    K_I: float = 0.0
    lamb_init: float = 0.0
    lamb_red: float = 0.0

    def __post_init__(self):
        if self.K_P < 0.0 or self.K_I < 0.0:
            raise AssertionError("controller gains must be non-negative")

    @staticmethod
    def from_params(params) -> "ControllerSettings":
        return ControllerSettings(K_P=params.K_P, K_I=params.K_I, lamb_init=params.lamb_init,
                                  lamb_red=params.lamb_red)


class Controller:
    """u_k = K_P e_k + K_I sum_{i<=k} e_i with e = ref - measured (controller.py:29-51)."""

    def __init__(self, settings: ControllerSettings, ref: float):
        self.settings = settings
        self.ref = ref
        self.value = settings.lamb_init
        self.error_sum = 0.0

    def reset(self):
        self.error_sum = 0.0

    def update(self, val: float) -> float:
        err = self.ref - val
        self.error_sum += err
        self.value = self.settings.K_P * err + self.settings.K_I * self.error_sum
        return self.value


class LogController:
    """The same law on logarithms; ``update`` returns exp(u_k) (controller.py:54-77).

    As in the reference, ``error_sum`` of this wrapper itself stays 0 (the integral lives in
    the inner controller), so the distance-ratio controller's ``error_sum > 0`` reset test
    never fires -- kept for parity.
    """

    def __init__(self, settings: ControllerSettings, ref: float):
        if not ref > 0.0:
            raise AssertionError("reference value must be positive")
        self.settings = settings
        self.controller = Controller(settings, math.log(ref))
        self.ref = ref
        self.error_sum = 0.0

    @property
    def value(self) -> float:
        return math.exp(self.controller.value)

    def update(self, val: float) -> float:
        if not val > 0.0:
            raise AssertionError("measured value must be positive")
        self.controller.update(math.log(val))
        return self.value
