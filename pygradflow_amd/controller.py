"""PI controllers for the step size, host side.

Same interface and arithmetic as the reference's ``pygradflow/controller.py:7-77``
(``ControllerSettings``, ``Controller``, ``LogController``): the distance-ratio step
controller feeds the measured contraction ``theta`` into a PI law on the log scale and
divides ``lambda = 1/dt`` by what comes back.
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
        if min(self.K_P, self.K_I) < 0.0:
            raise AssertionError("controller gains must be non-negative")

    @staticmethod
    def from_params(params) -> "ControllerSettings":
        fields = {name: getattr(params, name) for name in ("K_P", "K_I", "lamb_init", "lamb_red")}
        return ControllerSettings(**fields)


class _PILaw:
    """u_k = K_P e_k + K_I (e_0 + ... + e_k), e = target - measured."""

    def __init__(self, gains: ControllerSettings, target: float):
        self.k_p, self.k_i = gains.K_P, gains.K_I
        self.target = target
        self.integral = 0.0

    def feed(self, measured: float) -> float:
        e = self.target - measured
        self.integral += e
        return self.k_p * e + self.k_i * self.integral


class Controller:
    """Plain PI controller (reference controller.py:29-51): ``value`` starts at
    ``settings.lamb_init`` and holds the last output afterwards."""

    def __init__(self, settings: ControllerSettings, ref: float):
        self.settings = settings
        self.ref = ref
        self._law = _PILaw(settings, ref)
        self.value = settings.lamb_init

    @property
    def error_sum(self) -> float:
        return self._law.integral

    def reset(self):
        self._law.integral = 0.0

    def update(self, val: float) -> float:
        self.value = self._law.feed(val)
        return self.value


class LogController:
    """The same law on logarithms; ``update`` returns ``exp(u_k)`` (reference
    controller.py:54-77).  As there, ``error_sum`` of this object stays 0 -- the integral
    lives in the law underneath -- so the distance-ratio controller's
    ``error_sum > 0 -> reset`` branch never fires; kept that way for parity."""

    error_sum = 0.0

    def __init__(self, settings: ControllerSettings, ref: float):
        if not ref > 0.0:
            raise AssertionError("reference value must be positive")
        self.settings = settings
        self.ref = ref
        self._law = _PILaw(settings, math.log(ref))
        self._log_value = settings.lamb_init

    @property
    def value(self) -> float:
        return math.exp(self._log_value)

    def update(self, val: float) -> float:
        if not val > 0.0:
            raise AssertionError("measured value must be positive")
        self._log_value = self._law.feed(math.log(val))
        return self.value
