"""``LinearSolver``-shaped front end of the dense LDL^T factor on the GPU.

Mirrors the reference ABC (``pygradflow/linear_solver/linear_solver.py:18-31``):
the constructor factorises and raises ``LinearSolverError`` on failure;
``solve(rhs, trans=False, initial_sol=None)``, ``num_neg_eigvals()``, ``rcond()``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sps

from . import _lib


class HipLinearSolver:
    def __init__(self, matrix, symmetric: bool = False, device: int = 0):
        _lib.require_gpu()
        self.symmetric = symmetric
        dense = matrix.toarray() if sps.issparse(matrix) else np.asarray(matrix)
        dense = _lib.as_f64(dense)
        if dense.ndim != 2 or dense.shape[0] != dense.shape[1]:
            raise ValueError("square matrix expected")
        # symmetric=True: LDL^T of the lower triangle (the Symmetric step solver's reduced KKT
        # matrix); symmetric=False: LU with partial pivoting of the full matrix, which is what
        # the reference's LUSolver does for every matrix (lu_solver.py:9-17)
        self.shape = dense.shape
        self._lib = _lib.load()
        self._h = C.c_void_p()
        n = dense.shape[0]
        rc = self._lib.pgf_ls_create_dense(n, _lib.dptr(dense), max(n, 1), int(bool(symmetric)),
                                           device, C.byref(self._h))
        if rc != _lib.PGF_OK:
            self._h = C.c_void_p()
        what = "LDL^T" if symmetric else "LU"
        _lib.check(rc, None, f"{what} factorisation failed" if rc == _lib.PGF_SINGULAR else "pgf_ls_create_dense")

    def solve(self, rhs, trans: bool = False, initial_sol=None):
        rhs = _lib.as_f64(rhs)
        if rhs.shape != (self.shape[0],):
            raise ValueError("rhs shape mismatch")
        sol = np.empty_like(rhs)
        rc = self._lib.pgf_ls_solve(self._h, _lib.dptr(rhs), int(bool(trans)), _lib.dptr(sol))
        _lib.check(rc, None, "pgf_ls_solve")
        return sol

    def num_neg_eigvals(self):
        if not self.symmetric:
            return None  # an LU carries no inertia (reference LUSolver: base-class None)
        out = C.c_int(0)
        _lib.check(self._lib.pgf_ls_num_neg(self._h, C.byref(out)), None, "pgf_ls_num_neg")
        return out.value

    def factor_matrix(self):
        """LDL^T: unit-lower L (below the diagonal) and D (on it); LU: L below, U on and above
        the diagonal (rows in pivoted order) -- as factored on the device."""
        n = self.shape[0]
        out = np.zeros((n, n))
        if n:
            _lib.check(self._lib.pgf_ls_get_factor(self._h, _lib.dptr(out), n), None, "pgf_ls_get_factor")
        return np.tril(out) if self.symmetric else out

    def rcond(self):
        return None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.pgf_ls_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
