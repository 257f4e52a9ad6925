"""Host-side plan of the sparse (banded) mode.

For a FIXED sparsity pattern of H (n x n) and J (m x n) the plan holds a
bandwidth-reducing symmetric permutation of the full (n + m) KKT pattern (reverse
Cuthill-McKee, scipy.sparse.csgraph) and, for every stored entry of H and J, its slot in
the lower band of the permuted matrix.  This is host logic done once per pattern; all
per-step arithmetic (CSR products, band assembly, banded LDL^T, solves, update) runs in
libpgf_hip.so (csrc/pgf_sparse.hip).  Active variables keep their row / column as an
identity row, so the plan survives any churn of the active-set mask.

Replaces, for sparse problems, the scipy slicing + ``bmat`` assembly of the reference
(``symmetric_step_solver.py:27-39, 49-77``).
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sps
from scipy.sparse.csgraph import reverse_cuthill_mckee

from . import _lib

MAX_BANDWIDTH = 10  # kernels map one lane to one (row, column) pair of the band window


class BandPlan:
    def __init__(self, hess, jac, n, m):
        H = sps.csr_matrix(hess, dtype=np.float64)
        J = sps.csr_matrix(jac, dtype=np.float64) if m > 0 else sps.csr_matrix((0, n), dtype=np.float64)
        H.sum_duplicates()
        J.sum_duplicates()
        H.sort_indices()
        J.sort_indices()
        if H.shape != (n, n) or J.shape != (m, n):
            raise ValueError("derivative shapes do not match the problem")
        self.n, self.m = n, m
        N = n + m
        # symmetric pattern of the full KKT matrix (H's pattern symmetrised, diagonal present)
        Hp = sps.csr_matrix((np.ones(H.nnz), H.indices, H.indptr), shape=(n, n))
        Jp = sps.csr_matrix((np.ones(J.nnz), J.indices, J.indptr), shape=(m, n))
        pat = sps.bmat([[Hp + Hp.T + sps.identity(n), Jp.T], [Jp, sps.identity(m)]], format="csr")
        perm = reverse_cuthill_mckee(pat, symmetric_mode=True)  # perm[new] = old
        pos = np.empty(N, dtype=np.int64)
        pos[perm] = np.arange(N)
        coo = pat.tocoo()
        self.bw = int(np.max(np.abs(pos[coo.row] - pos[coo.col]))) if coo.nnz else 0
        self.ldb = (self.bw + 1 + 1) // 2 * 2
        self.pos = pos.astype(np.int32)
        # H entries: lower part in permuted order gets a slot, the mirror is skipped.  An
        # unsymmetric *pattern* (entry (i, j) stored without (j, i)) keeps its only copy.
        hrow = np.repeat(np.arange(n), np.diff(H.indptr))
        hcol = H.indices
        pi, pj = pos[hrow], pos[hcol]
        lower = pi >= pj
        mirror_present = np.asarray(Hp[hcol, hrow]).ravel() > 0
        use = lower | ~mirror_present
        a, b = np.maximum(pi, pj), np.abs(pi - pj)
        self.Hslot = np.where(use, a * self.ldb + b, -1).astype(np.int32)
        self.Hptr = H.indptr.astype(np.int32)
        self.Hrow = hrow.astype(np.int32)
        self.Hcol = hcol.astype(np.int32)
        # J entries: K[n + r][j] or its mirror, whichever is below the diagonal
        jrow = np.repeat(np.arange(m), np.diff(J.indptr))
        jcol = J.indices
        pa, pb = pos[n + jrow], pos[jcol]
        self.Jslot = (np.maximum(pa, pb) * self.ldb + np.abs(pa - pb)).astype(np.int32)
        self.Jptr = J.indptr.astype(np.int32)
        self.Jcol = jcol.astype(np.int32)
        # column-ordered copy of J's pattern for J' w without atomics
        order = np.lexsort((jrow, jcol))
        self.JTrow = jrow[order].astype(np.int32)
        self.JTmap = order.astype(np.int32)
        self.JTptr = np.concatenate(([0], np.cumsum(np.bincount(jcol, minlength=n)))).astype(np.int32)
        self.nnzH, self.nnzJ = int(H.nnz), int(J.nnz)
        self._Hpat = (H.indptr.copy(), H.indices.copy())
        self._Jpat = (J.indptr.copy(), J.indices.copy())

    @property
    def supported(self) -> bool:
        return self.bw <= MAX_BANDWIDTH

    def values(self, hess, jac):
        """Values of H, J in plan order; raises if the pattern differs from the plan's."""
        H = sps.csr_matrix(hess, dtype=np.float64)
        H.sum_duplicates()
        H.sort_indices()
        if H.nnz != self.nnzH or not (np.array_equal(H.indptr, self._Hpat[0])
                                      and np.array_equal(H.indices, self._Hpat[1])):
            H = self._conform(H, self._Hpat, (self.n, self.n))
        if self.m > 0:
            J = sps.csr_matrix(jac, dtype=np.float64)
            J.sum_duplicates()
            J.sort_indices()
            if J.nnz != self.nnzJ or not (np.array_equal(J.indptr, self._Jpat[0])
                                          and np.array_equal(J.indices, self._Jpat[1])):
                J = self._conform(J, self._Jpat, (self.m, self.n))
            jv = np.ascontiguousarray(J.data, dtype=np.float64)
        else:
            jv = np.zeros(0)
        return np.ascontiguousarray(H.data, dtype=np.float64), jv

    @staticmethod
    def _conform(mat, pat, shape):
        """Re-express ``mat`` on the plan's pattern (entries outside it are an error)."""
        ptr, idx = pat
        base = sps.csr_matrix((np.zeros(len(idx)), idx, ptr), shape=shape)
        full = (base + mat).tocsr()
        full.sort_indices()
        if full.nnz != len(idx):
            raise ValueError("sparsity pattern changed: rebuild the band plan")
        out = sps.csr_matrix((np.zeros(len(idx)), idx, ptr), shape=shape)
        out.data = np.asarray(full.data, dtype=np.float64)
        return out

    def upload(self, lib, handle):
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))  # noqa: E731
        rc = lib.pgf_sparse_set_pattern(
            handle, self.bw, ip(self.pos), self.nnzH, ip(self.Hptr), ip(self.Hrow), ip(self.Hcol),
            ip(self.Hslot), self.nnzJ, ip(self.Jptr), ip(self.Jcol), ip(self.Jslot), ip(self.JTptr),
            ip(self.JTrow), ip(self.JTmap))
        _lib.check(rc, handle, "pgf_sparse_set_pattern")


def wants_band(problem, hess, n, m, dense_limit=20000):
    """Sparse derivatives take the banded path when the problem is too large for the dense
    one, or when the problem asks for it (``pgf_force_band``; tests use that)."""
    if not sps.issparse(hess):
        return False
    return bool(getattr(problem, "pgf_force_band", False)) or (n + m > dense_limit)
