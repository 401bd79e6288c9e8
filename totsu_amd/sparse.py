"""`SparseMatOp`: a CSR matrix as a linear `Operator` (totsu_core/src/solver/operator.rs:11-156) for the trait-level
`Solver(F32HIP)` -- the user-defined-operator pattern of examples/imgnr_udef/src/prob_op_a.rs with the matrix kept
sparse on the device.  Both A and A^T are stored in CSR so that `op` and `trans_op` are deterministic gathers."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class _DevInts:
    def __init__(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        self.nbytes = a.nbytes
        nfl = max((a.nbytes + 3) // 4, 1)
        p = C.c_void_p()
        lib.thip_alloc(nfl, C.byref(p))
        self.ptr = p.value
        if a.nbytes:
            pad = np.zeros(nfl * 4, dtype=np.uint8)
            pad[:a.nbytes] = a.view(np.uint8)
            lib.thip_h2d(self.ptr, pad.ctypes.data, nfl)

    def free(self):
        if self.ptr is not None:
            lib.thip_free(self.ptr)
            self.ptr = None


class _Csr:
    def __init__(self, m):
        m = m.tocsr()
        m.sort_indices()
        self.shape = m.shape
        self.nnz = int(m.nnz)
        self.rowptr = _DevInts(m.indptr, np.int64)
        self.colidx = _DevInts(m.indices, np.int32)
        self.vals = _DevInts(m.data.astype(np.float32).view(np.int32), np.int32)

    def mv(self, alpha, x, beta, y, abs_mode=0):
        # abs mode ignores x (taken as all-ones): pass a pointer that does not alias y
        xp = self.vals.ptr if abs_mode else x.dev()
        lib.thip_spmv_csr(self.shape[0], self.shape[1], self.nnz, self.rowptr.ptr, self.colidx.ptr, self.vals.ptr,
                          float(alpha), xp, float(beta), y.dev(), abs_mode)

    def free(self):
        for d in (self.rowptr, self.colidx, self.vals):
            d.free()


class SparseMatOp:
    """Operator over a scipy.sparse matrix, for L = F32HIP."""

    def __init__(self, L, mat):
        assert getattr(L, "name", "") == "F32HIP"
        _lib.ensure_init()
        self.L = L
        self.n_row, self.n_col = mat.shape
        self.a = _Csr(mat)
        self.at = _Csr(mat.T)

    def size(self):
        return (self.n_row, self.n_col)

    def op(self, alpha, x, beta, y):
        assert x.len() == self.n_col and y.len() == self.n_row
        if self.n_row and self.n_col:
            self.a.mv(alpha, x, beta, y)
        else:
            self.L.scale(beta, y)

    def trans_op(self, alpha, x, beta, y):
        assert x.len() == self.n_row and y.len() == self.n_col
        if self.n_row and self.n_col:
            self.at.mv(alpha, x, beta, y)
        else:
            self.L.scale(beta, y)

    def absadd_cols(self, tau):          # tau[c] += sum_r |A(r,c)|  (operator.rs:82-113 reference semantics)
        assert tau.len() == self.n_col
        if self.n_row and self.n_col:
            self.at.mv(1.0, None, 1.0, tau, abs_mode=1)

    def absadd_rows(self, sigma):
        assert sigma.len() == self.n_row
        if self.n_row and self.n_col:
            self.a.mv(1.0, None, 1.0, sigma, abs_mode=1)

    def drop(self):
        self.a.free()
        self.at.free()
