"""`SparseMatOp`: a sparse matrix as a linear `Operator` (totsu_core/src/solver/operator.rs:11-156) for the trait-level
`Solver(F32HIP)` -- the user-defined-operator pattern of examples/imgnr_udef/src/prob_op_a.rs with the matrix kept
sparse on the device.  Default: ONE tiled copy serving `op` and `trans_op` (`SpTile`, thip_sptile_*); `two_copies=True`
keeps the round-5 form, CSR of A and of A^T (one right-hand side per call; the A/B leg of bench.py's sparse workloads).
Both forms give bitwise reproducible sums."""
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


class SpTile:
    """A sparse matrix held ONCE on the device in 4096 x 4096 tiles (thip_sptile_*, totsu_amd/csrc/thip_sptile.hip): both
    products stream the same stored entries.  Built from a scipy.sparse matrix through its CSC arrays on the host."""

    def __init__(self, mat):
        _lib.ensure_init()
        m = mat.tocsc()
        m.sort_indices()
        self.shape = m.shape
        self.nnz = int(m.nnz)
        colptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        rowidx = np.ascontiguousarray(m.indices, dtype=np.int32)
        vals = np.ascontiguousarray(m.data, dtype=np.float32)
        h = C.c_void_p()
        lib.thip_sptile_create(self.shape[0], self.shape[1], self.nnz, colptr.ctypes.data, rowidx.ctypes.data,
                               vals.ctypes.data, C.byref(h))
        self.h = h

    @staticmethod
    def from_csc_arrays(n_row, n_col, colptr, rowidx, vals):
        """host CSC arrays as they are (int64 / int32 / float32): no scipy object in between (bench.py's GB-sized operators)"""
        _lib.ensure_init()
        self = SpTile.__new__(SpTile)
        self.shape = (int(n_row), int(n_col))
        self.nnz = int(vals.size)
        assert colptr.dtype == np.int64 and rowidx.dtype == np.int32 and vals.dtype == np.float32
        assert colptr.flags.c_contiguous and rowidx.flags.c_contiguous and vals.flags.c_contiguous
        h = C.c_void_p()
        lib.thip_sptile_create(self.shape[0], self.shape[1], self.nnz, colptr.ctypes.data, rowidx.ctypes.data,
                               vals.ctypes.data, C.byref(h))
        self.h = h
        return self

    def mv(self, transpose, alpha, x, beta, y, abs_mode=0):
        xp = y.dev() if abs_mode else x.dev()          # abs mode ignores x (taken as all-ones)
        lib.thip_sptile_mv(self.h, 1 if transpose else 0, float(alpha), xp, float(beta), y.dev(), abs_mode)

    def info(self):
        nz, by = C.c_size_t(), C.c_size_t()
        t, i_n, i_t, s_n, s_t = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib.thip_sptile_info(self.h, C.byref(nz), C.byref(t), C.byref(i_n), C.byref(i_t), C.byref(s_n), C.byref(s_t), C.byref(by))
        nd, ni, bp = C.c_int(), C.c_size_t(), C.c_size_t()
        lib.thip_sptile_layout(self.h, C.byref(nd), C.byref(ni), C.byref(bp))
        return {"nnz": self.nnz, "nnz_stored": nz.value, "tiles": t.value, "items_n": i_n.value, "items_t": i_t.value,
                "slices_n": s_n.value, "slices_t": s_t.value, "device_bytes": by.value,
                "dense_tiles": nd.value, "indexed_entries": ni.value, "bytes_per_product": bp.value}

    def free(self):
        if getattr(self, "h", None) is not None:
            lib.thip_sptile_destroy(self.h)
            self.h = None


class SparseMatOp:
    """Operator over a scipy.sparse matrix, for L = F32HIP."""

    def __init__(self, L, mat, two_copies=False):
        """two_copies: the round-5 form (CSR of A and CSR of A^T, both gathers deterministic); default: one tiled copy"""
        assert getattr(L, "name", "") == "F32HIP"
        _lib.ensure_init()
        self.L = L
        self.n_row, self.n_col = mat.shape
        self.t = None if two_copies else SpTile(mat)
        self.a = _Csr(mat) if two_copies else None
        self.at = _Csr(mat.T) if two_copies else None

    def size(self):
        return (self.n_row, self.n_col)

    def op(self, alpha, x, beta, y):
        assert x.len() == self.n_col and y.len() == self.n_row
        if self.n_row and self.n_col:
            self.t.mv(False, alpha, x, beta, y) if self.t else self.a.mv(alpha, x, beta, y)
        else:
            self.L.scale(beta, y)

    def trans_op(self, alpha, x, beta, y):
        assert x.len() == self.n_row and y.len() == self.n_col
        if self.n_row and self.n_col:
            self.t.mv(True, alpha, x, beta, y) if self.t else self.at.mv(alpha, x, beta, y)
        else:
            self.L.scale(beta, y)

    def absadd_cols(self, tau):          # tau[c] += sum_r |A(r,c)|  (operator.rs:82-113 reference semantics)
        assert tau.len() == self.n_col
        if self.n_row and self.n_col:
            self.t.mv(True, 1.0, None, 1.0, tau, abs_mode=1) if self.t else self.at.mv(1.0, None, 1.0, tau, abs_mode=1)

    def absadd_rows(self, sigma):
        assert sigma.len() == self.n_row
        if self.n_row and self.n_col:
            self.t.mv(False, 1.0, None, 1.0, sigma, abs_mode=1) if self.t else self.a.mv(1.0, None, 1.0, sigma, abs_mode=1)

    def drop(self):
        for d in (self.t, self.a, self.at):
            if d is not None:
                d.free()
