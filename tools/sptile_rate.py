"""What the tiled sparse products (thip_sptile.hip) stream on three sparsity patterns: TB/s of stored entries (8 B each) for the
T product (A^T [y0 y1]) and the N product (A [x0 x1]), beside the round-5 CSR gathers (thip_spmv_csr, one right-hand side per call, two
copies of the matrix).  profiles/r06_sparse_product_rates.txt.
    python tools/sptile_rate.py"""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import totsu_amd as T  # noqa: E402
from totsu_amd import _lib  # noqa: E402
from totsu_amd._lib import lib  # noqa: E402
from totsu_amd.sparse import SpTile, _Csr  # noqa: E402


def csr_time(mat, reps=5):
    a = _Csr(mat)
    x = T.DeviceBuffer.from_host(np.ones(mat.shape[1], np.float32))
    y = T.DeviceBuffer(mat.shape[0])
    best = 1e30
    for r in range(reps + 1):
        lib.thip_sync()
        t0 = time.perf_counter()
        lib.thip_spmv_csr(mat.shape[0], mat.shape[1], a.nnz, a.rowptr.ptr, a.colidx.ptr, a.vals.ptr, 1.0, x.ptr, 0.0, y.ptr, 0)
        lib.thip_sync()
        if r:
            best = min(best, time.perf_counter() - t0)
    a.free(); x.free(); y.free()
    return best * 1e3


def main():
    _lib.init()
    rng = np.random.default_rng(0)
    cases = []
    # (a) the scaled l1reg_lp matrix's shape of sparsity: a dense 2l x l block between diagonals
    l = 8192
    blk = sp.csc_matrix(rng.standard_normal((2 * l, l)).astype(np.float32))
    d = sp.identity(l, dtype=np.float32, format="csc")
    top = sp.hstack([-d, blk[:l], sp.csc_matrix((l, l), dtype=np.float32)])
    A = sp.vstack([top, sp.hstack([-d, blk[l:], sp.csc_matrix((l, l), dtype=np.float32)]),
                   sp.hstack([sp.csc_matrix((l, l), dtype=np.float32), d, -d])]).tocsc()
    cases.append(("dense 16384 x 8192 block between diagonals (l1reg_lp's pattern)", A))
    # (b) uniformly random, 1 % dense
    m, n = 200_000, 100_000
    nnz = int(0.01 * m * n)
    r = rng.integers(0, m, nnz, dtype=np.int64)
    c = rng.integers(0, n, nnz, dtype=np.int64)
    B = sp.csc_matrix((rng.standard_normal(nnz).astype(np.float32), (r, c)), shape=(m, n))
    B.sum_duplicates()
    cases.append(("uniformly random, 1 %% dense, %d x %d" % (m, n), B))
    # (c) 2-D 5-point Laplacian on a 3000 x 3000 grid (9 M x 9 M, 45 M entries): the matrix-free pattern of imgnr_udef
    g = 3000
    e = np.ones(g, np.float32)
    L1 = sp.diags([-e[:-1], 2 * e, -e[:-1]], [-1, 0, 1], format="csc")
    Lp = (sp.kron(sp.identity(g, dtype=np.float32), L1) + sp.kron(L1, sp.identity(g, dtype=np.float32))).tocsc().astype(np.float32)
    cases.append(("5-point Laplacian on a %d x %d grid" % (g, g), Lp))
    for name, M in cases:
        M.sort_indices()
        t = SpTile(M)
        info = t.info()
        ms = (C.c_float * 4)()
        lib.thip_test_sptile_time(t.h, 5, ms)
        by = float(info["bytes_per_product"])
        tcsr_n = csr_time(M.tocsr())
        tcsr_t = csr_time(M.T.tocsr())
        print("%s: nnz %d (stored %d), %d tiles (%d without indices), slices N / T %d / %d, %.3f GB per product"
              % (name, M.nnz, info["nnz_stored"], info["tiles"], info["dense_tiles"], info["slices_n"], info["slices_t"], by / 1e9))
        print("    tiled copy, two right-hand sides:  T product %8.3f ms = %6.2f TB/s of entries   N product %8.3f ms = %6.2f TB/s"
              % (ms[0], by / ms[0] / 1e9, ms[1], by / ms[1] / 1e9))
        print("    CSR gathers, one right-hand side:  A^T (own copy) %8.3f ms = %6.2f TB/s   A %8.3f ms = %6.2f TB/s"
              % (tcsr_t, 8.0 * M.nnz / tcsr_t / 1e9, tcsr_n, 8.0 * M.nnz / tcsr_n / 1e9))
        t.free()


if __name__ == "__main__":
    main()
