"""Synthetic instances generated on the device from a counter-based generator keyed by (seed, stream, index)
(SURVEY.md 8d): every GPU shard -- and the CPU baseline through oracle.gen_matrix -- regenerates bit-identical
f32 entries without shipping the matrix.

SOCP (BASELINE.json configs[2]): n variables, `n_cones` second-order cones of 1 + `ni` rows each; in the
stacked conic form  min f^T x  s.t.  A x + s = b, s in prod SOC  the rows of cone i are [-c_i^T ; -G_i]
(totsu/src/problem/socp.rs:88-93), b = [d_i ; h_i]:
    G_i, c_i ~ N(0, 1/n),  h_i ~ N(0, 1),
    d_i = ||G_i x0 + h_i|| - c_i^T x0 + U(0.1, 1.1)      strictly feasible at x0 ~ N(0, I)
    f   = sum_i (t_i c_i + G_i^T w_i), t_i ~ U(0.5, 1.5), ||w_i|| < 0.9 t_i   strictly dual feasible => bounded
LP (configs[1], experimental/benchmark_lp/src/main.rs:14-57): c = -U(0,1), G = [-I ; U(0,1)], h = [0 ; U(0,1)].
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import lib
from .fused import DeviceBuffer

STREAM_A, STREAM_X0, STREAM_H, STREAM_D, STREAM_T, STREAM_W, STREAM_WS, STREAM_C = 1, 2, 3, 4, 5, 6, 7, 8


def shard_cones(n_cones, world, rank):
    """contiguous, cone-aligned row blocks (no cone straddles a boundary)"""
    base, rem = divmod(n_cones, world)
    c0 = rank * base + min(rank, rem)
    c1 = c0 + base + (1 if rank < rem else 0)
    return c0, c1


def _gen(n, seed, stream, idx0, kind, scale=1.0, shift=0.0):
    d = DeviceBuffer(n)
    lib.thip_gen_vector(d.ptr, n, seed, stream, idx0, kind, scale, shift)
    h = d.to_host()
    d.free()
    return h


class SocpInstance:
    """device-resident row shard of the synthetic SOCP"""

    def __init__(self, n, n_cones, ni=99, seed=0, rank=0, world=1, allreduce_host=None, first_cones=None):
        """first_cones: build the STANDALONE problem made of the first `first_cones` cones of the n_cones-cone instance
        (same matrix entries -- the generator's index still runs over all n_cones * (1 + ni) rows -- with the objective
        formed from these rows alone): the sub-instance bench.py's cpu_baseline leg times, and the one the full-n parity
        test compares with the oracle."""
        _lib.ensure_init()
        self.n, self.n_cones, self.ni, self.seed = n, n_cones, ni, seed
        rows = 1 + ni
        c0, c1 = shard_cones(n_cones, world, rank)
        if first_cones is not None:
            assert world == 1 and 0 < first_cones <= n_cones
            c0, c1 = 0, first_cones
        self.c0, self.c1 = c0, c1
        self.m_total = n_cones * rows
        self.m = (c1 - c0) * rows
        row0 = c0 * rows
        m = self.m
        self.mat_a = DeviceBuffer(max(m * n, 1))
        lib.thip_gen_matrix(self.mat_a.ptr, m, n, m, seed, STREAM_A, row0, 0, self.m_total, 1, -1.0 / math.sqrt(n), 0.0)
        # strictly feasible point and the right-hand side
        x0 = DeviceBuffer(n)
        lib.thip_gen_vector(x0.ptr, n, seed, STREAM_X0, 0, 1, 1.0, 0.0)
        w = DeviceBuffer(max(m, 1))
        lib.thip_transform_ge(0, m, n, 1.0, self.mat_a.ptr, x0.ptr, 0.0, w.ptr)
        wh = w.to_host()[:m].astype(np.float64).reshape(c1 - c0, rows)
        hh = _gen(max(m, 1), seed, STREAM_H, row0, 1)[:m].astype(np.float64).reshape(c1 - c0, rows)
        margin = _gen(max(c1 - c0, 1), seed, STREAM_D, c0, 0, 1.0, 0.1)[:c1 - c0].astype(np.float64)
        d = np.linalg.norm(hh[:, 1:] - wh[:, 1:], axis=1) + wh[:, 0] + margin
        b = hh.copy()
        b[:, 0] = d
        self.vec_b_host = b.reshape(-1).astype(np.float32)
        self.vec_b = DeviceBuffer.from_host(self.vec_b_host) if m else DeviceBuffer(1)
        # objective from a strictly dual-feasible point z = [t_i ; w_i]
        t = _gen(max(c1 - c0, 1), seed, STREAM_T, c0, 0, 1.0, 0.5)[:c1 - c0].astype(np.float64)
        wd = _gen(max(m, 1), seed, STREAM_W, row0, 1)[:m].astype(np.float64).reshape(c1 - c0, rows)[:, 1:]
        ws = _gen(max(c1 - c0, 1), seed, STREAM_WS, c0, 0)[:c1 - c0].astype(np.float64)
        wd = wd * (0.9 * t * ws / np.maximum(np.linalg.norm(wd, axis=1), 1e-9))[:, None]
        z = np.concatenate([t[:, None], wd], axis=1).reshape(-1).astype(np.float32)
        zd = DeviceBuffer.from_host(z) if m else DeviceBuffer(1)
        f = DeviceBuffer(n)
        lib.thip_transform_ge(1, m, n, -1.0, self.mat_a.ptr, zd.ptr, 0.0, f.ptr)
        fh = f.to_host()
        if allreduce_host is not None:
            fh = allreduce_host(fh)          # sum over the row shards
            lib.thip_h2d(f.ptr, fh.ctypes.data, n)
        self.vec_c_host = fh
        self.vec_c = f
        self.seg_type = [_lib.CONE_SOC] * (c1 - c0)
        self.seg_len = [rows] * (c1 - c0)
        for d_ in (x0, w, zd):
            d_.free()

    def free(self):
        for d in (self.mat_a, self.vec_b, self.vec_c):
            d.free()


def shard_cols(n, world, rank):
    """contiguous column blocks (a column-sharded sweep, thip_solver_set_column_shard)"""
    base, rem = divmod(n, world)
    c0 = rank * base + min(rank, rem)
    return c0, c0 + base + (1 if rank < rem else 0)


class SocpInstanceCols:
    """device-resident COLUMN shard of the same synthetic SOCP (identical entries: the generator is keyed by the global
    index): mat_a is m_total x n_local, vec_c this rank's block of f, vec_b and the cones the whole problem's.
    allreduce_host sums an m-vector over the ranks once, at construction (A x0)."""

    def __init__(self, n, n_cones, ni=99, seed=0, rank=0, world=1, allreduce_host=None):
        _lib.ensure_init()
        self.n, self.n_cones, self.ni, self.seed = n, n_cones, ni, seed
        rows = 1 + ni
        self.col0, self.col1 = shard_cols(n, world, rank)
        nl = self.n_local = self.col1 - self.col0
        m = self.m = self.m_total = n_cones * rows
        self.mat_a = DeviceBuffer(max(m * nl, 1))
        lib.thip_gen_matrix(self.mat_a.ptr, m, nl, m, seed, STREAM_A, 0, self.col0, m, 1, -1.0 / math.sqrt(n), 0.0)
        x0 = DeviceBuffer(max(nl, 1))
        lib.thip_gen_vector(x0.ptr, nl, seed, STREAM_X0, self.col0, 1, 1.0, 0.0)
        w = DeviceBuffer(m)
        lib.thip_transform_ge(0, m, nl, 1.0, self.mat_a.ptr, x0.ptr, 0.0, w.ptr)
        wv = w.to_host()
        if allreduce_host is not None:
            wv = allreduce_host(wv)           # A x0 summed over the column blocks
        wh = wv.astype(np.float64).reshape(n_cones, rows)
        hh = _gen(m, seed, STREAM_H, 0, 1).astype(np.float64).reshape(n_cones, rows)
        margin = _gen(n_cones, seed, STREAM_D, 0, 0, 1.0, 0.1).astype(np.float64)
        d = np.linalg.norm(hh[:, 1:] - wh[:, 1:], axis=1) + wh[:, 0] + margin
        b = hh.copy()
        b[:, 0] = d
        self.vec_b_host = b.reshape(-1).astype(np.float32)
        self.vec_b = DeviceBuffer.from_host(self.vec_b_host)
        t = _gen(n_cones, seed, STREAM_T, 0, 0, 1.0, 0.5).astype(np.float64)
        wd = _gen(m, seed, STREAM_W, 0, 1).astype(np.float64).reshape(n_cones, rows)[:, 1:]
        ws = _gen(n_cones, seed, STREAM_WS, 0, 0).astype(np.float64)
        wd = wd * (0.9 * t * ws / np.maximum(np.linalg.norm(wd, axis=1), 1e-9))[:, None]
        z = np.concatenate([t[:, None], wd], axis=1).reshape(-1).astype(np.float32)
        zd = DeviceBuffer.from_host(z)
        f = DeviceBuffer(max(nl, 1))
        lib.thip_transform_ge(1, m, nl, -1.0, self.mat_a.ptr, zd.ptr, 0.0, f.ptr)      # this rank's block of f: no exchange
        self.vec_c_host = f.to_host()[:nl]
        self.vec_c = f
        self.seg_type = [_lib.CONE_SOC] * n_cones
        self.seg_len = [rows] * n_cones
        for d_ in (x0, w, zd):
            d_.free()

    def free(self):
        for d in (self.mat_a, self.vec_b, self.vec_c):
            d.free()


class LpInstanceCols:
    """device-resident COLUMN shard of the benchmark_lp construction: mat_a = [-I ; U](:, col0:col1), 2 n x n_local"""

    def __init__(self, n, seed=0, rank=0, world=1):
        _lib.ensure_init()
        self.n, self.seed = n, seed
        m = self.m = self.m_total = 2 * n
        self.col0, self.col1 = shard_cols(n, world, rank)
        nl = self.n_local = self.col1 - self.col0
        self.mat_a = DeviceBuffer(max(m * nl, 1))
        lib.thip_gen_matrix(self.mat_a.ptr, m, nl, m, seed, STREAM_A, 0, self.col0, m, 0, 1.0, 0.0)
        # rows r < n are -I: the entry of local column c sits in row col0 + c, i.e. row0 + r == c with row0 = -col0
        lib.thip_gen_identity(self.mat_a.ptr, n, nl, m, (-self.col0) & 0xFFFFFFFFFFFFFFFF, -1.0)
        h = _gen(m, seed, STREAM_H, 0, 0)
        h[:n] = 0.0
        self.vec_b_host = h
        self.vec_b = DeviceBuffer.from_host(h)
        c = -_gen(max(nl, 1), seed, STREAM_C, self.col0, 0)[:nl]
        self.vec_c_host = c
        self.vec_c = DeviceBuffer.from_host(c) if nl else DeviceBuffer(1)
        self.seg_type = [_lib.CONE_RPOS]
        self.seg_len = [m]

    def free(self):
        for d in (self.mat_a, self.vec_b, self.vec_c):
            d.free()


class LpInstance:
    """device-resident row shard of the benchmark_lp construction (rows split evenly; nonneg cone is separable)"""

    def __init__(self, n, seed=0, rank=0, world=1, bf16_direct=False, block_cols=2048):
        """bf16_direct (True / "bf16" / "f16"): build the shard as a 16-bit Bf16Matrix from f32 column blocks of
        block_cols columns (identical entries, rounded); the f32 shard is never allocated -- for shards that only fit
        HBM at 16 bits per entry."""
        _lib.ensure_init()
        self.n, self.seed = n, seed
        self.m_total = 2 * n
        base, rem = divmod(self.m_total, world)
        r0 = rank * base + min(rank, rem)
        r1 = r0 + base + (1 if rank < rem else 0)
        self.r0, self.r1 = r0, r1
        m = self.m = r1 - r0
        if bf16_direct:
            from .fused import Bf16Matrix
            self.mat_a = Bf16Matrix(m, n, "f16" if bf16_direct == "f16" else "bf16")
            blk = DeviceBuffer(max(m * min(block_cols, n), 1))
            for c0 in range(0, n, block_cols):
                nc = min(block_cols, n - c0)
                lib.thip_gen_matrix(blk.ptr, m, nc, m, seed, STREAM_A, r0, c0, self.m_total, 0, 1.0, 0.0)
                if r0 < n:      # -I rows: (r0 + r == c0 + c) in the block's own column index
                    lib.thip_gen_identity(blk.ptr, min(n, r1) - r0, nc, m, (r0 - c0) & 0xFFFFFFFFFFFFFFFF, -1.0)
                self.mat_a.set_columns(c0, blk, nc)
            lib.thip_sync()
            blk.free()
        else:
            self.mat_a = DeviceBuffer(max(m * n, 1))
            lib.thip_gen_matrix(self.mat_a.ptr, m, n, m, seed, STREAM_A, r0, 0, self.m_total, 0, 1.0, 0.0)
            # rows r < n of the full matrix are -I: overwrite that part of the shard on the device (the first k rows
            # of the column-major shard share its base pointer and lda)
            if r0 < n:
                k = min(n, r1) - r0
                lib.thip_gen_identity(self.mat_a.ptr, k, n, m, r0, -1.0)
        h = _gen(max(m, 1), seed, STREAM_H, r0, 0)[:m]
        h[np.arange(r0, r1) < n] = 0.0
        self.vec_b_host = h
        self.vec_b = DeviceBuffer.from_host(h) if m else DeviceBuffer(1)
        c = -_gen(n, seed, STREAM_C, 0, 0)
        self.vec_c_host = c
        self.vec_c = DeviceBuffer.from_host(c)
        self.seg_type = [_lib.CONE_RPOS]
        self.seg_len = [m]

    def free(self):
        for d in (self.mat_a, self.vec_b, self.vec_c):
            d.free()


class SdpInstance:
    """SDP of the partitioning_sdp shape (BASELINE.json configs[3]): one PSD cone of order k, n dense symmetric
    F_i, in totsu's ProbSDP conic form (sdp.rs:250-331): A = [svec(F_0) .. svec(F_{n-1})] (sk x n), b = -svec(F_n),
    cone = PSD(sk).  Entries of svec(F_i) ~ N(0, 1/k); F_n = -I - sum x0_i F_i (strictly feasible at x0);
    c = -A^T svec(Y), Y = I + 0.1 R > 0 (strictly dual feasible => bounded).  Single GPU (a PSD cone does not shard)."""

    def __init__(self, n, k, seed=0):
        _lib.ensure_init()
        self.n, self.k, self.seed = n, k, seed
        sk = k * (k + 1) // 2
        self.m = self.m_total = sk
        self.mat_a = DeviceBuffer(sk * n)
        lib.thip_gen_matrix(self.mat_a.ptr, sk, n, sk, seed, STREAM_A, 0, 0, sk, 1, 1.0 / math.sqrt(k), 0.0)
        x0 = DeviceBuffer(n)
        lib.thip_gen_vector(x0.ptr, n, seed, STREAM_X0, 0, 1, 1.0 / math.sqrt(n), 0.0)
        w = DeviceBuffer(sk)
        lib.thip_transform_ge(0, sk, n, 1.0, self.mat_a.ptr, x0.ptr, 0.0, w.ptr)
        diag = np.array([c * (c + 1) // 2 + c for c in range(k)])
        b = w.to_host()
        b[diag] += 1.0                                   # b = -svec(F_n) = svec(I) + A x0
        self.vec_b_host = b
        self.vec_b = DeviceBuffer.from_host(b)
        yv = 0.1 / math.sqrt(k) * _gen(sk, seed, STREAM_W, 0, 1)
        yv[diag] += 1.0
        yd = DeviceBuffer.from_host(yv)
        f = DeviceBuffer(n)
        lib.thip_transform_ge(1, sk, n, -1.0, self.mat_a.ptr, yd.ptr, 0.0, f.ptr)
        self.vec_c_host = f.to_host()
        self.vec_c = f
        self.seg_type = [_lib.CONE_PSD]
        self.seg_len = [sk]
        for d_ in (x0, w, yd):
            d_.free()

    def free(self):
        for d in (self.mat_a, self.vec_b, self.vec_c):
            d.free()
