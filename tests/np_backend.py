"""A plain numpy LinAlgEx backend used ONLY by the CPU tests to exercise the host-side logic of totsu_amd
(generic Solver loop, MatOp, cones, problem builders, row sharding) without a GPU.  Its primitive semantics are
FloatGeneric's (totsu_core/src/floatgeneric.rs); eigen-decomposition through numpy."""
import math

import numpy as np


class NpSlice:
    __slots__ = ("a",)

    def __init__(self, a):
        self.a = a

    @staticmethod
    def new_ref(host):
        return NpSlice(np.asarray(host))

    @staticmethod
    def new_mut(host):
        assert isinstance(host, np.ndarray)
        return NpSlice(host)

    def len(self):
        return self.a.size

    def split(self, mid):
        return NpSlice(self.a[:mid]), NpSlice(self.a[mid:])

    def get_ref(self):
        return self.a

    def get_mut(self):
        return self.a

    def get(self, i):
        return float(self.a[i])

    def set(self, i, v):
        self.a[i] = v

    def drop(self):
        pass


def make_backend(dtype):
    class NP:
        F = dtype
        Sl = NpSlice
        name = "NP_" + np.dtype(dtype).name

        @staticmethod
        def norm(x):
            return float(math.sqrt(float(np.dot(x.a, x.a))))

        @staticmethod
        def copy(x, y):
            y.a[:] = x.a

        @staticmethod
        def scale(alpha, x):
            x.a *= dtype(alpha)

        @staticmethod
        def add(alpha, x, y):
            y.a += dtype(alpha) * x.a

        @staticmethod
        def adds(s, y):
            y.a += dtype(s)

        @staticmethod
        def abssum(x, incx):
            if incx == 0:
                return 0.0
            return float(np.abs(x.a[::incx]).sum())

        @staticmethod
        def transform_di(alpha, mat, x, beta, y):
            y.a[:] = dtype(alpha) * mat.a * x.a + dtype(beta) * y.a

        @staticmethod
        def transform_ge(transpose, n_row, n_col, alpha, mat, x, beta, y):
            a = mat.a.reshape((n_col, n_row)).T
            r = (a.T @ x.a) if transpose else (a @ x.a)
            y.a[:] = dtype(alpha) * r + dtype(beta) * y.a

        @staticmethod
        def transform_sp(n, alpha, mat, x, beta, y):
            s = np.zeros((n, n), dtype=dtype)
            iu = np.triu_indices(n)
            # packed upper by columns: (r, c) with r <= c at c(c+1)/2 + r
            for c in range(n):
                for r in range(c + 1):
                    s[r, c] = s[c, r] = mat.a[c * (c + 1) // 2 + r]
            y.a[:] = dtype(alpha) * (s @ x.a) + dtype(beta) * y.a

        @staticmethod
        def map_eig_worklen(n):
            return n + n * n

        @staticmethod
        def map_eig(mat, scale_diag, eps_zero, work, map):
            sn = mat.len()
            n = (int(math.sqrt(8 * sn + 1)) - 1) // 2
            s = np.zeros((n, n), dtype=np.float64)
            for c in range(n):
                for r in range(c + 1):
                    v = mat.a[c * (c + 1) // 2 + r]
                    s[r, c] = s[c, r] = v * (scale_diag if (r == c and scale_diag is not None) else 1.0)
            w, z = np.linalg.eigh(s)
            out = np.zeros((n, n))
            for i in range(n):
                e = map(float(w[i]))
                if e is not None:
                    out += e * np.outer(z[:, i], z[:, i])
            for c in range(n):
                for r in range(c + 1):
                    v = out[r, c]
                    if r == c and scale_diag is not None:
                        v = v / scale_diag
                    mat.a[c * (c + 1) // 2 + r] = v
    return NP


F64NP = make_backend(np.float64)
F32NP = make_backend(np.float32)
