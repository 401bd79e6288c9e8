"""`MatBuild<L>`: owns a column-major / packed-upper host array and builds matrices for the problem classes.
Mirror of totsu/src/matbuild/mod.rs:9-300."""
import math

import numpy as np

from .matop import MatOp, MatType


class MatBuild:
    def __init__(self, L, typ):                                    # mod.rs:22-28
        self.L = L
        self.typ = typ
        self.array = np.zeros(typ.len(), dtype=L.F)

    def clone(self):
        o = MatBuild(self.L, MatType(self.typ.kind, self.typ.n_row, self.typ.n_col))
        o.array = self.array.copy()
        return o

    def size(self):
        return self.typ.size()

    def as_op(self):                                               # mod.rs:36-39
        return MatOp(self.L, self.typ, self.array)

    def is_sympack(self):
        return self.typ.kind == MatType.SYMPACK

    def _index(self, r, c):                                        # mod.rs:254-279
        if self.typ.kind == MatType.GENERAL:
            assert r < self.typ.n_row and c < self.typ.n_col
            return c * self.typ.n_row + r
        assert r < self.typ.n_row and c < self.typ.n_row
        if r > c:
            r, c = c, r
        return c * (c + 1) // 2 + r

    def __getitem__(self, rc):
        return self.array[self._index(*rc)]

    def __setitem__(self, rc, v):
        self.array[self._index(*rc)] = v

    def set_by_fn(self, func):                                     # mod.rs:52-71
        if self.typ.kind == MatType.GENERAL:
            for c in range(self.typ.n_col):
                for r in range(self.typ.n_row):
                    self[(r, c)] = func(r, c)
        else:
            for c in range(self.typ.n_row):
                for r in range(c + 1):
                    self[(r, c)] = func(r, c)
        return self

    by_fn = set_by_fn

    def set_iter_colmaj(self, it):                                 # mod.rs:79-96
        it = iter(it)
        nr, nc = self.typ.size()
        for c in range(nc):
            for r in range(nr):
                try:
                    self[(r, c)] = next(it)
                except StopIteration:
                    return self
        return self

    iter_colmaj = set_iter_colmaj

    def set_iter_rowmaj(self, it):                                 # mod.rs:105-122
        it = iter(it)
        nr, nc = self.typ.size()
        for r in range(nr):
            for c in range(nc):
                try:
                    self[(r, c)] = next(it)
                except StopIteration:
                    return self
        return self

    iter_rowmaj = set_iter_rowmaj

    def set_array(self, a):
        """bulk fill from a numpy array: 2-D (n_row x n_col) for General, packed 1-D otherwise"""
        a = np.asarray(a, dtype=self.L.F)
        if self.typ.kind == MatType.GENERAL and a.ndim == 2:
            assert a.shape == self.typ.size()
            a = np.asfortranarray(a).ravel(order="F")
        assert a.size == self.array.size
        self.array[:] = a.ravel()
        return self

    def set_scale(self, alpha):                                    # mod.rs:130-134
        self.array *= self.L.F(alpha)
        return self

    scale = set_scale

    def set_scale_nondiag(self, alpha):                            # mod.rs:141-166
        a = self.L.F(alpha)
        if self.typ.kind == MatType.SYMPACK:
            n = self.typ.n_row
            for c in range(n - 1):
                i = self._index(c, c)
                ii = self._index(c + 1, c + 1)
                self.array[i + 1:ii] *= a
        else:
            nr, nc = self.typ.size()
            for c in range(nc):
                for r in range(nr):
                    if r != c:
                        self.array[c * nr + r] *= a
        return self

    scale_nondiag = set_scale_nondiag

    def set_reshape_colvec(self):                                  # mod.rs:168-177
        self.typ = MatType.General(self.array.size, 1)
        return self

    reshape_colvec = set_reshape_colvec

    def set_sqrt(self, eps_zero):                                  # mod.rs:179-206
        assert self.typ.kind == MatType.SYMPACK
        L = self.L
        n = self.typ.n_row
        work_vec = np.zeros(L.map_eig_worklen(n), dtype=L.F)
        work = L.Sl.new_mut(work_vec)
        arr = L.Sl.new_mut(self.array)
        try:
            if getattr(L, "name", "") == "F32HIP":
                L.map_eig(arr, None, eps_zero, work, "sqrt_pos")
            else:
                L.map_eig(arr, None, eps_zero, work, lambda e: math.sqrt(e) if e > 0.0 else None)
        finally:
            arr.drop()
            work.drop()
        return self

    sqrt = set_sqrt
