"""MatType / MatOp: a borrowed column-major (or packed-symmetric) array as a linear Operator.
Mirror of totsu_core/src/matop.rs:9-175."""


class MatType:
    GENERAL, SYMPACK = 0, 1

    def __init__(self, kind, a, b=None):
        self.kind = kind
        if kind == MatType.GENERAL:
            self.n_row, self.n_col = int(a), int(b)
        else:
            self.n_row = self.n_col = int(a)

    @staticmethod
    def General(n_row, n_col):
        return MatType(MatType.GENERAL, n_row, n_col)

    @staticmethod
    def SymPack(n):
        return MatType(MatType.SYMPACK, n)

    def len(self):                                   # matop.rs:20-26
        if self.kind == MatType.GENERAL:
            return self.n_row * self.n_col
        return self.n_row * (self.n_row + 1) // 2

    def size(self):                                  # matop.rs:29-35
        return (self.n_row, self.n_col)

    def __eq__(self, o):
        return isinstance(o, MatType) and (self.kind, self.n_row, self.n_col) == (o.kind, o.n_row, o.n_col)

    def __repr__(self):
        return ("General(%d, %d)" % (self.n_row, self.n_col)) if self.kind == 0 else ("SymPack(%d)" % self.n_row)


class MatOp:
    """matop.rs:43-175. `L` is a LinAlgEx backend class; `array` a host array (uploaded once, like
    `L::Sl::new_ref(array)` at matop.rs:66-74) or an existing slice of L."""

    def __init__(self, L, typ, array):
        self.L = L
        self.typ = typ
        self._own = not hasattr(array, "split")
        self.array = L.Sl.new_ref(array) if self._own else array
        assert typ.len() == self.array.len()

    def drop(self):
        if self._own:
            self.array.drop()

    def size(self):
        return self.typ.size()

    def _op_impl(self, transpose, alpha, x, beta, y):            # matop.rs:76-96
        L, t = self.L, self.typ
        if t.kind == MatType.GENERAL:
            if t.n_row > 0 and t.n_col > 0:
                L.transform_ge(transpose, t.n_row, t.n_col, alpha, self.array, x, beta, y)
            else:
                L.scale(beta, y)
        else:
            if t.n_row > 0:
                L.transform_sp(t.n_row, alpha, self.array, x, beta, y)
            else:
                L.scale(beta, y)

    def _absadd_impl(self, colwise, y):                           # matop.rs:98-138
        L, t = self.L, self.typ
        if t.kind == MatType.GENERAL:
            nr, nc = t.n_row, t.n_col
            assert y.len() == (nc if colwise else nr)
            if hasattr(L, "absadd_cols"):
                # one launch instead of n (or m) blocking asum calls
                if nr > 0 and nc > 0:
                    (L.absadd_cols if colwise else L.absadd_rows)(nr, nc, self.array, y)
                return
            ym = y.get_mut()
            if colwise:
                for i in range(nc):
                    _, rest = self.array.split(i * nr)
                    col, _ = rest.split(nr)
                    ym[i] = L.abssum(col, 1) + ym[i]
            else:
                for i in range(nr):
                    _, rest = self.array.split(i)
                    row, _ = rest.split(nr * nc - i)
                    ym[i] = L.abssum(row, nr) + ym[i]
        else:
            n = t.n_row
            assert y.len() == n
            if hasattr(L, "absadd_sympack"):
                if n > 0:
                    L.absadd_sympack(n, self.array, y)
                return
            ym = y.get_mut()
            s = 0
            for c in range(n):
                _, rest = self.array.split(s)
                col, _ = rest.split(c + 1)
                s += c + 1
                ym[c] = L.abssum(col, 1) + ym[c]
                cr = col.get_ref()
                for i in range(c):
                    ym[i] = ym[i] + abs(cr[i])

    # Operator (operator.rs:11-156)
    def op(self, alpha, x, beta, y):
        self._op_impl(False, alpha, x, beta, y)

    def trans_op(self, alpha, x, beta, y):
        self._op_impl(True, alpha, x, beta, y)

    def absadd_cols(self, tau):
        self._absadd_impl(True, tau)

    def absadd_rows(self, sigma):
        self._absadd_impl(False, sigma)

    def as_ref(self):
        return self.array.get_ref()
