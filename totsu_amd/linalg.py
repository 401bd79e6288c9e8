"""`F32HIP`: the LinAlgEx backend (host side) and `F32HIPSlice`: its SliceLike.

Mirrors, in Python over the C ABI, what a Rust `totsu_f32hip` crate implements:
  LinAlg    totsu_core/src/solver/linalg.rs:10-68
  LinAlgEx  totsu_core/src/linalg_ex.rs:7-66
  SliceLike totsu_core/src/solver/slicelike.rs:9-70
with the CUDA crate as behaviour model (totsu_f32cuda/src/f32cuda.rs, f32cuda_slice.rs).

Ownership: the caller owns host memory; `new_ref` / `new_mut` create a device mirror
(one upload); the DEVICE copy is the truth afterwards.  `get_ref()` downloads the range, `get_mut()`
downloads it and marks it host-dirty (uploaded again before the next device use); dropping a root
`new_mut` slice leaves the host buffer up to date (f32cuda_slice.rs:203-207).
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib


class _Root:
    __slots__ = ("host", "dev", "n", "mutable", "dirty", "alive")

    def __init__(self, host, mutable):
        _lib.ensure_init()
        self.host = host
        self.n = int(host.size)
        self.mutable = mutable
        p = C.c_void_p()
        lib.thip_alloc(self.n, C.byref(p))
        self.dev = p.value
        self.dirty = []          # host-dirty (off, len) ranges awaiting upload
        self.alive = True
        if self.n:
            lib.thip_h2d(self.dev, host.ctypes.data, self.n)

    def flush(self):
        if self.dirty:
            for off, ln in self.dirty:
                if ln:
                    lib.thip_h2d(self.dev + 4 * off, self.host.ctypes.data + 4 * off, ln)
            self.dirty = []

    def free(self):
        if self.alive:
            self.alive = False
            lib.thip_free(self.dev)


class F32HIPSlice:
    """SliceLike over a device mirror (slicelike.rs:9-70)."""
    __slots__ = ("root", "off", "n", "is_root")

    def __init__(self, root, off, n, is_root=False):
        self.root, self.off, self.n, self.is_root = root, off, n, is_root

    # -- SliceLike --
    @staticmethod
    def new_ref(host):
        host = _as_f32(host)
        return F32HIPSlice(_Root(host, False), 0, host.size, True)

    @staticmethod
    def new_mut(host):
        host = _as_f32(host, require_writable=True)
        return F32HIPSlice(_Root(host, True), 0, host.size, True)

    def len(self):
        return self.n

    def split(self, mid):
        assert 0 <= mid <= self.n
        return (F32HIPSlice(self.root, self.off, mid), F32HIPSlice(self.root, self.off + mid, self.n - mid))

    split_ref = split
    split_mut = split

    def get_ref(self):
        r = self.root
        r.flush()
        view = r.host[self.off:self.off + self.n]
        if self.n:
            lib.thip_d2h(view.ctypes.data, r.dev + 4 * self.off, self.n)
        return view

    def get_mut(self):
        view = self.get_ref()
        assert self.root.mutable
        self.root.dirty.append((self.off, self.n))
        return view

    def get(self, idx):
        out = C.c_float()
        self.root.flush()
        lib.thip_get(self.root.dev + 4 * self.off, idx, C.byref(out))
        return float(out.value)

    def set(self, idx, val):
        self.root.flush()
        lib.thip_set(self.root.dev + 4 * self.off, idx, float(val))

    def drop(self):
        """SliceLike::drop: a root mutable slice syncs the host buffer and releases the device mirror."""
        if self.is_root and self.root.alive:
            if self.root.mutable:
                self.get_ref()
            self.root.free()

    # -- device access for the backend --
    def dev(self):
        self.root.flush()
        return self.root.dev + 4 * self.off

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.drop()


def _as_f32(a, require_writable=False):
    if not isinstance(a, np.ndarray) or a.dtype != np.float32 or not a.flags.c_contiguous or a.ndim != 1:
        if require_writable:
            raise TypeError("new_mut needs a contiguous 1-D float32 numpy array (the caller owns it)")
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).ravel())
    return a


def splitm(s, *lens):
    """the splitm! / splitm_mut! macros (slicelike.rs:145-247): consecutive sub-slices"""
    out = []
    rest = s
    for ln in lens:
        head, rest = rest.split(ln)
        out.append(head)
    return out


class F32HIP:
    """LinAlgEx over libtotsu_f32hip.so; all methods are static like the trait's associated functions."""
    F = np.float32
    Sl = F32HIPSlice
    name = "F32HIP"

    # ---- LinAlg (linalg.rs:22-67) ----
    @staticmethod
    def norm(x):
        out = C.c_float()
        lib.thip_norm(x.len(), x.dev(), C.byref(out))
        return float(out.value)

    @staticmethod
    def copy(x, y):
        assert x.len() == y.len()
        lib.thip_copy(x.len(), x.dev(), y.dev())

    @staticmethod
    def scale(alpha, x):
        lib.thip_scale(x.len(), float(alpha), x.dev())

    @staticmethod
    def add(alpha, x, y):
        assert x.len() == y.len()
        lib.thip_add(x.len(), float(alpha), x.dev(), y.dev())

    @staticmethod
    def adds(s, y):
        lib.thip_adds(y.len(), float(s), y.dev())

    @staticmethod
    def abssum(x, incx):
        out = C.c_float()
        lib.thip_abssum(x.len(), x.dev(), int(incx), C.byref(out))
        return float(out.value)

    @staticmethod
    def transform_di(alpha, mat, x, beta, y):
        assert mat.len() == x.len() == y.len()
        lib.thip_transform_di(x.len(), float(alpha), mat.dev(), x.dev(), float(beta), y.dev())

    # ---- LinAlgEx (linalg_ex.rs:23-65) ----
    @staticmethod
    def transform_ge(transpose, n_row, n_col, alpha, mat, x, beta, y):
        assert mat.len() == n_row * n_col
        if transpose:
            assert x.len() == n_row and y.len() == n_col
        else:
            assert x.len() == n_col and y.len() == n_row
        lib.thip_transform_ge(int(bool(transpose)), n_row, n_col, float(alpha), mat.dev(), x.dev(), float(beta), y.dev())

    @staticmethod
    def transform_sp(n, alpha, mat, x, beta, y):
        assert mat.len() == n * (n + 1) // 2 and x.len() == n and y.len() == n
        lib.thip_transform_sp(n, float(alpha), mat.dev(), x.dev(), float(beta), y.dev())

    @staticmethod
    def map_eig_worklen(n):
        return int(_lib.load().thip_map_eig_worklen(n))

    @staticmethod
    def map_eig(mat, scale_diag, eps_zero, work, map):
        """`map` is either the string 'pos' / 'sqrt_pos' (evaluated on the device) or a Python callable
        e -> value | None (two-phase path with the eigenvalues visiting the host)."""
        sn = mat.len()
        n = (int(np.sqrt(8 * sn + 1)) - 1) // 2
        assert n * (n + 1) // 2 == sn
        assert work.len() >= F32HIP.map_eig_worklen(n)
        has = 0 if scale_diag is None else 1
        sc = 0.0 if scale_diag is None else float(scale_diag)
        if isinstance(map, str):
            kind = {"pos": 0, "sqrt_pos": 1}[map]
            lib.thip_map_eig(n, mat.dev(), has, sc, float(eps_zero), work.dev(), work.len(), kind)
            return
        w = np.zeros(max(n, 1), dtype=np.float32)
        lib.thip_eig_decompose(n, mat.dev(), has, sc, float(eps_zero), work.dev(), work.len(),
                               w.ctypes.data_as(_lib.fp))
        e = np.zeros(max(n, 1), dtype=np.float32)
        keep = np.zeros(max(n, 1), dtype=np.uint8)
        for i in range(n):
            r = map(float(w[i]))
            if r is not None:
                e[i] = r
                keep[i] = 1
        lib.thip_eig_rebuild(n, mat.dev(), has, sc, work.dev(), work.len(), e.ctypes.data_as(_lib.fp),
                             keep.ctypes.data_as(C.POINTER(C.c_uint8)))

    # ---- device-side extras used by the Hip* cones / builders (not part of the trait) ----
    @staticmethod
    def absadd_cols(n_row, n_col, mat, tau):
        lib.thip_absadd_cols(n_row, n_col, mat.dev(), tau.dev())

    @staticmethod
    def absadd_rows(n_row, n_col, mat, sigma):
        lib.thip_absadd_rows(n_row, n_col, mat.dev(), sigma.dev())

    @staticmethod
    def absadd_sympack(n, mat, y):
        lib.thip_absadd_sympack(n, mat.dev(), y.dev())

    @staticmethod
    def recip_max(eps_zero, x):
        lib.thip_recip_max(x.len(), float(eps_zero), x.dev())

    @staticmethod
    def sync():
        lib.thip_sync()
