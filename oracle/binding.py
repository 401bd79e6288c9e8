"""ctypes binding of oracle/libtotsu_oracle.so (see totsu_oracle.h for the reference citations)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libtotsu_oracle.so")

CONE_ZERO, CONE_RPOS, CONE_SOC, CONE_ROTSOC, CONE_PSD = 0, 1, 2, 3, 4
OK, UNBOUNDED, INFEASIBLE, EXCESS_ITER, INVALID_OP, WORK_SHORTAGE, CONE_FAILURE = range(7)
STATUS_NAMES = ["Ok", "Unbounded", "Infeasible", "ExcessIter", "InvalidOp", "WorkShortage", "ConeFailure"]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("totsu_oracle.c", "totsu_oracle.h")]
    if (not force and os.path.exists(_SO)
            and all(not os.path.exists(s) or os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    if not os.path.exists(src[0]):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libtotsu_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Param(C.Structure):
    _fields_ = [("max_iter", C.c_int64), ("eps_acc", C.c_double), ("eps_inf", C.c_double),
                ("eps_zero", C.c_double), ("log_period", C.c_int64)]


class TraceRec(C.Structure):
    _fields_ = [("iter", C.c_int64), ("kind", C.c_int32), ("v0", C.c_double), ("v1", C.c_double),
                ("v2", C.c_double)]


class Trace(C.Structure):
    _fields_ = [("rec", C.POINTER(TraceRec)), ("cap", C.c_size_t), ("len", C.c_size_t),
                ("iters", C.c_int64), ("norm_b", C.c_double), ("norm_c", C.c_double),
                ("snap_iters", C.POINTER(C.c_int64)), ("n_snap", C.c_size_t),
                ("snap_out", C.POINTER(C.c_double)), ("precond_out", C.POINTER(C.c_double))]


_lib = None
_dp = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oc_norm.restype = C.c_double
        L.oc_abssum.restype = C.c_double
        L.oc_map_eig_worklen.restype = C.c_size_t
        L.oc_map_eig_worklen_ql.restype = C.c_size_t
        L.oc_query_worklen.restype = C.c_size_t
        L.oc_rng_uniform.restype = C.c_float
        L.oc_rng_normal.restype = C.c_float
        L.oc_rng_hash.restype = C.c_uint64
        for f in ("oc_rng_uniform", "oc_rng_normal", "oc_rng_hash"):
            getattr(L, f).argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        _lib = L
    return _lib


def _d(a):
    """contiguous float64 array + pointer"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _sz(v):
    return C.c_size_t(int(v))


def param(max_iter=None, eps_acc=1e-6, eps_inf=1e-6, eps_zero=1e-12, log_period=10000):
    p = Param()
    lib().oc_param_default(C.byref(p))
    # defaults follow solver.rs:27-41 (ten.powi(-6) etc.); explicit values override
    if max_iter is not None:
        p.max_iter = int(max_iter)
    if eps_acc != 1e-6:
        p.eps_acc = eps_acc
    if eps_inf != 1e-6:
        p.eps_inf = eps_inf
    if eps_zero != 1e-12:
        p.eps_zero = eps_zero
    p.log_period = log_period
    return p


class Result:
    def __init__(self, status, x, y, trace, recs, snaps=None):
        self.status = status
        self.status_name = STATUS_NAMES[status]
        self.x = x
        self.y = y
        self.iters = trace.iters
        self.norm_b = trace.norm_b
        self.norm_c = trace.norm_c
        self.trace = recs      # list of (iter, kind, v0, v1, v2)
        self.snaps = snaps


def _mk_trace(cap, snap_iters=None, NM=0):
    t = Trace()
    keep = {}
    if cap:
        recs = (TraceRec * cap)()
        t.rec = C.cast(recs, C.POINTER(TraceRec))
        t.cap = cap
        keep["recs"] = recs
    if snap_iters is not None and len(snap_iters):
        si = np.ascontiguousarray(snap_iters, dtype=np.int64)
        so = np.zeros((len(si), NM), dtype=np.float64)
        t.snap_iters = si.ctypes.data_as(C.POINTER(C.c_int64))
        t.n_snap = len(si)
        t.snap_out = so.ctypes.data_as(_dp)
        keep["si"] = si
        keep["so"] = so
    if NM:
        pc = np.zeros(NM, dtype=np.float64)       # dp_tau (N) then dp_sigma (M) of calc_precond
        t.precond_out = pc.ctypes.data_as(_dp)
        keep["pc"] = pc
    return t, keep


def _finish(rc, x, y, t, keep):
    recs = []
    if "recs" in keep:
        n = min(t.len, t.cap)
        r = keep["recs"]
        recs = [(r[i].iter, r[i].kind, r[i].v0, r[i].v1, r[i].v2) for i in range(n)]
    res = Result(rc, x, y, t, recs, keep.get("so"))
    res.precond = keep.get("pc")
    return res


def solve_matop_cones(par, vec_c, mat_a, vec_b, seg_type, seg_len, use_ql=False, trace_cap=0,
                      snap_iters=None):
    """MatOp operators (col-major mat_a of shape m x n given as 1-D col-major or 2-D array)."""
    vec_c, pc = _d(vec_c)
    vec_b, pb = _d(vec_b)
    n, m = vec_c.size, vec_b.size
    mat_a = np.asarray(mat_a, dtype=np.float64)
    if mat_a.ndim == 2:
        mat_a = np.asfortranarray(mat_a).ravel(order="F")
    mat_a, pa = _d(mat_a)
    assert mat_a.size == m * n
    st = np.ascontiguousarray(seg_type, dtype=np.int32)
    sl = np.ascontiguousarray(seg_len, dtype=np.int64)
    x = np.zeros(n)
    y = np.zeros(m)
    t, keep = _mk_trace(trace_cap, snap_iters, (n + 2 * m + 1) + (n + m + 1))
    rc = lib().oc_solve_matop_cones(C.byref(par), _sz(n), _sz(m), pc, pa, pb, _sz(len(st)),
                                    st.ctypes.data_as(C.POINTER(C.c_int32)),
                                    sl.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int(int(use_ql)),
                                    x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.byref(t))
    return _finish(rc, x, y, t, keep)


def solve_csc_cones(par, vec_c, colptr, rowidx, vals, vec_b, seg_type, seg_len, use_ql=False, trace_cap=0, snap_iters=None):
    """A given sparse by columns (scipy's csc arrays: indptr / indices / data) as a user-defined Operator."""
    vec_c, pc = _d(vec_c)
    vec_b, pb = _d(vec_b)
    n, m = vec_c.size, vec_b.size
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int32)
    vals, pv = _d(vals)
    assert cp.size == n + 1 and ri.size == vals.size == int(cp[-1])
    st = np.ascontiguousarray(seg_type, dtype=np.int32)
    sl = np.ascontiguousarray(seg_len, dtype=np.int64)
    x = np.zeros(n)
    y = np.zeros(m)
    t, keep = _mk_trace(trace_cap, snap_iters, (n + 2 * m + 1) + (n + m + 1))
    rc = lib().oc_solve_csc_cones(C.byref(par), _sz(n), _sz(m), pc, cp.ctypes.data_as(C.POINTER(C.c_int64)),
                                  ri.ctypes.data_as(C.POINTER(C.c_int32)), pv, pb, _sz(len(st)),
                                  st.ctypes.data_as(C.POINTER(C.c_int32)), sl.ctypes.data_as(C.POINTER(C.c_int64)),
                                  C.c_int(int(use_ql)), x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.byref(t))
    return _finish(rc, x, y, t, keep)


def _colmaj(a, nr, nc):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        assert a.shape == (nr, nc), (a.shape, nr, nc)
        a = np.asfortranarray(a).ravel(order="F")
    assert a.size == nr * nc
    return np.ascontiguousarray(a)


def solve_lp(par, vec_c, mat_g, vec_h, mat_a, vec_b, trace_cap=0, snap_iters=None):
    vec_c, pc = _d(vec_c)
    vec_h, ph = _d(vec_h)
    vec_b, pb = _d(vec_b)
    n, m, p = vec_c.size, vec_h.size, vec_b.size
    mat_g, pg = _d(_colmaj(mat_g, m, n))
    mat_a, pa = _d(_colmaj(mat_a, p, n))
    x = np.zeros(n)
    y = np.zeros(m + p)
    t, keep = _mk_trace(trace_cap, snap_iters, (n + 2 * (m + p) + 1) + (n + m + p + 1))
    rc = lib().oc_solve_lp(C.byref(par), _sz(n), _sz(m), _sz(p), pc, pg, ph, pa, pb,
                           x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.byref(t))
    return _finish(rc, x, y, t, keep)


def solve_socp(par, vec_f, mats_g, vecs_h, vecs_c, scls_d, mat_a, vec_b, trace_cap=0, snap_iters=None):
    vec_f, pf = _d(vec_f)
    n = vec_f.size
    ni = np.array([np.asarray(h).size for h in vecs_h], dtype=np.int64)
    g = np.concatenate([_colmaj(G, int(k), n) for G, k in zip(mats_g, ni)] + [np.zeros(0)])
    h = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in vecs_h] + [np.zeros(0)])
    c = np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in vecs_c] + [np.zeros(0)])
    g, pg = _d(g)
    h, ph = _d(h)
    c, pcc = _d(c)
    d, pd = _d(scls_d)
    vec_b, pb = _d(vec_b)
    p = vec_b.size
    mat_a, pa = _d(_colmaj(mat_a, p, n))
    m = int(ni.sum()) + len(ni) + p
    x = np.zeros(n)
    y = np.zeros(m)
    t, keep = _mk_trace(trace_cap, snap_iters, (n + 2 * m + 1) + (n + m + 1))
    rc = lib().oc_solve_socp(C.byref(par), _sz(n), _sz(len(ni)), ni.ctypes.data_as(C.POINTER(C.c_int64)),
                             _sz(p), pf, pg, ph, pcc, pd, pa, pb,
                             x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.byref(t))
    return _finish(rc, x, y, t, keep)


def solve_sdp(par, vec_c, syms_f, mat_a, vec_b, eps_zero, use_ql=False, trace_cap=0, snap_iters=None):
    """syms_f: n+1 packed-upper (by columns) arrays of length k(k+1)/2, unscaled (sdp.rs:250-297)."""
    vec_c, pc = _d(vec_c)
    n = vec_c.size
    f = np.stack([np.asarray(s, dtype=np.float64).ravel() for s in syms_f])
    assert f.shape[0] == n + 1
    sk = f.shape[1]
    k = (int(np.sqrt(8 * sk + 1)) - 1) // 2
    assert k * (k + 1) // 2 == sk
    f, pf = _d(f)
    vec_b, pb = _d(vec_b)
    p = vec_b.size
    mat_a, pa = _d(_colmaj(mat_a, p, n))
    x = np.zeros(n)
    y = np.zeros(sk + p)
    m = sk + p
    t, keep = _mk_trace(trace_cap, snap_iters, (n + 2 * m + 1) + (n + m + 1))
    rc = lib().oc_solve_sdp(C.byref(par), _sz(n), _sz(k), _sz(p), pc, pf, pa, pb, C.c_double(eps_zero),
                            C.c_int(int(use_ql)), x.ctypes.data_as(_dp), y.ctypes.data_as(_dp), C.byref(t))
    return _finish(rc, x, y, t, keep)


# ---- primitives ---------------------------------------------------------------------------

def norm(x):
    x, p = _d(x)
    return lib().oc_norm(_sz(x.size), p)


def abssum(x, incx):
    x, p = _d(x)
    return lib().oc_abssum(_sz(x.size), p, _sz(incx))


def transform_di(alpha, d, x, beta, y):
    d, pd = _d(d)
    x, px = _d(x)
    y = np.array(y, dtype=np.float64)
    lib().oc_transform_di(_sz(x.size), C.c_double(alpha), pd, px, C.c_double(beta), y.ctypes.data_as(_dp))
    return y


def transform_ge(transpose, n_row, n_col, alpha, mat, x, beta, y):
    mat, pm = _d(mat)
    x, px = _d(x)
    y = np.array(y, dtype=np.float64)
    lib().oc_transform_ge(C.c_int(int(transpose)), _sz(n_row), _sz(n_col), C.c_double(alpha), pm, px,
                          C.c_double(beta), y.ctypes.data_as(_dp))
    return y


def transform_sp(n, alpha, mat, x, beta, y):
    mat, pm = _d(mat)
    x, px = _d(x)
    y = np.array(y, dtype=np.float64)
    lib().oc_transform_sp(_sz(n), C.c_double(alpha), pm, px, C.c_double(beta), y.ctypes.data_as(_dp))
    return y


def map_eig(mat, scale_diag, eps_zero, map_kind=0, use_ql=False):
    mat = np.array(mat, dtype=np.float64)
    sn = mat.size
    n = (int(np.sqrt(8 * sn + 1)) - 1) // 2
    L = lib()
    wl = L.oc_map_eig_worklen_ql(_sz(n)) if use_ql else L.oc_map_eig_worklen(_sz(n))
    work = np.zeros(max(wl, 1))
    fn = L.oc_map_eig_ql if use_ql else L.oc_map_eig
    fn(_sz(sn), mat.ctypes.data_as(_dp), C.c_int(0 if scale_diag is None else 1),
       C.c_double(0.0 if scale_diag is None else scale_diag), C.c_double(eps_zero),
       work.ctypes.data_as(_dp), C.c_int(map_kind))
    return mat


def proj(cone_type, x, dual_cone=False, eps_zero=1e-12, use_ql=False):
    x = np.array(x, dtype=np.float64)
    L = lib()
    px = x.ctypes.data_as(_dp)
    if cone_type == CONE_ZERO:
        L.oc_proj_zero(C.c_int(int(dual_cone)), _sz(x.size), px)
    elif cone_type == CONE_RPOS:
        L.oc_proj_rpos(_sz(x.size), px)
    elif cone_type == CONE_SOC:
        L.oc_proj_soc(_sz(x.size), px)
    elif cone_type == CONE_ROTSOC:
        L.oc_proj_rotsoc(_sz(x.size), px)
    elif cone_type == CONE_PSD:
        n = (int(np.sqrt(8 * x.size + 1)) - 1) // 2
        wl = L.oc_map_eig_worklen_ql(_sz(n)) if use_ql else L.oc_map_eig_worklen(_sz(n))
        work = np.zeros(max(wl, 1))
        rc = L.oc_proj_psd(_sz(x.size), px, C.c_double(eps_zero), work.ctypes.data_as(_dp), _sz(work.size),
                           C.c_int(int(use_ql)))
        assert rc == 0
    else:
        raise ValueError(cone_type)
    return x


def vec_to_mat(v, scale=None):
    v, pv = _d(v)
    n = (int(np.sqrt(8 * v.size + 1)) - 1) // 2
    m = np.zeros(n * n)
    lib().oc_vec_to_mat(_sz(n), pv, m.ctypes.data_as(_dp), C.c_int(0 if scale is None else 1),
                        C.c_double(scale or 0.0))
    return m


def mat_to_vec(m, scale=None):
    m = np.array(m, dtype=np.float64)
    n = int(round(np.sqrt(m.size)))
    v = np.zeros(n * (n + 1) // 2)
    lib().oc_mat_to_vec(_sz(n), m.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                        C.c_int(0 if scale is None else 1), C.c_double(scale or 0.0))
    return v, m


def scale_nondiag_sympack(packed, alpha):
    packed = np.array(packed, dtype=np.float64)
    n = (int(np.sqrt(8 * packed.size + 1)) - 1) // 2
    lib().oc_matbuild_scale_nondiag_sympack(_sz(n), packed.ctypes.data_as(_dp), C.c_double(alpha))
    return packed


def matop_absadd(typ, nr, nc, array, colwise, y):
    class MatOp(C.Structure):
        _fields_ = [("typ", C.c_int), ("nr", C.c_size_t), ("nc", C.c_size_t), ("array", _dp)]
    array, pa = _d(array)
    y = np.array(y, dtype=np.float64)
    m = MatOp(typ, nr, nc, pa)
    lib().oc_matop_absadd(C.byref(m), C.c_int(int(colwise)), y.ctypes.data_as(_dp))
    return y


def rng_uniform(seed, stream, idx):
    return float(lib().oc_rng_uniform(seed, stream, idx))


def rng_normal(seed, stream, idx):
    return float(lib().oc_rng_normal(seed, stream, idx))


def num_threads():
    return int(lib().oc_num_threads())


def set_num_threads(k):
    """OpenMP threads of the oracle from now on (small problems: a handful beats every core of a big host)"""
    lib().oc_set_num_threads(int(k))


def gen_matrix(n_row, n_col, seed, stream, row0, col0, ld_index, kind, scale, shift=0.0):
    """column-major (n_row x n_col) block of the synthetic matrix, f64 holding the exact f32 entries"""
    out = np.empty(n_row * n_col, dtype=np.float64)
    lib().oc_gen_matrix(out.ctypes.data_as(_dp), _sz(n_row), _sz(n_col), _sz(n_row), C.c_uint64(seed),
                        C.c_uint64(stream), C.c_uint64(row0), C.c_uint64(col0), C.c_uint64(ld_index),
                        C.c_int(kind), C.c_float(scale), C.c_float(shift))
    return out


def gen_vector(n, seed, stream, idx0, kind, scale=1.0, shift=0.0):
    out = np.empty(n, dtype=np.float64)
    lib().oc_gen_vector(out.ctypes.data_as(_dp), _sz(n), C.c_uint64(seed), C.c_uint64(stream), C.c_uint64(idx0),
                        C.c_int(kind), C.c_float(scale), C.c_float(shift))
    return out


def _quadratic_dense(n, syms_p, vecs_q, scls_r, tails, eps_zero):
    """Stacked form of ProbQP / ProbQCQP (totsu/src/problem/qp.rs:66-300, qcqp.rs:62-345): variables (x, t), c = [0; 1],
    block i rows [0 ; q_i^T, -(i == 0) ; -P_i^{1/2}, 0] with b = [1 ; -r_i ; 0_n] in a rotated SOC of 2 + n, then the
    tail blocks (G | A rows).  P^{1/2} by map_eig with the sqrt closure (matbuild/mod.rs:219-245).  The reference's
    absadd_* of these operators are plain |.| sums, so the MatOp solve on the stacked matrix is the same iteration
    (up to the summation order inside a row)."""
    rows, bs, st, sl = [], [], [], []
    for i, (P, q, r) in enumerate(zip(syms_p, vecs_q, scls_r)):
        ps = map_eig(np.asarray(P, dtype=np.float64), None, eps_zero, map_kind=1)
        full = np.zeros((n, n))
        for c in range(n):
            for rr in range(c + 1):
                full[rr, c] = full[c, rr] = ps[c * (c + 1) // 2 + rr]
        blk = np.zeros((2 + n, n + 1))
        blk[1, :n] = q
        blk[1, n] = -1.0 if i == 0 else 0.0
        blk[2:, :n] = -full
        rows.append(blk)
        bb = np.zeros(2 + n)
        bb[0], bb[1] = 1.0, -r
        bs.append(bb)
        st.append(CONE_ROTSOC)
        sl.append(2 + n)
    for mat, vec, typ in tails:
        mat = np.asarray(mat, dtype=np.float64).reshape((-1, n)) if np.size(mat) else np.zeros((0, n))
        blk = np.zeros((mat.shape[0], n + 1))
        blk[:, :n] = mat
        rows.append(blk)
        bs.append(np.asarray(vec, dtype=np.float64).ravel())
        st.append(typ)
        sl.append(mat.shape[0])
    a = np.vstack(rows)
    c = np.zeros(n + 1)
    c[n] = 1.0
    return c, a, np.concatenate(bs), st, sl


def solve_qp(par, sym_p, vec_q, mat_g, vec_h, mat_a, vec_b, eps_zero=1e-12, **kw):
    """ProbQP (qp.rs:304-440); sym_p packed upper by columns"""
    n = np.size(vec_q)
    c, a, b, st, sl = _quadratic_dense(n, [sym_p], [np.ravel(vec_q)], [0.0],
                                       [(mat_g, vec_h, CONE_RPOS), (mat_a, vec_b, CONE_ZERO)], eps_zero)
    return solve_matop_cones(par, c, a, b, st, sl, **kw)


def solve_qcqp(par, syms_p, vecs_q, scls_r, mat_a, vec_b, eps_zero=1e-12, **kw):
    """ProbQCQP (qcqp.rs:349-505)"""
    n = np.size(vecs_q[0])
    c, a, b, st, sl = _quadratic_dense(n, syms_p, [np.ravel(q) for q in vecs_q], scls_r,
                                       [(mat_a, vec_b, CONE_ZERO)], eps_zero)
    return solve_matop_cones(par, c, a, b, st, sl, **kw)
