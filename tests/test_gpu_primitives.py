"""GPU parity tests, primitive level: every LinAlg / LinAlgEx / Cone entry point of the C ABI against the CPU
oracle (oracle/) on the same seeded inputs.  Tolerances are f32 round-off of the operation, stated per test.
Edge cases follow the reference: zero-length and length-1 vectors, odd lengths, unaligned sub-slices (the
solver hands out slices at offsets n, n+m, n+2m: solver.rs:116-118), strided abssum (matop.rs:110-114),
zero-sized matrices (matop.rs:83-85)."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu

LENS = [0, 1, 2, 3, 63, 64, 65, 255, 1000, 4097, 100003]


@pytest.fixture(scope="module")
def L():
    from totsu_amd import F32HIP, _lib
    _lib.init()
    return F32HIP


def _sl(L, a, off=0):
    """device slice of `a` that starts `off` floats into a bigger allocation (unaligned base)"""
    a = np.asarray(a, dtype=np.float32)
    buf = np.zeros(a.size + off, dtype=np.float32)
    buf[off:] = a
    root = L.Sl.new_mut(buf)
    _, s = root.split(off)
    return root, s


@pytest.mark.parametrize("n", LENS)
@pytest.mark.parametrize("off", [0, 1, 3])
def test_level1(L, n, off):
    rng = np.random.default_rng(n * 7 + off)
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(n).astype(np.float32)
    d = rng.standard_normal(n).astype(np.float32)
    rx, sx = _sl(L, x, off)
    ry, sy = _sl(L, y, off)
    rd, sd = _sl(L, d, off)
    x64, y64, d64 = x.astype(np.float64), y.astype(np.float64), d.astype(np.float64)

    # norm / abssum: relative 1e-5 (f32 tree sums, f64 final stage)
    assert abs(L.norm(sx) - O.norm(x64)) <= 1e-5 * max(O.norm(x64), 1e-30)
    for inc in (1, 2, 3, 7):
        assert abs(L.abssum(sx, inc) - O.abssum(x64, inc)) <= 1e-5 * max(O.abssum(x64, inc), 1e-30)
    assert L.abssum(sx, 0) == 0.0

    L.copy(sx, sy)
    assert np.array_equal(sy.get_ref(), x)
    L.scale(-1.5, sy)
    assert np.array_equal(sy.get_ref(), (np.float32(-1.5) * x))
    L.add(0.25, sx, sy)
    assert np.allclose(sy.get_ref(), -1.25 * x64, rtol=1e-6, atol=1e-7)
    L.adds(2.0, sy)
    assert np.allclose(sy.get_ref(), -1.25 * x64 + 2.0, rtol=1e-6, atol=1e-6)
    ref = O.transform_di(0.7, d64, x64, -0.3, sy.get_ref().astype(np.float64))
    L.transform_di(0.7, sd, sx, -0.3, sy)
    assert np.allclose(sy.get_ref(), ref, rtol=1e-5, atol=1e-6)
    L.scale(0.0, sy)
    assert not sy.get_ref().any()
    if n:
        sy.set(n - 1, 3.5)
        assert sy.get(n - 1) == 3.5
    for r in (rx, ry, rd):
        r.drop()


SHAPES = [(1, 1), (3, 2), (4, 4), (99, 50), (100, 64), (256, 33), (1024, 17), (1028, 300), (4096, 129),
          (5000, 1030), (20000, 37), (8, 5000), (33000, 70),
          # the n x 1 / 1 x n operators of c, b (solver.rs:129-130, 599-603) and other extreme aspect ratios
          (100_000, 1), (1, 100_000), (2, 70_001), (300_001, 3)]


@pytest.mark.parametrize("scale", [1e-30, 1e-25, 1e-18, 1.0, 1e15, 1e18])
def test_norm_is_scale_invariant(L, scale):
    """LinAlg::norm through nrm2 in the reference's backends (f64lapack.rs, f32cuda.rs) neither underflows nor
    overflows on the squares: a vector of 1e-25s has the norm 1e-25 sqrt(n), not 0"""
    rng = np.random.default_rng(7)
    for n in (1, 99, 5000, 300_000):
        x = (rng.standard_normal(n) * scale).astype(np.float32)
        sx = L.Sl.new_ref(x)
        want = float(np.linalg.norm(x.astype(np.float64)))
        got = L.norm(sx)
        sx.drop()
        assert np.isfinite(got) and abs(got - want) <= 1e-5 * want, (n, got, want)


@pytest.mark.parametrize("shape", SHAPES)
def test_transform_ge(L, shape):
    nr, nc = shape
    rng = np.random.default_rng(nr * 131 + nc)
    a = rng.standard_normal((nr, nc)).astype(np.float32)
    am = L.Sl.new_ref(a.ravel(order="F"))
    for tr in (False, True):
        for off in (0, 1):
            x = rng.standard_normal(nr if tr else nc).astype(np.float32)
            y0 = rng.standard_normal(nc if tr else nr).astype(np.float32)
            rx, sx = _sl(L, x, off)
            ry, sy = _sl(L, y0, off)
            for alpha, beta in ((1.0, 0.0), (-0.7, 1.0), (0.5, -0.25)):
                ry2, sy2 = _sl(L, y0, off)
                L.transform_ge(tr, nr, nc, alpha, am, sx, beta, sy2)
                got = sy2.get_ref().astype(np.float64)
                ref = O.transform_ge(tr, nr, nc, alpha, a.astype(np.float64).ravel(order="F"), x, beta, y0)
                scale = np.abs(alpha) * (np.abs(a.T if tr else a).astype(np.float64) @ np.abs(x)) + np.abs(beta * y0)
                # f32 dot of length K: error <= ~K eps sum|a||x| worst case, sqrt(K) typical: 2e-6 * scale * 8
                assert np.all(np.abs(got - ref) <= 2e-5 * scale + 1e-6), (tr, off, alpha, beta)
                ry2.drop()
            rx.drop()
            ry.drop()
    am.drop()


def test_transform_ge_zero_sized_is_scale(L):
    # matop.rs:83-85
    from totsu_amd import MatOp, MatType
    y = np.arange(5, dtype=np.float32)
    ry, sy = _sl(L, y)
    m = MatOp(L, MatType.General(0, 5), np.zeros(0, dtype=np.float32))
    x0 = L.Sl.new_ref(np.zeros(0, dtype=np.float32))
    m.trans_op(2.0, x0, 0.5, sy)
    assert np.array_equal(sy.get_ref(), 0.5 * y)
    ry.drop()


@pytest.mark.parametrize("n", [1, 2, 5, 64, 130])
def test_transform_sp_and_absadd(L, n):
    rng = np.random.default_rng(n)
    sp = rng.standard_normal(n * (n + 1) // 2).astype(np.float32)
    x = rng.standard_normal(n).astype(np.float32)
    y0 = rng.standard_normal(n).astype(np.float32)
    sm = L.Sl.new_ref(sp)
    rx, sx = _sl(L, x)
    ry, sy = _sl(L, y0)
    L.transform_sp(n, 0.7, sm, sx, -0.3, sy)
    ref = O.transform_sp(n, 0.7, sp, x, -0.3, y0)
    assert np.allclose(sy.get_ref(), ref, rtol=1e-4, atol=1e-4)
    ry2, sy2 = _sl(L, y0)
    L.absadd_sympack(n, sm, sy2)
    ref = O.matop_absadd(1, n, n, sp, True, y0)   # typ 1 = SymPack
    assert np.allclose(sy2.get_ref(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(3, 2), (100, 64), (1000, 333), (4100, 257)])
def test_absadd_general(L, shape):
    # MatOp::absadd_impl, matop.rs:98-117
    nr, nc = shape
    rng = np.random.default_rng(nr + nc)
    a = rng.standard_normal((nr, nc)).astype(np.float32)
    am = L.Sl.new_ref(a.ravel(order="F"))
    t0 = rng.uniform(0, 1, nc).astype(np.float32)
    s0 = rng.uniform(0, 1, nr).astype(np.float32)
    rt, st = _sl(L, t0, 1)
    rs, ss = _sl(L, s0, 1)
    L.absadd_cols(nr, nc, am, st)
    L.absadd_rows(nr, nc, am, ss)
    reft = O.matop_absadd(0, nr, nc, a.astype(np.float64).ravel(order="F"), True, t0)
    refs = O.matop_absadd(0, nr, nc, a.astype(np.float64).ravel(order="F"), False, s0)
    assert np.allclose(st.get_ref(), reft, rtol=2e-5)
    assert np.allclose(ss.get_ref(), refs, rtol=2e-5)


def test_cones_single(L):
    from totsu_amd import ConeRPos, ConeRotSOC, ConeSOC, ConeZero
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 3, 100, 2049, 5000):
        x = rng.standard_normal(n).astype(np.float32)
        for cone, typ in ((ConeRPos(L), O.CONE_RPOS), (ConeSOC(L), O.CONE_SOC), (ConeRotSOC(L), O.CONE_ROTSOC),
                          (ConeZero(L), O.CONE_ZERO)):
            for dual in (False, True):
                for shift in (-3.0, 0.0, 3.0):       # force each branch of cone_soc.rs:49-61
                    xx = x.copy()
                    if n:
                        xx[0] += shift * max(np.linalg.norm(x), 1.0)
                    if typ == O.CONE_ROTSOC and n > 1:
                        xx[1] += shift * max(np.linalg.norm(x), 1.0)
                    r, s = _sl(L, xx, 1)
                    assert cone.proj(dual, s)
                    ref = O.proj(typ, xx.astype(np.float64), dual_cone=dual)
                    assert np.allclose(s.get_ref(), ref, rtol=2e-5, atol=2e-5 * max(np.abs(xx).max() if n else 0, 1)), (n, typ, dual, shift)
                    r.drop()


@pytest.mark.parametrize("scale", [1e-30, 1e-25, 1e-18, 1e15])
def test_soc_projection_is_scale_invariant(L, scale):
    """cone_soc.rs:47 takes LinAlg::norm (nrm2): the branch taken and the scaling factor do not depend on the scale of x"""
    from totsu_amd import ConeRotSOC, ConeSOC
    rng = np.random.default_rng(6)
    for n in (2, 3, 100, 2049):
        x = rng.standard_normal(n)
        for cone, typ in ((ConeSOC(L), O.CONE_SOC), (ConeRotSOC(L), O.CONE_ROTSOC)):
            for shift in (-3.0, 0.0, 3.0):
                xx = x.copy()
                xx[0] += shift * np.linalg.norm(x)
                if typ == O.CONE_ROTSOC:
                    xx[1] += shift * np.linalg.norm(x)
                xs = (xx * scale).astype(np.float32)
                r, s = _sl(L, xs, 1)
                assert cone.proj(False, s)
                ref = O.proj(typ, xs.astype(np.float64), dual_cone=False)
                assert np.allclose(s.get_ref(), ref, rtol=2e-5, atol=2e-5 * np.abs(xs).max()), (n, typ, shift)
                r.drop()


def test_soc_batched_ragged(L):
    import ctypes as C
    from totsu_amd._lib import lib
    from totsu_amd.fused import DeviceBuffer
    rng = np.random.default_rng(11)
    lens = [1, 100, 0, 2, 65, 3000, 7, 64, 1]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for rot in (0, 1):
        x = rng.standard_normal(offs[-1]).astype(np.float32)
        for i, ln in enumerate(lens):      # alternate inside / outside / polar
            if ln:
                x[offs[i]] += (i % 3 - 1) * 2.0 * np.linalg.norm(x[offs[i]:offs[i + 1]])
        d = DeviceBuffer.from_host(x)
        od = DeviceBuffer(2 * len(offs))
        lib.thip_h2d(od.ptr, offs.ctypes.data, 2 * len(offs))      # int64 as pairs of floats
        lib.thip_proj_soc_batched(d.ptr, od.ptr, len(lens), rot, max(lens))
        got = d.to_host()
        ref = x.astype(np.float64).copy()
        for i in range(len(lens)):
            ref[offs[i]:offs[i + 1]] = O.proj(O.CONE_ROTSOC if rot else O.CONE_SOC, ref[offs[i]:offs[i + 1]])
        assert np.allclose(got, ref, rtol=2e-5, atol=1e-4)
        # group-min over the same ragged groups (solver.rs:509-520)
        t = rng.uniform(0.1, 1, offs[-1]).astype(np.float32)
        dt = DeviceBuffer.from_host(t)
        lib.thip_group_min_batched(dt.ptr, od.ptr, len(lens), max(lens))
        gt = dt.to_host()
        for i in range(len(lens)):
            if lens[i]:
                assert np.all(gt[offs[i]:offs[i + 1]] == t[offs[i]:offs[i + 1]].min())
        d.free(); od.free(); dt.free()


def test_rng_matches_oracle_bitwise(L):
    from totsu_amd._lib import lib
    from totsu_amd.fused import DeviceBuffer
    d = DeviceBuffer(1000)
    lib.thip_gen_vector(d.ptr, 1000, 7, 3, 12345, 0, 1.0, 0.0)
    u = d.to_host()
    lib.thip_gen_vector(d.ptr, 1000, 7, 3, 12345, 1, 1.0, 0.0)
    g = d.to_host()
    for i in (0, 1, 17, 999):
        assert u[i] == np.float32(O.rng_uniform(7, 3, 12345 + i))
        assert g[i] == np.float32(O.rng_normal(7, 3, 12345 + i))
    m = DeviceBuffer(12 * 5)
    lib.thip_gen_matrix(m.ptr, 12, 5, 12, 1, 2, 100, 10, 1000, 1, 0.5, 0.0)
    mm = m.to_host().reshape((5, 12)).T
    assert mm[3, 2] == np.float32(0.5) * np.float32(O.rng_normal(1, 2, (10 + 2) * 1000 + 100 + 3))
    d.free(); m.free()


@pytest.mark.parametrize("rot", [0, 1])
def test_soc_batched_moreau_decomposition_at_configs2_layout(L, rot):
    # BASELINE.json configs[2]: 1000 cones of 1 + 99 rows in one launch.  Size-independent properties of a projection
    # onto a self-dual cone K (cone_soc.rs:38-65, cone_rotsoc.rs:38-65): P(P(x)) = P(x);  x = P(x) - P(-x);  <P(x), P(-x)> = 0
    from totsu_amd._lib import lib
    from totsu_amd.fused import DeviceBuffer
    rng = np.random.default_rng(3 + rot)
    ncone, ln = 1000, 100
    offs = (np.arange(ncone + 1) * ln).astype(np.int64)
    x = rng.standard_normal(ncone * ln).astype(np.float32)
    x[::ln] *= rng.choice([0.1, 1.0, 10.0], ncone).astype(np.float32) * 5.0       # inside / boundary region / outside mix
    od = DeviceBuffer(2 * len(offs))
    lib.thip_h2d(od.ptr, offs.ctypes.data, 2 * len(offs))

    def proj(v):
        d = DeviceBuffer.from_host(v)
        lib.thip_proj_soc_batched(d.ptr, od.ptr, ncone, rot, ln)
        out = d.to_host()
        d.free()
        return out

    p, q = proj(x), proj(-x)
    scale = np.abs(x).reshape(ncone, ln).max(axis=1).repeat(ln)
    assert np.all(np.abs(proj(p) - p) <= 2e-6 * scale)
    assert np.all(np.abs(p - q - x) <= 4e-6 * scale)
    dots = (p.astype(np.float64) * q).reshape(ncone, ln).sum(axis=1)
    norms = np.linalg.norm(p.reshape(ncone, ln), axis=1) * np.linalg.norm(q.reshape(ncone, ln), axis=1)
    assert np.all(np.abs(dots) <= 1e-5 * (norms + 1e-30) + 1e-30)
    od.free()


# ---- deferred / grouped transform_ge (thip_lazy.hip) ----------------------------------------------------------------
import ctypes as C      # noqa: E402


@pytest.fixture(autouse=True)
def _lazy_off_after_each_test():
    # deferred execution is OPT-IN (thip_set_lazy_gemv): the tests below switch it on for themselves; the library's
    # default (off) is restored whatever happens
    yield
    from totsu_amd._lib import lib
    lib.thip_set_lazy_gemv(0)


def _dev(L, a):
    return L.Sl.new_mut(np.ascontiguousarray(a, dtype=np.float32).copy())


def _blocks(rng, n, nis):
    return [rng.standard_normal((ni, n)).astype(np.float32) for ni in nis]


@pytest.mark.parametrize("lazy", [1, 0])
def test_grouped_block_products_match_numpy(L, lazy):
    """the call pattern of ProbSOCPOpA::op / ::trans_op (socp.rs:77-130): one transform_ge per c_i (n x 1, transposed /
    plain) and per G_i, outputs = consecutive slices of one vector (op) or ONE accumulated vector (trans_op)"""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(5)
    n, nis = 300, [99, 1, 7, 260, 33, 99, 2, 513]
    Gs, cs = _blocks(rng, n, nis), [rng.standard_normal(n).astype(np.float32) for _ in nis]
    m = sum(1 + k for k in nis)
    D = lambda a: _dev(L, np.asfortranarray(a).ravel(order="F"))
    dG, dc = [D(g) for g in Gs], [D(c) for c in cs]
    x, ym = rng.standard_normal(n).astype(np.float32), rng.standard_normal(m).astype(np.float32)
    dx, dy = _dev(L, x), _dev(L, ym)
    lib.thip_set_lazy_gemv(lazy)
    d0, f0 = C.c_int64(), C.c_int64()
    lib.thip_lazy_gemv_stats(C.byref(d0), C.byref(f0))
    alpha, beta = -0.7, 0.3
    done, ref = 0, ym.astype(np.float64).copy()
    for g, c, dg, dcc in zip(Gs, cs, dG, dc):
        ni = g.shape[0]
        lib.thip_transform_ge(1, n, 1, -alpha, dcc.dev(), dx.dev(), beta, dy.dev() + 4 * done)
        lib.thip_transform_ge(0, ni, n, -alpha, dg.dev(), dx.dev(), beta, dy.dev() + 4 * (done + 1))
        ref[done] = beta * ref[done] - alpha * (c.astype(np.float64) @ x)
        ref[done + 1:done + 1 + ni] = beta * ref[done + 1:done + 1 + ni] - alpha * (g.astype(np.float64) @ x)
        done += 1 + ni
    got = dy.get_ref().copy()            # a download: runs whatever is pending
    assert np.allclose(got, ref, rtol=2e-5, atol=2e-5 * np.abs(ref).max())
    # trans_op: y_n = beta y_n + sum_i ( -alpha c_i x_i[0] - alpha G_i^T x_i[1:] )
    yn = rng.standard_normal(n).astype(np.float32)
    dyn = _dev(L, yn)
    lib.thip_scale(n, beta, dyn.dev())
    refn = beta * yn.astype(np.float64)
    done = 0
    for g, c, dg, dcc in zip(Gs, cs, dG, dc):
        ni = g.shape[0]
        lib.thip_transform_ge(0, n, 1, -alpha, dcc.dev(), dy.dev() + 4 * done, 1.0, dyn.dev())
        lib.thip_transform_ge(1, ni, n, -alpha, dg.dev(), dy.dev() + 4 * (done + 1), 1.0, dyn.dev())
        refn += -alpha * c.astype(np.float64) * got[done] - alpha * (g.astype(np.float64).T @ got[done + 1:done + 1 + ni])
        done += 1 + ni
    gotn = dyn.get_ref().copy()
    assert np.allclose(gotn, refn, rtol=2e-5, atol=2e-5 * np.abs(refn).max())
    # ProbSOCPOpB::trans_op (socp.rs:219-246): one number accumulates a scale, and per cone an add and a dot
    hs = [rng.standard_normal(k).astype(np.float32) for k in nis]
    dh = [_dev(L, h) for h in hs]
    ds = _dev(L, np.array([0.37], np.float32))
    lib.thip_scale(1, beta, ds.dev())
    refs = beta * 0.37
    done = 0
    for i, (h, dhh) in enumerate(zip(hs, dh)):
        ni = len(h)
        lib.thip_add(1, alpha * (i + 1.5), dy.dev() + 4 * done, ds.dev())
        lib.thip_transform_ge(1, ni, 1, alpha, dhh.dev(), dy.dev() + 4 * (done + 1), 1.0, ds.dev())
        refs += alpha * (i + 1.5) * float(got[done]) + alpha * float(h.astype(np.float64) @ got[done + 1:done + 1 + ni])
        done += 1 + ni
    gots = float(ds.get_ref()[0])
    assert abs(gots - refs) <= 2e-5 * (1 + abs(refs)), (gots, refs)
    d1, f1 = C.c_int64(), C.c_int64()
    lib.thip_lazy_gemv_stats(C.byref(d1), C.byref(f1))
    if lazy:
        # 4 + 2 recorded calls per block, the two scales: three grouped flushes in all
        assert d1.value - d0.value == 6 * len(nis) + 2 and f1.value - f0.value == 3
    else:
        assert d1.value == d0.value
    for s in dG + dc + dh + [dx, dy, dyn, ds]:
        s.drop()


def test_deferred_products_respect_data_hazards(L):
    """a product that reads (or overwrites) what a pending one writes must see it done: y1 = A x ; y2 = B y1 ;
    x <- C y2 (overwrites an input of the first) ; y1 <- 2 y1 + A x"""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(6)
    k = 40
    A, B, Cm = (rng.standard_normal((k, k)).astype(np.float32) for _ in range(3))
    x = rng.standard_normal(k).astype(np.float32)
    D = lambda a: _dev(L, np.asfortranarray(a).ravel(order="F"))
    dA, dB, dC = D(A), D(B), D(Cm)
    dx, y1, y2 = _dev(L, x), _dev(L, np.zeros(k, np.float32)), _dev(L, np.zeros(k, np.float32))
    lib.thip_set_lazy_gemv(1)
    lib.thip_transform_ge(0, k, k, 1.0, dA.dev(), dx.dev(), 0.0, y1.dev())
    lib.thip_transform_ge(0, k, k, 1.0, dB.dev(), y1.dev(), 0.0, y2.dev())        # RAW on y1
    lib.thip_transform_ge(1, k, k, 1.0, dC.dev(), y2.dev(), 0.0, dx.dev())        # RAW on y2, WAR on x
    lib.thip_transform_ge(0, k, k, 1.0, dA.dev(), dx.dev(), 2.0, y1.dev())        # WAW on y1, RAW on x
    r1 = A.astype(np.float64) @ x
    r2 = B.astype(np.float64) @ r1
    rx = Cm.astype(np.float64).T @ r2
    r1b = 2.0 * r1 + A.astype(np.float64) @ rx
    assert np.allclose(y2.get_ref(), r2, rtol=1e-4, atol=1e-4 * np.abs(r2).max())
    assert np.allclose(dx.get_ref(), rx, rtol=1e-4, atol=1e-4 * np.abs(rx).max())
    assert np.allclose(y1.get_ref(), r1b, rtol=1e-4, atol=1e-4 * np.abs(r1b).max())
    for s in (dA, dB, dC, dx, y1, y2):
        s.drop()


def test_grouped_products_at_scale_match_the_stacked_matrix(L):
    """200 blocks of 99 x 20 000 (+ 200 row vectors), 1.6 GB: the block-by-block products of a composite operator through
    the deferred / grouped path against ONE transform_ge on the stacked matrix (the fused loop's form), both directions"""
    from totsu_amd._lib import lib
    from totsu_amd.fused import DeviceBuffer as D
    n, nb, ni = 20_000, 200, 99
    rows = 1 + ni
    m = nb * rows
    A = D(m * n)
    lib.thip_gen_matrix(A.ptr, m, n, m, 0, 1, 0, 0, m, 1, 1.0, 0.0)
    x, y = D(n), D(m)
    lib.thip_gen_vector(x.ptr, n, 0, 2, 0, 1, 1.0, 0.0)
    lib.thip_gen_vector(y.ptr, m, 0, 3, 0, 1, 1.0, 0.0)
    # per-block copies (own contiguous arrays, as the reference's MatOps hold them): row 0 of a block -> c_i, rows 1.. -> G_i
    G, Cv = D(nb * ni * n), D(nb * n)
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    lib.thip_sync()
    for i in range(nb):
        assert hip.hipMemcpy2D(G.ptr + 4 * i * ni * n, 4 * ni, A.ptr + 4 * (i * rows + 1), 4 * m, 4 * ni, n, 3) == 0
        assert hip.hipMemcpy2D(Cv.ptr + 4 * i * n, 4, A.ptr + 4 * (i * rows), 4 * m, 4, n, 3) == 0
    lib.thip_set_lazy_gemv(1)
    # op: y_blk = A_blk x
    out, ref = D(m, zero=True), D(m, zero=True)
    for i in range(nb):
        lib.thip_transform_ge(1, n, 1, 1.0, Cv.ptr + 4 * i * n, x.ptr, 0.0, out.ptr + 4 * i * rows)
        lib.thip_transform_ge(0, ni, n, 1.0, G.ptr + 4 * i * ni * n, x.ptr, 0.0, out.ptr + 4 * (i * rows + 1))
    lib.thip_transform_ge(0, m, n, 1.0, A.ptr, x.ptr, 0.0, ref.ptr)        # 1.6 GB: runs at once, after the record
    a, b = out.to_host(), ref.to_host()
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
    # trans_op: z = sum_blk A_blk^T y_blk
    outn, refn = D(n, zero=True), D(n, zero=True)
    lib.thip_scale(n, 0.0, outn.ptr)
    for i in range(nb):
        lib.thip_transform_ge(0, n, 1, 1.0, Cv.ptr + 4 * i * n, y.ptr + 4 * i * rows, 1.0, outn.ptr)
        lib.thip_transform_ge(1, ni, n, 1.0, G.ptr + 4 * i * ni * n, y.ptr + 4 * (i * rows + 1), 1.0, outn.ptr)
    lib.thip_transform_ge(1, m, n, 1.0, A.ptr, y.ptr, 0.0, refn.ptr)
    a, b = outn.to_host(), refn.to_host()
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max()
    for d in (A, x, y, G, Cv, out, ref, outn, refn):
        d.free()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_deferred_execution_fuzz_against_eager(L, seed):
    """random sequences of transform_ge / scale / add on random (overlapping) windows of three vectors, with matrices
    and vectors of every shape class (matrix, row vector, column vector): deferred + grouped execution must give what
    one launch per call gives (same values up to the order of f32 additions)"""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(100 + seed)
    NV = 700
    mats = []
    for _ in range(10):
        nr, nc = int(rng.choice([1, 1, 3, 17, 64, 130, 300])), int(rng.choice([1, 1, 5, 40, 129, 260]))
        mats.append((nr, nc, rng.standard_normal((nr, nc)).astype(np.float32)))
    dm = [_dev(L, np.asfortranarray(a).ravel(order="F")) for _, _, a in mats]
    init = [rng.standard_normal(NV).astype(np.float32) for _ in range(3)]
    ops = []
    for _ in range(120):
        kind = rng.choice(["ge", "ge", "ge", "scale", "add"])
        if kind == "ge":
            k = int(rng.integers(len(mats)))
            nr, nc, _ = mats[k]
            tr = int(rng.integers(2))
            xin, yout = (nr, nc) if tr else (nc, nr)
            vi, vo = int(rng.integers(3)), int(rng.integers(3))
            ox, oy = int(rng.integers(0, NV - xin + 1)), int(rng.integers(0, NV - yout + 1))
            if vi == vo and ox < oy + yout and oy < ox + xin:
                continue                      # a product must not alias its own input and output (as in the reference)
            ops.append(("ge", k, tr, float(rng.choice([1.0, -0.5, 0.25])), vi, ox, float(rng.choice([0.0, 1.0, 1.0, 0.5])), vo, oy))
        elif kind == "scale":
            ln = int(rng.choice([1, 7, 300]))
            ops.append(("scale", float(rng.choice([0.0, 0.5, 1.0, -1.0])), int(rng.integers(3)), int(rng.integers(0, NV - ln + 1)), ln))
        else:
            ln = int(rng.choice([1, 9, 200]))
            vi, vo = int(rng.integers(3)), int(rng.integers(3))
            ox, oy = int(rng.integers(0, NV - ln + 1)), int(rng.integers(0, NV - ln + 1))
            if vi == vo and ox != oy and ox < oy + ln and oy < ox + ln:
                continue
            ops.append(("add", float(rng.choice([1.0, -2.0, 0.125])), vi, ox, vo, oy, ln))

    def run(lazy):
        lib.thip_set_lazy_gemv(lazy)
        vs = [_dev(L, v) for v in init]
        for op in ops:
            if op[0] == "ge":
                _, k, tr, al, vi, ox, be, vo, oy = op
                nr, nc, _ = mats[k]
                lib.thip_transform_ge(tr, nr, nc, al, dm[k].dev(), vs[vi].dev() + 4 * ox, be, vs[vo].dev() + 4 * oy)
            elif op[0] == "scale":
                _, al, vi, ox, ln = op
                lib.thip_scale(ln, al, vs[vi].dev() + 4 * ox)
            else:
                _, al, vi, ox, vo, oy, ln = op
                lib.thip_add(ln, al, vs[vi].dev() + 4 * ox, vs[vo].dev() + 4 * oy)
        out = [v.get_ref().copy() for v in vs]
        for v in vs:
            v.drop()
        return out

    eager, lazy = run(0), run(1)
    for a, b in zip(eager, lazy):
        assert np.all(np.isfinite(a)) == np.all(np.isfinite(b))
        sc = max(1.0, float(np.abs(a[np.isfinite(a)]).max()) if np.isfinite(a).any() else 1.0)
        assert np.allclose(a, b, rtol=2e-4, atol=2e-4 * sc, equal_nan=True), np.nanmax(np.abs(a - b)) / sc
    for d in dm:
        d.drop()


def test_deferred_execution_is_opt_in_and_suspended_on_a_caller_stream(L):
    """the header's ordering contract: by default, and always while a caller-provided stream is installed, every call has
    been ENQUEUED when it returns (a caller interleaving its own work on the stream must see it ordered)"""
    from totsu_amd._lib import lib
    on = C.c_int(-1)
    lib.thip_get_lazy_gemv(C.byref(on))
    assert on.value == 0                                   # the library's default
    k = 24
    rng = np.random.default_rng(3)
    A = rng.standard_normal((k, k)).astype(np.float32)
    dA, dx, dy = _dev(L, np.asfortranarray(A).ravel(order="F")), _dev(L, np.ones(k, np.float32)), _dev(L, np.zeros(k, np.float32))
    d0, f0, d1, f1 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    lib.thip_lazy_gemv_stats(C.byref(d0), C.byref(f0))
    lib.thip_transform_ge(0, k, k, 1.0, dA.dev(), dx.dev(), 0.0, dy.dev())
    lib.thip_lazy_gemv_stats(C.byref(d1), C.byref(f1))
    assert d1.value == d0.value                            # not deferred: off by default
    # opted in, but with a caller stream installed: still not deferred, and the caller's own copy on that stream,
    # enqueued right behind the call, sees the product
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p()
    assert hip.hipStreamCreateWithFlags(C.byref(stream), 1) == 0
    lib.thip_sync()
    lib.thip_set_lazy_gemv(1)
    lib.thip_set_stream(stream)
    lib.thip_transform_ge(0, k, k, 2.0, dA.dev(), dx.dev(), 0.0, dy.dev())
    lib.thip_lazy_gemv_stats(C.byref(d1), C.byref(f1))
    assert d1.value == d0.value
    host = np.zeros(k, np.float32)
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    assert hip.hipMemcpyAsync(host.ctypes.data, dy.dev(), 4 * k, 2, stream) == 0
    assert hip.hipStreamSynchronize(stream) == 0
    assert np.allclose(host, 2.0 * A.sum(axis=1), rtol=1e-5, atol=1e-5)
    lib.thip_set_stream(None)
    # back on the library's own stream the opt-in takes effect, and thip_get_stream runs the record before handing it out
    lib.thip_transform_ge(0, k, k, 3.0, dA.dev(), dx.dev(), 0.0, dy.dev())
    lib.thip_lazy_gemv_stats(C.byref(d1), C.byref(f1))
    assert d1.value == d0.value + 1
    own = lib.thip_get_stream()
    assert own
    assert hip.hipMemcpyAsync(host.ctypes.data, dy.dev(), 4 * k, 2, C.c_void_p(own)) == 0
    assert hip.hipStreamSynchronize(C.c_void_p(own)) == 0
    assert np.allclose(host, 3.0 * A.sum(axis=1), rtol=1e-5, atol=1e-5)
    hip.hipStreamDestroy(stream)
    for s_ in (dA, dx, dy):
        s_.drop()


def _plan_stats(lib):
    h, m = C.c_int64(), C.c_int64()
    lib.thip_lazy_plan_stats(C.byref(h), C.byref(m))
    return h.value, m.value


def test_repeated_call_sequence_replays_its_plans_with_new_factors(L):
    """The loop repeats its call sequence every iteration: the second pass must hit the plans of the first (no analysis, no
    table build) -- also when alpha / beta differ from pass to pass (criteria_conv passes 1 / tau, solver.rs:594-597) and
    when the data changed -- and give what one launch per call gives.  An N and a T product of the same block in one
    segment share one read of it (pairing); a run of projections on disjoint slices is one launch."""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(21)
    n, nis = 300, [5, 99, 1, 64, 130, 7]
    m = sum(1 + k for k in nis)
    G = [rng.standard_normal((k, n)).astype(np.float32) for k in nis]
    cv = [rng.standard_normal(n).astype(np.float32) for _ in nis]
    dG = [_dev(L, np.asfortranarray(g).ravel(order="F")) for g in G]
    dc = [_dev(L, v) for v in cv]

    def one_pass(alpha, beta, x, y, xt, yt, proj):
        # ProbSOCPOpA::op then ::trans_op (socp.rs:77-130) on the same blocks, then the cones of ProbSOCPCone::proj
        off = 0
        for i, k in enumerate(nis):
            lib.thip_transform_ge(1, n, 1, -alpha, dc[i].dev(), x.dev(), beta, y.dev() + 4 * off)
            lib.thip_transform_ge(0, k, n, -alpha, dG[i].dev(), x.dev(), beta, y.dev() + 4 * (off + 1))
            off += 1 + k
        lib.thip_scale(n, beta, yt.dev())
        off = 0
        for i, k in enumerate(nis):
            lib.thip_transform_ge(0, n, 1, -alpha, dc[i].dev(), xt.dev() + 4 * off, 1.0, yt.dev())
            lib.thip_transform_ge(1, k, n, -alpha, dG[i].dev(), xt.dev() + 4 * (off + 1), 1.0, yt.dev())
            off += 1 + k
        off = 0
        for k in nis:
            lib.thip_proj_soc(1 + k, proj.dev() + 4 * off)
            off += 1 + k

    def run(lazy, passes):
        lib.thip_set_lazy_gemv(lazy)
        x = _dev(L, np.zeros(n, np.float32))
        y = _dev(L, np.zeros(m, np.float32))
        xt = _dev(L, np.zeros(m, np.float32))
        yt = _dev(L, np.zeros(n, np.float32))
        pj = _dev(L, np.zeros(m, np.float32))
        outs = []
        for p_, (alpha, beta, seed) in enumerate(passes):
            r2 = np.random.default_rng(seed)
            for d, ln in ((x, n), (xt, m), (pj, m), (y, m), (yt, n)):
                h = r2.standard_normal(ln).astype(np.float32)
                lib.thip_h2d(d.dev(), h.ctypes.data, ln)
            one_pass(alpha, beta, x, y, xt, yt, pj)
            outs.append((y.get_ref().copy(), yt.get_ref().copy(), pj.get_ref().copy()))
        for d in (x, y, xt, yt, pj):
            d.drop()
        return outs

    passes = [(1.0, 0.0, 1), (1.0, 0.0, 2), (0.37, 0.0, 3), (2.5, -0.75, 4), (2.5, -0.75, 5)]
    eager = run(0, passes)
    h0, m0 = _plan_stats(lib)
    lazy = run(1, passes)
    h1, m1 = _plan_stats(lib)
    lib.thip_set_lazy_gemv(0)
    # first pass builds its plans (misses); the later ones replay them -- with new factors in passes 3 and 4
    assert m1 - m0 >= 2 and h1 - h0 >= 2 * (len(passes) - 2), (h1 - h0, m1 - m0)
    for (ya, yta, pa), (yb, ytb, pb) in zip(eager, lazy):
        sc = max(1.0, float(np.abs(ya).max()), float(np.abs(yta).max()))
        assert np.allclose(ya, yb, rtol=2e-5, atol=2e-5 * sc), np.abs(ya - yb).max() / sc
        assert np.allclose(yta, ytb, rtol=2e-5, atol=2e-5 * sc), np.abs(yta - ytb).max() / sc
        assert np.array_equal(pa, pb)            # the projection of a slice does not depend on how it was launched
    for d in dG + dc:
        d.drop()


def test_deferred_projections_respect_order_and_overlaps(L):
    """projections join the record: a product that reads a slice pending projection, a projection of a slice a pending
    product writes, two projections of overlapping slices, kinds that alternate -- all must equal one launch per call"""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(33)
    k = 48
    A = rng.standard_normal((k, k)).astype(np.float32)
    dA = _dev(L, np.asfortranarray(A).ravel(order="F"))

    def run(lazy):
        lib.thip_set_lazy_gemv(lazy)
        v = _dev(L, np.linspace(-2, 3, 4 * k).astype(np.float32))
        w = _dev(L, np.zeros(k, np.float32))
        lib.thip_proj_soc(k, v.dev())                                  # slice 0
        lib.thip_proj_rotsoc(k, v.dev() + 4 * k)                       # another kind: new run
        lib.thip_proj_soc(k, v.dev() + 4 * 2 * k)                      # back to SOC
        lib.thip_proj_soc(k // 2, v.dev() + 4 * (2 * k + 10))          # overlaps the previous slice: must come after it
        lib.thip_transform_ge(0, k, k, 1.0, dA.dev(), v.dev(), 0.0, w.dev())        # reads a projected slice
        lib.thip_proj_rpos(k, w.dev())                                 # projects what the product writes
        lib.thip_transform_ge(1, k, k, 0.5, dA.dev(), w.dev(), 1.0, v.dev() + 4 * 3 * k)
        lib.thip_proj_zero(0, 7, v.dev() + 4 * 3 * k)
        lib.thip_proj_zero(1, 7, v.dev() + 4 * (3 * k + 7))            # dual cone: identity
        out = (v.get_ref().copy(), w.get_ref().copy())
        v.drop(); w.drop()
        return out

    (va, wa), (vb, wb) = run(0), run(1)
    lib.thip_set_lazy_gemv(0)
    assert np.allclose(va, vb, rtol=1e-5, atol=1e-5) and np.allclose(wa, wb, rtol=1e-5, atol=1e-5)
    assert np.all(wa >= 0) and np.all(va[3 * k:3 * k + 7] == 0)
    dA.drop()


def test_big_block_pair_shares_one_pass(L):
    """a block above 64 MB (the single G of a ProbLP, lp.rs:76-98) recorded with both of its products: the flush runs ONE
    dual GEMV over it for both; results equal the two separate products"""
    from totsu_amd._lib import lib
    D = __import__("totsu_amd").DeviceBuffer
    nr, nc = 6000, 3000                       # 18e6 entries = 72 MB
    A = D(nr * nc)
    lib.thip_gen_matrix(A.ptr, nr, nc, nr, 7, 1, 0, 0, nr, 1, 1.0, 0.0)
    x, xt = D(nc), D(nr)
    lib.thip_gen_vector(x.ptr, nc, 7, 2, 0, 1, 1.0, 0.0)
    lib.thip_gen_vector(xt.ptr, nr, 7, 3, 0, 1, 1.0, 0.0)
    res = {}
    for lazy in (0, 1):
        lib.thip_set_lazy_gemv(lazy)
        y, yt = D(nr, zero=True), D(nc, zero=True)
        lib.thip_transform_ge(1, nr, nc, -1.0, A.ptr, xt.ptr, 0.0, yt.ptr)
        lib.thip_transform_ge(0, nr, nc, 2.0, A.ptr, x.ptr, 0.0, y.ptr)
        res[lazy] = (y.to_host(), yt.to_host())
        y.free(); yt.free()
    lib.thip_set_lazy_gemv(0)
    for a, b in zip(res[0], res[1]):
        assert np.allclose(a, b, rtol=1e-4, atol=1e-4 * np.abs(a).max())
    for d in (A, x, xt):
        d.free()


def test_read_ahead_of_the_reference_cone_loop(L):
    """The reference's ConeSOC::proj, literally (cone_soc.rs:38-65): per cone get(0) + norm(v) on the host, then scale / set.
    With deferred execution on, the reads of a pass are learnt and the next pass fetches them all at once; results must
    be those of one blocking read per call -- also when the data changes from pass to pass, when a pass takes another
    branch, and when a write lands in a range still to be read (the cache must be dropped, not served)."""
    from totsu_amd._lib import lib
    rng = np.random.default_rng(77)
    lens = [5, 100, 100, 33, 1, 64, 100, 7, 100, 100, 12, 100, 100, 100, 50, 100, 100, 3, 100, 100]      # 20 cones
    total = sum(lens)

    def cone_pass(x, poke=None):
        off = 0
        for i, ln in enumerate(lens):
            s0, hn = C.c_float(), C.c_float()
            lib.thip_get(x.dev() + 4 * off, 0, C.byref(s0))
            lib.thip_norm(ln - 1, x.dev() + 4 * (off + 1), C.byref(hn))
            vs, nv = s0.value, hn.value
            if nv <= -vs:
                lib.thip_scale(ln - 1, 0.0, x.dev() + 4 * (off + 1))
                lib.thip_set(x.dev() + 4 * off, 0, 0.0)
            elif nv > vs:
                lib.thip_scale(ln - 1, (1.0 + vs / nv) / 2.0, x.dev() + 4 * (off + 1))
                lib.thip_set(x.dev() + 4 * off, 0, (nv + vs) / 2.0)
            if poke is not None and i == poke[0]:
                # a write into a cone still to be read: the read-ahead must not serve its stale value
                lib.thip_set(x.dev() + 4 * poke[1], 0, 50.0)
            off += ln
        return x.get_ref().copy()

    def run(lazy):
        lib.thip_set_lazy_gemv(lazy)
        x = _dev(L, np.zeros(total, np.float32))
        outs = []
        for p_ in range(6):
            h = np.random.default_rng(100 + p_).standard_normal(total).astype(np.float32) * (3.0 if p_ % 2 else 0.3)
            if p_ == 4:
                h[0] = 40.0                      # cone 0 inside: another branch than the pass before
            lib.thip_h2d(x.dev(), h.ctypes.data, total)
            outs.append(cone_pass(x, poke=(2, sum(lens[:9])) if p_ == 5 else None))
        x.drop()
        return outs

    eager = run(0)
    s0, f0, s1, f1 = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
    lib.thip_lazy_read_stats(C.byref(s0), C.byref(f0))
    lazy = run(1)
    lib.thip_lazy_read_stats(C.byref(s1), C.byref(f1))
    lib.thip_set_lazy_gemv(0)
    for a, b in zip(eager, lazy):
        assert np.allclose(a, b, rtol=2e-6, atol=2e-6), np.abs(a - b).max()
    # pass 0 learns (40 blocking reads); passes 1 .. 5 are fetched at once (the poked one falls back part of the way)
    assert f1.value - f0.value >= 4 and s1.value - s0.value >= 4 * 2 * len(lens), (s1.value - s0.value, f1.value - f0.value)
    # the SOC projection really happened: every cone ends inside its cone
    off = 0
    for ln in lens:
        blk = lazy[3][off:off + ln]
        assert np.linalg.norm(blk[1:]) <= blk[0] * (1 + 1e-5) + 1e-6
        off += ln
