"""GPU parity tests for the reduced-precision STORAGE of A: bf16 elements, f32 accumulation.
The stored matrix is a different (rounded) matrix, so parity is stated against the CPU oracle run ON THE ROUNDED
MATRIX (bit-exact conversion first, then products, then iterates), and the distance to the exact-matrix answer is
measured separately."""
import ctypes as C

import numpy as np
import pytest

import oracle as O
from problems import benchmark_lp, random_socp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import totsu_amd
    from totsu_amd import _lib
    _lib.init()
    return totsu_amd


def bf16_bits(a):
    """round-to-nearest-even f32 -> bf16 bit patterns (numpy restatement of the conversion rule)"""
    b = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    r = (b + 0x7fff + ((b >> 16) & 1)) >> 16
    special = (b & 0x7f800000) == 0x7f800000
    r = np.where(special, (b >> 16) | np.where((b & 0xffff) != 0, np.uint64(0x40), np.uint64(0)), r)
    return r.astype(np.uint16)


def bf16_round(a):
    return (bf16_bits(a).astype(np.uint32) << 16).view(np.float32).reshape(np.shape(a))


class U16Buffer:
    """device array of 16-bit patterns (allocated as floats)"""

    def __init__(self, T, count):
        self.count = count
        self.buf = T.DeviceBuffer((count + 1) // 2 + 4, zero=True)

    def to_host(self):
        return self.buf.to_host().view(np.uint16)[:self.count]


@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (8, 5), (13, 9), (256, 17), (2049, 33)])
def test_to_bf16_is_bit_exact(T, shape):
    from totsu_amd._lib import lib
    m, n = shape
    rng = np.random.default_rng(m * 131 + n)
    a = (rng.standard_normal((m, n)) * 10.0 ** rng.integers(-30, 30, (m, n))).astype(np.float32)
    flat = a.ravel(order="F")
    # ties, subnormals, signed zeros, the largest finite value (rounds to inf), inf and nan
    specials = np.array([1.00390625, 1.01171875, -1.00390625, 1e-40, -1e-45, 0.0, -0.0, 3.4028235e38, np.inf, -np.inf,
                         np.nan], np.float32)
    k = min(flat.size, specials.size)
    flat[:k] = specials[:k]
    a = flat.reshape((n, m)).T
    ld = (m + 7) // 8 * 8
    src = T.DeviceBuffer.from_host(np.asfortranarray(a).ravel(order="F"))
    dst = U16Buffer(T, ld * n)
    lib.thip_to_bf16(m, n, src.ptr, dst.buf.ptr, ld)
    got = dst.to_host().reshape((n, ld)).T
    want = bf16_bits(a)
    nan = np.isnan(a)
    assert np.array_equal(got[:m][~nan], want[~nan])
    assert np.all(np.isnan((got[:m][nan].astype(np.uint32) << 16).view(np.float32)))
    assert not got[m:].any()                       # padding rows are zeros


@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (8, 8), (100, 30), (2048, 40), (2049, 65), (5000, 1500), (64, 9000),
                                   (20_000, 700)])
def test_gemv_bf16_equals_products_of_the_rounded_matrix(T, shape):
    from totsu_amd._lib import lib
    m, n = shape
    rng = np.random.default_rng(m + 7 * n)
    a = rng.standard_normal((m, n)).astype(np.float32)
    ar = bf16_round(a).astype(np.float64)
    ld = (m + 7) // 8 * 8
    src = T.DeviceBuffer.from_host(np.asfortranarray(a).ravel(order="F"))
    a16 = U16Buffer(T, ld * n)
    lib.thip_to_bf16(m, n, src.ptr, a16.buf.ptr, ld)
    x = rng.standard_normal(n).astype(np.float32)
    y0 = rng.standard_normal(m).astype(np.float32)
    dx, dy = T.DeviceBuffer.from_host(x), T.DeviceBuffer.from_host(y0)
    lib.thip_transform_ge_bf16(0, m, n, 1.5, a16.buf.ptr, ld, dx.ptr, -0.5, dy.ptr)
    want = 1.5 * ar @ x.astype(np.float64) - 0.5 * y0
    scale = np.abs(ar) @ np.abs(x.astype(np.float64)) + np.abs(y0)
    assert np.all(np.abs(dy.to_host() - want) <= 4e-6 * scale + 1e-30)
    v = rng.standard_normal(m).astype(np.float32)
    w0 = rng.standard_normal(n).astype(np.float32)
    dv, dw = T.DeviceBuffer.from_host(v), T.DeviceBuffer.from_host(w0)
    lib.thip_transform_ge_bf16(1, m, n, -2.0, a16.buf.ptr, ld, dv.ptr, 1.0, dw.ptr)
    want = -2.0 * ar.T @ v.astype(np.float64) + w0
    scale = 2.0 * np.abs(ar).T @ np.abs(v.astype(np.float64)) + np.abs(w0)
    assert np.all(np.abs(dw.to_host() - want) <= 4e-6 * scale + 1e-30)


def test_gemv_bf16_unaligned_leading_dimension_takes_the_scalar_path(T):
    from totsu_amd._lib import lib
    m, n = 37, 11
    rng = np.random.default_rng(5)
    a = rng.standard_normal((m, n)).astype(np.float32)
    bits = bf16_bits(a)                                   # ld = m = 37: not a multiple of 8
    raw = np.zeros(((m * n + 1) // 2 + 4) * 2, np.uint16)
    raw[:m * n] = bits.ravel(order="F")
    a16 = T.DeviceBuffer.from_host(raw.view(np.float32))
    x = rng.standard_normal(n).astype(np.float32)
    dx, dy = T.DeviceBuffer.from_host(x), T.DeviceBuffer(m, zero=True)
    lib.thip_transform_ge_bf16(0, m, n, 1.0, a16.ptr, m, dx.ptr, 0.0, dy.ptr)
    want = bf16_round(a).astype(np.float64) @ x
    assert np.allclose(dy.to_host(), want, rtol=0, atol=1e-5 * np.abs(want).max())


def _rounded(dense):
    import copy
    d = copy.copy(dense)
    d.mat_a = bf16_round(np.asarray(dense.mat_a, np.float32))
    return d


def _mb(T, typ):
    return T.MatBuild(T.F32HIP, typ)


def _socp(T, n, cones, seed):
    f, Gs, hs, cs, d = random_socp(n, cones, seed=seed)
    return T.ProbSOCP(_mb(T, T.MatType.General(n, 1)).set_array(f.reshape(-1, 1)),
                      [_mb(T, T.MatType.General(G.shape[0], n)).set_array(G) for G in Gs],
                      [_mb(T, T.MatType.General(len(h_), 1)).set_array(h_.reshape(-1, 1)) for h_ in hs],
                      [_mb(T, T.MatType.General(n, 1)).set_array(c_.reshape(-1, 1)) for c_ in cs], d,
                      _mb(T, T.MatType.General(0, n)), _mb(T, T.MatType.General(0, 1)))


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried", "sweep"])
def test_iterates_with_bf16_storage_follow_the_oracle_on_the_rounded_matrix(T, schedule):
    # (the one-pass kernel wants >= 80 columns for its two-column panels of a 16-bit matrix: a larger instance for it)
    dense = (_socp(T, 100, [5, 1, 0, 17, 99, 3, 32], seed=2) if schedule == "sweep" else _socp(T, 30, [5, 1, 0, 17, 99, 3], seed=2)).dense()
    iters, tols = [0, 1, 9, 99], [2e-5, 2e-5, 1e-4, 2e-3]
    dr = _rounded(dense)
    ro = O.solve_matop_cones(O.param(max_iter=max(iters) + 2, eps_acc=1e-30), dr.vec_c, dr.mat_a, dr.vec_b, dr.seg_type,
                             dr.seg_len, snap_iters=iters, trace_cap=max(iters) + 3, use_ql=True)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(dense, p, schedule, a_storage="bf16", sweep_min_bytes=0)      # the UNROUNDED matrix goes in
    assert fs.passes()[1] == dense.n * dense.m * 2 and fs.schedule_in_use() == schedule
    assert fs.passes()[0] == {"reference": 6, "fused": 3, "carried": 2, "sweep": 1}[schedule]
    N = dense.n + 2 * dense.m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.abs(x - rx).max() <= tol * max(np.abs(rx).max(), 1e-6), (schedule, it)
        assert np.abs(y - ry).max() <= tol * max(np.abs(ry).max(), 1e-6), (schedule, it)
    fs.destroy()


@pytest.mark.parametrize("n,cones,seed", [(30, [5, 1, 0, 17, 99, 3], 2), (80, [20] * 10, 5)])
def test_bf16_passes_then_f32_passes_finish_on_the_exact_matrix(T, n, cones, seed):
    # bf16 storage solves a NEARBY problem at half the bytes per pass (objective off by ~3e-4 relative: outside the
    # 1e-4 gate of SURVEY.md 8d/8f-4); switching the running iteration to the f32 matrix (thip_solver_set_a_storage +
    # thip_solver_resume) finishes on the exact one, and the answer is then as good as an all-f32 solve
    dense = _socp(T, n, cones, seed).dense()
    ro = O.solve_matop_cones(O.param(max_iter=2_000_000, eps_acc=1e-7), dense.vec_c, dense.mat_a, dense.vec_b,
                             dense.seg_type, dense.seg_len)
    assert ro.status == 0
    obj = float(np.dot(dense.vec_c, ro.x))
    gap = lambda x: abs(float(np.dot(dense.vec_c, x)) - obj) / abs(obj)
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 2_000_000, 1e-4
    for sched in ("fused", "carried") + (("sweep",) if n >= 80 else ()):
        fs = T.FusedSolver.from_dense(dense, p, sched, sweep_min_bytes=0)
        assert fs.schedule_in_use() == sched
        r0 = fs.run(-1, poll_every=64)
        assert r0.state == 0 and gap(fs.solution()[0]) < 1e-4
        fs.destroy()
        fs = T.FusedSolver.from_dense(dense, p, sched, a_storage="bf16", sweep_min_bytes=0)
        assert fs.schedule_in_use() == sched
        r1 = fs.run(-1, poll_every=64)
        assert r1.state == 0
        gap16 = gap(fs.solution()[0])
        assert 1e-5 < gap16 < 3e-3, gap16                     # a nearby problem, not the same one
        fs.set_a_storage("f32")
        assert fs.passes()[1] == dense.n * dense.m * 4 and fs.schedule_in_use() == sched
        fs.resume()
        r2 = fs.run(-1, poll_every=64)
        assert r2.state == 0
        gap32 = gap(fs.solution()[0])
        assert gap32 < 1e-4, (sched, gap16, gap32)
        assert r2.iters - r1.iters < 0.5 * r1.iters, (r1.iters, r2.iters)   # the f32 phase is the short one
        fs.destroy()


def test_resume_with_tighter_eps_continues_the_same_solve(T):
    dense = _socp(T, 30, [5, 1, 0, 17, 99, 3], 2).dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 2_000_000, 1e-3
    fs = T.FusedSolver.from_dense(dense, p, "carried")
    r1 = fs.run(-1, poll_every=16)
    assert r1.state == 0
    p2 = T.SolverParam()
    p2.max_iter, p2.eps_acc = 2_000_000, 1e-4
    fs.resume(p2)
    r2 = fs.run(-1, poll_every=16)
    assert r2.state == 0 and r2.iters > r1.iters and max(r2.cri) <= 1e-4
    x2, _ = fs.solution()
    fs.destroy()
    fs = T.FusedSolver.from_dense(dense, p2, "carried")
    r3 = fs.run(-1, poll_every=16)
    x3, _ = fs.solution()
    fs.destroy()
    # stopping and resuming costs one 1/tau scaling round trip (1 ulp per element): same iteration count +-1, same answer
    assert abs(r3.iters - r2.iters) <= 2, (r2.iters, r3.iters)
    assert np.allclose(x2, x3, rtol=1e-4, atol=1e-5 * np.abs(x3).max())


def test_resume_after_excess_iter_reaches_the_same_answer(T):
    dense = _socp(T, 30, [5, 1, 0, 17, 99, 3], 2).dense()
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 100, 1e-4
    fs = T.FusedSolver.from_dense(dense, p, "carried")
    r1 = fs.run(-1, poll_every=16)
    assert r1.state == 3 and r1.iters == 99              # SolverError::ExcessIter at i + 1 >= max_iter (solver.rs:424-432)
    p2 = T.SolverParam()
    p2.max_iter, p2.eps_acc = 1_000_000, 1e-4
    fs.resume(p2)
    r2 = fs.run(-1, poll_every=16)
    assert r2.state == 0
    x2, _ = fs.solution()
    fs.destroy()
    fs = T.FusedSolver.from_dense(dense, p2, "carried")
    r3 = fs.run(-1, poll_every=16)
    x3, _ = fs.solution()
    fs.destroy()
    assert abs(r3.iters - r2.iters) <= 2, (r2.iters, r3.iters)
    assert np.allclose(x2, x3, rtol=1e-4, atol=1e-5 * np.abs(x3).max())


def test_caller_built_bf16_matrix_needs_no_f32_copy(T):
    # thip_solver_set_a_bf16: the matrix arrives as bf16 column blocks; same bits as the library's own conversion, so the
    # two solves are identical -- and the block-wise generator of synth.LpInstance produces exactly those bits
    from totsu_amd import synth
    n = 48
    inst = synth.LpInstance(n, seed=5)
    direct = synth.LpInstance(n, seed=5, bf16_direct=True, block_cols=7)
    conv = T.Bf16Matrix.from_f32(inst.mat_a, inst.m, n)
    a = direct.mat_a._buf.to_host().view(np.uint16)[:conv.ld16 * n]
    b = conv._buf.to_host().view(np.uint16)[:conv.ld16 * n]
    assert np.array_equal(a, b)
    assert np.array_equal(b.reshape(n, conv.ld16)[:, :inst.m], bf16_bits(inst.mat_a.to_host()[:inst.m * n]).reshape(n, inst.m))
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 200_000
    res = []
    for mat, kw in ((inst.mat_a, dict(a_storage="bf16")), (direct.mat_a, {})):
        fs = T.FusedSolver(n, inst.m, mat, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried", **kw)
        assert fs.a_storage == "bf16" and fs.passes()[1] == 2 * n * inst.m
        r = fs.run(-1, poll_every=64)
        x, y = fs.solution()
        res.append((r.iters, x.copy(), y.copy()))
        if not kw:
            with pytest.raises(_lib_error(T)):
                fs.set_a_storage("f32")                    # there is no f32 matrix to go back to
        fs.destroy()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    conv.free(); direct.free(); inst.free()


def _lib_error(T):
    from totsu_amd._lib import ThipError
    return ThipError


# ---- f16 storage with one power-of-two scale per column: 8x finer rounding than bf16 at the same bytes ----------

def f16_quantize(a):
    """numpy restatement of thip_to_f16: (bit patterns as uint16 (m, n), inv_scale (n,), dequantised f64 matrix)"""
    a = np.asarray(a, np.float32)
    mx = np.abs(a).max(axis=0)
    e = np.frexp(mx)[1]                                    # mx = f * 2^e, f in [0.5, 1)
    inv = np.where(mx > 0, np.ldexp(np.float32(1.0), e - 14), np.float32(1.0)).astype(np.float32)
    q = (a / inv[None, :]).astype(np.float16)              # exact scaling by a power of two, then round to nearest even
    return q.view(np.uint16), inv, q.astype(np.float64) * inv[None, :].astype(np.float64)


@pytest.mark.parametrize("shape", [(1, 1), (7, 3), (8, 5), (13, 9), (256, 17), (2049, 33)])
def test_to_f16_is_bit_exact(T, shape):
    from totsu_amd._lib import lib
    m, n = shape
    rng = np.random.default_rng(m * 17 + n)
    a = (rng.standard_normal((m, n)) * 10.0 ** rng.integers(-20, 20, (1, n))).astype(np.float32)
    a[:, n // 2] = 0.0                                      # an all-zero column keeps scale 1
    if m > 4:
        a[1, 0] = a[:, 0].max() * 2.0 ** -20                # far below the column's top: lands in f16's subnormals
    ld = (m + 7) // 8 * 8
    src = T.DeviceBuffer.from_host(np.asfortranarray(a).ravel(order="F"))
    dst, inv = U16Buffer(T, ld * n), T.DeviceBuffer(n)
    lib.thip_to_f16(m, n, src.ptr, dst.buf.ptr, ld, inv.ptr)
    bits, want_inv, _ = f16_quantize(a)
    got = dst.to_host().reshape((n, ld)).T
    assert np.array_equal(inv.to_host(), want_inv)
    assert np.array_equal(got[:m], bits) and not got[m:].any()
    top = np.abs(bits.view(np.float16).astype(np.float32)).max(axis=0)
    assert np.all((top[want_inv != 1.0] >= 2.0 ** 13) & (top[want_inv != 1.0] <= 2.0 ** 14))


@pytest.mark.parametrize("shape", [(1, 1), (5, 7), (8, 8), (100, 30), (2049, 65), (5000, 1500), (64, 9000), (20_000, 700)])
def test_gemv_f16_equals_products_of_the_dequantised_matrix(T, shape):
    from totsu_amd._lib import lib
    m, n = shape
    rng = np.random.default_rng(m + 11 * n)
    a = (rng.standard_normal((m, n)) * 10.0 ** rng.integers(-6, 6, (1, n))).astype(np.float32)
    _, _, ar = f16_quantize(a)
    ld = (m + 7) // 8 * 8
    src = T.DeviceBuffer.from_host(np.asfortranarray(a).ravel(order="F"))
    a16, inv = U16Buffer(T, ld * n), T.DeviceBuffer(n)
    lib.thip_to_f16(m, n, src.ptr, a16.buf.ptr, ld, inv.ptr)
    x = rng.standard_normal(n).astype(np.float32)
    y0 = rng.standard_normal(m).astype(np.float32)
    dx, dy = T.DeviceBuffer.from_host(x), T.DeviceBuffer.from_host(y0)
    lib.thip_transform_ge_f16(0, m, n, 1.5, a16.buf.ptr, ld, inv.ptr, dx.ptr, -0.5, dy.ptr)
    want = 1.5 * ar @ x.astype(np.float64) - 0.5 * y0
    scale = np.abs(ar) @ np.abs(x.astype(np.float64)) + np.abs(y0)
    assert np.all(np.abs(dy.to_host() - want) <= 4e-6 * scale + 1e-30)
    v = rng.standard_normal(m).astype(np.float32)
    w0 = rng.standard_normal(n).astype(np.float32)
    dv, dw = T.DeviceBuffer.from_host(v), T.DeviceBuffer.from_host(w0)
    lib.thip_transform_ge_f16(1, m, n, -2.0, a16.buf.ptr, ld, inv.ptr, dv.ptr, 1.0, dw.ptr)
    want = -2.0 * ar.T @ v.astype(np.float64) + w0
    scale = 2.0 * np.abs(ar).T @ np.abs(v.astype(np.float64)) + np.abs(w0)
    assert np.all(np.abs(dw.to_host() - want) <= 4e-6 * scale + 1e-30)


@pytest.mark.parametrize("schedule", ["reference", "fused", "carried", "sweep"])
def test_iterates_with_f16_storage_follow_the_oracle_on_the_dequantised_matrix(T, schedule):
    import copy
    dense = (_socp(T, 100, [5, 1, 0, 17, 99, 3, 32], seed=2) if schedule == "sweep" else _socp(T, 30, [5, 1, 0, 17, 99, 3], seed=2)).dense()
    iters, tols = [0, 1, 9, 99], [2e-5, 2e-5, 1e-4, 2e-3]
    dr = copy.copy(dense)
    A = np.asarray(dense.mat_a, np.float32).reshape((dense.n, dense.m)).T
    dr.mat_a = np.asfortranarray(f16_quantize(A)[2]).ravel(order="F")
    ro = O.solve_matop_cones(O.param(max_iter=max(iters) + 2, eps_acc=1e-30), dr.vec_c, dr.mat_a, dr.vec_b, dr.seg_type,
                             dr.seg_len, snap_iters=iters, trace_cap=max(iters) + 3, use_ql=True)
    p = T.SolverParam()
    p.eps_acc = 1e-30
    fs = T.FusedSolver.from_dense(dense, p, schedule, a_storage="f16", sweep_min_bytes=0)
    assert fs.passes()[1] == dense.n * dense.m * 2 and fs.schedule_in_use() == schedule
    N = dense.n + 2 * dense.m + 1
    done = 0
    for q, (it, tol) in enumerate(zip(iters, tols)):
        fs.run(it + 1 - done, poll_every=64)
        done = it + 1
        x, y = fs.iterate()
        rx, ry = ro.snaps[q][:N], ro.snaps[q][N:]
        assert np.abs(x - rx).max() <= tol * max(np.abs(rx).max(), 1e-6), (schedule, it)
        assert np.abs(y - ry).max() <= tol * max(np.abs(ry).max(), 1e-6), (schedule, it)
    fs.destroy()


@pytest.mark.parametrize("n,cones,seed", [(30, [5, 1, 0, 17, 99, 3], 2), (80, [20] * 10, 5)])
def test_f16_storage_alone_is_inside_the_objective_gate(T, n, cones, seed):
    # "reduced-precision A storage ... with the 1e-4 objective-gap test as gate".  bf16 misses that
    # gate at these sizes (2.6e-4, test above); column-scaled f16 rounds 8x finer and passes it without an f32 phase
    dense = _socp(T, n, cones, seed).dense()
    ro = O.solve_matop_cones(O.param(max_iter=2_000_000, eps_acc=1e-7), dense.vec_c, dense.mat_a, dense.vec_b,
                             dense.seg_type, dense.seg_len)
    obj = float(np.dot(dense.vec_c, ro.x))
    p = T.SolverParam()
    p.max_iter, p.eps_acc = 2_000_000, 1e-4
    fs = T.FusedSolver.from_dense(dense, p, "carried", a_storage="f16")
    r = fs.run(-1, poll_every=64)
    x, _ = fs.solution()
    assert r.state == 0
    gap = abs(float(np.dot(dense.vec_c, x)) - obj) / abs(obj)
    assert gap < 1e-4, gap
    fs.set_a_storage("bf16")                 # the owned 16-bit copy is rebuilt in the other format
    fs.resume()
    r2 = fs.run(-1, poll_every=64)
    assert r2.state == 0 and fs.a_storage == "bf16"
    fs.destroy()


def test_caller_built_f16_matrix_equals_the_library_conversion(T):
    from totsu_amd import synth
    n = 48
    inst = synth.LpInstance(n, seed=5)
    direct = synth.LpInstance(n, seed=5, bf16_direct="f16", block_cols=7)
    conv = T.Bf16Matrix.from_f32(inst.mat_a, inst.m, n, "f16")
    assert np.array_equal(direct.mat_a._buf.to_host().view(np.uint16)[:conv.ld16 * n],
                          conv._buf.to_host().view(np.uint16)[:conv.ld16 * n])
    assert np.array_equal(direct.mat_a._inv.to_host(), conv._inv.to_host())
    p = T.SolverParam()
    p.eps_acc, p.max_iter = 1e-4, 200_000
    res = []
    for mat, kw in ((inst.mat_a, dict(a_storage="f16")), (direct.mat_a, {})):
        fs = T.FusedSolver(n, inst.m, mat, inst.vec_b, inst.vec_c, inst.seg_type, inst.seg_len, p, "carried", **kw)
        assert fs.a_storage == "f16" and fs.passes()[1] == 2 * n * inst.m
        r = fs.run(-1, poll_every=64)
        x, y = fs.solution()
        res.append((r.iters, x.copy(), y.copy()))
        fs.destroy()
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    conv.free(); direct.free(); inst.free()


# ---- the one-pass kernel on a 16-bit-stored A (thip_sweep16.hip) -------------------------------------------------------

@pytest.mark.parametrize("kind", ["bf16", "f16"])
@pytest.mark.parametrize("m,n,members,w", [(4096, 3000, 0, 2), (20000, 2500, 0, 2), (3584, 2400, 0, 1), (100_000, 800, 0, 2), (12504, 1000, 0, 4),
                                           (1000, 2500, 0, 1), (248, 120, 0, 1), (28_672, 5200, 4, 2), (28_672, 5200, 2, 1), (21_504, 10_400, 2, 2),
                                           (57_344, 5200, 8, 4)])
def test_sweep_kernel_on_a_16_bit_matrix_vs_numpy(T, kind, m, n, members, w):
    """sweep_k<..., ELEM = bf16 / f16> alone (thip_test_sweep) against numpy f64 on the ROUNDED matrix: one, two and four columns
    per panel, 1 .. 4 eight-row slots per thread, every group size, the u update on and off"""
    import ctypes as C
    from totsu_amd import _lib
    D = T.DeviceBuffer
    rng = np.random.default_rng(m + 7 * n)
    A = (rng.standard_normal((m, n)) * np.exp(rng.uniform(-3, 3, n))[None, :] / np.sqrt(n)).astype(np.float32)      # columns of different scales
    mat = T.Bf16Matrix.from_f32(np.asfortranarray(A).ravel(order="F"), m, n, kind)
    Ar = (bf16_round(A) if kind == "bf16" else f16_quantize(A)[2]).astype(np.float64)
    host = dict(v=rng.standard_normal(m), xy=rng.standard_normal(m), c=rng.standard_normal(n), su=rng.random(n) + 0.5,
                tx=rng.random(n) + 0.5, u=rng.standard_normal(n), xx=rng.standard_normal(n), gp=rng.standard_normal(n))
    host = {k: np.asarray(a, np.float32) for k, a in host.items()}
    for first in (0, 1):
        bufs = {k: D.from_host(a) for k, a in host.items()}
        outs = {k: D(sz, zero=True) for k, sz in dict(xx_out=n, hn=m + 8, h3=m + 8).items()}
        t = _lib.SweepTest()
        t.m, t.n, t.lda = (m + 7) // 8 * 8, n, mat.ld16          # the library-made copy has zero rows up to a multiple of 8
        t.mat_a, t.v, t.xy, t.c, t.su, t.tx = mat.ptr, bufs["v"].ptr, bufs["xy"].ptr, bufs["c"].ptr, bufs["su"].ptr, bufs["tx"].ptr
        t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = bufs["u"].ptr, None, bufs["xx"].ptr, None, outs["xx_out"].ptr, None
        t.gp, t.hn, t.h3 = bufs["gp"].ptr, outs["hn"].ptr, outs["h3"].ptr
        t.kappa, t.rtau, t.first, t.reps, t.force_members = -0.37, 0.81, first, 1, members
        t.elem, t.inv_s, t.variant = (1 if kind == "bf16" else 2), mat.inv_ptr, w      # (16-bit: `variant` = columns per panel)
        ms, info = (C.c_float * 2)(), (C.c_int * 8)()
        sums = (C.c_float * 4)()
        t.host_sums = sums
        if m % 8:
            # v / x_y behind row m must read as zero (the solver's arena guarantees it): longer buffers here
            for k in ("v", "xy"):
                bufs[k].free()
                bufs[k] = D.from_host(np.concatenate([host[k], np.zeros(8, np.float32)]))
            t.v, t.xy = bufs["v"].ptr, bufs["xy"].ptr
        _lib.lib.thip_test_sweep(C.byref(t), ms, info)
        assert info[0] == 0, list(info)
        g3 = Ar.T @ host["xy"].astype(np.float64)
        gT = Ar.T @ host["v"].astype(np.float64)
        u_ref = host["u"].astype(np.float64) if first else host["u"] + host["su"] * (-(host["gp"] - 2 * g3) - host["c"] * 0.81)
        x_ref = host["xx"] + host["tx"] * (gT + host["c"] * (-0.37))
        got = {"u": bufs["u"].to_host(), "x": outs["xx_out"].to_host(), "gp": bufs["gp"].to_host(),
               "hn": outs["hn"].to_host()[:m], "h3": outs["h3"].to_host()[:m]}
        ref = {"u": u_ref, "x": x_ref, "gp": g3, "hn": Ar @ u_ref, "h3": Ar @ x_ref}
        for k in ref:
            err = np.abs(got[k] - ref[k]).max() / (np.abs(ref[k]).max() + 1e-30)
            assert err < 1e-5, (kind, m, n, members, w, first, k, err, list(info))
        c64, xx64 = host["c"].astype(np.float64), host["xx"].astype(np.float64)
        want = [((c64 + g3) ** 2).sum(), c64 @ xx64, c64 @ u_ref, c64 @ (xx64 - 2 * x_ref)]
        scale = [want[0], np.abs(c64) @ np.abs(xx64), np.abs(c64) @ np.abs(u_ref), np.abs(c64) @ (np.abs(xx64) + 2 * np.abs(x_ref))]
        for q in range(4):
            assert abs(sums[q] - want[q]) <= 3e-5 * scale[q], (kind, m, n, members, w, first, q, sums[q], want[q])
        for b in list(bufs.values()) + list(outs.values()):
            b.free()
    mat.free()
