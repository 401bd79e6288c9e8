"""Seeded problem constructions shared by tests and bench."""
import numpy as np


def benchmark_lp(sz, seed=0):
    """experimental/benchmark_lp/src/main.rs:14-57: n = sz, m = 2 sz, c = -U(0,1), G = [-I; U(0,1)], h = [0; U(0,1)]"""
    rng = np.random.default_rng(seed)
    n, m = sz, 2 * sz
    c = -rng.uniform(0, 1, n)
    G = np.vstack([-np.eye(n), rng.uniform(0, 1, (n, n))])
    h = np.concatenate([np.zeros(n), rng.uniform(0, 1, n)])
    return c.astype(np.float32), G.astype(np.float32), h.astype(np.float32)


def random_socp(n, cones, seed=0):
    """strictly feasible, bounded SOCP (SURVEY.md 8d, C3 construction)"""
    rng = np.random.default_rng(seed)
    mats_g, vecs_h, vecs_c, d = [], [], [], []
    x0 = rng.standard_normal(n)
    for ni in cones:
        G = (rng.standard_normal((ni, n)) / np.sqrt(n)).astype(np.float32)
        h = rng.standard_normal(ni).astype(np.float32)
        c = (rng.standard_normal(n) / np.sqrt(n)).astype(np.float32)
        mats_g.append(G)
        vecs_h.append(h)
        vecs_c.append(c)
        d.append(np.float32(np.linalg.norm(G @ x0 + h) - c @ x0 + rng.uniform(0.1, 1.1)))
    f = np.zeros(n)
    for G, c in zip(mats_g, vecs_c):
        t = rng.uniform(0.5, 1.5)
        w = rng.standard_normal(G.shape[0])
        w *= 0.9 * t * rng.uniform(0, 1) / max(np.linalg.norm(w), 1e-9)
        f += t * c + G.T @ w
    return f.astype(np.float32), mats_g, vecs_h, vecs_c, d


def random_sdp(n, k, seed=0):
    """min c^T x s.t. sum_i x_i F_i + F_n <= 0 (totsu ProbSDP form, sdp.rs:222-248; cf. test_sdp1), strictly
    feasible and bounded"""
    rng = np.random.default_rng(seed)

    def sym():
        b = rng.standard_normal((k, k))
        return (b + b.T) / 2
    Fs = [sym() for _ in range(n)]
    x0 = rng.standard_normal(n)
    # F_n makes x0 strictly feasible: sum x0_i F_i + F_n = -I
    Fn = -np.eye(k) - sum(x * F for x, F in zip(x0, Fs))
    # c from a random Y > 0 : c_i = -tr(F_i Y)  (dual feasible => bounded)
    b = rng.standard_normal((k, k))
    Y = b @ b.T / k + 0.1 * np.eye(k)
    c = np.array([-np.trace(F @ Y) for F in Fs])

    def pack(S):
        return np.array([S[r, cc] for cc in range(k) for r in range(cc + 1)], dtype=np.float32)
    return c.astype(np.float32), [pack(F) for F in Fs] + [pack(Fn)]


def l1reg_lp(l=20, seed=0, lam=0.2):
    """BASELINE.json configs[0]: the construction of examples/l1reg_lp/src/main.rs:50-116 (L1-regularised L1-error
    kernel regression as an LP; gaussian kernel sigma^2 = 1/8, lambda = 0.2; n = 3l + 1, m = 4l, p = 0).  The
    reference draws the sample points from Xoshiro256StarStar::seed_from_u64(0), whose stream is not reproducible
    here (crate not vendored): the points come from numpy's generator instead -- same construction, other data."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (2, l))
    y = np.cos(5.0 * x[0]) * np.cos(7.0 * x[1])
    n, m = 3 * l + 1, 4 * l
    c = np.zeros(n)
    c[:l] = 1.0
    c[2 * l:3 * l] = lam
    G = np.zeros((m, n))
    for i in range(l):
        G[i, i] = -1.0
        G[l + i, i] = -1.0
        G[2 * l + i, l + i] = 1.0
        G[3 * l + i, l + i] = -1.0
        G[2 * l + i, 2 * l + i] = -1.0
        G[3 * l + i, 2 * l + i] = -1.0
        G[i, 3 * l] = 1.0
        G[l + i, 3 * l] = -1.0
    d2 = ((x[:, :, None] - x[:, None, :]) ** 2).sum(axis=0)
    K = np.exp(-d2 / (1.0 / 8.0))
    G[:l, l:2 * l] = K
    G[l:2 * l, l:2 * l] = -K
    h = np.zeros(m)
    h[:l] = y
    h[l:2 * l] = -y
    return c, G, h


def partitioning_sdp(x_num, y_num, seed=0):
    """BASELINE.json configs[3] template: the construction of examples/partitioning_sdp/src/main.rs:21-79 (max-cut style
    SDP relaxation on an x_num x y_num grid graph): minimise sum W_ij X_ij over packed X (n = l(l+1)/2 variables),
    X >= 0 (F_k = -E_ij, F_n = 0), diag(X) = 1 (A picks the diagonal entries, b = 1).  Edge weights ~ N(0,1) from
    numpy's generator (the reference's Xoshiro stream is not reproducible here)."""
    rng = np.random.default_rng(seed)
    l = x_num * y_num
    n = l * (l + 1) // 2

    def pidx(r, c):
        return c * (c + 1) // 2 + r
    w = np.zeros(n)
    for i in range(l):
        x, y = divmod(i, y_num)
        if x < x_num - 1:
            w[pidx(i, i + y_num)] = rng.standard_normal()
        if y < y_num - 1:
            w[pidx(i, i + 1)] = rng.standard_normal()
    syms_f = [np.zeros(n) for _ in range(n + 1)]
    kk = 0
    for j in range(l):
        for i in range(j + 1):
            syms_f[kk][pidx(i, j)] = -1.0
            kk += 1
    mat_a = np.zeros((l, n))
    j = 0
    for i in range(l):
        mat_a[i, j] = 1.0
        j += i + 2
    return w, syms_f, mat_a, np.ones(l)


def sym_pack(S):
    """full symmetric matrix -> packed upper triangle by columns (MatType::SymPack storage, matbuild/mod.rs:254-279)"""
    S = np.asarray(S)
    return np.concatenate([S[:c + 1, c] for c in range(S.shape[0])])


def svm_qp(l=50, seed=0):
    """The QP of examples/svm_qp/src/main.rs:47-106 (dual of a hard-margin SVM with a Gaussian kernel, sigma^2 = 1/8, on
    l random points of the unit square labelled +1 inside a ring 0.25 < r < 0.4 around the centre): minimise
    0.5 a^T P a - 1^T a  s.t. a >= 0, y^T a = 0, P_ij = y_i y_j k(x_i, x_j).  Sample points from numpy's generator (the
    reference's Xoshiro stream is not reproducible here)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, (2, l))
    r = np.hypot(x[0] - 0.5, x[1] - 0.5)
    y = np.where((r > 0.25) & (r < 0.4), 1.0, -1.0)
    d2 = ((x[:, :, None] - x[:, None, :]) ** 2).sum(axis=0)
    P = np.outer(y, y) * np.exp(-d2 * 8.0)
    return dict(sym_p=P, vec_q=-np.ones(l), mat_g=-np.eye(l), vec_h=np.zeros(l), mat_a=y.reshape(1, l), vec_b=np.zeros(1))


def trajplan_qcqp(t_cap=30, a_cap=90.0):
    """The QCQP of examples/trajplan_qcqp/src/main.rs:19-151: a 2-D trajectory on t_cap time grids minimising the total
    squared velocity, every acceleration bounded by a_cap, start / end positions with zero velocity and two way
    points.  Unknowns: x-coordinates then y-coordinates (n = 2 t_cap); m = t_cap - 2 quadratic constraints; p = 12."""
    n, m, dt = 2 * t_cap, t_cap - 2, 1.0 / t_cap
    D = np.zeros((n, n))
    for i in range(t_cap - 1):
        for off in (0, t_cap):
            D[off + i, off + i], D[off + i, off + i + 1] = -1.0 / dt, 1.0 / dt
    syms_p, scls_r = [D.T @ D], [0.0]
    for i in range(m):
        D2 = np.zeros((n, n))
        for off in (0, t_cap):
            D2[off + i, off + i:off + i + 3] = np.array([1.0, -2.0, 1.0]) / (dt * dt)
        syms_p.append(D2.T @ D2)
        scls_r.append(-0.5 * a_cap * a_cap)
    A, b = np.zeros((12, n)), np.zeros(12)
    x_s, x_m1, x_m2, x_t = (0.0, 0.0), (0.5, -1.5), (0.25, 1.5), (1.0, 1.0)
    A[0, 0], b[0] = 1.0, x_s[0]
    A[1, t_cap], b[1] = 1.0, x_s[0]                       # the reference uses x_s.0 for both coordinates (main.rs:101-106)
    A[2, 0], A[2, 1] = -1.0, 1.0
    A[3, t_cap], A[3, t_cap + 1] = -1.0, 1.0
    A[4, t_cap - 1], b[4] = 1.0, x_t[0]
    A[5, 2 * t_cap - 1], b[5] = 1.0, x_t[1]
    A[6, t_cap - 2], A[6, t_cap - 1] = -1.0, 1.0
    A[7, 2 * t_cap - 2], A[7, 2 * t_cap - 1] = -1.0, 1.0
    t1, t2 = t_cap // 3, t_cap * 2 // 3
    A[8, t1], b[8] = 1.0, x_m1[0]
    A[9, t_cap + t1], b[9] = 1.0, x_m1[1]
    A[10, t2], b[10] = 1.0, x_m2[0]
    A[11, t_cap + t2], b[11] = 1.0, x_m2[1]
    return dict(syms_p=syms_p, vecs_q=[np.zeros(n) for _ in range(m + 1)], scls_r=scls_r, mat_a=A, vec_b=b)


def toruscompl_socp(x_num=9, y_num=7, vol_ratio=0.2):
    """The SOCP of examples/toruscompl_socp/src/main.rs:43-268: compliance minimisation of a planar truss on an
    x_num x y_num grid of nodes (left column clamped, unit downward load on the middle node of the right column).
    Members: right and up neighbours of every node, plus the four diagonals and (1,-1)/(-1,-1) from nodes with odd x
    and even y.  Unknowns [x (cross sections), q (member forces), w (energies)], n = 3 l.  3 l + 1 cones: l of
    1 + 2 rows (w_i x_i >= v_i q_i^2 / 2 as a second-order cone), then 2 l + 1 cones WITHOUT rows (0 <= x_i <= 1 and the
    volume bound, i.e. one-dimensional cones c^T x + d >= 0), and one equality row per free degree of freedom."""
    nodes = [(x, y) for x in range(x_num) for y in range(y_num)]
    idx = {xy: k for k, xy in enumerate(nodes)}
    members = []
    for (hx, hy) in nodes:
        steps = [(1, 0), (0, 1), (1, 1), (-1, 1), (1, -1), (-1, -1)] if (hx % 2 == 1 and hy % 2 == 0) else [(1, 0), (0, 1)]
        for dx, dy in steps:
            if (hx + dx, hy + dy) in idx:
                members.append((idx[(hx, hy)], idx[(hx + dx, hy + dy)]))
    load = {k: (0.0, 0.0) for k in range(len(nodes))}
    for y in range(y_num):
        load[idx[(0, y)]] = None                                   # clamped: no degrees of freedom
    load[idx[(x_num - 1, y_num // 2)]] = (0.0, -1.0)
    dof_of, dof = {}, 0
    for k in range(len(nodes)):
        dof_of[k] = dof
        if load[k] is not None:
            dof += 2
    l = len(members)
    n = 3 * l
    length = np.array([np.hypot(nodes[h][0] - nodes[t][0], nodes[h][1] - nodes[t][1]) for h, t in members])
    f = np.concatenate([np.zeros(2 * l), np.ones(l)])
    mats_g, vecs_h, vecs_c, d = [], [], [], []
    for i in range(l):
        G = np.zeros((2, n))
        G[0, i], G[0, 2 * l + i], G[1, l + i] = -1.0, 1.0, np.sqrt(2.0 * length[i])
        c = np.zeros(n)
        c[i], c[2 * l + i] = 1.0, 1.0
        mats_g.append(G); vecs_h.append(np.zeros(2)); vecs_c.append(c); d.append(0.0)
    for sign, off in ((1.0, 0.0), (-1.0, 1.0)):
        for i in range(l):
            c = np.zeros(n)
            c[i] = sign
            mats_g.append(np.zeros((0, n))); vecs_h.append(np.zeros(0)); vecs_c.append(c); d.append(off)
    c = np.zeros(n)
    c[:l] = -length
    mats_g.append(np.zeros((0, n))); vecs_h.append(np.zeros(0)); vecs_c.append(c); d.append(vol_ratio * length.sum())
    A, b = np.zeros((dof, n)), np.zeros(dof)
    for i, (h, t) in enumerate(members):
        beta = np.array([nodes[t][0] - nodes[h][0], nodes[t][1] - nodes[h][1]], float)
        beta /= np.hypot(*beta)
        if load[h] is not None:
            A[dof_of[h], l + i] -= beta[0]; A[dof_of[h] + 1, l + i] -= beta[1]
        if load[t] is not None:
            A[dof_of[t], l + i] += beta[0]; A[dof_of[t] + 1, l + i] += beta[1]
    for k in range(len(nodes)):
        if load[k] is not None:
            b[dof_of[k]], b[dof_of[k] + 1] = load[k]
    return dict(vec_f=f, mats_g=mats_g, vecs_h=vecs_h, vecs_c=vecs_c, scls_d=d, mat_a=A, vec_b=b, members=members,
                length=length)
