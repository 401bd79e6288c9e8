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
