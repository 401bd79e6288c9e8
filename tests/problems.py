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
