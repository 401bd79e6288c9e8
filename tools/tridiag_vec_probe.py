"""numpy restatement of the device tridiagonal eigen-solver planned for the closure path of map_eig:
eigenvalues by Sturm-count multisection in f64, one eigenvector per eigenvalue from the twisted factorisation of
T - lambda I (the dlar1v step of MRRR, without the representation tree), then a Newton-Schulz polish of the
back-transformed vectors in f32.  What it measures: orthogonality before / after the polish and the residual, on the
spectra the tests use (random, rank-deficient, clustered, Wilkinson, diagonal).  Run: python tools/tridiag_vec_probe.py"""
import numpy as np
import scipy.linalg as sla


def tridiag_f32(a):
    """Householder tridiagonalisation in f32 (the noise floor the f64 stage sees)."""
    h, q = sla.hessenberg(a.astype(np.float32), calc_q=True)
    d = np.diag(h).astype(np.float64)
    e = np.diag(h, -1).astype(np.float64)
    return d, e, q.astype(np.float32)


def split_blocks(d, e):
    n = d.size
    eps = np.finfo(np.float64).eps
    cut = [0]
    for i in range(n - 1):
        if abs(e[i]) <= eps * (abs(d[i]) + abs(d[i + 1])):
            cut.append(i + 1)
    cut.append(n)
    return [(cut[i], cut[i + 1]) for i in range(len(cut) - 1)]


def sturm_count(d, e2, sig, pivmin):
    """number of eigenvalues < sig, vectorised over sig"""
    q = d[0] - sig
    q = np.where(np.abs(q) < pivmin, -pivmin, q)
    c = (q < 0).astype(np.int64)
    for i in range(1, d.size):
        q = d[i] - sig - e2[i - 1] / q
        q = np.where(np.abs(q) < pivmin, -pivmin, q)
        c += q < 0
    return c


def bisect_block(d, e, rounds=11, fan=64):
    n = d.size
    if n == 1:
        return d.copy()
    e2 = e * e
    r = np.abs(np.concatenate([[0], e])) + np.abs(np.concatenate([e, [0]]))
    gl, gu = (d - r).min(), (d + r).max()
    tn = max(abs(gl), abs(gu))
    pivmin = np.finfo(np.float64).tiny * max(1.0, e2.max())
    gl -= 2.1 * tn * np.finfo(np.float64).eps * n + 2.1 * pivmin
    gu += 2.1 * tn * np.finfo(np.float64).eps * n + 2.1 * pivmin
    lo = np.full(n, gl)
    hi = np.full(n, gu)
    k = np.arange(n)
    for _ in range(rounds):
        t = (np.arange(fan) + 1.0) / (fan + 1.0)
        sig = lo[:, None] + (hi - lo)[:, None] * t[None, :]            # n x fan
        cnt = sturm_count(d[:, None, None], e2[:, None, None], sig, pivmin)   # eigenvalues < sig
        le = cnt <= k[:, None]                                           # sig is a lower bound for eigenvalue k
        idx = le.sum(axis=1)                                             # le is monotone: first idx are true
        sig_ext = np.concatenate([lo[:, None], sig, hi[:, None]], axis=1)
        lo, hi = sig_ext[k, idx], sig_ext[k, idx + 1]
    return 0.5 * (lo + hi)


def twisted_vectors(d, e, lam):
    """one vector per lam (vectorised over lam): z with (T - lam) z = gamma_r e_r"""
    n = d.size
    m = lam.size
    if n == 1:
        return np.ones((1, m))
    tn = max(np.abs(d).max(), np.abs(e).max())
    piv = np.finfo(np.float64).eps * tn * 1e-3 + np.finfo(np.float64).tiny
    dp = np.empty((n, m))
    lp = np.empty((n - 1, m))
    dp[0] = d[0] - lam
    for i in range(n - 1):
        x = np.where(np.abs(dp[i]) < piv, np.copysign(piv, dp[i]) + (dp[i] == 0) * piv, dp[i])
        lp[i] = e[i] / x
        dp[i + 1] = (d[i + 1] - lam) - lp[i] * e[i]
    dm = np.empty((n, m))
    um = np.empty((n - 1, m))
    dm[n - 1] = d[n - 1] - lam
    for i in range(n - 2, -1, -1):
        x = np.where(np.abs(dm[i + 1]) < piv, np.copysign(piv, dm[i + 1]) + (dm[i + 1] == 0) * piv, dm[i + 1])
        um[i] = e[i] / x
        dm[i] = (d[i] - lam) - um[i] * e[i]
    gam = dp + dm - (d[:, None] - lam[None, :])
    r = np.abs(gam).argmin(axis=0)
    z = np.zeros((n, m))
    cols = np.arange(m)
    z[r, cols] = 1.0
    # upwards from r, downwards from r (masked, all columns in lockstep like the lanes of the kernel)
    for i in range(n - 2, -1, -1):
        act = i < r
        z[i] = np.where(act, -lp[i] * z[i + 1], z[i])
    for i in range(n - 1):
        act = i >= r
        z[i + 1] = np.where(act, -um[i] * z[i], z[i + 1])
    z /= np.linalg.norm(z, axis=0)
    return z


def solve(a, polish=2):
    n = a.shape[0]
    d, e, q = tridiag_f32(a)
    lam = np.empty(n)
    v = np.zeros((n, n))
    for (b0, b1) in split_blocks(d, e):
        l = bisect_block(d[b0:b1], e[b0:b1 - 1])
        lam[b0:b1] = l
        v[b0:b1, b0:b1] = twisted_vectors(d[b0:b1], e[b0:b1 - 1], l)
    z = (q @ v.astype(np.float32)).astype(np.float32)
    stats = {"orth_T": np.abs(v.T @ v - np.eye(n)).max()}
    g = z.T @ z
    stats["orth0"] = np.abs(g - np.eye(n, dtype=np.float32)).max()
    stats["fro0"] = np.linalg.norm(g - np.eye(n))
    for _ in range(polish):
        z = (z @ (1.5 * np.eye(n, dtype=np.float32) - 0.5 * g)).astype(np.float32)
        g = z.T @ z
    stats["orth"] = np.abs(g - np.eye(n)).max()
    a32 = a.astype(np.float32).astype(np.float64)
    z64 = z.astype(np.float64)
    nrm = np.abs(np.linalg.eigvalsh(a32)).max()
    stats["resid"] = np.abs(a32 @ z64 - z64 * lam).max() / nrm
    stats["recon"] = np.abs((z64 * lam) @ z64.T - a32).max() / nrm
    stats["eig"] = np.abs(np.sort(lam) - np.linalg.eigvalsh(a32)).max() / nrm
    return stats


def cases(n, rng):
    b = rng.standard_normal((n, n))
    yield "random", (b + b.T) / 2
    q, _ = np.linalg.qr(b)
    w = np.zeros(n)
    w[: n // 4] = rng.uniform(0.5, 2.0, n // 4)
    w[n // 4: n // 2] = -rng.uniform(0.5, 2.0, n // 2 - n // 4)
    yield "rank-deficient", (q * w) @ q.T
    w = np.where(np.arange(n) < n // 2, 1.0, 2.0)
    yield "two clusters", (q * w) @ q.T
    yield "wishart+0.05I", b @ b.T / n + 0.05 * np.eye(n)
    yield "identity", np.eye(n)
    yield "diagonal repeated", np.diag(np.repeat([1.0, 2.0, 3.0, -1.0], (n + 3) // 4)[:n])
    m = (n - 1) // 2
    wl = np.diag(np.abs(np.arange(n) - m).astype(float)) + np.diag(np.ones(n - 1), 1) + np.diag(np.ones(n - 1), -1)
    yield "wilkinson (already tridiagonal)", wl
    yield "q wilkinson q^T", q @ wl @ q.T
    yield "zero", np.zeros((n, n))
    yield "rank one", np.outer(b[0], b[0])
    yield "random * 1e-18", (b + b.T) / 2 * 1e-18
    yield "log-uniform spectrum 1e-8..1", (q * np.exp(rng.uniform(np.log(1e-8), 0, n))) @ q.T


if __name__ == "__main__":
    import sys
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(1)
    for name, a in cases(n, rng):
        s = solve(a)
        print("%-34s" % name, " ".join("%s=%.1e" % kv for kv in s.items()))
