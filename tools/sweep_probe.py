"""One-pass kernel alone (thip_test_sweep): correctness against numpy f64 at a few shapes, then the time of a sweep
at a given shape (default BASELINE configs[2]: m = 100 000, n = 50 000).

    python tools/sweep_probe.py [--check] [--m M --n N --reps R]

The geometry is sweep_plan()'s default (one column per panel, the fewest workgroups per column the rows allow);
THIP_SWEEP_CLASS=0|1 picks the other forms (totsu_amd/csrc/thip_sweep.hip), THIP_SWEEP_DBG the
experiment switches of a -DSW_DEBUG build, and a -DSW_PROFILE build prints the service wave's phase stamps."""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run_case(lib, torch, m, n, first, comp, seed=0, reps=1, check=True, lda=None):
    from totsu_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(seed)
    lda = lda or m
    A = torch.randn((n, lda), generator=g, device="cuda", dtype=torch.float32) / np.sqrt(n)   # column-major: row j = column j

    def rv(k, scale=1.0):
        return torch.randn(k, generator=g, device="cuda", dtype=torch.float32) * scale
    v, xy = rv(m), rv(m)
    c, su, tx = rv(n), torch.rand(n, generator=g, device="cuda") + 0.5, torch.rand(n, generator=g, device="cuda") + 0.5
    u, xx, gp = rv(n), rv(n), rv(n)
    ku = rv(n, 1e-7) if comp else None
    kx = rv(n, 1e-7) if comp else None
    u0, ku0, gp0 = u.clone(), (ku.clone() if comp else None), gp.clone()
    xx_out, kx_out = torch.zeros(n, device="cuda"), (torch.zeros(n, device="cuda") if comp else None)
    hn, h3 = torch.zeros(m, device="cuda"), torch.zeros(m, device="cuda")
    kappa, rtau = -0.37, 0.81
    t = _lib.SweepTest()
    t.m, t.n, t.lda = m, n, lda
    P = lambda x: x.data_ptr() if x is not None else None
    t.mat_a, t.v, t.xy, t.c, t.su, t.tx = P(A), P(v), P(xy), P(c), P(su), P(tx)
    t.u, t.ku, t.xx_in, t.kx_in, t.xx_out, t.kx_out = P(u), P(ku), P(xx), P(kx), P(xx_out), P(kx_out)
    t.gp, t.hn, t.h3 = P(gp), P(hn), P(h3)
    t.kappa, t.rtau, t.first, t.reps = kappa, rtau, int(first), reps
    ms = (C.c_float * 2)()
    info = (C.c_int * 8)()
    torch.cuda.synchronize()
    lib.thip_test_sweep(C.byref(t), ms, info)
    torch.cuda.synchronize()
    res = {"ms_best": ms[0], "ms_avg": ms[1], "err": info[0], "G": info[1], "groups": info[2], "panels": info[3]}
    if info[0] != 0:
        return res
    if check:
        Ad = A[:, :m].double()
        gT = Ad @ v.double()
        g3 = Ad @ xy.double()
        if first:
            u_new = u0.double()
        else:
            inc = su.double() * (-(gp0.double() - 2 * g3) - c.double() * rtau)
            u_new = u0.double() + (inc - (ku0.double() if comp else 0.0))
        x_new = xx.double() + (tx.double() * (gT + c.double() * kappa) - (kx.double() if comp else 0.0))
        hN_ref = Ad.t() @ u_new
        h3_ref = Ad.t() @ x_new

        def rel(a, b):
            return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
        res["e_u"] = rel(u, u_new)
        res["e_x"] = rel(xx_out, x_new)
        res["e_gp"] = rel(gp, g3)
        res["e_hn"] = rel(hn, hN_ref)
        res["e_h3"] = rel(h3, h3_ref)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--m", type=int, default=100_000)
    ap.add_argument("--n", type=int, default=50_000)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    import torch
    from totsu_amd import _lib
    _lib.init(0)
    lib = _lib.lib
    if a.check:
        bad = 0
        for (m, n) in [(4096, 30000), (20000, 10000), (3584, 24000), (100000, 3000), (12500, 8000), (1000, 25000)]:
            for first in (0, 1):
                for comp in (0, 1):
                    r = run_case(lib, torch, m, n, first, comp, seed=m + n + first)
                    errs = [r.get(k, 1.0) for k in ("e_u", "e_x", "e_gp", "e_hn", "e_h3")]
                    ok = r["err"] == 0 and max(errs) < 2e-5
                    bad += not ok
                    print("m %6d n %6d first %d comp %d: %s  %s" % (m, n, first, comp, "ok " if ok else "BAD", r), flush=True)
        print("check:", "PASS" if bad == 0 else "FAIL (%d)" % bad)
    r = run_case(lib, torch, a.m, a.n, 1, 1, reps=a.reps, check=False)
    gb = 4.0 * a.m * a.n / 1e9
    print("time m %d n %d class/variant %s: best %.4f ms (%.0f GB/s), avg %.4f ms (%.0f GB/s) %s"
          % (a.m, a.n, os.environ.get("THIP_SWEEP_CLASS", "-") + "/" + os.environ.get("THIP_SWEEP_VARIANT", "0"), r["ms_best"], gb / r["ms_best"] * 1e3, r["ms_avg"],
             gb / r["ms_avg"] * 1e3, r))


if __name__ == "__main__":
    main()
