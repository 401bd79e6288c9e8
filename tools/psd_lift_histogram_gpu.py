"""How many LIFTING steps of the degree-7 polar chain (thip_eig.hip polar_project7: 8 by default, gain 5.644 per step, band
[0.15, 1.85], S_0 = 1.85 M / ||M||_F) does each PSD projection of a real solve need?  The trait-level Solver(F32HIP) runs the
bench's SDP construction (one PSD cone of order k, n variables) on the GPU; every `stride`-th call of ConePSD::proj copies its
INPUT to the host, where numpy takes the smallest relative eigenvalue |lambda| / ||M||_F above 2e-8 (below that a wrong sign costs
less than f32 round-off) and the number L of lifting steps that brings it into the band: 0.15 / (1.85 * 5.644^L) <= |lambda| / ||M||_F.
    python tools/psd_lift_histogram_gpu.py 500 500 4000 10      ->  profiles/r06_psd_lifting_steps_needed_k500.txt
The question (VERDICT r5, next #3a): would a data-dependent step count pay?  A launch of the chain serves the x_y and the x_s
projection of an iteration together, so what can be skipped is the MINIMUM over the pair of (8 - L)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    k, n, iters, stride = (int(v) for v in sys.argv[1:5])
    import totsu_amd as T
    from totsu_amd import _lib
    from problems import random_sdp
    _lib.init()
    L = T.F32HIP
    c, syms = random_sdp(n, k, seed=0)
    mb = lambda typ: T.MatBuild(L, typ)
    sdp = T.ProbSDP(mb(T.MatType.General(n, 1)).set_array(c.reshape(-1, 1)), [mb(T.MatType.SymPack(k)).set_array(s) for s in syms],
                    mb(T.MatType.General(0, n)), mb(T.MatType.General(0, 1)), 1e-12)
    prob = sdp.problem()
    cone = sdp._cone
    inner = cone.proj
    iu_r = np.array([r for cc in range(k) for r in range(cc + 1)])
    iu_c = np.array([cc for cc in range(k) for r in range(cc + 1)])
    rec = []           # (call index, L needed, smallest relative eigenvalue)
    calls = [0]

    def proj(dual_cone, x):
        i = calls[0]
        calls[0] += 1
        if (i // 2) % stride == 0:
            v = x.get_ref().astype(np.float64)
            M = np.zeros((k, k))
            w = np.where(iu_r == iu_c, 1.0, 1.0 / math.sqrt(2.0))
            M[iu_r, iu_c] = v * w
            M[iu_c, iu_r] = v * w
            fro = np.linalg.norm(M)
            if fro > 0:
                rel = np.abs(np.linalg.eigvalsh(M)) / fro
                rel = rel[rel > 2e-8]
                mn = rel.min() if rel.size else 1.0
                need = max(0, math.ceil(math.log(0.15 / (1.85 * mn)) / math.log(5.644)))
                rec.append((i, need, mn))
        return inner(dual_cone, x)
    cone.proj = proj
    s = T.Solver(L)
    s.fused = None                 # the trait-level loop: one ConePSD::proj call per block and iteration
    s.param.eps_acc, s.param.max_iter = 1e-3, iters
    try:
        s.solve(prob)
        print("converged at iteration", s.iters)
    except Exception as e:
        print("stopped:", repr(e)[:80], "iterations", getattr(s, "iters", None))
    rec = np.array(rec)
    need = rec[:, 1].astype(int)
    print("k = %d, n = %d: %d projections sampled (every %d-th iteration, both blocks)" % (k, n, len(rec), stride))
    print("histogram of lifting steps needed, L = 0 .. 10:", np.bincount(np.minimum(need, 10), minlength=11).tolist())
    pair = np.maximum(need[0::2][:len(need) // 2], need[1::2][:len(need) // 2])
    print("per ITERATION (max over the x_y / x_s pair: what a launch serves):", np.bincount(np.minimum(pair, 10), minlength=11).tolist())
    third = max(len(pair) // 3, 1)
    for tag, sl in (("first third", slice(0, third)), ("middle third", slice(third, 2 * third)), ("last third", slice(2 * third, None))):
        p = pair[sl]
        print("  %-13s mean %.2f  min %d  max %d   steps of 8 that could be skipped on average: %.2f" % (tag, p.mean(), p.min(), p.max(), np.mean(np.maximum(8 - p, 0))))
    print("quantiles of log10(smallest relative eigenvalue): 5%% %.2f  50%% %.2f  95%% %.2f" % tuple(np.quantile(np.log10(rec[:, 2]), [0.05, 0.5, 0.95])))


if __name__ == "__main__":
    main()
