"""The five cones of totsu_core (cone_zero.rs, cone_rpos.rs, cone_soc.rs, cone_rotsoc.rs, cone_psd.rs) over a
LinAlg backend `L`.  Backends that expose device projections (F32HIP) get them used instead of the host
loops the reference runs through `get_mut()` (option (b)); any other backend (e.g. the numpy
backend of the CPU tests) takes the reference's own sequence of L calls."""
import math

from ._lib import lib


# False: every cone runs the reference's LITERAL sequence of SliceLike / LinAlg calls on any backend (ConeRPos: a host
# loop over get_mut(); ConeSOC: get + norm + scale + set; ConePSD: map_eig with a host closure) -- what an unchanged
# totsu_core executes over the F32HIP backend.  True (default): the device projections (thip_proj_*) on F32HIP.
DEVICE_FAST_PATHS = True


def _is_hip(L):
    return DEVICE_FAST_PATHS and getattr(L, "name", "") == "F32HIP"


class ConeZero:
    def __init__(self, L):
        self.L = L

    def proj(self, dual_cone, x):                                 # cone_zero.rs:38-44
        if not dual_cone:
            self.L.scale(0.0, x)
        return True

    def product_group(self, dp_tau, group):                       # cone_zero.rs:46-49
        pass


class ConeRPos:
    def __init__(self, L):
        self.L = L

    def proj(self, dual_cone, x):                                 # cone_rpos.rs:38-45
        if _is_hip(self.L):
            lib.thip_proj_rpos(x.len(), x.dev())
        else:
            xm = x.get_mut()
            for i in range(len(xm)):
                xm[i] = max(xm[i], 0.0)
        return True

    def product_group(self, dp_tau, group):
        pass


class ConeSOC:
    def __init__(self, L):
        self.L = L

    def proj(self, dual_cone, x):                                 # cone_soc.rs:38-65
        L = self.L
        if x.len() > 0:
            if _is_hip(L):
                lib.thip_proj_soc(x.len(), x.dev())
                return True
            s, v = x.split(1)
            val_s = s.get(0)
            norm_v = L.norm(v)
            if norm_v <= -val_s:
                L.scale(0.0, v)
                s.set(0, 0.0)
            elif norm_v <= val_s:
                pass
            else:
                alpha = (1.0 + val_s / norm_v) / 2.0
                L.scale(alpha, v)
                s.set(0, (norm_v + val_s) / 2.0)
        return True

    def product_group(self, dp_tau, group):                       # cone_soc.rs:67-70
        group(dp_tau)


class ConeRotSOC:
    def __init__(self, L):
        self.L = L
        self.soc = ConeSOC(L)

    def proj(self, dual_cone, x):                                 # cone_rotsoc.rs:38-65
        if x.len() > 0:
            if _is_hip(self.L):
                lib.thip_proj_rotsoc(x.len(), x.dev())
                return True
            if x.len() == 1:
                x.set(0, max(x.get(0), 0.0))
            else:
                fsqrt2 = math.sqrt(2.0)
                r, s = x.get(0), x.get(1)
                x.set(0, (r + s) / fsqrt2)
                x.set(1, (r - s) / fsqrt2)
                self.soc.proj(dual_cone, x)
                r, s = x.get(0), x.get(1)
                x.set(0, (r + s) / fsqrt2)
                x.set(1, (r - s) / fsqrt2)
        return True

    def product_group(self, dp_tau, group):
        group(dp_tau)


class ConePSD:
    """cone_psd.rs:22-85.  `work` is a host array the caller owns (like `&'a mut [F]`)."""

    def __init__(self, L, work, eps_zero):
        self.L = L
        self.work = L.Sl.new_mut(work)
        self.eps_zero = eps_zero

    @staticmethod
    def query_worklen(L, nvars):                                  # cone_psd.rs:32-38
        n = (int(math.sqrt(8 * nvars + 1)) - 1) // 2
        assert n * (n + 1) // 2 == nvars
        return L.map_eig_worklen(n)

    def drop(self):
        self.work.drop()

    def proj(self, dual_cone, x):                                 # cone_psd.rs:56-79
        L = self.L
        if self.work.len() < ConePSD.query_worklen(L, x.len()):
            return False
        fsqrt2 = math.sqrt(2.0)
        if _is_hip(L):
            L.map_eig(x, fsqrt2, self.eps_zero, self.work, "pos")
        else:
            L.map_eig(x, fsqrt2, self.eps_zero, self.work, lambda e: e if e > 0.0 else None)
        return True

    def product_group(self, dp_tau, group):                       # cone_psd.rs:81-84
        group(dp_tau)
