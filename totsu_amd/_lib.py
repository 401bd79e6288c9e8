"""ctypes binding of libtotsu_f32hip.so (include/totsu_f32hip.h).

There is NO CPU fallback: importing this module without the built library raises, and every entry point
raises if the library reports an error (the reference backends assert on library status, f32cuda.rs:38).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libtotsu_f32hip.so")

E_INVALID, E_NOTINIT, E_NOGPU, E_WORK, E_NOCONV, E_TIMEOUT = 10001, 10002, 10003, 10004, 10005, 10006

ST_RUNNING, ST_OK, ST_UNBOUNDED, ST_INFEASIBLE, ST_EXCESS_ITER, ST_INVALID_OP, ST_WORK_SHORTAGE, ST_CONE_FAILURE = \
    -1, 0, 1, 2, 3, 4, 5, 6
SCHED_REFERENCE, SCHED_FUSED, SCHED_CARRIED, SCHED_SWEEP = 0, 1, 2, 3
STATE_COMPENSATED, STATE_PLAIN = 0, 1
CONE_ZERO, CONE_RPOS, CONE_SOC, CONE_ROTSOC, CONE_PSD = 0, 1, 2, 3, 4

fp = C.POINTER(C.c_float)


class ThipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("totsu_f32hip error %d: %s" % (code, msg))
        self.code = code


class Param(C.Structure):
    _fields_ = [("max_iter", C.c_int64), ("eps_acc", C.c_float), ("eps_inf", C.c_float),
                ("eps_zero", C.c_float), ("log_period", C.c_int64), ("state_arith", C.c_int32), ("reserved", C.c_int32)]


class Problem(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("mat_a", C.c_void_p), ("vec_b", C.c_void_p),
                ("vec_c", C.c_void_p), ("vec_b_rowabs", C.c_void_p), ("n_seg", C.c_size_t),
                ("host_seg_type", C.POINTER(C.c_int32)), ("host_seg_len", C.POINTER(C.c_int64))]


class Status(C.Structure):
    _fields_ = [("state", C.c_int32), ("iter", C.c_int64), ("kind", C.c_int32), ("cri", C.c_float * 3),
                ("tau", C.c_float), ("kappa", C.c_float), ("norm_b", C.c_float), ("norm_c", C.c_float)]


class SweepTest(C.Structure):
    _fields_ = [("m", C.c_size_t), ("n", C.c_size_t), ("lda", C.c_size_t)] + \
               [(k, C.c_void_p) for k in ("mat_a", "v", "xy", "c", "su", "tx", "u", "ku", "xx_in", "kx_in", "xx_out",
                                          "kx_out", "gp", "hn", "h3")] + \
               [("kappa", C.c_float), ("rtau", C.c_float), ("first", C.c_int32), ("reps", C.c_int32),
                ("force_members", C.c_int32), ("pub_agent", C.c_int32), ("variant", C.c_int32), ("elem", C.c_int32), ("inv_s", C.c_void_p),
                ("host_sums", C.POINTER(C.c_float))]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

# name -> (restype, argtypes); every symbol include/totsu_f32hip.h declares
_sz, _f, _i, _vp, _u64 = C.c_size_t, C.c_float, C.c_int, C.c_void_p, C.c_uint64
PROTOTYPES = {
    "thip_init": (_i, [_i]),
    "thip_shutdown": (_i, []),
    "thip_device_count": (_i, [C.POINTER(_i)]),
    "thip_set_stream": (_i, [_vp]),
    "thip_get_stream": (_vp, []),
    "thip_sync": (_i, []),
    "thip_last_error": (C.c_char_p, []),
    "thip_version": (C.c_char_p, []),
    "thip_alloc": (_i, [_sz, C.POINTER(_vp)]),
    "thip_alloc_zeroed": (_i, [_sz, C.POINTER(_vp)]),
    "thip_free": (_i, [_vp]),
    "thip_h2d": (_i, [_vp, _vp, _sz]),
    "thip_d2h": (_i, [_vp, _vp, _sz]),
    "thip_get": (_i, [_vp, _sz, fp]),
    "thip_set": (_i, [_vp, _sz, _f]),
    "thip_norm": (_i, [_sz, _vp, fp]),
    "thip_copy": (_i, [_sz, _vp, _vp]),
    "thip_scale": (_i, [_sz, _f, _vp]),
    "thip_add": (_i, [_sz, _f, _vp, _vp]),
    "thip_adds": (_i, [_sz, _f, _vp]),
    "thip_abssum": (_i, [_sz, _vp, _sz, fp]),
    "thip_transform_di": (_i, [_sz, _f, _vp, _vp, _f, _vp]),
    "thip_transform_ge": (_i, [_i, _sz, _sz, _f, _vp, _vp, _f, _vp]),
    "thip_transform_sp": (_i, [_sz, _f, _vp, _vp, _f, _vp]),
    "thip_set_lazy_gemv": (_i, [_i]),
    "thip_get_lazy_gemv": (_i, [C.POINTER(_i)]),
    "thip_lazy_gemv_stats": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "thip_lazy_plan_stats": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "thip_lazy_read_stats": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "thip_to_bf16": (_i, [_sz, _sz, _vp, _vp, _sz]),
    "thip_to_f16": (_i, [_sz, _sz, _vp, _vp, _sz, _vp]),
    "thip_transform_ge_f16": (_i, [_i, _sz, _sz, _f, _vp, _sz, _vp, _vp, _f, _vp]),
    "thip_transform_ge_bf16": (_i, [_i, _sz, _sz, _f, _vp, _sz, _vp, _f, _vp]),
    "thip_spmv_csr": (_i, [_sz, _sz, _sz, _vp, _vp, _vp, _f, _vp, _f, _vp, _i]),
    "thip_sptile_create": (_i, [_sz, _sz, _sz, _vp, _vp, _vp, C.POINTER(_vp)]),
    "thip_sptile_destroy": (_i, [_vp]),
    "thip_sptile_mv": (_i, [_vp, _i, _f, _vp, _f, _vp, _i]),
    "thip_sptile_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                              C.POINTER(_sz)]),
    "thip_sptile_layout": (_i, [_vp, C.POINTER(_i), C.POINTER(_sz), C.POINTER(_sz)]),
    "thip_map_eig_worklen": (_sz, [_sz]),
    "thip_map_eig": (_i, [_sz, _vp, _i, _f, _f, _vp, _sz, _i]),
    "thip_eig_decompose": (_i, [_sz, _vp, _i, _f, _f, _vp, _sz, fp]),
    "thip_eig_rebuild": (_i, [_sz, _vp, _i, _f, _vp, _sz, fp, C.POINTER(C.c_uint8)]),
    "thip_eig_engine_info": (_i, [C.POINTER(_i), C.POINTER(_i), fp]),
    "thip_test_eig_force": (_i, [_i]),
    "thip_norm_dev": (_i, [_sz, _vp, _vp]),
    "thip_dot_dev": (_i, [_sz, _vp, _vp, _vp]),
    "thip_abssum_dev": (_i, [_sz, _vp, _sz, _vp]),
    "thip_absadd_cols": (_i, [_sz, _sz, _vp, _vp]),
    "thip_absadd_rows": (_i, [_sz, _sz, _vp, _vp]),
    "thip_absadd_sympack": (_i, [_sz, _vp, _vp]),
    "thip_recip_max": (_i, [_sz, _f, _vp]),
    "thip_copy_block": (_i, [_i, _sz, _sz, _f, _vp, _vp, _sz]),
    "thip_proj_zero": (_i, [_i, _sz, _vp]),
    "thip_proj_rpos": (_i, [_sz, _vp]),
    "thip_proj_soc": (_i, [_sz, _vp]),
    "thip_proj_rotsoc": (_i, [_sz, _vp]),
    "thip_proj_soc_batched": (_i, [_vp, _vp, _sz, _i, _sz]),
    "thip_proj_psd": (_i, [_sz, _vp, _f, _vp, _sz]),
    "thip_group_min_batched": (_i, [_vp, _vp, _sz, _sz]),
    "thip_comm_unique_id": (_i, [C.POINTER(C.c_uint8)]),
    "thip_comm_init": (_i, [_i, _i, C.POINTER(C.c_uint8)]),
    "thip_comm_allreduce": (_i, [_vp, _sz]),
    "thip_comm_count": (_i, [C.POINTER(_i)]),
    "thip_comm_destroy": (_i, []),
    "thip_solver_use_rccl": (_i, [_vp]),
    "thip_oneshot_init": (_i, [_i, _i, _sz, C.POINTER(C.c_uint8)]),
    "thip_oneshot_connect": (_i, [C.POINTER(C.c_uint8)]),
    "thip_oneshot_allreduce": (_i, [_vp, _sz]),
    "thip_oneshot_error": (_i, [C.POINTER(_i)]),
    "thip_oneshot_destroy": (_i, []),
    "thip_solver_use_oneshot": (_i, [_vp]),
    "thip_solver_set_gemv_autotune": (_i, [_vp, _i]),
    "thip_solver_set_lda_pad": (_i, [_vp, _i]),
    "thip_solver_create": (_i, [C.POINTER(Problem), C.POINTER(Param), _i, C.POINTER(_vp)]),
    "thip_solver_set_csr": (_i, [_vp, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thip_solver_set_sptile": (_i, [_vp, _vp]),
    "thip_solver_set_allreduce": (_i, [_vp, ALLREDUCE_FN, _vp]),
    "thip_solver_set_overlap": (_i, [_vp, _i]),
    "thip_solver_overlap_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_sz)]),
    "thip_test_spin_allreduce": (_i, [_vp, _i]),
    "thip_solver_set_a_storage": (_i, [_vp, _i]),
    "thip_solver_set_a_bf16": (_i, [_vp, _vp, _sz]),
    "thip_solver_set_a_f16": (_i, [_vp, _vp, _sz, _vp]),
    "thip_solver_resume": (_i, [_vp]),
    "thip_solver_set_param": (_i, [_vp, _vp]),
    "thip_solver_init": (_i, [_vp]),
    "thip_solver_run": (_i, [_vp, C.c_int64, C.c_int64, C.POINTER(Status)]),
    "thip_solver_status": (_i, [_vp, C.POINTER(Status)]),
    "thip_solver_solution": (_i, [_vp, _vp, _vp]),
    "thip_solver_iterate": (_i, [_vp, _vp, _vp]),
    "thip_solver_precond": (_i, [_vp, _vp, _vp]),
    "thip_solver_destroy": (_i, [_vp]),
    "thip_solver_passes": (_i, [_vp, C.POINTER(_i), C.POINTER(_sz)]),
    "thip_solver_schedule_in_use": (_i, [_vp, C.POINTER(_i)]),
    "thip_solver_set_sweep_min_bytes": (_i, [_vp, _sz]),
    "thip_solver_set_column_shard": (_i, [_vp, _i]),
    "thip_sweep_probe": (_i, [_sz, _sz, _sz, _i, C.POINTER(_i)]),
    "thip_solver_sweep_plan": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_f)]),
    "thip_test_sweep": (_i, [_vp, C.POINTER(_f), C.POINTER(_i)]),
    "thip_stream_probe": (_i, [_vp, _sz, _i, C.POINTER(_f), C.POINTER(_f)]),
    "thip_solver_sweep_faults": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(C.c_int64)]),
    "thip_test_sweep_fault": (_i, [_vp, _i, C.c_int64, _i]),
    "thip_solver_set_sweep_publish": (_i, [_vp, _i]),
    "thip_sweep_publish_selftest": (_i, [_i, C.POINTER(_i), C.POINTER(_i)]),
    "thip_test_gemm_sym": (_i, [_i, _i, _f, _vp, _vp, _f, _vp, _f, _vp]),
    "thip_test_gemm_chain": (_i, [_i, _i, _i, _i, _i, _f, _vp, _vp, _f, _vp, _f, _vp]),
    "thip_test_chain_probe": (_i, [_i, _i, _i, C.POINTER(_f)]),
    "thip_test_sptile_time": (_i, [_vp, _i, C.POINTER(_f)]),
    "thip_test_gemm_dual": (_i, [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "thip_solver_gemv_plan": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_f)]),
    "thip_prof_enable": (_i, [_i]),
    "thip_prof_read": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "thip_prof_read_psd": (_i, [C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "thip_gen_vector": (_i, [_vp, _sz, _u64, _u64, _u64, _i, _f, _f]),
    "thip_gen_identity": (_i, [_vp, _sz, _sz, _sz, _u64, _f]),
    "thip_gen_matrix": (_i, [_vp, _sz, _sz, _sz, _u64, _u64, _u64, _u64, _u64, _i, _f, _f]),
}

_NOCHECK = {"thip_last_error", "thip_version", "thip_get_stream", "thip_map_eig_worklen"}

_cdll = None


def load():
    """Loads the shared library (without touching the GPU). Raises if it has not been built."""
    global _cdll
    if _cdll is not None:
        return _cdll
    if not os.path.exists(SO_PATH):
        raise ImportError("totsu_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % SO_PATH)
    try:
        # share one HIP runtime with torch when torch is used in the same process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        f = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        f.restype = res
        f.argtypes = args
    _cdll = lib
    return lib


class _Checked:
    """attribute access returns a wrapper that raises ThipError on a non-zero return code"""

    def __getattr__(self, name):
        lib = load()
        f = getattr(lib, name)
        if name in _NOCHECK:
            return f

        def call(*a):
            rc = f(*a)
            if rc != 0:
                raise ThipError(rc, lib.thip_last_error().decode(errors="replace"))
            return rc
        call.__name__ = name
        setattr(self, name, call)
        return call


lib = _Checked()
_inited = False


def init(device=None):
    """thip_init on the device of this rank (LOCAL_RANK) -- raises ThipError(E_NOGPU) without a GPU."""
    global _inited
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    lib.thip_init(int(device))
    _inited = True


class lazy_calls:
    """with lazy_calls(): the trait-level hosts' opt-in to deferred, batched small calls (thip_set_lazy_gemv) for the
    duration of their own call sequence; the previous setting is restored on exit"""

    def __enter__(self):
        prev = C.c_int(0)
        lib.thip_get_lazy_gemv(C.byref(prev))
        self.prev = prev.value
        lib.thip_set_lazy_gemv(1)
        return self

    def __exit__(self, *exc):
        lib.thip_set_lazy_gemv(self.prev)
        return False


def ensure_init():
    if not _inited:
        init()
