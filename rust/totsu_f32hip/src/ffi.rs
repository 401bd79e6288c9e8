//! Raw bindings of `include/totsu_f32hip.h` (what `bindgen` emits).  AUTHORED, NOT COMPILED.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct thip_param { pub max_iter: i64, pub eps_acc: f32, pub eps_inf: f32, pub eps_zero: f32, pub log_period: i64,
                         pub state_arith: i32, pub reserved: i32 }
pub const THIP_STATE_COMPENSATED: i32 = 0;
pub const THIP_STATE_PLAIN: i32 = 1;

#[repr(C)]
pub struct thip_problem {
    pub n: usize, pub m: usize,
    pub mat_a: *const f32, pub vec_b: *const f32, pub vec_c: *const f32, pub vec_b_rowabs: *const f32,
    pub n_seg: usize, pub host_seg_type: *const i32, pub host_seg_len: *const i64,
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
pub struct thip_status {
    pub state: i32, pub iter: i64, pub kind: i32, pub cri: [f32; 3],
    pub tau: f32, pub kappa: f32, pub norm_b: f32, pub norm_c: f32,
}

#[repr(C)]
pub struct thip_sweep_test {
    pub m: usize, pub n: usize, pub lda: usize,
    pub mat_a: *const f32, pub v: *const f32, pub xy: *const f32, pub c: *const f32, pub su: *const f32, pub tx: *const f32,
    pub u: *mut f32, pub ku: *mut f32,
    pub xx_in: *const f32, pub kx_in: *const f32,
    pub xx_out: *mut f32, pub kx_out: *mut f32, pub gp: *mut f32, pub hn: *mut f32, pub h3: *mut f32,
    pub kappa: f32, pub rtau: f32, pub first: i32, pub reps: i32,
    pub force_members: i32, pub pub_agent: i32, pub variant: i32, pub elem: i32,
    pub inv_s: *const f32,
    pub host_sums: *mut f32,
}

pub enum thip_solver {}
pub enum thip_sptile {}
pub type thip_allreduce_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, dev_buf: *mut f32, n: usize, stream: *mut c_void) -> c_int>;

pub const THIP_CONE_ZERO: i32 = 0;
pub const THIP_CONE_RPOS: i32 = 1;
pub const THIP_CONE_SOC: i32 = 2;
pub const THIP_CONE_ROTSOC: i32 = 3;
pub const THIP_CONE_PSD: i32 = 4;
pub const THIP_SCHED_REFERENCE: c_int = 0;
pub const THIP_SCHED_FUSED: c_int = 1;
pub const THIP_SCHED_CARRIED: c_int = 2;
pub const THIP_SCHED_SWEEP: c_int = 3;
pub const THIP_OVERLAP_OFF: c_int = 0;
pub const THIP_OVERLAP_LOCAL_ROWS: c_int = 1;
pub const THIP_OVERLAP_COLUMN_PIPELINE: c_int = 2;
pub const THIP_OVERLAP_COLUMN_INORDER: c_int = 3;

extern "C" {
    pub fn thip_init(device: c_int) -> c_int;
    pub fn thip_shutdown() -> c_int;
    pub fn thip_sync() -> c_int;
    pub fn thip_last_error() -> *const c_char;
    pub fn thip_alloc(n: usize, out: *mut *mut f32) -> c_int;
    pub fn thip_free(p: *mut f32) -> c_int;
    pub fn thip_h2d(dst: *mut f32, host_src: *const f32, n: usize) -> c_int;
    pub fn thip_d2h(host_dst: *mut f32, src: *const f32, n: usize) -> c_int;
    pub fn thip_get(x: *const f32, idx: usize, host_out: *mut f32) -> c_int;
    pub fn thip_set(x: *mut f32, idx: usize, val: f32) -> c_int;

    pub fn thip_norm(n: usize, x: *const f32, host_out: *mut f32) -> c_int;
    pub fn thip_copy(n: usize, x: *const f32, y: *mut f32) -> c_int;
    pub fn thip_scale(n: usize, alpha: f32, x: *mut f32) -> c_int;
    pub fn thip_add(n: usize, alpha: f32, x: *const f32, y: *mut f32) -> c_int;
    pub fn thip_adds(n: usize, s: f32, y: *mut f32) -> c_int;
    pub fn thip_abssum(len: usize, x: *const f32, incx: usize, host_out: *mut f32) -> c_int;
    pub fn thip_transform_di(n: usize, alpha: f32, d: *const f32, x: *const f32, beta: f32, y: *mut f32) -> c_int;

    pub fn thip_transform_ge(transpose: c_int, n_row: usize, n_col: usize, alpha: f32, mat: *const f32,
                             x: *const f32, beta: f32, y: *mut f32) -> c_int;
    pub fn thip_transform_sp(n: usize, alpha: f32, mat: *const f32, x: *const f32, beta: f32, y: *mut f32) -> c_int;
    pub fn thip_map_eig_worklen(n: usize) -> usize;
    pub fn thip_map_eig(n: usize, mat: *mut f32, has_scale: c_int, scale_diag: f32, eps_zero: f32,
                        work: *mut f32, worklen: usize, map_kind: c_int) -> c_int;
    pub fn thip_eig_decompose(n: usize, mat: *mut f32, has_scale: c_int, scale_diag: f32, eps_zero: f32,
                              work: *mut f32, worklen: usize, host_w: *mut f32) -> c_int;
    pub fn thip_eig_rebuild(n: usize, mat: *mut f32, has_scale: c_int, scale_diag: f32, work: *mut f32,
                            worklen: usize, host_e: *const f32, host_keep: *const u8) -> c_int;
    pub fn thip_eig_engine_info(host_engine: *mut c_int, host_polish: *mut c_int, host_cert: *mut f32) -> c_int;

    pub fn thip_absadd_cols(n_row: usize, n_col: usize, mat: *const f32, tau: *mut f32) -> c_int;
    pub fn thip_absadd_rows(n_row: usize, n_col: usize, mat: *const f32, sigma: *mut f32) -> c_int;
    pub fn thip_recip_max(n: usize, eps_zero: f32, x: *mut f32) -> c_int;
    pub fn thip_copy_block(transposed: c_int, n_row: usize, n_col: usize, sign: f32, src: *const f32, dst: *mut f32,
                           ld_dst: usize) -> c_int;
    pub fn thip_set_lazy_gemv(on: c_int) -> c_int;
    pub fn thip_get_lazy_gemv(host_on: *mut c_int) -> c_int;
    pub fn thip_lazy_gemv_stats(host_deferred: *mut i64, host_flushes: *mut i64) -> c_int;
    pub fn thip_lazy_plan_stats(host_hits: *mut i64, host_misses: *mut i64) -> c_int;
    pub fn thip_lazy_read_stats(host_served: *mut i64, host_fetches: *mut i64) -> c_int;
    pub fn thip_proj_zero(dual_cone: c_int, n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_rpos(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_soc(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_rotsoc(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_psd(sn: usize, x: *mut f32, eps_zero: f32, work: *mut f32, worklen: usize) -> c_int;

    pub fn thip_solver_create(prob: *const thip_problem, par: *const thip_param, schedule: c_int,
                              out: *mut *mut thip_solver) -> c_int;
    pub fn thip_solver_set_allreduce(s: *mut thip_solver, f: thip_allreduce_fn, ctx: *mut c_void) -> c_int;
    pub fn thip_solver_set_overlap(s: *mut thip_solver, mode: c_int) -> c_int;
    pub fn thip_solver_overlap_info(s: *mut thip_solver, host_mode: *mut c_int, host_launches_per_pass: *mut c_int,
                                    host_split_col: *mut usize) -> c_int;
    pub fn thip_solver_init(s: *mut thip_solver) -> c_int;
    pub fn thip_solver_run(s: *mut thip_solver, max_steps: i64, poll_every: i64, host_status: *mut thip_status) -> c_int;
    pub fn thip_solver_solution(s: *mut thip_solver, host_x: *mut f32, host_y: *mut f32) -> c_int;
    pub fn thip_solver_destroy(s: *mut thip_solver) -> c_int;

    // ---- the rest of include/totsu_f32hip.h: context, device-scalar reductions, batched cones, sparse and
    // ---- reduced-precision operators, solver controls, RCCL communicator, generators, profiling ----
    pub fn thip_version() -> *const c_char;
    pub fn thip_device_count(host_count: *mut c_int) -> c_int;
    pub fn thip_set_stream(hip_stream: *mut c_void) -> c_int;
    pub fn thip_get_stream() -> *mut c_void;
    pub fn thip_alloc_zeroed(n: usize, out: *mut *mut f32) -> c_int;

    pub fn thip_norm_dev(n: usize, x: *const f32, dev_out: *mut f32) -> c_int;
    pub fn thip_dot_dev(n: usize, x: *const f32, y: *const f32, dev_out: *mut f32) -> c_int;
    pub fn thip_abssum_dev(len: usize, x: *const f32, incx: usize, dev_out: *mut f32) -> c_int;
    pub fn thip_absadd_sympack(n: usize, mat: *const f32, y: *mut f32) -> c_int;
    pub fn thip_spmv_csr(n_row: usize, n_col: usize, nnz: usize, dev_rowptr: *const i64, dev_colidx: *const i32,
                         vals: *const f32, alpha: f32, x: *const f32, beta: f32, y: *mut f32, abs_mode: c_int) -> c_int;
    pub fn thip_sptile_create(n_row: usize, n_col: usize, nnz: usize, host_colptr: *const i64, host_rowidx: *const i32,
                              host_vals: *const f32, out: *mut *mut thip_sptile) -> c_int;
    pub fn thip_sptile_destroy(mat: *mut thip_sptile) -> c_int;
    pub fn thip_sptile_mv(mat: *mut thip_sptile, transpose: c_int, alpha: f32, x: *const f32, beta: f32, y: *mut f32,
                          abs_mode: c_int) -> c_int;
    pub fn thip_sptile_info(mat: *const thip_sptile, host_nnz_stored: *mut usize, host_tiles: *mut c_int,
                            host_items_n: *mut c_int, host_items_t: *mut c_int, host_slices_n: *mut c_int,
                            host_slices_t: *mut c_int, host_bytes: *mut usize) -> c_int;
    pub fn thip_sptile_layout(mat: *const thip_sptile, host_dense_tiles: *mut c_int, host_indexed_entries: *mut usize,
                              host_bytes_per_product: *mut usize) -> c_int;
    pub fn thip_to_bf16(n_row: usize, n_col: usize, mat: *const f32, mat16: *mut u16, ld16: usize) -> c_int;
    pub fn thip_transform_ge_bf16(transpose: c_int, n_row: usize, n_col: usize, alpha: f32, mat16: *const u16,
                                  ld16: usize, x: *const f32, beta: f32, y: *mut f32) -> c_int;

    pub fn thip_proj_soc_batched(x: *mut f32, dev_offs: *const i64, n_cones: usize, rotated: c_int, max_len: usize) -> c_int;
    pub fn thip_group_min_batched(dp_tau: *mut f32, dev_offs: *const i64, n_groups: usize, max_len: usize) -> c_int;

    pub fn thip_solver_set_csr(s: *mut thip_solver, nnz: usize, dev_rowptr: *const i64, dev_colidx: *const i32,
                               dev_vals: *const f32, dev_t_rowptr: *const i64, dev_t_colidx: *const i32,
                               dev_t_vals: *const f32) -> c_int;
    pub fn thip_solver_set_sptile(s: *mut thip_solver, mat: *mut thip_sptile) -> c_int;
    pub fn thip_solver_set_a_storage(s: *mut thip_solver, a_kind: c_int) -> c_int;
    pub fn thip_solver_set_a_bf16(s: *mut thip_solver, mat16: *const u16, ld16: usize) -> c_int;
    pub fn thip_solver_set_a_f16(s: *mut thip_solver, mat16: *const u16, ld16: usize, inv_scale: *const f32) -> c_int;
    pub fn thip_to_f16(n_row: usize, n_col: usize, mat: *const f32, mat16: *mut u16, ld16: usize, inv_scale: *mut f32) -> c_int;
    pub fn thip_transform_ge_f16(transpose: c_int, n_row: usize, n_col: usize, alpha: f32, mat16: *const u16, ld16: usize,
                                 inv_scale: *const f32, x: *const f32, beta: f32, y: *mut f32) -> c_int;
    pub fn thip_solver_set_param(s: *mut thip_solver, par: *const thip_param) -> c_int;
    pub fn thip_solver_resume(s: *mut thip_solver) -> c_int;
    pub fn thip_solver_status(s: *mut thip_solver, host_status: *mut thip_status) -> c_int;
    pub fn thip_solver_iterate(s: *mut thip_solver, host_x: *mut f32, host_y: *mut f32) -> c_int;
    pub fn thip_solver_precond(s: *mut thip_solver, host_dp_tau: *mut f32, host_dp_sigma: *mut f32) -> c_int;
    pub fn thip_solver_passes(s: *const thip_solver, host_passes: *mut c_int, host_bytes_per_pass: *mut usize) -> c_int;
    pub fn thip_solver_schedule_in_use(s: *mut thip_solver, host_schedule: *mut c_int) -> c_int;
    pub fn thip_solver_set_sweep_min_bytes(s: *mut thip_solver, bytes: usize) -> c_int;
    pub fn thip_solver_set_column_shard(s: *mut thip_solver, on: c_int) -> c_int;
    pub fn thip_sweep_probe(m: usize, n_local: usize, lda: usize, elem: c_int, host_ok: *mut c_int) -> c_int;
    pub fn thip_solver_sweep_plan(s: *mut thip_solver, host_members: *mut c_int, host_cols_per_panel: *mut c_int, host_slots: *mut c_int, host_ms: *mut f32) -> c_int;
    pub fn thip_stream_probe(dev_ptr: *const c_void, bytes: usize, reps: c_int, host_best_ms: *mut f32, host_avg_ms: *mut f32) -> c_int;
    pub fn thip_solver_sweep_faults(s: *mut thip_solver, host_faults: *mut c_int, host_last_word: *mut c_int, host_restored_iter: *mut i64) -> c_int;
    pub fn thip_solver_set_sweep_publish(s: *mut thip_solver, agent_scope: c_int) -> c_int;
    pub fn thip_sweep_publish_selftest(mode: c_int, host_agent_scope: *mut c_int, host_info: *mut c_int) -> c_int;
    pub fn thip_solver_gemv_plan(s: *const thip_solver, host_nj: *mut c_int, host_blocks: *mut c_int, host_ms: *mut f32) -> c_int;

    pub fn thip_comm_unique_id(host_id128: *mut u8) -> c_int;
    pub fn thip_comm_init(rank: c_int, world: c_int, host_id128: *const u8) -> c_int;
    pub fn thip_comm_allreduce(dev_buf: *mut f32, n: usize) -> c_int;
    pub fn thip_comm_count(host_ranks: *mut c_int) -> c_int;
    pub fn thip_comm_destroy() -> c_int;
    pub fn thip_solver_use_rccl(s: *mut thip_solver) -> c_int;
    pub fn thip_oneshot_init(rank: c_int, world: c_int, max_floats: usize, host_handle64: *mut u8) -> c_int;
    pub fn thip_oneshot_connect(host_handles: *const u8) -> c_int;
    pub fn thip_oneshot_allreduce(dev_buf: *mut f32, n: usize) -> c_int;
    pub fn thip_oneshot_error(host_err: *mut c_int) -> c_int;
    pub fn thip_oneshot_destroy() -> c_int;
    pub fn thip_solver_use_oneshot(s: *mut thip_solver) -> c_int;
    pub fn thip_solver_set_gemv_autotune(s: *mut thip_solver, on: c_int) -> c_int;
    pub fn thip_solver_set_lda_pad(s: *mut thip_solver, floats: c_int) -> c_int;

    pub fn thip_gen_vector(out: *mut f32, n: usize, seed: u64, stream: u64, idx0: u64, kind: c_int, scale: f32, shift: f32) -> c_int;
    pub fn thip_gen_matrix(out: *mut f32, n_row: usize, n_col: usize, lda: usize, seed: u64, stream: u64, row0: u64,
                           col0: u64, ld_index: u64, kind: c_int, scale: f32, shift: f32) -> c_int;
    pub fn thip_gen_identity(out: *mut f32, n_row: usize, n_col: usize, lda: usize, row0: u64, value: f32) -> c_int;

    pub fn thip_prof_enable(on: c_int) -> c_int;
    pub fn thip_prof_read(host_launches: *mut i64, host_total_ms: *mut f64) -> c_int;
    pub fn thip_prof_read_psd(host_spans: *mut i64, host_total_ms: *mut f64) -> c_int;
}

/// Test hooks and timing probes (include/totsu_f32hip_test.h): exported by the same library, not part of the interface
/// this crate wraps -- declared only for the crate's own device tests.
#[cfg(feature = "test-hooks")]
extern "C" {
    pub fn thip_test_eig_force(engine: c_int) -> c_int;
    pub fn thip_test_spin_allreduce(s: *mut thip_solver, latency_us: c_int) -> c_int;
    pub fn thip_test_sweep(t: *const thip_sweep_test, host_ms: *mut f32, host_info: *mut c_int) -> c_int;
    pub fn thip_test_sweep_fault(s: *mut thip_solver, kind: c_int, after_sweeps: i64, spin_max: c_int) -> c_int;
    pub fn thip_test_gemm_sym(n: c_int, ld: c_int, alpha: f32, a: *const f32, b: *const f32, beta: f32, d: *const f32,
                              gamma: f32, c: *mut f32) -> c_int;
    pub fn thip_test_gemm_chain(shape: c_int, kernel: c_int, n: c_int, ld: c_int, nb: c_int, alpha: f32, x: *const f32,
                                y: *const f32, beta: f32, d: *const f32, gamma: f32, c: *mut f32) -> c_int;
    pub fn thip_test_chain_probe(mode: c_int, ld: c_int, reps: c_int, host_us: *mut f32) -> c_int;
    pub fn thip_test_sptile_time(mat: *mut thip_sptile, reps: c_int, host_ms: *mut f32) -> c_int;
    pub fn thip_test_gemm_dual(kernel: c_int, n: c_int, ld: c_int, nb: c_int, a: *const f32, b0: *const f32, b1: *const f32,
                               coef: *const f32, o0: *mut f32, o1: *mut f32) -> c_int;
}

pub const THIP_A_F32: c_int = 0;
pub const THIP_A_BF16: c_int = 1;
pub const THIP_A_F16: c_int = 2;

/// The reference backends assert on library status (totsu_f32cuda/src/f32cuda.rs:38): so does this one.
pub fn chk(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(thip_last_error()) }.to_string_lossy().into_owned();
        panic!("totsu_f32hip: error {}: {}", rc, msg);
    }
}
