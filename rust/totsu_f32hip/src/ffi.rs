//! Raw bindings of `include/totsu_f32hip.h` (what `bindgen` emits).  AUTHORED, NOT COMPILED.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct thip_param { pub max_iter: i64, pub eps_acc: f32, pub eps_inf: f32, pub eps_zero: f32, pub log_period: i64 }

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

pub enum thip_solver {}
pub type thip_allreduce_fn = Option<unsafe extern "C" fn(ctx: *mut c_void, dev_buf: *mut f32, n: usize, stream: *mut c_void) -> c_int>;

pub const THIP_SCHED_REFERENCE: c_int = 0;
pub const THIP_SCHED_FUSED: c_int = 1;
pub const THIP_SCHED_CARRIED: c_int = 2;

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

    pub fn thip_absadd_cols(n_row: usize, n_col: usize, mat: *const f32, tau: *mut f32) -> c_int;
    pub fn thip_absadd_rows(n_row: usize, n_col: usize, mat: *const f32, sigma: *mut f32) -> c_int;
    pub fn thip_recip_max(n: usize, eps_zero: f32, x: *mut f32) -> c_int;
    pub fn thip_proj_zero(dual_cone: c_int, n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_rpos(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_soc(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_rotsoc(n: usize, x: *mut f32) -> c_int;
    pub fn thip_proj_psd(sn: usize, x: *mut f32, eps_zero: f32, work: *mut f32, worklen: usize) -> c_int;

    pub fn thip_solver_create(prob: *const thip_problem, par: *const thip_param, schedule: c_int,
                              out: *mut *mut thip_solver) -> c_int;
    pub fn thip_solver_set_allreduce(s: *mut thip_solver, f: thip_allreduce_fn, ctx: *mut c_void) -> c_int;
    pub fn thip_solver_init(s: *mut thip_solver) -> c_int;
    pub fn thip_solver_run(s: *mut thip_solver, max_steps: i64, poll_every: i64, host_status: *mut thip_status) -> c_int;
    pub fn thip_solver_solution(s: *mut thip_solver, host_x: *mut f32, host_y: *mut f32) -> c_int;
    pub fn thip_solver_destroy(s: *mut thip_solver) -> c_int;
}

/// The reference backends assert on library status (totsu_f32cuda/src/f32cuda.rs:38): so does this one.
pub fn chk(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { std::ffi::CStr::from_ptr(thip_last_error()) }.to_string_lossy().into_owned();
        panic!("totsu_f32hip: error {}: {}", rc, msg);
    }
}
