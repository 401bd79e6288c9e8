//! `HipProbLP` / `HipProbSOCP` and `FusedSolver`: the drop-in BY ALIAS.  AUTHORED, NOT COMPILED; the same design is
//! compiled and tested in C++ (`include/totsu_f32hip_prob.hpp`, `examples/hip_prob_demo.cpp`,
//! `tests/test_gpu_hosts.py`) and in Python (`totsu_amd/problem.py`, `Solver.fused`).
//!
//! A caller written against `totsu`'s builders changes one type name:
//!
//! ```ignore
//! // before: let mut lp = ProbLP::<F32HIP>::new(vec_c, mat_g, vec_h, mat_a, vec_b);
//! let mut lp = HipProbLP::new(vec_c, mat_g, vec_h, mat_a, vec_b);        // same arguments (lp.rs:259-307)
//! let rslt = HipSolver::new().par(|p| { p.eps_acc = 1e-3; }).solve(lp.problem())?;   // same call shape
//! ```
//!
//! `problem()` returns a `HipProblem`: the reference's tuple (for a call-by-call solve through `Solver<F32HIP>`)
//! PLUS a dense description of the same problem -- stacked A (device), b, c, cone segments -- which
//! `HipSolver::solve` hands to the device-resident loop (`thip_solver_*`).  Measured on the C++ twin at BASELINE
//! configs[1]: 3471 iter/s through the alias vs 3368 driving the fused loop directly; the call-by-call route gets 787.
use std::os::raw::c_int;
use totsu::prelude::*;
use totsu::{MatBuild, ProbLP, ProbSOCP};
use crate::f32hip::F32HIP;
use crate::f32hip_slice::F32HIPSlice;
use crate::ffi::*;

/// dense description for the fused loop (`thip_problem` + owned device buffers)
pub struct Dense {
    pub n: usize, pub m: usize,
    a: *mut f32, b: *mut f32, c: *mut f32, b_rowabs: *mut f32,
    seg_type: Vec<i32>, seg_len: Vec<i64>,
}
impl Drop for Dense {
    fn drop(&mut self) { for p in [self.a, self.b, self.c, self.b_rowabs] { if !p.is_null() { chk(unsafe { thip_free(p) }); } } }
}

fn upload(v: &[f32]) -> *mut f32 {
    let mut d = std::ptr::null_mut();
    chk(unsafe { thip_alloc(v.len().max(1), &mut d) });
    if !v.is_empty() { chk(unsafe { thip_h2d(d, v.as_ptr(), v.len()) }); }
    d
}

/// rows of `blk` (nr x n, column-major) go to rows r0.. of the stacked m x n matrix; `transposed`: an n-vector as ONE row
fn stack(dst: *mut f32, m: usize, r0: usize, blk: &MatBuild<F32HIP>, n: usize, sign: f32, transposed: bool) {
    let nr = if transposed { 1 } else { blk.size().0 };
    if nr == 0 { return; }
    let sl = F32HIPSlice::new_ref(blk.as_ref());            // mirror cache: one upload per host array (see f32hip_slice.rs)
    chk(unsafe { thip_copy_block(transposed as c_int, nr, n, sign, sl.get_dev(), dst.add(r0), m) });
}

pub struct HipProbLP { inner: ProbLP<F32HIP>, dense: Dense }
impl HipProbLP {
    /// lp.rs:259-307, same arguments
    pub fn new(vec_c: MatBuild<F32HIP>, mat_g: MatBuild<F32HIP>, vec_h: MatBuild<F32HIP>,
               mat_a: MatBuild<F32HIP>, vec_b: MatBuild<F32HIP>) -> Self {
        let (n, m, p) = (vec_c.size().0, vec_h.size().0, vec_b.size().0);
        let mut a = std::ptr::null_mut();
        chk(unsafe { thip_alloc(((m + p) * n).max(1), &mut a) });
        stack(a, m + p, 0, &mat_g, n, 1., false);
        stack(a, m + p, m, &mat_a, n, 1., false);
        let b: Vec<f32> = vec_h.as_ref().iter().chain(vec_b.as_ref().iter()).cloned().collect();
        let babs: Vec<f32> = b.iter().map(|v| v.abs()).collect();
        let dense = Dense { n, m: m + p, a, b: upload(&b), c: upload(vec_c.as_ref()), b_rowabs: upload(&babs),
                            seg_type: vec![THIP_CONE_RPOS, THIP_CONE_ZERO], seg_len: vec![m as i64, p as i64] };
        HipProbLP { inner: ProbLP::new(vec_c, mat_g, vec_h, mat_a, vec_b), dense }
    }
    pub fn problem(&mut self) -> HipProblem<'_, impl FnOnce(&Solver<F32HIP>) -> Result<(&[f32], &[f32]), SolverError> + '_> {
        let inner = &mut self.inner;
        HipProblem { dense: &self.dense, by_calls: move |s: &Solver<F32HIP>| s.solve(inner.problem()) }
    }
}

pub struct HipProbSOCP { inner: ProbSOCP<F32HIP>, dense: Dense }
impl HipProbSOCP {
    /// socp.rs:376-428, same arguments.  Stacked rows of cone i: [-c_i^T ; -G_i], b = [d_i ; h_i] (socp.rs:88-93)
    pub fn new(vec_f: MatBuild<F32HIP>, mats_g: Vec<MatBuild<F32HIP>>, vecs_h: Vec<MatBuild<F32HIP>>,
               vecs_c: Vec<MatBuild<F32HIP>>, scls_d: Vec<f32>, mat_a: MatBuild<F32HIP>, vec_b: MatBuild<F32HIP>) -> Self {
        let (n, p) = (vec_f.size().0, vec_b.size().0);
        let m: usize = mats_g.iter().map(|g| 1 + g.size().0).sum::<usize>() + p;
        let mut a = std::ptr::null_mut();
        chk(unsafe { thip_alloc((m * n).max(1), &mut a) });
        let (mut r0, mut b, mut babs, mut st, mut sl) = (0, vec![], vec![], vec![], vec![]);
        for i in 0..mats_g.len() {
            let ni = mats_g[i].size().0;
            stack(a, m, r0, &vecs_c[i], n, -1., true);
            stack(a, m, r0 + 1, &mats_g[i], n, -1., false);
            b.push(scls_d[i]); babs.push(scls_d[i]);                    // socp.rs:259-279 adds d_i, not |d_i|
            for v in vecs_h[i].as_ref() { b.push(*v); babs.push(v.abs()); }
            st.push(THIP_CONE_SOC); sl.push((1 + ni) as i64);
            r0 += 1 + ni;
        }
        stack(a, m, r0, &mat_a, n, 1., false);
        for v in vec_b.as_ref() { b.push(*v); babs.push(v.abs()); }
        st.push(THIP_CONE_ZERO); sl.push(p as i64);
        let dense = Dense { n, m, a, b: upload(&b), c: upload(vec_f.as_ref()), b_rowabs: upload(&babs), seg_type: st, seg_len: sl };
        HipProbSOCP { inner: ProbSOCP::new(vec_f, mats_g, vecs_h, vecs_c, scls_d, mat_a, vec_b), dense }
    }
    pub fn problem(&mut self) -> HipProblem<'_, impl FnOnce(&Solver<F32HIP>) -> Result<(&[f32], &[f32]), SolverError> + '_> {
        let inner = &mut self.inner;
        HipProblem { dense: &self.dense, by_calls: move |s: &Solver<F32HIP>| s.solve(inner.problem()) }
    }
}

/// what `Hip*::problem()` returns: the dense description, and the call-by-call route kept as a closure
pub struct HipProblem<'a, C> { pub dense: &'a Dense, pub by_calls: C }

/// RAII over `thip_solver_*` (the C++ twin is `totsu::FusedSolver`, include/totsu_f32hip.hpp)
pub struct FusedSolver { h: *mut thip_solver, n: usize, m: usize }
impl FusedSolver {
    pub fn new(d: &Dense, par: &SolverParam<f32>, schedule: c_int) -> Self {
        Self::with_state_arith(d, par, schedule, THIP_STATE_COMPENSATED)
    }
    /// `state_arith`: THIP_STATE_COMPENSATED (the library's default: Kahan terms on the iterate updates) or
    /// THIP_STATE_PLAIN (the reference's literal f32 additions, solver.rs:542,560)
    pub fn with_state_arith(d: &Dense, par: &SolverParam<f32>, schedule: c_int, state_arith: i32) -> Self {
        let prob = thip_problem { n: d.n, m: d.m, mat_a: d.a, vec_b: d.b, vec_c: d.c, vec_b_rowabs: d.b_rowabs,
                                  n_seg: d.seg_type.len(), host_seg_type: d.seg_type.as_ptr(), host_seg_len: d.seg_len.as_ptr() };
        let p = thip_param { max_iter: par.max_iter.map_or(-1, |v| v as i64), eps_acc: par.eps_acc, eps_inf: par.eps_inf,
                             eps_zero: par.eps_zero, log_period: 0, state_arith, reserved: 0 };
        let mut h = std::ptr::null_mut();
        chk(unsafe { thip_solver_create(&prob, &p, schedule, &mut h) });
        chk(unsafe { thip_solver_init(h) });
        FusedSolver { h, n: d.n, m: d.m }
    }
    /// the schedule the next run executes: THIP_SCHED_CARRIED when THIP_SCHED_SWEEP (one pass over A per iteration) cannot
    /// take the problem (include/totsu_f32hip.h: thip_solver_schedule_in_use)
    pub fn schedule_in_use(&mut self) -> c_int {
        let mut v: c_int = 0;
        chk(unsafe { thip_solver_schedule_in_use(self.h, &mut v) });
        v
    }
    /// recoveries from a one-pass kernel that gave up (thip_solver_sweep_faults): how many in this solve, the kernel's error
    /// word of the last one, the iteration of the snapshot the run went back to (-1: none).  The answer of `solve` is valid
    /// either way: the run restored its snapshot and went on with the 2-pass schedule.
    pub fn sweep_faults(&mut self) -> (c_int, c_int, i64) {
        let (mut k, mut w, mut it): (c_int, c_int, i64) = (0, 0, -1);
        chk(unsafe { thip_solver_sweep_faults(self.h, &mut k, &mut w, &mut it) });
        (k, w, it)
    }
    /// runs to termination; Ok((x, y)) or the reference's SolverError (solver_error.rs:3-17)
    pub fn solve(&mut self) -> Result<(Vec<f32>, Vec<f32>), SolverError> {
        let mut st: thip_status = unsafe { std::mem::zeroed() };
        chk(unsafe { thip_solver_run(self.h, -1, 64, &mut st) });
        match st.state {
            0 => { let (mut x, mut y) = (vec![0f32; self.n], vec![0f32; self.m]);
                   chk(unsafe { thip_solver_solution(self.h, x.as_mut_ptr(), y.as_mut_ptr()) }); Ok((x, y)) }
            1 => Err(SolverError::Unbounded), 2 => Err(SolverError::Infeasible), 3 => Err(SolverError::ExcessIter),
            4 => Err(SolverError::InvalidOp), 5 => Err(SolverError::WorkShortage), _ => Err(SolverError::ConeFailure),
        }
    }
}
impl Drop for FusedSolver { fn drop(&mut self) { unsafe { thip_solver_destroy(self.h) }; } }

/// `Solver::<F32HIP>` with the dispatch: a `HipProblem` goes to the device-resident loop
pub struct HipSolver { pub inner: Solver<F32HIP>, pub fused: bool, pub state_arith: i32 }
impl HipSolver {
    pub fn new() -> Self { HipSolver { inner: Solver::new(), fused: true, state_arith: THIP_STATE_COMPENSATED } }
    /// the reference's literal f32 iterate arithmetic instead of the compensated default
    pub fn plain_state(mut self) -> Self { self.state_arith = THIP_STATE_PLAIN; self }
    pub fn par<P: FnOnce(&mut SolverParam<f32>)>(mut self, f: P) -> Self { f(&mut self.inner.param); self }
    pub fn solve<C>(self, prob: HipProblem<'_, C>) -> Result<(Vec<f32>, Vec<f32>), SolverError>
    where C: FnOnce(&Solver<F32HIP>) -> Result<(&[f32], &[f32]), SolverError> {
        if self.fused { FusedSolver::with_state_arith(prob.dense, &self.inner.param, THIP_SCHED_SWEEP, self.state_arith).solve() }
        else { (prob.by_calls)(&self.inner).map(|(x, y)| (x.to_vec(), y.to_vec())) }
    }
}
