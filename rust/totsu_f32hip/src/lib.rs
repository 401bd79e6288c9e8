//! `totsu_f32hip`: an AMD Instinct MI355X (gfx950) backend for `totsu_core`'s first-order conic solver.
//!
//! AUTHORED, NOT COMPILED -- the environment this repository is built in has no Rust toolchain.  The C ABI these
//! modules bind (`include/totsu_f32hip.h`) is exercised by the Python mirror and the GPU tests of the repository.
//!
//! ```ignore
//! use totsu::prelude::*;
//! use totsu::*;
//! use totsu_f32hip::F32HIP;
//! type La = F32HIP;
//! let s = Solver::<La>::new().par(|p| { p.eps_acc = 1e-3; });
//! let mut lp = ProbLP::<La>::new(vec_c, mat_g, vec_h, mat_a, vec_b);      // unchanged problem builders
//! let rslt = s.solve(lp.problem()).unwrap();
//! ```
pub mod ffi;
pub mod f32hip;
pub mod f32hip_slice;
pub mod cones;
pub mod prob;

pub use f32hip::F32HIP;
pub use f32hip_slice::F32HIPSlice;
pub use cones::{HipConePSD, HipConeRPos, HipConeSOC};
pub use prob::{FusedSolver, HipProbLP, HipProbSOCP, HipSolver};

/// Selects the GPU (cuda_mgr.rs:30-60 hard-codes device 0; any device here).  Call once per thread before use.
/// The crate drives the library through this API only and never installs a stream of its own, so it opts into the
/// deferred, batched execution of small calls (`thip_set_lazy_gemv`, off by default in the library): the per-block loops
/// of `ProbSOCPOpA/OpB` and `ProbSOCPCone::proj` then run as a handful of launches.
pub fn init(device: i32) {
    ffi::chk(unsafe { ffi::thip_init(device) });
    ffi::chk(unsafe { ffi::thip_set_lazy_gemv(1) });
}
