//! `F32HIP`: `LinAlg` + `LinAlgEx` (totsu_core/src/solver/linalg.rs:10-68, linalg_ex.rs:7-66) on gfx950 kernels.
//! AUTHORED, NOT COMPILED.  One call per trait function, like F64LAPACK (totsu_f64lapack/src/f64lapack.rs).
use std::os::raw::c_int;
use totsu_core::solver::{LinAlg, SliceLike};
use totsu_core::LinAlgEx;
use crate::f32hip_slice::F32HIPSlice;
use crate::ffi::*;

#[derive(Clone)]
pub struct F32HIP;

fn tri_order(sn: usize) -> usize {
    let n = (((8 * sn + 1) as f64).sqrt() as usize - 1) / 2;
    assert_eq!(n * (n + 1) / 2, sn);
    n
}

impl LinAlg for F32HIP {
    type F = f32;
    type Sl = F32HIPSlice;

    fn norm(x: &F32HIPSlice) -> f32 { let mut r = 0f32; chk(unsafe { thip_norm(x.len(), x.get_dev(), &mut r) }); r }
    fn copy(x: &F32HIPSlice, y: &mut F32HIPSlice) {
        assert_eq!(x.len(), y.len());
        chk(unsafe { thip_copy(x.len(), x.get_dev(), y.get_dev_mut()) })
    }
    fn scale(alpha: f32, x: &mut F32HIPSlice) { chk(unsafe { thip_scale(x.len(), alpha, x.get_dev_mut()) }) }
    fn add(alpha: f32, x: &F32HIPSlice, y: &mut F32HIPSlice) {
        assert_eq!(x.len(), y.len());
        chk(unsafe { thip_add(x.len(), alpha, x.get_dev(), y.get_dev_mut()) })
    }
    fn adds(s: f32, y: &mut F32HIPSlice) { chk(unsafe { thip_adds(y.len(), s, y.get_dev_mut()) }) }
    fn abssum(x: &F32HIPSlice, incx: usize) -> f32 {
        let mut r = 0f32;
        chk(unsafe { thip_abssum(x.len(), x.get_dev(), incx, &mut r) });
        r
    }
    fn transform_di(alpha: f32, mat: &F32HIPSlice, x: &F32HIPSlice, beta: f32, y: &mut F32HIPSlice) {
        assert_eq!(mat.len(), x.len());
        assert_eq!(mat.len(), y.len());
        chk(unsafe { thip_transform_di(x.len(), alpha, mat.get_dev(), x.get_dev(), beta, y.get_dev_mut()) })
    }
}

impl LinAlgEx for F32HIP {
    fn transform_ge(transpose: bool, n_row: usize, n_col: usize, alpha: f32, mat: &F32HIPSlice, x: &F32HIPSlice,
                    beta: f32, y: &mut F32HIPSlice) {
        assert_eq!(mat.len(), n_row * n_col);
        if transpose { assert_eq!(x.len(), n_row); assert_eq!(y.len(), n_col); }
        else { assert_eq!(x.len(), n_col); assert_eq!(y.len(), n_row); }
        chk(unsafe { thip_transform_ge(transpose as c_int, n_row, n_col, alpha, mat.get_dev(), x.get_dev(), beta, y.get_dev_mut()) })
    }
    fn transform_sp(n: usize, alpha: f32, mat: &F32HIPSlice, x: &F32HIPSlice, beta: f32, y: &mut F32HIPSlice) {
        assert_eq!(mat.len(), n * (n + 1) / 2);
        assert_eq!(x.len(), n);
        assert_eq!(y.len(), n);
        chk(unsafe { thip_transform_sp(n, alpha, mat.get_dev(), x.get_dev(), beta, y.get_dev_mut()) })
    }
    fn map_eig_worklen(n: usize) -> usize { unsafe { thip_map_eig_worklen(n) } }
    fn map_eig<M>(mat: &mut F32HIPSlice, scale_diag: Option<f32>, eps_zero: f32, work: &mut F32HIPSlice, map: M)
    where M: Fn(f32) -> Option<f32> {
        let n = tri_order(mat.len());
        assert!(work.len() >= Self::map_eig_worklen(n));
        let (has, sc) = scale_diag.map_or((0, 0f32), |s| (1, s));
        let wl = work.len();
        // eigenvalues visit the host so that the arbitrary closure can be applied (linalg_ex.rs:64-65)
        let mut w = vec![0f32; n.max(1)];
        chk(unsafe { thip_eig_decompose(n, mat.get_dev_mut(), has, sc, eps_zero, work.get_dev_mut(), wl, w.as_mut_ptr()) });
        let (mut e, mut keep) = (vec![0f32; n.max(1)], vec![0u8; n.max(1)]);
        for i in 0..n { if let Some(v) = map(w[i]) { e[i] = v; keep[i] = 1; } }
        chk(unsafe { thip_eig_rebuild(n, mat.get_dev_mut(), has, sc, work.get_dev_mut(), wl, e.as_ptr(), keep.as_ptr()) });
    }
}
