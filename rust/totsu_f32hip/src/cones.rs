//! Device cones for the hot loop.  `ConeRPos::proj` in totsu_core is a host loop over `get_mut()`
//! (cone_rpos.rs:40) that no backend can intercept; these types implement `Cone<F32HIP>` on the device and are
//! what `HipProbLP` / `HipProbSOCP` / `HipProbSDP` aliases would use.  AUTHORED, NOT COMPILED.
use totsu_core::solver::{Cone, SliceLike, SliceMut};
use crate::f32hip::F32HIP;
use crate::f32hip_slice::F32HIPSlice;
use crate::ffi::*;

pub struct HipConeRPos;
impl Cone<F32HIP> for HipConeRPos {
    fn proj(&mut self, _dual_cone: bool, x: &mut F32HIPSlice) -> Result<(), ()> {
        chk(unsafe { thip_proj_rpos(x.len(), x.get_dev_mut()) });
        Ok(())
    }
    fn product_group<G: Fn(&mut F32HIPSlice) + Copy>(&self, _dp_tau: &mut F32HIPSlice, _group: G) {}
}

pub struct HipConeSOC;
impl Cone<F32HIP> for HipConeSOC {
    fn proj(&mut self, _dual_cone: bool, x: &mut F32HIPSlice) -> Result<(), ()> {
        chk(unsafe { thip_proj_soc(x.len(), x.get_dev_mut()) });
        Ok(())
    }
    fn product_group<G: Fn(&mut F32HIPSlice) + Copy>(&self, dp_tau: &mut F32HIPSlice, group: G) { group(dp_tau); }
}

/// PSD projection on the matrix cores (no eigenvalue round trip); cone_psd.rs:22-85 semantics incl. work shortage.
pub struct HipConePSD<'a> { work: SliceMut<'a, F32HIPSlice>, eps_zero: f32 }
impl<'a> HipConePSD<'a> {
    pub fn query_worklen(nvars: usize) -> usize {
        let n = (((8 * nvars + 1) as f64).sqrt() as usize - 1) / 2;
        assert_eq!(n * (n + 1) / 2, nvars);
        unsafe { thip_map_eig_worklen(n) }
    }
    pub fn new(work: &'a mut [f32], eps_zero: f32) -> Self { HipConePSD { work: F32HIPSlice::new_mut(work), eps_zero } }
}
impl<'a> Cone<F32HIP> for HipConePSD<'a> {
    fn proj(&mut self, _dual_cone: bool, x: &mut F32HIPSlice) -> Result<(), ()> {
        if self.work.len() < Self::query_worklen(x.len()) { return Err(()); }
        let wl = self.work.len();
        chk(unsafe { thip_proj_psd(x.len(), x.get_dev_mut(), self.eps_zero, self.work.get_dev_mut(), wl) });
        Ok(())
    }
    fn product_group<G: Fn(&mut F32HIPSlice) + Copy>(&self, dp_tau: &mut F32HIPSlice, group: G) { group(dp_tau); }
}
