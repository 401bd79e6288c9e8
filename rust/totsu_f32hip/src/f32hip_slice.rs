//! `F32HIPSlice`: `SliceLike` (totsu_core/src/solver/slicelike.rs:9-70) over a device mirror.
//! AUTHORED, NOT COMPILED.  Same contract as the Python `F32HIPSlice` (totsu_amd/linalg.py), which is tested:
//! the DEVICE copy is the truth after `new_ref`/`new_mut`; `get_ref` downloads the range, `get_mut` downloads it
//! and marks it host-dirty (uploaded again before the next device use); dropping the root `new_mut` slice leaves
//! the caller's host buffer up to date and releases the device memory.
use std::cell::RefCell;
use std::rc::Rc;
use totsu_core::solver::{SliceLike, SliceMut, SliceRef};
use crate::ffi::*;

struct Root {
    dev: *mut f32,
    host: *mut f32,          // the caller's buffer (const for new_ref)
    n: usize,
    mutable: bool,
    dirty: RefCell<Vec<(usize, usize)>>,   // host-dirty (offset, len) ranges
}

impl Root {
    fn flush(&self) {
        for (off, len) in self.dirty.borrow_mut().drain(..) {
            if len > 0 { chk(unsafe { thip_h2d(self.dev.add(off), self.host.add(off), len) }); }
        }
    }
}

impl Drop for Root {
    fn drop(&mut self) { chk(unsafe { thip_free(self.dev) }); }
}

pub struct F32HIPSlice {
    root: Rc<Root>,
    off: usize,
    len: usize,
    is_root: bool,
    slot: usize,
}

thread_local! {
    // child slices need stable addresses for the lifetime of their SliceRef/SliceMut wrapper
    static ARENA: RefCell<Vec<Option<Box<F32HIPSlice>>>> = RefCell::new(Vec::new());
}

fn register<'a>(mut s: F32HIPSlice) -> &'a mut F32HIPSlice {
    ARENA.with(|a| {
        let mut a = a.borrow_mut();
        let slot = a.iter().position(|e| e.is_none()).unwrap_or_else(|| { a.push(None); a.len() - 1 });
        s.slot = slot;
        a[slot] = Some(Box::new(s));
        let p: *mut F32HIPSlice = &mut **a[slot].as_mut().unwrap();
        unsafe { &mut *p }
    })
}

thread_local! {
    // read-only mirrors keyed by the host array: `MatBuild::as_op()` is taken twice per G_i by ProbSOCP::problem
    // (socp.rs:450,463) and once per `problem()` call by every builder -- one upload per array, shared while alive
    // (totsu_f32cuda uploads on every new_ref, f32cuda_slice.rs:89-113).  The C++ twin is `MirrorCache`.
    static MIRRORS: RefCell<std::collections::HashMap<(usize, usize), std::rc::Weak<Root>>> = RefCell::new(Default::default());
}

fn new_root<'a>(host: *mut f32, n: usize, mutable: bool) -> &'a mut F32HIPSlice {
    if !mutable {
        let hit = MIRRORS.with(|m| m.borrow().get(&(host as usize, n)).and_then(|w| w.upgrade()));
        if let Some(root) = hit { return register(F32HIPSlice { root, off: 0, len: n, is_root: true, slot: 0 }); }
    }
    let mut dev = std::ptr::null_mut();
    chk(unsafe { thip_alloc(n, &mut dev) });
    if n > 0 { chk(unsafe { thip_h2d(dev, host, n) }); }
    let root = Rc::new(Root { dev, host, n, mutable, dirty: RefCell::new(Vec::new()) });
    if !mutable { MIRRORS.with(|m| { m.borrow_mut().insert((host as usize, n), Rc::downgrade(&root)); }); }
    register(F32HIPSlice { root, off: 0, len: n, is_root: true, slot: 0 })
}

impl F32HIPSlice {
    /// device pointer for read access (pending host writes are uploaded first)
    pub fn get_dev(&self) -> *const f32 { self.root.flush(); unsafe { self.root.dev.add(self.off) } }
    /// device pointer for write access
    pub fn get_dev_mut(&mut self) -> *mut f32 { self.root.flush(); unsafe { self.root.dev.add(self.off) } }
    fn child<'a>(&self, off: usize, len: usize) -> &'a mut F32HIPSlice {
        register(F32HIPSlice { root: self.root.clone(), off, len, is_root: false, slot: 0 })
    }
    fn download(&self) {
        self.root.flush();
        if self.len > 0 {
            chk(unsafe { thip_d2h(self.root.host.add(self.off), self.root.dev.add(self.off), self.len) });
        }
    }
}

impl SliceLike for F32HIPSlice {
    type F = f32;

    fn new_ref(s: &[f32]) -> SliceRef<'_, F32HIPSlice> {
        unsafe { SliceRef::new(new_root(s.as_ptr() as *mut f32, s.len(), false)) }
    }
    fn new_mut(s: &mut [f32]) -> SliceMut<'_, F32HIPSlice> {
        unsafe { SliceMut::new(new_root(s.as_mut_ptr(), s.len(), true)) }
    }
    fn split_ref(&self, mid: usize) -> (SliceRef<'_, F32HIPSlice>, SliceRef<'_, F32HIPSlice>) {
        assert!(mid <= self.len);
        unsafe { (SliceRef::new(self.child(self.off, mid)), SliceRef::new(self.child(self.off + mid, self.len - mid))) }
    }
    fn split_mut(&mut self, mid: usize) -> (SliceMut<'_, F32HIPSlice>, SliceMut<'_, F32HIPSlice>) {
        assert!(mid <= self.len);
        unsafe { (SliceMut::new(self.child(self.off, mid)), SliceMut::new(self.child(self.off + mid, self.len - mid))) }
    }
    fn drop(&self) {
        if self.is_root && self.root.mutable { self.download(); }       // host buffer up to date again
        ARENA.with(|a| { a.borrow_mut()[self.slot] = None; });           // last Rc<Root> frees the device memory
    }
    fn len(&self) -> usize { self.len }
    fn get_ref(&self) -> &[f32] {
        self.download();
        unsafe { std::slice::from_raw_parts(self.root.host.add(self.off), self.len) }
    }
    fn get_mut(&mut self) -> &mut [f32] {
        assert!(self.root.mutable);
        self.download();
        self.root.dirty.borrow_mut().push((self.off, self.len));
        unsafe { std::slice::from_raw_parts_mut(self.root.host.add(self.off), self.len) }
    }
    // get/set of one element without the default implementation's four child slices (slicelike.rs:54-69)
    fn get(&self, idx: usize) -> f32 { let mut v = 0f32; chk(unsafe { thip_get(self.get_dev(), idx, &mut v) }); v }
    fn set(&mut self, idx: usize, val: f32) { chk(unsafe { thip_set(self.get_dev_mut(), idx, val) }); }
}
