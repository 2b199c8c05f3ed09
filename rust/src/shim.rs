//! Plumbing between the public types and the C ABI: a per-thread engine context, seed repacking, the hook
//! trampoline and the status -> panic / BuildErr mapping.
use crate::hip_ffi as ffi;
use crate::HookCtx;
use ndarray as nd;
use std::ffi::CStr;
use std::os::raw::{c_int, c_void};

/// One `ws_ctx` per host thread: a context is single-threaded, while the transform structs are plain data and
/// therefore `Send + Sync` like the reference's (lib.rs:65-67).  Created on first use on HIP device
/// `WS_HIP_DEVICE` (default 0).
pub(crate) struct HipCtx(pub *mut ffi::ws_ctx);

impl HipCtx {
    fn new() -> Self {
        let device = std::env::var("WS_HIP_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        unsafe {
            assert_eq!(ffi::ws_abi_version(), ffi::WS_ABI_VERSION, "libws_hip.so speaks another ABI version");
            let mut p = std::ptr::null_mut();
            let rc = ffi::ws_ctx_create(device, &mut p);
            if rc != ffi::WS_OK {
                // there is no CPU fallback: without a HIP device the transform cannot run
                panic!("ws_ctx_create(device {device}) failed: {}", strerror(rc));
            }
            HipCtx(p)
        }
    }
}

impl Drop for HipCtx {
    fn drop(&mut self) {
        unsafe { ffi::ws_ctx_destroy(self.0) }
    }
}

thread_local! {
    static CTX: HipCtx = HipCtx::new();
}

pub(crate) fn with_ctx<R>(f: impl FnOnce(*mut ffi::ws_ctx) -> R) -> R {
    CTX.with(|c| f(c.0))
}

pub(crate) fn strerror(rc: c_int) -> String {
    unsafe { CStr::from_ptr(ffi::ws_strerror(rc)).to_string_lossy().into_owned() }
}

/// Status of a transform call -> what the reference does in the same situation.
pub(crate) fn check(ctx: *mut ffi::ws_ctx, rc: c_int, what: &str) {
    if rc == ffi::WS_OK {
        return;
    }
    let detail = unsafe { CStr::from_ptr(ffi::ws_last_error(ctx)).to_string_lossy().into_owned() };
    if rc == ffi::WS_ERR_SEED_OOB {
        // lib.rs:1366 / 1676: `output[*seed_idx] = ..` panics with ndarray's index error
        panic!("ndarray: index out of bounds ({detail})");
    }
    panic!("{what}: {} ({rc}): {detail}", strerror(rc));
}

/// `&[(usize, usize)]` -> row, col, row, col ... as u64 (the layout of a Rust tuple is unspecified).
pub(crate) fn pack_seeds(seeds: &[(usize, usize)]) -> Vec<u64> {
    let mut v = Vec::with_capacity(seeds.len() * 2);
    for &(r, c) in seeds {
        v.push(r as u64);
        v.push(c as u64);
    }
    v
}

/// The `seeds` slice of HookCtx: (colour, (row, col)) with colour = index + 1 (lib.rs:1671-1672).
pub(crate) fn seed_colours(seeds: &[(usize, usize)]) -> Vec<(usize, (usize, usize))> {
    seeds.iter().enumerate().map(|(i, &s)| (i + 1, s)).collect()
}

/// What the trampoline needs: the user's hook, the seed list in HookCtx form, the results so far.
pub(crate) struct HookState<'a, T> {
    pub hook: fn(HookCtx) -> T,
    pub seeds: &'a [(usize, (usize, usize))],
    pub results: Vec<T>,
}

/// `ws_level_cb`: builds the HookCtx of lib.rs:1796-1804 / 1510-1518 around the engine's host planes (valid only
/// during the call, like the reference's views) and stores the hook's result.
pub(crate) unsafe extern "C" fn trampoline<T>(
    user: *mut c_void,
    water_level: u8,
    max_water_level: u8,
    image: *const u8,
    labels: *const u64,
    h: usize,
    w: usize,
) {
    const _: () = assert!(std::mem::size_of::<usize>() == 8, "labels cross the ABI as u64 == usize");
    let st = &mut *(user as *mut HookState<T>);
    let image = nd::ArrayView2::from_shape_ptr((h, w), image);
    let colours = nd::ArrayView2::from_shape_ptr((h, w), labels as *const usize);
    let ctx = HookCtx { water_level, max_water_level, image, colours, seeds: st.seeds };
    st.results.push((st.hook)(ctx));
}

/// A C-contiguous copy only when the view is not already one (unit column stride, non-negative row stride >= w).
pub(crate) fn standard<'a>(input: &'a nd::ArrayView2<'a, u8>) -> (nd::CowArray<'a, u8, nd::Ix2>, usize) {
    let (h, w) = input.dim();
    let s = input.strides();
    let ok = (w <= 1 || s[1] == 1) && (h <= 1 || s[0] >= w as isize);
    if ok {
        let stride = if h > 1 { s[0] as usize } else { w.max(1) };
        (nd::CowArray::from(input.view()), stride)
    } else {
        (nd::CowArray::from(input.as_standard_layout().into_owned()), w.max(1))
    }
}
