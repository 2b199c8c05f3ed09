//! rustronomy-watershed's public interface (v0.4.1) with the transforms running on an AMD MI355X through the
//! C ABI of include/ws_hip.h.  "lib.rs:N" = line N of the reference's src/lib.rs.
//!
//! Same names, signatures and error behaviour as the reference: `TransformBuilder` (lib.rs:908-1047), `BuildErr`
//! (lib.rs:1051-1065), `HookCtx` (lib.rs:844-862), `Watershed<T>` (lib.rs:1206-1238), `WatershedUtils`
//! (lib.rs:1069-1198), `MergingWatershed` / `SegmentingWatershed`.  Deviations, all documented in ws_hip.h:
//! the tie-break where lakes meet is the first coloured neighbour in down, right, left, up order (a legal outcome
//! of lib.rs:249-253); `SegmentingWatershed::transform` returns the labels after the last level instead of
//! panicking (lib.rs:1821); merged-lake ids are the smallest seed colour of the lake.  Out of scope here: the
//! `plots`, `progress`, `debug` and `jemalloc` features.
//!
//! This crate is source only in this repository: its build image has no Rust toolchain.
mod hip_ffi;
mod shim;

use ndarray as nd;
use num_traits::{Num, ToPrimitive};
use std::os::raw::{c_int, c_void};

pub const UNCOLOURED: usize = 0; // lib.rs:138
pub const NORMAL_MAX: u8 = 254; // lib.rs:139
pub const ALWAYS_FILL: u8 = 0; // lib.rs:140
pub const NEVER_FILL: u8 = 255; // lib.rs:141

pub mod prelude {
    pub use crate::{MergingWatershed, SegmentingWatershed, TransformBuilder, Watershed, WatershedUtils};
}

/// lib.rs:844-850
#[derive(Clone)]
pub struct HookCtx<'a> {
    pub water_level: u8,
    pub max_water_level: u8,
    pub image: nd::ArrayView2<'a, u8>,
    pub colours: nd::ArrayView2<'a, usize>,
    pub seeds: &'a [(usize, (usize, usize))],
}

/// lib.rs:1051-1065
#[derive(Debug, Clone)]
pub enum BuildErr {
    MaxToHigh(u8),
    MaxToLow(u8),
}

impl std::error::Error for BuildErr {}
impl std::fmt::Display for BuildErr {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        match self {
            BuildErr::MaxToHigh(v) => {
                write!(f, "Maximum water level set to {v}, which is higher than the maximum allowed value {NORMAL_MAX}")
            }
            // the reference's message names NEVER_FILL here too (lib.rs:1062)
            BuildErr::MaxToLow(v) => {
                write!(f, "Maximum water level set to {v}, which is lower than the minimum allowed value {NEVER_FILL}")
            }
        }
    }
}

/// The options every transform struct carries (the reference's private fields, lib.rs:1297-1311, 1609-1621).
#[derive(Clone, Copy)]
struct Options {
    max_water_level: u8,
    edge_correction: bool,
    seed_shift: bool,
}

impl Options {
    fn ffi(&self) -> hip_ffi::ws_options {
        hip_ffi::ws_options {
            max_water_level: self.max_water_level,
            edge_correction: self.edge_correction as u8,
            engine: 0,   // WS_ENGINE_AUTO
            tie_rule: 0, // WS_TIE_FIRST_DRLU
            seed_shift: self.seed_shift as u8,
            reserved: [0; 3],
        }
    }
    /// shape of every label plane the transform hands out (lib.rs:1640-1666: outputs stay padded)
    fn plane(&self, h: usize, w: usize) -> (usize, usize) {
        let e = if self.edge_correction { 2 } else { 0 };
        (h + e, w + e)
    }
}

/// lib.rs:908-923
#[derive(Clone)]
pub struct TransformBuilder<T = ()> {
    opt: Options,
    wlvl_hook: Option<fn(HookCtx) -> T>,
}

impl Default for TransformBuilder<()> {
    fn default() -> Self {
        TransformBuilder::new()
    }
}

impl<T> TransformBuilder<T> {
    /// lib.rs:936-946: max level 254, no edge correction, no hook
    pub const fn new() -> Self {
        TransformBuilder {
            opt: Options { max_water_level: NORMAL_MAX, edge_correction: false, seed_shift: false },
            wlvl_hook: None,
        }
    }
    /// lib.rs:950
    pub const fn set_max_water_lvl(mut self, max_water_lvl: u8) -> Self {
        self.opt.max_water_level = max_water_lvl;
        self
    }
    /// lib.rs:958.  No padded image copy is made on the device: the ring of zeros is virtual.
    pub const fn enable_edge_correction(mut self) -> Self {
        self.opt.edge_correction = true;
        self
    }
    /// lib.rs:967
    pub const fn set_wlvl_hook(mut self, hook: fn(HookCtx) -> T) -> Self {
        self.wlvl_hook = Some(hook);
        self
    }
    /// Not in the reference (`ws_options.seed_shift`): with edge correction, move every seed by (+1, +1) onto the
    /// pixel it was found at, instead of indexing the padded plane with the caller's coordinates (lib.rs:1675-1677).
    pub const fn shift_seeds_into_padded_plane(mut self) -> Self {
        self.opt.seed_shift = true;
        self
    }

    fn validate(&self) -> Result<(), BuildErr> {
        // lib.rs:999-1004, 1026-1030; ws_options_validate returns the same two errors to non-Rust callers
        if self.opt.max_water_level > NORMAL_MAX {
            Err(BuildErr::MaxToHigh(self.opt.max_water_level))
        } else if self.opt.max_water_level <= ALWAYS_FILL {
            Err(BuildErr::MaxToLow(self.opt.max_water_level))
        } else {
            Ok(())
        }
    }
    /// lib.rs:998-1020
    pub fn build_merging(self) -> Result<MergingWatershed<T>, BuildErr> {
        self.validate()?;
        Ok(MergingWatershed { opt: self.opt, wlvl_hook: self.wlvl_hook })
    }
    /// lib.rs:1024-1046
    pub fn build_segmenting(self) -> Result<SegmentingWatershed<T>, BuildErr> {
        self.validate()?;
        Ok(SegmentingWatershed { opt: self.opt, wlvl_hook: self.wlvl_hook })
    }
}

/// lib.rs:1069-1198
pub trait WatershedUtils {
    /// lib.rs:1081-1087
    fn pre_processor<T, D>(&self, img: nd::ArrayView<T, D>) -> nd::Array<u8, D>
    where
        T: Num + Copy + ToPrimitive + PartialOrd + 'static,
        D: nd::Dimension,
    {
        self.pre_processor_with_max::<NORMAL_MAX, T, D>(img)
    }

    /// lib.rs:1134-1173.  Element types the ABI lists (f32, f64, i32, u16, i16, u8) are quantised on the GPU with
    /// the reference's arithmetic (zero-seeded min / max folds, f64, `is_normal` quirks: NaN, -inf, subnormals and
    /// exact 0 -> NEVER_FILL, +inf -> ALWAYS_FILL); any other `T` goes through f64 on the host first, which is
    /// what the reference's `to_f64()` does per element anyway.
    fn pre_processor_with_max<const MAX: u8, T, D>(&self, img: nd::ArrayView<T, D>) -> nd::Array<u8, D>
    where
        T: Num + Copy + ToPrimitive + PartialOrd + 'static,
        D: nd::Dimension,
    {
        assert!(MAX < NEVER_FILL); // lib.rs:1143
        assert!(MAX > ALWAYS_FILL); // lib.rs:1144
        use std::any::TypeId;
        let t = TypeId::of::<T>();
        let dtype = [
            (TypeId::of::<f32>(), hip_ffi::WS_F32),
            (TypeId::of::<f64>(), hip_ffi::WS_F64),
            (TypeId::of::<i32>(), hip_ffi::WS_I32),
            (TypeId::of::<u16>(), hip_ffi::WS_U16),
            (TypeId::of::<i16>(), hip_ffi::WS_I16),
            (TypeId::of::<u8>(), hip_ffi::WS_U8),
        ]
        .iter()
        .find(|(id, _)| *id == t)
        .map(|&(_, d)| d);
        let mut out = nd::Array::<u8, D>::zeros(img.raw_dim());
        let n = img.len();
        let std_img = img.as_standard_layout();
        let out_ptr = out.as_slice_mut().expect("fresh array is contiguous").as_mut_ptr();
        shim::with_ctx(|ctx| unsafe {
            let rc = match dtype {
                Some(d) => hip_ffi::ws_pre_processor(ctx, std_img.as_ptr() as *const c_void, d, n, MAX, out_ptr),
                None => {
                    let wide: Vec<f64> = std_img.iter().map(|x| x.to_f64().unwrap()).collect();
                    hip_ffi::ws_pre_processor(ctx, wide.as_ptr() as *const c_void, hip_ffi::WS_F64, n, MAX, out_ptr)
                }
            };
            shim::check(ctx, rc, "ws_pre_processor");
        });
        out
    }

    /// lib.rs:1178-1197: strict 8-neighbour local MAXIMA of the interior (as the reference's code does, whatever
    /// its name says), in row-major order.
    fn find_local_minima(&self, img: nd::ArrayView2<u8>) -> Vec<(usize, usize)> {
        let (h, w) = img.dim();
        if h < 3 || w < 3 {
            return Vec::new(); // no 3x3 window (lib.rs:1183)
        }
        let (std_img, stride) = shim::standard(&img);
        // at most one strict maximum per 2x2 block
        let cap = ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1);
        let mut rc_pairs = vec![0u64; 2 * cap];
        let mut n = 0usize;
        shim::with_ctx(|ctx| unsafe {
            let rc = hip_ffi::ws_find_local_minima(ctx, std_img.as_ptr(), h, w, stride, rc_pairs.as_mut_ptr(), cap, &mut n);
            shim::check(ctx, rc, "ws_find_local_minima");
        });
        rc_pairs[..2 * n].chunks_exact(2).map(|p| (p[0] as usize, p[1] as usize)).collect()
    }
}

impl<T> WatershedUtils for MergingWatershed<T> {}
impl<T> WatershedUtils for SegmentingWatershed<T> {}

/// lib.rs:1206-1238
pub trait Watershed<T = ()> {
    fn transform(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> nd::Array2<usize>;
    fn transform_with_hook(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<T>;
    fn transform_to_list(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, Vec<usize>)>;
    fn transform_history(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, nd::Array2<usize>)>;
}

/// lib.rs:1297-1311
pub struct MergingWatershed<T = ()> {
    opt: Options,
    wlvl_hook: Option<fn(HookCtx) -> T>,
}

/// lib.rs:1609-1621
pub struct SegmentingWatershed<T = ()> {
    opt: Options,
    wlvl_hook: Option<fn(HookCtx) -> T>,
}

// ---- the three ABI call shapes both transforms share ------------------------------------------------------------

/// `ws_segment_with_hook` / `ws_merge_with_hook`: lib.rs:1638-1808 / 1328-1522.  The whole transform runs even
/// without a hook (the reference does the same and returns an empty Vec, lib.rs:1796-1807).
fn run_with_hook<U>(merging: bool, opt: &Options, hook: Option<fn(HookCtx) -> U>, input: nd::ArrayView2<u8>,
                    seeds: &[(usize, usize)], final_labels: Option<&mut nd::Array2<usize>>) -> Vec<U> {
    let (h, w) = input.dim();
    let (std_img, stride) = shim::standard(&input);
    let packed = shim::pack_seeds(seeds);
    let colours = shim::seed_colours(seeds);
    let o = opt.ffi();
    let out_ptr = match final_labels {
        Some(a) => {
            assert_eq!(a.dim(), opt.plane(h, w));
            a.as_slice_mut().expect("standard layout").as_mut_ptr() as *mut u64
        }
        None => std::ptr::null_mut(),
    };
    match hook {
        Some(f) => {
            let mut st = shim::HookState { hook: f, seeds: &colours, results: Vec::with_capacity(opt.max_water_level as usize + 1) };
            let user = &mut st as *mut _ as *mut c_void;
            let cb: hip_ffi::ws_level_cb = Some(shim::trampoline::<U>);
            shim::with_ctx(|ctx| unsafe {
                let rc = if merging {
                    hip_ffi::ws_merge_with_hook(ctx, std_img.as_ptr(), h, w, stride, packed.as_ptr(), seeds.len(), &o, cb, user, out_ptr)
                } else {
                    hip_ffi::ws_segment_with_hook(ctx, std_img.as_ptr(), h, w, stride, packed.as_ptr(), seeds.len(), &o, cb, user, out_ptr)
                };
                shim::check(ctx, rc, "transform_with_hook");
            });
            st.results
        }
        None => {
            let user = std::ptr::null_mut();
            shim::with_ctx(|ctx| unsafe {
                let rc = if merging {
                    hip_ffi::ws_merge_with_hook(ctx, std_img.as_ptr(), h, w, stride, packed.as_ptr(), seeds.len(), &o, None, user, out_ptr)
                } else {
                    hip_ffi::ws_segment_with_hook(ctx, std_img.as_ptr(), h, w, stride, packed.as_ptr(), seeds.len(), &o, None, user, out_ptr)
                };
                shim::check(ctx, rc, "transform_with_hook");
            });
            Vec::new()
        }
    }
}

/// `ws_transform_to_list` + the dense expansion the reference's return type asks for: per level a `Vec<usize>` of
/// length pixels + 1 with `v[c]` = area of lake `c`, `v[0]` = uncoloured pixels (lib.rs:628-635, 1551-1561,
/// 1837-1847).  The engine returns (colour, area) records; the records of a level are sorted by colour.
fn run_to_list(merging: bool, opt: &Options, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, Vec<usize>)> {
    let (h, w) = input.dim();
    let (ph, pw) = opt.plane(h, w);
    let (std_img, stride) = shim::standard(&input);
    let packed = shim::pack_seeds(seeds);
    let o = opt.ffi();
    let levels = opt.max_water_level as usize + 1;
    let mut offsets = vec![0u64; levels + 1];
    let mut uncoloured = vec![0u64; levels];
    // one record per live lake and level; half of seeds x levels covers a random field.  A guess that is too small
    // costs a second transform (WS_ERR_CAPACITY reports the exact count), never a wrong answer.
    let mut cap = (seeds.len().max(1) * levels / 2 + 1024).min(1 << 26);
    let mut lakes: Vec<hip_ffi::ws_lake>;
    let mut n = 0usize;
    loop {
        lakes = vec![hip_ffi::ws_lake::default(); cap];
        let rc = shim::with_ctx(|ctx| unsafe {
            let rc = hip_ffi::ws_transform_to_list(ctx, merging as c_int, std_img.as_ptr(), h, w, stride, packed.as_ptr(),
                                                   seeds.len(), &o, lakes.as_mut_ptr(), cap, &mut n, offsets.as_mut_ptr(),
                                                   uncoloured.as_mut_ptr());
            if rc != hip_ffi::WS_ERR_CAPACITY {
                shim::check(ctx, rc, "ws_transform_to_list");
            }
            rc
        });
        if rc == hip_ffi::WS_ERR_CAPACITY && n > cap {
            cap = n;
            continue;
        }
        break;
    }
    (0..levels)
        .map(|l| {
            let mut sizes = vec![0usize; ph * pw + 1];
            sizes[UNCOLOURED] = uncoloured[l] as usize;
            for rec in &lakes[offsets[l] as usize..offsets[l + 1] as usize] {
                sizes[rec.colour as usize] = rec.area as usize;
            }
            (l as u8, sizes)
        })
        .collect()
}

fn history_hook(ctx: HookCtx) -> (u8, nd::Array2<usize>) {
    (ctx.water_level, ctx.colours.to_owned()) // lib.rs:1545, 1831
}

impl<T> Watershed<T> for SegmentingWatershed<T> {
    /// lib.rs:1810-1822 with the intended semantics: the labels after the last water level (`ws_segment`).
    fn transform(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> nd::Array2<usize> {
        let (h, w) = input.dim();
        let (std_img, stride) = shim::standard(&input);
        let packed = shim::pack_seeds(seeds);
        let o = self.opt.ffi();
        let mut out = nd::Array2::<usize>::zeros(self.opt.plane(h, w));
        shim::with_ctx(|ctx| unsafe {
            let rc = hip_ffi::ws_segment(ctx, std_img.as_ptr(), h, w, stride, packed.as_ptr(), seeds.len(), &o,
                                         out.as_mut_ptr() as *mut u64);
            shim::check(ctx, rc, "ws_segment");
        });
        out
    }
    /// lib.rs:1638-1808
    fn transform_with_hook(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<T> {
        run_with_hook(false, &self.opt, self.wlvl_hook, input, seeds, None)
    }
    /// lib.rs:1837-1847
    fn transform_to_list(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, Vec<usize>)> {
        run_to_list(false, &self.opt, input, seeds)
    }
    /// lib.rs:1824-1835
    fn transform_history(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, nd::Array2<usize>)> {
        run_with_hook(false, &self.opt, Some(history_hook as fn(HookCtx) -> _), input, seeds, None)
    }
}

impl<T> SegmentingWatershed<T> {
    /// Not in the reference: its README's call pair (lib.rs:73-86), `transform(input, &find_local_minima(input))`, as ONE
    /// call of the library (`ws_segment_minima`): the seeds never cross PCIe as (usize, usize) pairs unless asked for.
    /// Returns the labels and, with `want_seeds`, the seed list the labels are numbered by (colour i + 1 = entry i).
    pub fn transform_from_minima(&self, input: nd::ArrayView2<u8>, want_seeds: bool) -> (nd::Array2<usize>, Vec<(usize, usize)>) {
        let (h, w) = input.dim();
        let (std_img, stride) = shim::standard(&input);
        let o = self.opt.ffi();
        let mut out = nd::Array2::<usize>::zeros(self.opt.plane(h, w));
        // at most one strict maximum per 2 x 2 block of the interior
        let cap = if want_seeds && h >= 3 && w >= 3 { ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1) } else { 0 };
        let mut packed = vec![0u64; 2 * cap];
        let mut n = 0usize;
        shim::with_ctx(|ctx| unsafe {
            let rc = hip_ffi::ws_segment_minima(ctx, std_img.as_ptr(), h, w, stride, &o, out.as_mut_ptr() as *mut u64,
                                                if cap != 0 { packed.as_mut_ptr() } else { std::ptr::null_mut() }, cap, &mut n);
            shim::check(ctx, rc, "ws_segment_minima");
        });
        let seeds = if want_seeds { packed.chunks_exact(2).take(n).map(|p| (p[0] as usize, p[1] as usize)).collect() } else { Vec::new() };
        (out, seeds)
    }

    /// Not in the reference: what its integration tests loop over (tests/integration.rs:267,356 -- `find_local_minima` +
    /// `transform` for one slice of a cube after the other) as ONE call of the library (`ws_segment_batch`): the slices take
    /// turns on internal contexts, so one slice's upload, another's transform and a third's label copy overlap.  Slice k of
    /// the result is exactly `transform(cube[k], &find_local_minima(cube[k]))`.
    pub fn transform_cube(&self, cube: nd::ArrayView3<u8>) -> nd::Array3<usize> {
        let (n, h, w) = cube.dim();
        let std_cube = cube.as_standard_layout();      // contiguous (slice, row, column); a copy only if the view is strided
        let o = self.opt.ffi();
        let (ph, pw) = self.opt.plane(h, w);
        let mut out = nd::Array3::<usize>::zeros((n, ph, pw));
        let mut failed = 0usize;
        shim::with_ctx(|ctx| unsafe {
            let rc = hip_ffi::ws_segment_batch(ctx, std_cube.as_ptr(), n, h, w, w, h * w, std::ptr::null(), std::ptr::null(), &o,
                                               out.as_mut_ptr() as *mut u64, std::ptr::null_mut(), &mut failed);
            shim::check(ctx, rc, "ws_segment_batch");
        });
        out
    }
}

impl<T> Watershed<T> for MergingWatershed<T> {
    /// lib.rs:1524-1536: a stub in the reference (zeros, interior 123, seeds ignored); kept for drop-in fidelity.
    fn transform(&self, input: nd::ArrayView2<u8>, _seeds: &[(usize, usize)]) -> nd::Array2<usize> {
        let (h, w) = input.dim();
        let mut out = nd::Array2::<usize>::zeros((h, w));
        let rc = unsafe { hip_ffi::ws_merge_transform_stub(h, w, out.as_mut_ptr() as *mut u64) };
        assert_eq!(rc, hip_ffi::WS_OK);
        out
    }
    /// lib.rs:1328-1522
    fn transform_with_hook(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<T> {
        run_with_hook(true, &self.opt, self.wlvl_hook, input, seeds, None)
    }
    /// lib.rs:1551-1561
    fn transform_to_list(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, Vec<usize>)> {
        run_to_list(true, &self.opt, input, seeds)
    }
    /// lib.rs:1538-1549
    fn transform_history(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> Vec<(u8, nd::Array2<usize>)> {
        run_with_hook(true, &self.opt, Some(history_hook as fn(HookCtx) -> _), input, seeds, None)
    }
}

impl<T> MergingWatershed<T> {
    /// Not in the reference: the merged label plane after the last level (canonical ids: the smallest seed colour of
    /// every lake).  The reference's own `transform` is the stub above.
    pub fn transform_final(&self, input: nd::ArrayView2<u8>, seeds: &[(usize, usize)]) -> nd::Array2<usize> {
        let (h, w) = input.dim();
        let mut out = nd::Array2::<usize>::zeros(self.opt.plane(h, w));
        run_with_hook::<()>(true, &self.opt, None, input, seeds, Some(&mut out));
        out
    }
}

#[cfg(test)]
mod tests {
    // needs an MI355X and libws_hip.so: `WS_HIP_LIB_DIR=.. cargo test`; mirrors README.md:55-70 of the reference
    use super::prelude::*;
    use ndarray as nd;
    use ndarray_rand::{rand_distr::Uniform, RandomExt};

    #[test]
    fn quickstart() {
        let rf = nd::Array2::<u8>::random((512, 512), Uniform::new(0, 254));
        let watershed = TransformBuilder::default().build_segmenting().unwrap();
        let mins = watershed.find_local_minima(rf.view());
        let output = watershed.transform(rf.view(), &mins);
        assert_eq!(output.dim(), (512, 512));
        for (i, &(r, c)) in mins.iter().enumerate() {
            assert_eq!(output[(r, c)], i + 1);
        }
        let lists = TransformBuilder::default().build_merging().unwrap().transform_to_list(rf.view(), &mins);
        assert_eq!(lists.len(), 255);
        assert!(lists.iter().all(|(_, v)| v.len() == 512 * 512 + 1 && v.iter().sum::<usize>() == 512 * 512));
    }

    #[test]
    fn builder_errors() {
        assert!(matches!(TransformBuilder::default().set_max_water_lvl(255).build_segmenting(), Err(crate::BuildErr::MaxToHigh(255))));
        assert!(matches!(TransformBuilder::default().set_max_water_lvl(0).build_merging(), Err(crate::BuildErr::MaxToLow(0))));
    }
}
