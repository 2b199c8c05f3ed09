//! extern "C" declarations of include/ws_hip.h (ABI version 3), one to one.
//! tests/test_abi_cpu.py checks that no function of the header is missing here.
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const WS_ABI_VERSION: c_int = 3;

#[repr(C)]
pub struct ws_ctx {
    _private: [u8; 0],
}

/// TransformBuilder's runtime options as plain data (8 bytes).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ws_options {
    pub max_water_level: u8,
    pub edge_correction: u8,
    pub engine: u8,
    pub tie_rule: u8,
    /// 0: seeds index the padded plane with the caller's coordinates (reference behaviour); 1: moved by (+1, +1)
    pub seed_shift: u8,
    pub reserved: [u8; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ws_lake {
    pub colour: u64,
    pub area: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct ws_stats {
    pub relax_passes: u32,
    pub resolve_passes: u32,
    pub sweep_steps: u32,
    pub merge_levels: u32,
    pub tiles_run_relax: u64,
    pub tiles_run_resolve: u64,
    pub ms_relax: f32,
    pub ms_resolve: f32,
    pub ms_sweep: f32,
    pub ms_other: f32,
    pub ms_total: f32,
    pub launches_relax: u32,
    pub launches_resolve: u32,
    pub launches_sweep: u32,
    pub relax_tile_iterations: u32,
    pub graph_launches: u32,
}

/// A set of ranks (one context and one device each) that transform one field, or one batch, together.
#[repr(C)]
pub struct ws_group {
    _private: [u8; 0],
}

/// One rank's row block of a tiled field, device resident (ws_segment_tiled_device).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ws_tile_block {
    pub d_img: *const u8,
    pub d_seeds_rc: *const u32,
    pub d_colours: *const u32,
    pub n_seeds: usize,
    pub first_colour: u32,
    pub reserved: u32,
    pub d_labels: *mut u32,
}

/// One rank's tile of a field cut in both directions, device resident (ws_segment_tiled2d_device).
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ws_tile_block2d {
    pub d_img: *const u8,
    pub img_stride: usize,
    pub d_seeds_rc: *const u32,
    pub d_colours: *const u32,
    pub n_seeds: usize,
    pub d_labels: *mut u32,
}

/// One rank's slices of a batch, device resident (ws_segment_batch_group); seed_offsets lives on the host.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct ws_batch_part {
    pub d_cube: *const u8,
    pub d_seeds_rc: *const u32,
    pub seed_offsets: *const usize,
    pub n_slices: usize,
    pub d_labels: *mut u32,
}

pub type ws_level_cb = Option<
    unsafe extern "C" fn(
        user: *mut c_void,
        water_level: u8,
        max_water_level: u8,
        image: *const u8,
        labels: *const u64,
        h: usize,
        w: usize,
    ),
>;

pub const WS_OK: c_int = 0;
pub const WS_ERR_BAD_ARG: c_int = -1;
pub const WS_ERR_MAX_TOO_HIGH: c_int = -2;
pub const WS_ERR_MAX_TOO_LOW: c_int = -3;
pub const WS_ERR_SEED_OOB: c_int = -4;
pub const WS_ERR_HIP: c_int = -5;
pub const WS_ERR_OOM: c_int = -6;
pub const WS_ERR_NO_DEVICE: c_int = -7;
pub const WS_ERR_CAPACITY: c_int = -8;
pub const WS_ERR_RING_OVERFLOW: c_int = -9;
pub const WS_ERR_TOO_LARGE: c_int = -10;
pub const WS_ERR_UNSUPPORTED: c_int = -11;
pub const WS_ERR_RCCL: c_int = -12;
pub const WS_RCCL_ID_BYTES: usize = 128;

/// ws_dtype
pub const WS_F32: c_int = 0;
pub const WS_F64: c_int = 1;
pub const WS_I32: c_int = 2;
pub const WS_U16: c_int = 3;
pub const WS_I16: c_int = 4;
pub const WS_U8: c_int = 5;

extern "C" {
    // ---- context
    pub fn ws_abi_version() -> c_int;
    pub fn ws_strerror(status: c_int) -> *const c_char;
    pub fn ws_ctx_create(device: c_int, out: *mut *mut ws_ctx) -> c_int;
    pub fn ws_ctx_create_on_stream(device: c_int, hip_stream: *mut c_void, out: *mut *mut ws_ctx) -> c_int;
    pub fn ws_ctx_destroy(ctx: *mut ws_ctx);
    pub fn ws_last_error(ctx: *const ws_ctx) -> *const c_char;
    pub fn ws_ctx_set_profiling(ctx: *mut ws_ctx, enabled: c_int) -> c_int;
    pub fn ws_ctx_get_stats(ctx: *const ws_ctx, out: *mut ws_stats) -> c_int;
    pub fn ws_ctx_synchronize(ctx: *mut ws_ctx) -> c_int;
    pub fn ws_ctx_set_batch_pixel_limit(ctx: *mut ws_ctx, max_px: usize) -> c_int;
    pub fn ws_ctx_set_seam_repair_min_pixels(ctx: *mut ws_ctx, min_px: usize) -> c_int;
    pub fn ws_ctx_set_live_list_min_colours(ctx: *mut ws_ctx, min_colours: usize) -> c_int;
    pub fn ws_ctx_set_persistent_pass(ctx: *mut ws_ctx, mode: c_int) -> c_int;
    pub fn ws_ctx_set_host_threads(ctx: *mut ws_ctx, n_threads: c_int) -> c_int;
    pub fn ws_options_default(out: *mut ws_options) -> c_int;
    pub fn ws_options_validate(opt: *const ws_options) -> c_int;

    // ---- host-buffer entry points: what the shim binds
    pub fn ws_find_local_minima(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        out_rc: *mut u64, cap: usize, n_found: *mut usize) -> c_int;
    pub fn ws_segment(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        seeds_rc: *const u64, n_seeds: usize, opt: *const ws_options, out_labels: *mut u64) -> c_int;
    pub fn ws_segment_u32(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        seeds_rc: *const u64, n_seeds: usize, opt: *const ws_options, out_labels: *mut u32) -> c_int;
    pub fn ws_segment_minima(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        opt: *const ws_options, out_labels: *mut u64, seeds_rc: *mut u64, cap: usize, n_seeds: *mut usize) -> c_int;
    pub fn ws_segment_minima_u32(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        opt: *const ws_options, out_labels: *mut u32, seeds_rc: *mut u64, cap: usize, n_seeds: *mut usize) -> c_int;
    pub fn ws_segment_minima_device(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        opt: *const ws_options, d_labels: *mut u32, d_seeds_rc: *mut u32, cap: usize, n_seeds: *mut usize) -> c_int;
    pub fn ws_segment_with_hook(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        seeds_rc: *const u64, n_seeds: usize, opt: *const ws_options, cb: ws_level_cb, user: *mut c_void,
        out_labels: *mut u64) -> c_int;
    pub fn ws_merge_with_hook(ctx: *mut ws_ctx, img: *const u8, h: usize, w: usize, row_stride: usize,
        seeds_rc: *const u64, n_seeds: usize, opt: *const ws_options, cb: ws_level_cb, user: *mut c_void,
        out_labels: *mut u64) -> c_int;
    pub fn ws_transform_to_list(ctx: *mut ws_ctx, merging: c_int, img: *const u8, h: usize, w: usize,
        row_stride: usize, seeds_rc: *const u64, n_seeds: usize, opt: *const ws_options, lakes: *mut ws_lake,
        cap: usize, n_lakes: *mut usize, offsets: *mut u64, uncoloured: *mut u64) -> c_int;
    pub fn ws_lists_from_arrival_device(ctx: *mut ws_ctx, merging: c_int, d_keys: *const u32, d_seg_labels: *const u32, h: usize, w: usize, n_seeds: usize,
                                        opt: *const ws_options, d_lakes: *mut ws_lake, cap: usize, n_lakes: *mut usize, offsets: *mut u64, uncoloured: *mut u64) -> c_int;
    pub fn ws_merge_transform_stub(h: usize, w: usize, out_labels: *mut u64) -> c_int;
    pub fn ws_segment_batch(ctx: *mut ws_ctx, cube: *const u8, n_slices: usize, h: usize, w: usize, row_stride: usize, slice_stride: usize,
                            seeds_rc: *const u64, seed_offsets: *const usize, opt: *const ws_options, out_labels: *mut u64,
                            n_seeds: *mut usize, failed_slice: *mut usize) -> c_int;
    pub fn ws_segment_batch_host(g: *mut ws_group, cube: *const u8, n_slices: usize, h: usize, w: usize, row_stride: usize, slice_stride: usize,
                                 seeds_rc: *const u64, seed_offsets: *const usize, opt: *const ws_options, out_labels: *mut u64,
                                 n_seeds: *mut usize, failed_slice: *mut usize) -> c_int;
    pub fn ws_pre_processor(ctx: *mut ws_ctx, data: *const c_void, dtype: c_int, n_elems: usize, max_value: u8,
        out: *mut u8) -> c_int;

    // ---- device-resident pipelines: u8 images, u32 (row, col) seed pairs and u32 labels stay in HBM
    pub fn ws_find_local_minima_device(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        d_out_rc: *mut u32, cap: usize, n_found: *mut usize) -> c_int;
    pub fn ws_segment_device(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        d_seeds_rc: *const u32, n_seeds: usize, opt: *const ws_options, d_labels: *mut u32) -> c_int;
    pub fn ws_segment_device_begin(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        d_seeds_rc: *const u32, n_seeds: usize, opt: *const ws_options, d_labels: *mut u32) -> c_int;
    pub fn ws_segment_device_end(ctx: *mut ws_ctx) -> c_int;
    pub fn ws_segment_batch_device(ctx: *mut ws_ctx, d_cube: *const u8, n_slices: usize, h: usize, w: usize,
        row_stride: usize, slice_stride: usize, d_seeds_rc: *const u32, seed_offsets: *const usize,
        opt: *const ws_options, d_labels: *mut u32, failed_slice: *mut usize) -> c_int;
    pub fn ws_merge_device(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        d_seeds_rc: *const u32, n_seeds: usize, opt: *const ws_options, d_labels: *mut u32) -> c_int;
    pub fn ws_merge_device_begin(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        d_seeds_rc: *const u32, n_seeds: usize, opt: *const ws_options, d_labels: *mut u32) -> c_int;
    pub fn ws_merge_device_end(ctx: *mut ws_ctx) -> c_int;
    pub fn ws_transform_to_list_device(ctx: *mut ws_ctx, merging: c_int, d_img: *const u8, h: usize, w: usize,
        row_stride: usize, d_seeds_rc: *const u32, n_seeds: usize, opt: *const ws_options, d_lakes: *mut ws_lake,
        cap: usize, n_lakes: *mut usize, offsets: *mut u64, uncoloured: *mut u64) -> c_int;
    pub fn ws_last_arrival_device(ctx: *mut ws_ctx, d_keys: *mut *const u32, h: *mut usize, w: *mut usize) -> c_int;
    pub fn ws_copy_last_arrival_device(ctx: *mut ws_ctx, d_dst: *mut u32, n_elems: usize) -> c_int;
    pub fn ws_level_snapshot_device(ctx: *mut ws_ctx, d_labels: *const u32, water_level: u8, d_out: *mut u32) -> c_int;
    pub fn ws_pre_processor_device(ctx: *mut ws_ctx, d_data: *const c_void, dtype: c_int, n_elems: usize,
        max_value: u8, d_out: *mut u8) -> c_int;
    pub fn ws_random_field_device(ctx: *mut ws_ctx, d_img: *mut u8, h: usize, w: usize, row_stride: usize,
        seed: u64) -> c_int;

    // ---- one field tiled over several GPUs (row blocks with halo rows; the caller exchanges the halos)
    pub fn ws_block_init(ctx: *mut ws_ctx, h: usize, w: usize, d_seeds_rc: *const u32, d_colours: *const u32,
        n_seeds: usize, d_keys: *mut u32, d_labels: *mut u32) -> c_int;
    pub fn ws_block_relax(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        max_water_level: u8, d_keys: *mut u32, changed: *mut c_int) -> c_int;
    pub fn ws_block_resolve(ctx: *mut ws_ctx, d_keys: *const u32, d_labels: *mut u32, h: usize, w: usize,
        changed: *mut c_int) -> c_int;
    pub fn ws_block_resolve_ring(ctx: *mut ws_ctx, d_keys: *const u32, d_labels: *mut u32, h: usize, w: usize) -> c_int;
    // ... fast form for strictly increasing seed lists: one table exchange for the labels instead of rounds
    pub fn ws_block_begin(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize, max_water_level: u8,
        d_seeds_rc: *const u32, n_seeds: usize, first_colour: u32, d_keys: *mut u32) -> c_int;
    pub fn ws_block_relax_halo(ctx: *mut ws_ctx, d_img: *const u8, h: usize, w: usize, row_stride: usize,
        max_water_level: u8, halo_top: c_int, halo_bottom: c_int, d_keys: *mut u32) -> c_int;
    pub fn ws_block_resolve_local(ctx: *mut ws_ctx, d_keys: *const u32, d_labels: *mut u32, h: usize, w: usize,
        halo_top: c_int, halo_bottom: c_int) -> c_int;
    pub fn ws_block_export_boundary(ctx: *mut ws_ctx, d_labels: *const u32, h: usize, w: usize, halo_top: c_int,
        halo_bottom: c_int, rank: usize, d_rows: *mut u32) -> c_int;
    pub fn ws_block_import_boundary(ctx: *mut ws_ctx, d_table: *const u32, world: usize, rank: usize, d_labels: *mut u32,
        h: usize, w: usize, halo_top: c_int, halo_bottom: c_int) -> c_int;
    // ... and the merging transform's final labels across the blocks (one all-gather of boundary (colour, root) pairs)
    pub fn ws_block_merge_local(ctx: *mut ws_ctx, d_labels: *const u32, h: usize, w: usize, row0: usize, field_rows: usize,
        n_colours_total: usize, d_parent: *mut u32) -> c_int;
    pub fn ws_block_merge_export(ctx: *mut ws_ctx, d_labels: *const u32, h: usize, w: usize, d_parent: *mut u32,
        d_pairs: *mut u32) -> c_int;
    pub fn ws_block_merge_import(ctx: *mut ws_ctx, d_pairs: *const u32, n_pairs: usize, d_parent: *mut u32) -> c_int;
    pub fn ws_block_merge_relabel(ctx: *mut ws_ctx, d_labels: *const u32, n: usize, d_parent: *mut u32,
        n_colours_total: usize, d_out: *mut u32) -> c_int;

    // ---- several GPUs driven inside the library: local groups (all ranks in this process) and RCCL groups (one rank per process)
    pub fn ws_group_create_local(n_ranks: c_int, devices: *const c_int, out: *mut *mut ws_group) -> c_int;
    pub fn ws_group_rccl_unique_id(id: *mut c_void) -> c_int;
    pub fn ws_group_create_rccl(device: c_int, rank: c_int, world: c_int, id: *const c_void, out: *mut *mut ws_group) -> c_int;
    pub fn ws_group_destroy(g: *mut ws_group);
    pub fn ws_group_info(g: *const ws_group, world: *mut c_int, n_local: *mut c_int, first_local: *mut c_int) -> c_int;
    pub fn ws_group_last_error(g: *const ws_group) -> *const c_char;
    pub fn ws_group_selftest(g: *mut ws_group) -> c_int;
    pub fn ws_tile_rows(h: usize, rank: c_int, world: c_int, r0: *mut usize, r1: *mut usize, lo: *mut usize, hi: *mut usize) -> c_int;
    pub fn ws_segment_tiled(g: *mut ws_group, img: *const u8, h: usize, w: usize, row_stride: usize, seeds_rc: *const u64,
        n_seeds: usize, opt: *const ws_options, merging: c_int, out_labels: *mut u64, exchange_rounds: *mut u32) -> c_int;
    pub fn ws_segment_tiled_device(g: *mut ws_group, field_h: usize, w: usize, n_seeds_total: usize, blocks: *const ws_tile_block,
        opt: *const ws_options, merging: c_int, exchange_rounds: *mut u32) -> c_int;
    pub fn ws_transform_to_list_tiled(g: *mut ws_group, merging: c_int, img: *const u8, h: usize, w: usize, row_stride: usize, seeds_rc: *const u64, n_seeds: usize,
                                      opt: *const ws_options, lakes: *mut ws_lake, cap: usize, n_lakes: *mut usize, offsets: *mut u64, uncoloured: *mut u64,
                                      exchange_rounds: *mut u32) -> c_int;
    pub fn ws_transform_to_list_tiled_device(g: *mut ws_group, field_h: usize, w: usize, n_seeds_total: usize, blocks: *const ws_tile_block, opt: *const ws_options,
                                             merging: c_int, d_lakes: *mut ws_lake, cap: usize, n_lakes: *mut usize, offsets: *mut u64, uncoloured: *mut u64,
                                             exchange_rounds: *mut u32) -> c_int;
    pub fn ws_tile_grid(h: usize, w: usize, rank: c_int, py: c_int, px: c_int, rows: *mut usize, cols: *mut usize) -> c_int;
    pub fn ws_segment_tiled2d_device(g: *mut ws_group, field_h: usize, field_w: usize, py: c_int, px: c_int, n_seeds_total: usize,
        blocks: *const ws_tile_block2d, opt: *const ws_options, merging: c_int, exchange_rounds: *mut u32) -> c_int;
    pub fn ws_transform_to_list_tiled2d_device(g: *mut ws_group, field_h: usize, field_w: usize, py: c_int, px: c_int, n_seeds_total: usize, blocks: *const ws_tile_block2d,
                                               opt: *const ws_options, merging: c_int, d_lakes: *mut ws_lake, cap: usize, n_lakes: *mut usize, offsets: *mut u64,
                                               uncoloured: *mut u64, exchange_rounds: *mut u32) -> c_int;
    pub fn ws_segment_tiled2d(g: *mut ws_group, img: *const u8, h: usize, w: usize, row_stride: usize, seeds_rc: *const u64,
        n_seeds: usize, opt: *const ws_options, py: c_int, px: c_int, merging: c_int, out_labels: *mut u64, exchange_rounds: *mut u32) -> c_int;
    pub fn ws_segment_batch_group(g: *mut ws_group, h: usize, w: usize, parts: *const ws_batch_part, opt: *const ws_options,
        failed_rank: *mut usize, failed_slice: *mut usize) -> c_int;
}
