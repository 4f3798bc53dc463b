//! Raw binding of `include/jpegenc_mi355x.h` (ABI version 1), one declaration per exported function.
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_void};

pub const JPEGENC_OK: c_int = 0;
pub const JPEGENC_ERR_INVALID_APP_SEGMENT: c_int = 1;
pub const JPEGENC_ERR_APP_SEGMENT_TOO_LARGE: c_int = 2;
pub const JPEGENC_ERR_ICC_TOO_LARGE: c_int = 3;
pub const JPEGENC_ERR_BAD_IMAGE_DATA: c_int = 4;
pub const JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS: c_int = 5;
pub const JPEGENC_ERR_WRITE: c_int = 6;
pub const JPEGENC_ERR_INVALID_ARGUMENT: c_int = 7;
pub const JPEGENC_ERR_HIP: c_int = 8;
pub const JPEGENC_ERR_NO_DEVICE: c_int = 9;
pub const JPEGENC_ERR_BUFFER_TOO_SMALL: c_int = 10;

pub const JPEGENC_ORDER_MCU: c_int = 0;
pub const JPEGENC_ORDER_PLANAR: c_int = 1;
pub const JPEGENC_FDCT_SCALAR: c_int = 0;
pub const JPEGENC_FDCT_SIMD: c_int = 1;
pub const JPEGENC_Q_CUSTOM: c_int = 9;

#[repr(C)]
pub struct jpegenc_qtable {
    // == QuantizationTable (src/quantization.rs:209-213)
    pub table: [u16; 64],
    pub reciprocals: [i32; 64],
    pub corrections: [i32; 64],
}

#[repr(C)]
pub struct jpegenc_layout {
    pub num_components: i32,
    pub max_h: i32,
    pub max_v: i32,
    pub h: [i32; 4],
    pub v: [i32; 4],
    pub table: [i32; 4],
    pub blocks: [u64; 4],
    pub total_blocks: u64,
    pub mcus: u64,
}

#[repr(C)]
pub struct jpegenc_huffman_spec {
    pub bits: [u8; 16],
    pub values: [u8; 256],
    pub num_values: i32,
}

#[repr(C)]
pub struct jpegenc_scan {
    pub component: i32,
    pub with_dc: i32,
    pub ac_start: i32,
    pub ac_end: i32,
    pub restart_interval: i32,
}

#[repr(C)]
pub struct jpegenc_plane {
    pub d_data: *const c_void,
    pub pitch: usize,
    pub pixel_stride: i32,
    pub invert: i32,
    pub shift: i32,
    pub reserved: i32,
}

pub enum jpegenc_encoder {}
pub enum jpegenc_scan_lanes {}
pub type jpegenc_write_fn = unsafe extern "C" fn(user: *mut c_void, data: *const u8, len: usize) -> c_int;
pub type jpegenc_fill_row_fn = unsafe extern "C" fn(user: *mut c_void, y: u16, planes: *const *mut u8);
pub type jpegenc_tile_callback =
    unsafe extern "C" fn(user: *mut c_void, frame_index: c_int, coeffs: *const i16, num_blocks: usize) -> c_int;

extern "C" {
    pub fn jpegenc_abi_version() -> c_int;
    pub fn jpegenc_device_count() -> c_int;
    pub fn jpegenc_last_error() -> *const c_char;
    pub fn jpegenc_status_string(status: c_int) -> *const c_char;

    pub fn jpegenc_qtable_init(out: *mut jpegenc_qtable, table_type: c_int, custom: *const u16, quality: c_int, luma: c_int) -> c_int;
    pub fn jpegenc_sampling_factor_from_factors(horizontal: c_int, vertical: c_int) -> c_int;
    pub fn jpegenc_bytes_per_pixel(color_type: c_int) -> c_int;
    pub fn jpegenc_layout_init(out: *mut jpegenc_layout, width: c_int, height: c_int, color_type: c_int, h: c_int, v: c_int, order: c_int) -> c_int;

    pub fn jpegenc_blocks_device(d_pixels: *const c_void, pixel_frame_stride: usize, num_frames: c_int, width: c_int, height: c_int,
                                 color_type: c_int, h: c_int, v: c_int, tables: *const jpegenc_qtable, order: c_int, fdct_variant: c_int,
                                 d_coeffs: *mut c_void, coeff_frame_stride: usize, stream: *mut c_void) -> c_int;
    pub fn jpegenc_blocks_stream(device: c_int, frames: *const *const u8, frame_len: usize, num_frames: c_int, width: c_int, height: c_int,
                                 color_type: c_int, h: c_int, v: c_int, tables: *const jpegenc_qtable, order: c_int, fdct_variant: c_int,
                                 callback: jpegenc_tile_callback, user: *mut c_void) -> c_int;
    pub fn jpegenc_blocks_stream_release() -> c_int;
    pub fn jpegenc_blocks_host(device: c_int, pixels: *const u8, pixels_len: usize, width: c_int, height: c_int, color_type: c_int,
                               h: c_int, v: c_int, tables: *const jpegenc_qtable, order: c_int, fdct_variant: c_int,
                               coeffs: *mut i16, capacity: usize) -> c_int;
    pub fn jpegenc_histogram_device(d_coeffs: *const c_void, layout: *const jpegenc_layout, progressive_scans: c_int,
                                    d_freq: *mut c_void, stream: *mut c_void) -> c_int;

    pub fn jpegenc_scan_workspace_size(layout: *const jpegenc_layout, scan: *const jpegenc_scan, num_frames: c_int) -> usize;
    pub fn jpegenc_scan_max_bytes(layout: *const jpegenc_layout, scan: *const jpegenc_scan) -> usize;
    pub fn jpegenc_scan_device(d_coeffs: *const c_void, coeff_frame_stride: usize, num_frames: c_int, layout: *const jpegenc_layout,
                               scan: *const jpegenc_scan, tables: *const [jpegenc_huffman_spec; 2], d_out: *mut c_void,
                               out_frame_stride: usize, d_out_lengths: *mut u32, d_workspace: *mut c_void, workspace_bytes: usize,
                               stream: *mut c_void) -> c_int;

    pub fn jpegenc_pixels_scan_fused(width: c_int, height: c_int, color_type: c_int, h: c_int, v: c_int) -> c_int;
    pub fn jpegenc_pixels_scan_device(d_pixels: *const c_void, pixel_frame_stride: usize, num_frames: c_int, width: c_int, height: c_int,
                                      color_type: c_int, h: c_int, v: c_int, tables: *const jpegenc_qtable, fdct_variant: c_int,
                                      restart_interval: c_int, huffman: *const [jpegenc_huffman_spec; 2], d_coeffs: *mut c_void,
                                      coeff_frame_stride: usize, d_out: *mut c_void, out_frame_stride: usize, d_out_lengths: *mut u32,
                                      d_workspace: *mut c_void, workspace_bytes: usize, stream: *mut c_void) -> c_int;

    pub fn jpegenc_scan_lanes_new(out: *mut *mut jpegenc_scan_lanes, device: c_int, width: c_int, height: c_int, color_type: c_int, h: c_int, v: c_int,
                                  restart_interval: c_int, max_frames_per_call: c_int) -> c_int;
    pub fn jpegenc_scan_lanes_submit(lanes: *mut jpegenc_scan_lanes, d_pixels: *const c_void, pixel_frame_stride: usize, num_frames: c_int,
                                     tables: *const jpegenc_qtable, fdct_variant: c_int, huffman: *const [jpegenc_huffman_spec; 2],
                                     d_out: *mut c_void, out_frame_stride: usize, d_out_lengths: *mut u32, producer_stream: *mut c_void) -> c_int;
    pub fn jpegenc_scan_lanes_join(lanes: *mut jpegenc_scan_lanes, stream: *mut c_void) -> c_int;
    pub fn jpegenc_scan_lanes_free(lanes: *mut jpegenc_scan_lanes);
    pub fn jpegenc_pixels_scan_dense(layout: *const jpegenc_layout, scan_bytes: usize) -> c_int;

    pub fn jpegenc_encoder_new(quality: c_int) -> *mut jpegenc_encoder;
    pub fn jpegenc_encoder_free(e: *mut jpegenc_encoder);
    pub fn jpegenc_encoder_set_device(e: *mut jpegenc_encoder, device: c_int) -> c_int;
    pub fn jpegenc_encoder_set_fdct_variant(e: *mut jpegenc_encoder, variant: c_int) -> c_int;
    pub fn jpegenc_encoder_set_device_entropy(e: *mut jpegenc_encoder, enable: c_int) -> c_int;
    pub fn jpegenc_encoder_set_register_cache(e: *mut jpegenc_encoder, bytes: usize) -> c_int;
    pub fn jpegenc_encoder_set_numa_bind(e: *mut jpegenc_encoder, enable: c_int) -> c_int;
    pub fn jpegenc_encoder_set_batch_upload(e: *mut jpegenc_encoder, mode: c_int) -> c_int;
    pub fn jpegenc_encoder_batch_shard_info(e: *mut jpegenc_encoder, shard: c_int, device: *mut c_int, batch_workers: *mut c_int, upload_mode: *mut c_int,
                                            register_cache_bytes: *mut usize, pool_workers: *mut c_int) -> c_int;
    pub fn jpegenc_encoder_set_batch_workers(e: *mut jpegenc_encoder, threads: c_int) -> c_int;
    pub fn jpegenc_encoder_batch_workers(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_set_batch_round_frames(e: *mut jpegenc_encoder, frames: c_int) -> c_int;
    pub fn jpegenc_encoder_set_density(e: *mut jpegenc_encoder, unit: c_int, x: u16, y: u16) -> c_int;
    pub fn jpegenc_encoder_density(e: *const jpegenc_encoder, unit: *mut c_int, x: *mut u16, y: *mut u16) -> c_int;
    pub fn jpegenc_encoder_set_sampling_factor(e: *mut jpegenc_encoder, sf: c_int) -> c_int;
    pub fn jpegenc_encoder_sampling_factor(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_set_quantization_tables(e: *mut jpegenc_encoder, luma: c_int, luma_custom: *const u16, chroma: c_int,
                                                   chroma_custom: *const u16) -> c_int;
    pub fn jpegenc_encoder_quantization_tables(e: *const jpegenc_encoder, types: *mut c_int) -> c_int;
    pub fn jpegenc_encoder_set_progressive(e: *mut jpegenc_encoder, on: c_int) -> c_int;
    pub fn jpegenc_encoder_set_progressive_scans(e: *mut jpegenc_encoder, scans: c_int) -> c_int;
    pub fn jpegenc_encoder_progressive_scans(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_set_restart_interval(e: *mut jpegenc_encoder, interval: u16) -> c_int;
    pub fn jpegenc_encoder_restart_interval(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_set_optimized_huffman_tables(e: *mut jpegenc_encoder, on: c_int) -> c_int;
    pub fn jpegenc_encoder_optimized_huffman_tables(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_add_app_segment(e: *mut jpegenc_encoder, nr: c_int, data: *const u8, len: usize) -> c_int;
    pub fn jpegenc_encoder_add_icc_profile(e: *mut jpegenc_encoder, data: *const u8, len: usize) -> c_int;
    pub fn jpegenc_encoder_add_exif_metadata(e: *mut jpegenc_encoder, data: *const u8, len: usize) -> c_int;

    pub fn jpegenc_encoder_encode(e: *mut jpegenc_encoder, data: *const u8, len: usize, width: c_int, height: c_int, color_type: c_int,
                                  sink: jpegenc_write_fn, user: *mut c_void) -> c_int;
    pub fn jpegenc_encoder_block_order(e: *const jpegenc_encoder) -> c_int;
    pub fn jpegenc_encoder_encode_coefficients(e: *mut jpegenc_encoder, coeffs: *const i16, num_blocks: usize, width: c_int, height: c_int,
                                               color_type: c_int, sink: jpegenc_write_fn, user: *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_device(e: *mut jpegenc_encoder, d_pixels: *const c_void, width: c_int, height: c_int, color_type: c_int,
                                         sink: jpegenc_write_fn, user: *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_batch_device(e: *mut jpegenc_encoder, d_frames: *const c_void, frame_stride: usize, num_frames: c_int,
                                               width: c_int, height: c_int, color_type: c_int, sink: jpegenc_write_fn,
                                               users: *const *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_to_buffer(e: *mut jpegenc_encoder, data: *const u8, len: usize, width: c_int, height: c_int,
                                            color_type: c_int, out: *mut u8, capacity: usize, out_len: *mut usize) -> c_int;
    pub fn jpegenc_encoder_encode_to_file(e: *mut jpegenc_encoder, path: *const c_char, data: *const u8, len: usize, width: c_int,
                                          height: c_int, color_type: c_int) -> c_int;
    pub fn jpegenc_encoder_encode_image(e: *mut jpegenc_encoder, jpeg_color_type: c_int, width: c_int, height: c_int,
                                        fill_row: jpegenc_fill_row_fn, image_user: *mut c_void, sink: jpegenc_write_fn,
                                        sink_user: *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_planes_device(e: *mut jpegenc_encoder, jpeg_color_type: c_int, width: c_int, height: c_int,
                                                planes: *const jpegenc_plane, planes_subsampled: c_int, sink: jpegenc_write_fn,
                                                user: *mut c_void) -> c_int;
    // a pool of such surfaces (planes: num_frames x 4 descriptors, frame-major), the batch sharing its launches
    pub fn jpegenc_encoder_encode_planes_batch_device(e: *mut jpegenc_encoder, jpeg_color_type: c_int, width: c_int, height: c_int,
                                                      planes: *const jpegenc_plane, num_frames: c_int, planes_subsampled: c_int,
                                                      sink: jpegenc_write_fn, users: *const *mut c_void) -> c_int;
    pub fn jpegenc_packed_planes(surface_format: c_int, d_planes: *const *const c_void, pitches: *const usize, planes: *mut jpegenc_plane) -> c_int;
    pub fn jpegenc_encoder_encode_batch(e: *mut jpegenc_encoder, frames: *const *const u8, frame_len: usize, num_frames: c_int,
                                        width: c_int, height: c_int, color_type: c_int, sink: jpegenc_write_fn,
                                        users: *const *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_batch_to_buffers(e: *mut jpegenc_encoder, frames: *const *const u8, frame_len: usize, num_frames: c_int,
                                                   width: c_int, height: c_int, color_type: c_int, outs: *const *mut u8,
                                                   capacities: *const usize, lengths: *mut usize) -> c_int;
    pub fn jpegenc_encoder_encode_batch_device_to_buffers(e: *mut jpegenc_encoder, d_frames: *const c_void, frame_stride: usize,
                                                          num_frames: c_int, width: c_int, height: c_int, color_type: c_int,
                                                          outs: *const *mut u8, capacities: *const usize, lengths: *mut usize) -> c_int;

    pub fn jpegenc_host_alloc(bytes: usize, out: *mut *mut c_void) -> c_int;
    pub fn jpegenc_host_free(p: *mut c_void) -> c_int;
    pub fn jpegenc_host_register(p: *mut c_void, bytes: usize) -> c_int;
    pub fn jpegenc_host_unregister(p: *mut c_void) -> c_int;
    pub fn jpegenc_host_copy(dst: *mut c_void, src: *const c_void, bytes: usize) -> c_int;
    pub fn jpegenc_encoder_batch_worker_info(e: *mut jpegenc_encoder, worker: c_int, staging: *mut *const c_void, staging_bytes: *mut usize,
                                             last_cpu: *mut c_int) -> c_int;
    pub fn jpegenc_shard_frames(num_frames: c_int, num_shards: c_int, shard: c_int, indices: *mut c_int, capacity: c_int) -> c_int;
    pub fn jpegenc_encoder_encode_batch_multi(e: *mut jpegenc_encoder, devices: *const c_int, num_devices: c_int,
                                              frames: *const *const u8, frame_len: usize, num_frames: c_int, width: c_int,
                                              height: c_int, color_type: c_int, sink: jpegenc_write_fn,
                                              users: *const *mut c_void) -> c_int;
    pub fn jpegenc_encoder_encode_batch_multi_to_buffers(e: *mut jpegenc_encoder, devices: *const c_int, num_devices: c_int,
                                                         frames: *const *const u8, frame_len: usize, num_frames: c_int, width: c_int,
                                                         height: c_int, color_type: c_int, outs: *const *mut u8,
                                                         capacities: *const usize, lengths: *mut usize) -> c_int;

    pub fn jpegenc_rgb_to_ycbcr(r: u8, g: u8, b: u8, out: *mut u8);
    pub fn jpegenc_cmyk_to_ycck(c: u8, m: u8, y: u8, k: u8, out: *mut u8);
}
