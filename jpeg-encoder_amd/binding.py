"""ctypes binding of libjpegenc_mi355x.so — the same C ABI a Rust/cgo/JNI host would bind
(include/jpegenc_mi355x.h).  Python here is plumbing for tests and bench.py only.

The library must exist: there is no CPU fallback and no silent degradation.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JPEGENC_LIB: a diagnostic variant build (tools/diag/ab_bench.sh) instead of the in-tree library
LIB_PATH = os.environ.get("JPEGENC_LIB") or os.path.join(_HERE, "libjpegenc_mi355x.so")
# the -DJPEGENC_DIAG build of the same sources (csrc/diag_env.h): the only library that reads the diagnostic switches
# (JPEGENC_FUSED, JPEGENC_PACK_WINDOW_WORDS, ...); tests that force a rare path start a child process with JPEGENC_LIB = this
DIAG_LIB_PATH = os.path.join(_HERE, "libjpegenc_mi355x_diag.so")

# enum jpegenc_color_type == reference `enum ColorType` order (src/encoder.rs:72-99)
LUMA, RGB, RGBA, BGR, BGRA, YCBCR, CMYK, CMYK_AS_YCCK, YCCK = range(9)
RGB565, BGR565 = 9, 10          # extensions: 16-bit packed RGB, unpacked on the device (include/jpegenc_mi355x.h)
J_LUMA, J_YCBCR, J_CMYK, J_YCCK = range(4)
ORDER_MCU, ORDER_PLANAR = 0, 1
FDCT_SCALAR, FDCT_SIMD = 0, 1
UPLOAD_STAGED, UPLOAD_REGISTER_AHEAD = 0, 1
PLANES_FULL, PLANES_SUBSAMPLED, PLANES_SUBSAMPLED_H = 0, 1, 2        # planes_subsampled of the described-surface calls
(Q_DEFAULT, Q_FLAT, Q_CUSTOM_MS_SSIM, Q_CUSTOM_PSNR_HVS, Q_IMAGE_MAGICK, Q_KLEIN_SILVERSTEIN_CARNEY,
 Q_DENTAL_XRAYS, Q_VISUAL_DETECTION_MODEL, Q_IMPROVED_DETECTION_MODEL, Q_CUSTOM) = range(10)
DENSITY_PIXEL_ASPECT_RATIO, DENSITY_INCHES, DENSITY_CENTIMETERS = range(3)
(OK, ERR_INVALID_APP_SEGMENT, ERR_APP_SEGMENT_TOO_LARGE, ERR_ICC_TOO_LARGE, ERR_BAD_IMAGE_DATA,
 ERR_ZERO_IMAGE_DIMENSIONS, ERR_WRITE, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE,
 ERR_BUFFER_TOO_SMALL) = range(11)
BPP = {LUMA: 1, RGB: 3, RGBA: 4, BGR: 3, BGRA: 4, YCBCR: 3, CMYK: 4, CMYK_AS_YCCK: 4, YCCK: 4, RGB565: 2, BGR565: 2}


def sampling_factor(h, v):
    """SamplingFactor discriminant (src/encoder.rs:120-153): (h << 4) | v."""
    return (h << 4) | v


F_1_1, F_2_1, F_1_2, F_2_2 = 0x11, 0x21, 0x12, 0x22
F_4_1, F_4_2, F_1_4, F_2_4 = 0x41, 0x42, 0x14, 0x24

# every symbol include/jpegenc_mi355x.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "jpegenc_abi_version", "jpegenc_device_count", "jpegenc_last_error", "jpegenc_status_string",
    "jpegenc_qtable_init", "jpegenc_bytes_per_pixel", "jpegenc_sampling_factor_from_factors", "jpegenc_layout_init",
    "jpegenc_blocks_device", "jpegenc_blocks_host", "jpegenc_blocks_stream", "jpegenc_blocks_stream_release", "jpegenc_histogram_device",
    "jpegenc_scan_workspace_size", "jpegenc_scan_max_bytes", "jpegenc_scan_device",
    "jpegenc_pixels_scan_fused", "jpegenc_pixels_scan_dense", "jpegenc_pixels_scan_device",
    "jpegenc_scan_lanes_new", "jpegenc_scan_lanes_submit", "jpegenc_scan_lanes_join", "jpegenc_scan_lanes_free",
    "jpegenc_encoder_set_device_entropy", "jpegenc_encoder_set_register_cache", "jpegenc_encoder_set_numa_bind", "jpegenc_encoder_set_batch_upload", "jpegenc_encoder_set_batch_round_frames",
    "jpegenc_encoder_set_batch_workers", "jpegenc_encoder_batch_workers", "jpegenc_encoder_batch_shard_info",
    "jpegenc_encoder_new", "jpegenc_encoder_free", "jpegenc_encoder_set_device",
    "jpegenc_encoder_set_fdct_variant", "jpegenc_encoder_set_density", "jpegenc_encoder_density",
    "jpegenc_encoder_set_sampling_factor", "jpegenc_encoder_sampling_factor",
    "jpegenc_encoder_set_quantization_tables", "jpegenc_encoder_quantization_tables",
    "jpegenc_encoder_set_progressive", "jpegenc_encoder_set_progressive_scans",
    "jpegenc_encoder_progressive_scans", "jpegenc_encoder_set_restart_interval",
    "jpegenc_encoder_restart_interval", "jpegenc_encoder_set_optimized_huffman_tables",
    "jpegenc_encoder_optimized_huffman_tables", "jpegenc_encoder_add_app_segment",
    "jpegenc_encoder_add_icc_profile", "jpegenc_encoder_add_exif_metadata",
    "jpegenc_encoder_encode", "jpegenc_encoder_encode_device", "jpegenc_encoder_encode_batch_device", "jpegenc_encoder_encode_batch_device_to_buffers", "jpegenc_encoder_encode_to_buffer", "jpegenc_encoder_encode_to_file",
    "jpegenc_encoder_encode_image", "jpegenc_encoder_block_order", "jpegenc_encoder_encode_coefficients",
    "jpegenc_encoder_encode_batch", "jpegenc_encoder_encode_batch_to_buffers",
    "jpegenc_encoder_encode_planes_device", "jpegenc_encoder_encode_planes_batch_device", "jpegenc_packed_planes",
    "jpegenc_host_alloc", "jpegenc_host_free", "jpegenc_host_register", "jpegenc_host_unregister", "jpegenc_host_copy", "jpegenc_encoder_batch_worker_info",
    "jpegenc_shard_frames", "jpegenc_encoder_encode_batch_multi", "jpegenc_encoder_encode_batch_multi_to_buffers",
    "jpegenc_rgb_to_ycbcr", "jpegenc_cmyk_to_ycck",
]


class QTable(C.Structure):
    _fields_ = [("table", C.c_uint16 * 64), ("reciprocals", C.c_int32 * 64), ("corrections", C.c_int32 * 64)]


class Layout(C.Structure):
    _fields_ = [("num_components", C.c_int32), ("max_h", C.c_int32), ("max_v", C.c_int32),
                ("h", C.c_int32 * 4), ("v", C.c_int32 * 4), ("table", C.c_int32 * 4),
                ("blocks", C.c_uint64 * 4), ("total_blocks", C.c_uint64), ("mcus", C.c_uint64)]


class Plane(C.Structure):
    _fields_ = [("d_data", C.c_void_p), ("pitch", C.c_size_t), ("pixel_stride", C.c_int32), ("invert", C.c_int32),
                ("shift", C.c_int32), ("reserved", C.c_int32)]


def _plane(t):
    """(device_ptr, pitch, pixel_stride, invert[, shift]) -> Plane"""
    ptr, pitch, stride, inv = t[:4]
    return Plane(ptr, pitch, stride, 1 if inv else 0, t[4] if len(t) > 4 else 0, 0)


(SURFACE_I420, SURFACE_YV12, SURFACE_NV12, SURFACE_NV21, SURFACE_YUYV, SURFACE_UYVY, SURFACE_P010, SURFACE_P016,
 SURFACE_I010) = range(9)


def packed_planes(surface_format, ptrs, pitches):
    """jpegenc_packed_planes -> ([(ptr, pitch, pixel_stride, invert, shift)] * 3, sampling factor the layout is subsampled for)."""
    n = len(ptrs)
    p = (C.c_void_p * max(n, 3))(*ptrs)
    s = (C.c_size_t * max(n, 3))(*pitches)
    arr = (Plane * 4)()
    f = lib().jpegenc_packed_planes
    f.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(Plane)]
    rc = f(surface_format, p, s, arr)
    if rc < 0:
        check(-rc)
    return [(arr[i].d_data, arr[i].pitch, arr[i].pixel_stride, arr[i].invert, arr[i].shift) for i in range(3)], rc


class Scan(C.Structure):
    _fields_ = [("component", C.c_int32), ("with_dc", C.c_int32), ("ac_start", C.c_int32), ("ac_end", C.c_int32),
                ("restart_interval", C.c_int32)]


WRITE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint8), C.c_size_t)
FILL_ROW_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint16, C.POINTER(C.POINTER(C.c_uint8)))


class JpegEncError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"jpegenc status {status}: {message}")
        self.status = status


_lib = None


def lib():
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # PyTorch wheels bundle their own libamdhip64/libhsa-runtime64.  Two HIP runtimes in one
        # process cannot both own the GPU, so when torch is installed let it load first: the
        # dynamic loader then resolves our libamdhip64.so.7 dependency to the copy already mapped.
        # (A torch-free host — the Rust/C embedding — simply gets /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        l.jpegenc_last_error.restype = C.c_char_p
        l.jpegenc_status_string.restype = C.c_char_p
        l.jpegenc_status_string.argtypes = [C.c_int]
        l.jpegenc_qtable_init.argtypes = [C.POINTER(QTable), C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int]
        l.jpegenc_layout_init.argtypes = [C.POINTER(Layout)] + [C.c_int] * 6
        l.jpegenc_blocks_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.POINTER(QTable), C.c_int, C.c_int,
                                            C.c_void_p, C.c_size_t, C.c_void_p]
        l.jpegenc_blocks_host.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(QTable), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        l.jpegenc_blocks_stream.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.POINTER(QTable), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        l.jpegenc_histogram_device.argtypes = [C.c_void_p, C.POINTER(Layout), C.c_int, C.c_void_p, C.c_void_p]
        l.jpegenc_encoder_new.restype = C.c_void_p
        l.jpegenc_encoder_new.argtypes = [C.c_int]
        l.jpegenc_encoder_free.argtypes = [C.c_void_p]
        l.jpegenc_encoder_free.restype = None
        l.jpegenc_scan_workspace_size.restype = C.c_size_t
        l.jpegenc_scan_workspace_size.argtypes = [C.POINTER(Layout), C.POINTER(Scan), C.c_int]
        l.jpegenc_scan_max_bytes.restype = C.c_size_t
        l.jpegenc_scan_max_bytes.argtypes = [C.POINTER(Layout), C.POINTER(Scan)]
        l.jpegenc_scan_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(Layout), C.POINTER(Scan),
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                          C.c_void_p]
        l.jpegenc_pixels_scan_fused.argtypes = [C.c_int] * 5
        l.jpegenc_pixels_scan_device.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.POINTER(QTable), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        for name in ("set_device", "set_device_entropy", "set_numa_bind", "set_batch_round_frames", "set_batch_workers", "set_fdct_variant", "set_sampling_factor", "set_progressive",
                     "set_progressive_scans", "set_optimized_huffman_tables"):
            getattr(l, "jpegenc_encoder_" + name).argtypes = [C.c_void_p, C.c_int]
        for name in ("sampling_factor", "progressive_scans", "restart_interval", "optimized_huffman_tables", "batch_workers"):
            getattr(l, "jpegenc_encoder_" + name).argtypes = [C.c_void_p]
        l.jpegenc_encoder_set_restart_interval.argtypes = [C.c_void_p, C.c_uint16]
        l.jpegenc_encoder_set_density.argtypes = [C.c_void_p, C.c_int, C.c_uint16, C.c_uint16]
        l.jpegenc_encoder_density.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_uint16),
                                              C.POINTER(C.c_uint16)]
        l.jpegenc_encoder_set_quantization_tables.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint16),
                                                              C.c_int, C.POINTER(C.c_uint16)]
        l.jpegenc_encoder_quantization_tables.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.jpegenc_encoder_add_app_segment.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        l.jpegenc_encoder_add_icc_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.jpegenc_encoder_add_exif_metadata.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        l.jpegenc_encoder_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                             WRITE_FN, C.c_void_p]
        l.jpegenc_encoder_encode_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, WRITE_FN, C.c_void_p]
        l.jpegenc_encoder_encode_to_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int,
                                                       C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        l.jpegenc_encoder_encode_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, FILL_ROW_FN,
                                                   C.c_void_p, WRITE_FN, C.c_void_p]
        l.jpegenc_encoder_encode_batch.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int,
                                                   C.c_int, C.c_int, C.c_int, WRITE_FN, C.POINTER(C.c_void_p)]
        l.jpegenc_encoder_encode_batch_to_buffers.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int,
                                                              C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                              C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        l.jpegenc_encoder_encode_planes_device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Plane), C.c_int, WRITE_FN, C.c_void_p]
        l.jpegenc_shard_frames.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
        l.jpegenc_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        l.jpegenc_host_free.argtypes = [C.c_void_p]
        l.jpegenc_host_register.argtypes = [C.c_void_p, C.c_size_t]
        l.jpegenc_host_unregister.argtypes = [C.c_void_p]
        if hasattr(l, "jpegenc_host_copy"):          # (A/B runs load older builds through JPEGENC_LIB)
            l.jpegenc_host_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.jpegenc_encoder_encode_batch_multi_to_buffers.argtypes = [
            C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        l.jpegenc_encoder_encode_batch_multi.argtypes = [
            C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
            WRITE_FN, C.POINTER(C.c_void_p)]
        l.jpegenc_rgb_to_ycbcr.argtypes = [C.c_uint8] * 3 + [C.POINTER(C.c_uint8)]
        l.jpegenc_cmyk_to_ycck.argtypes = [C.c_uint8] * 4 + [C.POINTER(C.c_uint8)]
        _lib = l
    return _lib


def check(status):
    if status != OK:
        raise JpegEncError(status, lib().jpegenc_last_error().decode(errors="replace"))


def device_count():
    return lib().jpegenc_device_count()


class HostBuffer:
    """jpegenc_host_alloc: page-locked host memory as a numpy uint8 array (`.array`); frames handed to the batch entry
    points from such memory are uploaded in place, without the workers' staging copy.  Freed by close() / at collection."""

    def __init__(self, nbytes):
        import numpy as np
        self._ptr = C.c_void_p()
        check(lib().jpegenc_host_alloc(nbytes, C.byref(self._ptr)))
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(self._ptr.value)) if nbytes else np.zeros(0, np.uint8)

    def close(self):
        if getattr(self, "_ptr", None) is not None and self._ptr.value:
            self.array = None
            lib().jpegenc_host_free(self._ptr)
            self._ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown
            pass


def host_register(array):
    """jpegenc_host_register on a writable C-contiguous numpy array: its pages stay locked until host_unregister(array)."""
    check(lib().jpegenc_host_register(array.ctypes.data, array.nbytes))


def host_unregister(array):
    check(lib().jpegenc_host_unregister(array.ctypes.data))


def shard_frames(num_frames, num_shards, shard):
    """jpegenc_shard_frames: the indices of the frames shard `shard` of `num_shards` encodes (frame k -> shard
    k % num_shards).  Needs no GPU; the ONE sharding rule of the library, bench.py and the gloo test."""
    n = lib().jpegenc_shard_frames(num_frames, num_shards, shard, None, 0)
    if n < 0:
        raise JpegEncError(-n, lib().jpegenc_last_error().decode(errors="replace"))
    idx = (C.c_int * max(n, 1))()
    lib().jpegenc_shard_frames(num_frames, num_shards, shard, idx, n)
    return [idx[i] for i in range(n)]


def qtables(quality, types=(Q_DEFAULT, Q_DEFAULT), customs=(None, None)):
    """[luma, chroma] QuantizationTable::new_with_quality (src/encoder.rs:528-531)."""
    arr = (QTable * 2)()
    for i in range(2):
        cust = None
        if types[i] == Q_CUSTOM:
            cust = (C.c_uint16 * 64)(*[int(v) for v in customs[i]])
        check(lib().jpegenc_qtable_init(C.byref(arr[i]), types[i], cust, quality, 1 if i == 0 else 0))
    return arr


def layout(width, height, color_type, hs, vs, order):
    L = Layout()
    check(lib().jpegenc_layout_init(C.byref(L), width, height, color_type, hs, vs, order))
    return L


def blocks_host(pixels, width, height, color_type, hs, vs, quality=None, order=ORDER_MCU,
                variant=FDCT_SCALAR, q=None, device=0):
    """numpy pixels -> (nblocks, 64) int16 via H2D + kernel + D2H (jpegenc_blocks_host)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
    if q is None:
        q = qtables(quality)
    total = 0
    if width > 0 and height > 0 and px.size >= width * height * BPP[color_type]:
        total = layout(width, height, color_type, hs, vs, order).total_blocks
    out = np.empty((max(total, 1), 64), dtype=np.int16)
    check(lib().jpegenc_blocks_host(device, px.ctypes.data, px.size, width, height, color_type, hs, vs, q,
                                    order, variant, out.ctypes.data, out.size))
    return out[:total]


TILE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int16), C.c_size_t)


def blocks_stream(frame_ptrs, frame_len, width, height, color_type, hs, vs, q, on_tile, order=ORDER_MCU,
                  variant=FDCT_SCALAR, device=0):
    """jpegenc_blocks_stream: host frames (addresses in `frame_ptrs`) -> on_tile(index, ndarray view) per
    frame, in order, with uploads / kernel / downloads of neighbouring frames overlapped.  The view is
    only valid inside the callback; a non-zero / raising callback aborts the stream."""
    n = len(frame_ptrs)
    ptrs = (C.c_void_p * max(n, 1))(*frame_ptrs)
    err = []

    def trampoline(_user, index, coeffs, num_blocks):
        try:
            view = np.ctypeslib.as_array(coeffs, shape=(num_blocks, 64))
            return int(on_tile(index, view) or 0)
        except BaseException as exc:          # never unwind through C
            err.append(exc)
            return -1
    cb = TILE_CALLBACK(trampoline)
    rc = lib().jpegenc_blocks_stream(device, ptrs, frame_len, n, width, height, color_type, hs, vs, q, order,
                                     variant, cb, None)
    if err:
        raise err[0]
    check(rc)


def blocks_stream_release():
    """jpegenc_blocks_stream_release: free the streams and buffers the last successful blocks_stream call left for the next."""
    check(lib().jpegenc_blocks_stream_release())


def blocks_device(d_pixels_ptr, pixel_frame_stride, num_frames, width, height, color_type, hs, vs, q,
                  order, variant, d_coeffs_ptr, coeff_frame_stride, stream_ptr=0):
    """Raw device pointers (e.g. torch.Tensor.data_ptr()); asynchronous on `stream_ptr`."""
    check(lib().jpegenc_blocks_device(d_pixels_ptr, pixel_frame_stride, num_frames, width, height, color_type,
                                      hs, vs, q, order, variant, d_coeffs_ptr, coeff_frame_stride, stream_ptr))


def baseline_scan(component=-1, restart_interval=0):
    return Scan(component, 1, 1, 64, restart_interval)


def scan_device(d_coeffs_ptr, coeff_frame_stride, num_frames, L, scan, d_out_ptr, out_frame_stride, d_lengths_ptr,
                d_workspace_ptr, workspace_bytes, stream_ptr=0, tables=None):
    """Device entropy coding of one scan; tables=None -> Annex K.3 defaults."""
    check(lib().jpegenc_scan_device(d_coeffs_ptr, coeff_frame_stride, num_frames, C.byref(L), C.byref(scan), tables,
                                    d_out_ptr, out_frame_stride, d_lengths_ptr, d_workspace_ptr, workspace_bytes,
                                    stream_ptr))


def pixels_scan_device(d_pixels_ptr, pixel_frame_stride, num_frames, width, height, color_type, hs, vs, q, d_out_ptr,
                       out_frame_stride, d_lengths_ptr, d_workspace_ptr, workspace_bytes, stream_ptr=0, variant=FDCT_SCALAR,
                       restart_interval=0, d_coeffs_ptr=None, coeff_frame_stride=0, tables=None):
    """jpegenc_pixels_scan_device: pixels in HBM -> entropy-coded interleaved baseline scan in HBM (fused kernel for
    the RGB family; d_coeffs_ptr is only needed where pixels_scan_fused() is False)."""
    check(lib().jpegenc_pixels_scan_device(d_pixels_ptr, pixel_frame_stride, num_frames, width, height, color_type, hs, vs, q,
                                           variant, restart_interval, tables, d_coeffs_ptr, coeff_frame_stride, d_out_ptr,
                                           out_frame_stride, d_lengths_ptr, d_workspace_ptr, workspace_bytes, stream_ptr))


class ScanLanes:
    """jpegenc_scan_lanes: the two-stream pattern of jpegenc_pixels_scan_device behind one call site.  submit() returns at once and
    alternates between two internal lanes (stream + workspace each); join(stream) makes `stream` wait for everything submitted."""

    def __init__(self, width, height, color_type, hs, vs, max_frames_per_call, device=0, restart_interval=0):
        l = lib()
        l.jpegenc_scan_lanes_new.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int] * 8
        l.jpegenc_scan_lanes_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(QTable), C.c_int, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        l.jpegenc_scan_lanes_join.argtypes = [C.c_void_p, C.c_void_p]
        l.jpegenc_scan_lanes_free.argtypes = [C.c_void_p]
        l.jpegenc_scan_lanes_free.restype = None
        self._h = C.c_void_p()
        check(l.jpegenc_scan_lanes_new(C.byref(self._h), device, width, height, color_type, hs, vs, restart_interval, max_frames_per_call))

    def submit(self, d_pixels_ptr, pixel_frame_stride, num_frames, q, d_out_ptr, out_frame_stride, d_lengths_ptr, producer_stream_ptr=0,
               variant=FDCT_SCALAR, tables=None):
        check(lib().jpegenc_scan_lanes_submit(self._h, d_pixels_ptr, pixel_frame_stride, num_frames, q, variant, tables, d_out_ptr,
                                              out_frame_stride, d_lengths_ptr, producer_stream_ptr))

    def join(self, stream_ptr=0):
        check(lib().jpegenc_scan_lanes_join(self._h, stream_ptr))

    def close(self):
        if self._h:
            lib().jpegenc_scan_lanes_free(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter shutdown
            pass


def pixels_scan_dense(layout, scan_bytes):
    """jpegenc_pixels_scan_dense: a frame of this layout whose scan came to scan_bytes is dense content - the two-kernel pair is faster."""
    f = lib().jpegenc_pixels_scan_dense
    f.argtypes = [C.POINTER(Layout), C.c_size_t]
    return bool(f(C.byref(layout), int(scan_bytes)))


def pixels_scan_fused(width, height, color_type, hs, vs):
    return bool(lib().jpegenc_pixels_scan_fused(width, height, color_type, hs, vs))


def scan_workspace_size(L, scan, num_frames):
    return lib().jpegenc_scan_workspace_size(C.byref(L), C.byref(scan), num_frames)


def scan_max_bytes(L, scan):
    return lib().jpegenc_scan_max_bytes(C.byref(L), C.byref(scan))


def histogram_device(d_coeffs_ptr, L, progressive_scans, d_freq_ptr, stream_ptr=0):
    check(lib().jpegenc_histogram_device(d_coeffs_ptr, C.byref(L), progressive_scans, d_freq_ptr, stream_ptr))


class Encoder:
    """Python mirror of `struct Encoder` (src/encoder.rs:213-515) over the C handle API.

    Method names and argument meaning follow the Rust crate so the tests read like the
    reference's own (src/lib.rs:188-553).  Errors surface as JpegEncError with the status that
    mirrors the EncodingError variant.
    """

    def __init__(self, quality, device=0):
        self._h = lib().jpegenc_encoder_new(quality)
        if not self._h:
            raise MemoryError("jpegenc_encoder_new failed")
        check(lib().jpegenc_encoder_set_device(self._h, device))

    def close(self):
        if getattr(self, "_h", None):
            lib().jpegenc_encoder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except TypeError:          # interpreter shutdown: module globals are already gone
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_density(self, unit, x, y):
        check(lib().jpegenc_encoder_set_density(self._h, unit, x, y))

    def density(self):
        u, x, y = C.c_int(), C.c_uint16(), C.c_uint16()
        check(lib().jpegenc_encoder_density(self._h, C.byref(u), C.byref(x), C.byref(y)))
        return u.value, x.value, y.value

    def set_sampling_factor(self, sf):
        check(lib().jpegenc_encoder_set_sampling_factor(self._h, sf))

    def sampling_factor(self):
        return lib().jpegenc_encoder_sampling_factor(self._h)

    def set_quantization_tables(self, luma, chroma, luma_custom=None, chroma_custom=None):
        lc = (C.c_uint16 * 64)(*[int(v) for v in luma_custom]) if luma_custom is not None else None
        cc = (C.c_uint16 * 64)(*[int(v) for v in chroma_custom]) if chroma_custom is not None else None
        check(lib().jpegenc_encoder_set_quantization_tables(self._h, luma, lc, chroma, cc))

    def quantization_tables(self):
        t = (C.c_int * 2)()
        check(lib().jpegenc_encoder_quantization_tables(self._h, t))
        return t[0], t[1]

    def set_progressive(self, progressive):
        check(lib().jpegenc_encoder_set_progressive(self._h, 1 if progressive else 0))

    def set_progressive_scans(self, scans):
        check(lib().jpegenc_encoder_set_progressive_scans(self._h, scans))

    def progressive_scans(self):
        n = lib().jpegenc_encoder_progressive_scans(self._h)
        return n if n else None

    def set_restart_interval(self, interval):
        check(lib().jpegenc_encoder_set_restart_interval(self._h, interval))

    def restart_interval(self):
        n = lib().jpegenc_encoder_restart_interval(self._h)
        return n if n else None

    def set_optimized_huffman_tables(self, optimize):
        check(lib().jpegenc_encoder_set_optimized_huffman_tables(self._h, 1 if optimize else 0))

    def optimized_huffman_tables(self):
        return bool(lib().jpegenc_encoder_optimized_huffman_tables(self._h))

    def set_batch_upload(self, mode):
        """UPLOAD_STAGED (default) or UPLOAD_REGISTER_AHEAD: how batch calls bring pageable frames of more than 2 MB to the device."""
        lib().jpegenc_encoder_set_batch_upload.argtypes = [C.c_void_p, C.c_int]
        check(lib().jpegenc_encoder_set_batch_upload(self._h, int(mode)))

    def set_register_cache(self, nbytes):
        lib().jpegenc_encoder_set_register_cache.argtypes = [C.c_void_p, C.c_size_t]
        check(lib().jpegenc_encoder_set_register_cache(self._h, nbytes))

    def set_device_entropy(self, enable):
        check(lib().jpegenc_encoder_set_device_entropy(self._h, 1 if enable else 0))

    def batch_worker_info(self):
        """[(staging address, staging bytes, last cpu)] of the handle's batch workers (jpegenc_encoder_batch_worker_info)."""
        f = lib().jpegenc_encoder_batch_worker_info
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        out, n, i = [], 1, 0
        while i < n:
            p, nb, cpu = C.c_void_p(), C.c_size_t(), C.c_int()
            n = f(self._h, i, C.byref(p), C.byref(nb), C.byref(cpu))
            if n <= 0:
                break
            out.append((p.value or 0, nb.value, cpu.value))
            i += 1
        return out

    def set_batch_workers(self, threads):
        """Host threads this handle's batch calls keep busy at once, the caller's included (0 = automatic: at most 4 where the device
        codes the scans).  One process per GPU on a shared host: pass the rank's share of the CPUs (batch.rank_cpu_share)."""
        check(lib().jpegenc_encoder_set_batch_workers(self._h, int(threads)))

    def batch_workers(self):
        return int(lib().jpegenc_encoder_batch_workers(self._h))

    def batch_shard_info(self):
        """[{device, batch_workers, upload_mode, register_cache_bytes, pool_workers}] of the per-device children of encode_batch_multi."""
        f = lib().jpegenc_encoder_batch_shard_info
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]
        out, n, i = [], 1, 0
        while i < n:
            dev, bw, up, pw, rc = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
            n = f(self._h, i, C.byref(dev), C.byref(bw), C.byref(up), C.byref(rc), C.byref(pw))
            if n <= 0:
                break
            out.append({"device": dev.value, "batch_workers": bw.value, "upload_mode": up.value, "register_cache_bytes": rc.value, "pool_workers": pw.value})
            i += 1
        return out

    def set_batch_round_frames(self, frames):
        check(lib().jpegenc_encoder_set_batch_round_frames(self._h, int(frames)))

    def set_numa_bind(self, enable):
        check(lib().jpegenc_encoder_set_numa_bind(self._h, 1 if enable else 0))

    def set_fdct_variant(self, variant):
        check(lib().jpegenc_encoder_set_fdct_variant(self._h, variant))

    def add_app_segment(self, nr, data):
        check(lib().jpegenc_encoder_add_app_segment(self._h, nr, bytes(data), len(data)))

    def add_icc_profile(self, data):
        check(lib().jpegenc_encoder_add_icc_profile(self._h, bytes(data), len(data)))

    def add_exif_metadata(self, data):
        check(lib().jpegenc_encoder_add_exif_metadata(self._h, bytes(data), len(data)))

    def encode(self, data, width, height, color_type):
        """Encoder::encode -> bytes (the sink is an in-memory Vec<u8>)."""
        px = np.ascontiguousarray(data, dtype=np.uint8).reshape(-1)
        chunks = []

        def sink(_user, ptr, n):
            chunks.append(C.string_at(ptr, n))
            return 0

        cb = WRITE_FN(sink)
        check(lib().jpegenc_encoder_encode(self._h, px.ctypes.data, px.size, width, height, color_type, cb, None))
        return b"".join(chunks)

    def block_order(self):
        """The block order this encoder's mode consumes (ORDER_MCU for interleaved baseline, ORDER_PLANAR otherwise)."""
        fn = lib().jpegenc_encoder_block_order
        fn.argtypes = [C.c_void_p]
        return fn(self._h)

    def encode_coefficients(self, coeffs, width, height, color_type):
        """The host half only: quantised zig-zag blocks (num_blocks x 64 int16, in block_order()) -> JPEG bytes.  No GPU."""
        co = np.ascontiguousarray(coeffs, dtype=np.int16).reshape(-1, 64)
        chunks = []

        def sink(_user, ptr, n):
            chunks.append(C.string_at(ptr, n))
            return 0

        cb = WRITE_FN(sink)
        fn = lib().jpegenc_encoder_encode_coefficients
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, WRITE_FN, C.c_void_p]
        check(fn(self._h, co.ctypes.data, co.shape[0], width, height, color_type, cb, None))
        return b"".join(chunks)

    def encode_to_buffer(self, pixels, width, height, color_type, out):
        """jpegenc_encoder_encode_to_buffer into a caller-owned uint8 array; returns the file size."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
        n = C.c_size_t(0)
        fn = lib().jpegenc_encoder_encode_to_buffer
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        check(fn(self._h, px.ctypes.data, px.size, width, height, color_type, out.ctypes.data, out.size, C.byref(n)))
        return n.value

    def encode_to_file(self, path, pixels, width, height, color_type):
        """Encoder::new_file(path, q) + encode: the file is created first, then written."""
        px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
        lib().jpegenc_encoder_encode_to_file.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int]
        check(lib().jpegenc_encoder_encode_to_file(self._h, os.fsencode(path), px.ctypes.data, px.size, width, height, color_type))

    def encode_device(self, d_pixels_ptr, width, height, color_type):
        """Encode an image that is already in device memory (raw pointer, e.g. torch data_ptr())."""
        chunks = []

        def sink(_user, ptr, n):
            chunks.append(C.string_at(ptr, n))
            return 0

        cb = WRITE_FN(sink)
        check(lib().jpegenc_encoder_encode_device(self._h, d_pixels_ptr, width, height, color_type, cb, None))
        return b"".join(chunks)

    def encode_planes_device(self, jpeg_color_type, width, height, planes, planes_subsampled=False):
        """jpegenc_encoder_encode_planes_device: planes = [(device_ptr, pitch, pixel_stride, invert), ...] per component."""
        arr = (Plane * 4)()
        for i, t in enumerate(planes):
            arr[i] = _plane(t)
        chunks = []

        def sink(_user, ptr, n):
            chunks.append(C.string_at(ptr, n))
            return 0

        cb = WRITE_FN(sink)
        check(lib().jpegenc_encoder_encode_planes_device(self._h, jpeg_color_type, width, height, arr, int(planes_subsampled), cb, None))
        return b"".join(chunks)

    def encode_planes_batch_device(self, jpeg_color_type, width, height, frames, planes_subsampled=False):
        """jpegenc_encoder_encode_planes_batch_device: frames = [[(device_ptr, pitch, pixel_stride, invert), ...] per frame]
        -> list of bytes."""
        n = len(frames)
        arr = (Plane * (4 * max(n, 1)))()
        for f, planes in enumerate(frames):
            for i, t in enumerate(planes):
                arr[4 * f + i] = _plane(t)
        outs = [[] for _ in range(n)]

        def sink(user, ptr, nbytes):
            outs[(user or 0)].append(C.string_at(ptr, nbytes))
            return 0

        cb = WRITE_FN(sink)
        users = (C.c_void_p * max(n, 1))(*[i for i in range(n)])
        fn = lib().jpegenc_encoder_encode_planes_batch_device
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Plane), C.c_int, C.c_int, WRITE_FN, C.POINTER(C.c_void_p)]
        check(fn(self._h, jpeg_color_type, width, height, arr, n, int(planes_subsampled), cb, users))
        return [b"".join(o) for o in outs]

    def encode_batch_device(self, d_frames_ptr, frame_stride, num_frames, width, height, color_type):
        """Device-resident batch (raw pointer, frames `frame_stride` bytes apart) -> list of bytes."""
        outs = [[] for _ in range(num_frames)]

        def sink(user, ptr, nbytes):
            outs[(user or 0)].append(C.string_at(ptr, nbytes))
            return 0

        cb = WRITE_FN(sink)
        users = (C.c_void_p * max(num_frames, 1))(*[i for i in range(num_frames)])
        lib().jpegenc_encoder_encode_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                                             C.c_int, WRITE_FN, C.POINTER(C.c_void_p)]
        check(lib().jpegenc_encoder_encode_batch_device(self._h, d_frames_ptr, frame_stride, num_frames, width, height,
                                                        color_type, cb, users))
        return [b"".join(o) for o in outs]

    def encode_image(self, jpeg_color_type, width, height, fill_buffers):
        """Encoder::encode_image with a user ImageBuffer: fill_buffers(y) -> list of per-plane rows."""
        chunks = []

        def sink(_user, ptr, n):
            chunks.append(C.string_at(ptr, n))
            return 0

        def fill(_user, y, planes):
            rows = fill_buffers(y)
            for i, row in enumerate(rows):
                r = np.ascontiguousarray(row, dtype=np.uint8)
                C.memmove(planes[i], r.ctypes.data, width)

        cb, fb = WRITE_FN(sink), FILL_ROW_FN(fill)
        check(lib().jpegenc_encoder_encode_image(self._h, jpeg_color_type, width, height, fb, None, cb, None))
        return b"".join(chunks)

    def encode_batch_to_buffers(self, frames, width, height, color_type, capacity):
        """Batch encode with no Python in the inner loop: returns a list of bytes objects."""
        arrs = [np.ascontiguousarray(f, dtype=np.uint8).reshape(-1) for f in frames]
        n = len(arrs)
        outs = [np.empty(capacity, dtype=np.uint8) for _ in range(n)]
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        optrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * max(n, 1))(*([capacity] * n))
        lens = (C.c_size_t * max(n, 1))()
        flen = arrs[0].size if n else 0
        check(lib().jpegenc_encoder_encode_batch_to_buffers(self._h, ptrs, flen, n, width, height, color_type,
                                                           optrs, caps, lens))
        return [outs[i][:lens[i]].tobytes() for i in range(n)], outs, list(lens)

    def encode_batch_into(self, frames, width, height, color_type, outs, devices=None):
        """Batch encode into caller-owned uint8 arrays `outs` (no copies, no Python per frame inside the call):
        returns the list of file lengths.  devices=None -> this handle's GPU (jpegenc_encoder_encode_batch_to_buffers),
        else the multi-device entry point."""
        n = len(frames)
        ptrs = (C.c_void_p * max(n, 1))(*[f.ctypes.data for f in frames])
        optrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs[:n]])
        caps = (C.c_size_t * max(n, 1))(*[o.size for o in outs[:n]])
        lens = (C.c_size_t * max(n, 1))()
        flen = frames[0].size if n else 0
        if devices is None:
            check(lib().jpegenc_encoder_encode_batch_to_buffers(self._h, ptrs, flen, n, width, height, color_type, optrs, caps, lens))
        else:
            devs = (C.c_int * len(devices))(*devices)
            check(lib().jpegenc_encoder_encode_batch_multi_to_buffers(self._h, devs, len(devices), ptrs, flen, n, width, height,
                                                                     color_type, optrs, caps, lens))
        return [lens[i] for i in range(n)]

    def encode_batch_multi_to_buffers(self, devices, frames, width, height, color_type, capacity):
        """jpegenc_encoder_encode_batch_multi_to_buffers: the batch sharded over `devices` (HIP indices, repeats
        allowed) from this one process -> list of bytes, in frame order."""
        arrs = [np.ascontiguousarray(f, dtype=np.uint8).reshape(-1) for f in frames]
        n = len(arrs)
        outs = [np.empty(capacity, dtype=np.uint8) for _ in range(n)]
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        optrs = (C.c_void_p * max(n, 1))(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * max(n, 1))(*([capacity] * n))
        lens = (C.c_size_t * max(n, 1))()
        devs = (C.c_int * len(devices))(*devices)
        flen = arrs[0].size if n else 0
        check(lib().jpegenc_encoder_encode_batch_multi_to_buffers(self._h, devs, len(devices), ptrs, flen, n, width, height,
                                                                 color_type, optrs, caps, lens))
        return [outs[i][:lens[i]].tobytes() for i in range(n)]

    def encode_batch(self, frames, width, height, color_type):
        """frames: list of equally sized uint8 arrays -> list of bytes (frame-parallel on one GPU)."""
        arrs = [np.ascontiguousarray(f, dtype=np.uint8).reshape(-1) for f in frames]
        n = len(arrs)
        outs = [[] for _ in range(n)]

        def sink(user, ptr, nbytes):
            outs[(user or 0)].append(C.string_at(ptr, nbytes))
            return 0

        cb = WRITE_FN(sink)
        ptrs = (C.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrs])
        users = (C.c_void_p * max(n, 1))(*[i for i in range(n)])
        flen = arrs[0].size if n else 0
        check(lib().jpegenc_encoder_encode_batch(self._h, ptrs, flen, n, width, height, color_type, cb, users))
        return [b"".join(o) for o in outs]
