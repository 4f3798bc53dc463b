"""ctypes view of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/jpegenc_oracle.h).  The product (jpeg-encoder_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

LUMA, RGB, RGBA, BGR, BGRA, YCBCR, CMYK, CMYK_AS_YCCK, YCCK = range(9)
ORDER_MCU, ORDER_PLANAR = 0, 1
FDCT_SCALAR, FDCT_SIMD = 0, 1
Q_DEFAULT, Q_FLAT, Q_MSSSIM, Q_PSNRHVS, Q_IMAGEMAGICK, Q_KLEIN, Q_DENTAL, Q_VISUAL, Q_IMPROVED, Q_CUSTOM = range(10)
OK, ERR_INVALID_APP_SEGMENT, ERR_APP_SEGMENT_TOO_LARGE, ERR_ICC_TOO_LARGE, ERR_BAD_IMAGE_DATA, \
    ERR_ZERO_DIMENSIONS, ERR_WRITE, ERR_INVALID_ARGUMENT = range(8)
BPP = {LUMA: 1, RGB: 3, RGBA: 4, BGR: 3, BGRA: 4, YCBCR: 3, CMYK: 4, CMYK_AS_YCCK: 4, YCCK: 4}


class QTable(C.Structure):
    _fields_ = [("table", C.c_uint16 * 64), ("recip", C.c_int32 * 64), ("corr", C.c_int32 * 64)]


class Layout(C.Structure):
    _fields_ = [("ncomp", C.c_int), ("hmax", C.c_int), ("vmax", C.c_int),
                ("h", C.c_int * 4), ("v", C.c_int * 4), ("qsel", C.c_int * 4)]


class Config(C.Structure):
    _fields_ = [("quality", C.c_int), ("hs", C.c_int), ("vs", C.c_int),
                ("qpreset", C.c_int * 2), ("qcustom", (C.c_uint16 * 64) * 2),
                ("progressive_scans", C.c_int), ("restart_interval", C.c_int),
                ("optimize_huffman", C.c_int), ("density_unit", C.c_int),
                ("density_x", C.c_int), ("density_y", C.c_int), ("fdct_variant", C.c_int),
                ("n_app", C.c_int), ("app_data", C.c_void_p * 64), ("app_len", C.c_int * 64),
                ("app_nr", C.c_int * 64)]


def build(force=False, extra_cflags=None, out_path=None):
    """Compile the oracle with gcc (no GPU, no reference tree needed)."""
    out = out_path or _LIB_PATH
    src = os.path.join(_HERE, "jpegenc_oracle.c")
    src_avx2 = os.path.join(_HERE, "jpegenc_oracle_avx2.c")
    src_hw = os.path.join(_HERE, "fdct_avx2_hw.c")
    src_micro = os.path.join(_HERE, "criterion_micro.c")
    deps = [src, src_avx2, src_hw, src_micro, os.path.join(_HERE, "jpegenc_oracle.h"), os.path.join(_HERE, "orc_tables.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    flags = extra_cflags or ["-mavx2"]
    subprocess.check_call(["gcc", "-O3", "-fPIC", "-std=gnu11", *flags, "-shared", "-o", out, src, src_avx2, src_hw, src_micro])
    return out


_lib = None


def lib(path=None):
    global _lib
    if path is not None:
        return _bind(C.CDLL(path))
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(C.CDLL(_LIB_PATH))
    return _lib


def _bind(l):
    u8p, i16p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16)
    l.orc_rgb_to_ycbcr.argtypes = [C.c_uint8] * 3 + [u8p]
    l.orc_cmyk_to_ycck.argtypes = [C.c_uint8] * 4 + [u8p]
    l.orc_fdct.argtypes = [i16p, C.c_int]
    l.orc_fdct_avx2_hw.argtypes = [i16p]
    l.orc_fdct_avx2_hw_many.argtypes = [C.c_void_p, C.c_long]
    l.orc_fdct_avx2_hw_compare.argtypes = [C.c_void_p, C.c_long, C.c_int, C.POINTER(C.c_long)]
    l.orc_fdct_avx2_hw_compare.restype = C.c_long
    l.orc_bench_fdct_ns.argtypes = [C.c_int, C.c_double]
    l.orc_bench_fdct_ns.restype = C.c_double
    l.orc_bench_ycbcr_ms.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int]
    l.orc_bench_ycbcr_ms.restype = C.c_double
    l.orc_qtable_init.argtypes = [C.POINTER(QTable), C.c_int, C.POINTER(C.c_uint16), C.c_int, C.c_int]
    l.orc_quantize.argtypes = [C.POINTER(QTable), C.c_int16, C.c_int]
    l.orc_quantize.restype = C.c_int16
    l.orc_quantize_block.argtypes = [C.POINTER(QTable), i16p, i16p]
    l.orc_num_bits.argtypes = [C.c_int16]
    l.orc_get_code.argtypes = [C.c_int16, C.POINTER(C.c_int), C.POINTER(C.c_uint)]
    l.orc_layout_init.argtypes = [C.POINTER(Layout), C.c_int, C.c_int, C.c_int]
    l.orc_jpeg_color_type.argtypes = [C.c_int]
    l.orc_block_counts.argtypes = [C.c_int, C.c_int, C.POINTER(Layout), C.c_int, C.POINTER(C.c_size_t)]
    l.orc_block_counts.restype = C.c_size_t
    l.orc_encode_blocks.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(QTable), C.c_int, C.c_int, C.c_void_p]
    l.orc_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(Layout), C.c_int, C.c_void_p]
    l.orc_huffman_optimized.argtypes = [C.POINTER(C.c_uint32), u8p, u8p]
    l.orc_huffman_lookup.argtypes = [u8p, u8p, C.c_int, u8p, C.POINTER(C.c_uint16)]
    l.orc_config_default.argtypes = [C.POINTER(Config), C.c_int]
    l.orc_encode_jpeg.argtypes = [C.POINTER(Config), C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    l.orc_zigzag.restype = u8p
    return l


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error code {code}")
        self.code = code


def rgb_to_ycbcr(r, g, b):
    out = (C.c_uint8 * 3)()
    lib().orc_rgb_to_ycbcr(r, g, b, out)
    return tuple(out)


def cmyk_to_ycck(c, m, y, k):
    out = (C.c_uint8 * 4)()
    lib().orc_cmyk_to_ycck(c, m, y, k, out)
    return tuple(out)


def fdct(block, variant=FDCT_SCALAR):
    a = np.ascontiguousarray(block, dtype=np.int16).copy()
    assert a.size == 64
    lib().orc_fdct(a.ctypes.data_as(C.POINTER(C.c_int16)), variant)
    return a


def qtable(quality, luma, preset=Q_DEFAULT, custom=None):
    t = QTable()
    cust = None
    if preset == Q_CUSTOM:
        cust = (C.c_uint16 * 64)(*[int(v) for v in custom])
    lib().orc_qtable_init(C.byref(t), preset, cust, quality, 1 if luma else 0)
    return t


def qtables(quality, presets=(Q_DEFAULT, Q_DEFAULT), customs=(None, None)):
    arr = (QTable * 2)()
    for i in range(2):
        t = qtable(quality, i == 0, presets[i], customs[i])
        C.memmove(C.byref(arr[i]), C.byref(t), C.sizeof(QTable))
    return arr


def quantize(t, v, idx):
    return lib().orc_quantize(C.byref(t), v, idx)


def num_bits(v):
    return lib().orc_num_bits(v)


def get_code(v):
    s, b = C.c_int(), C.c_uint()
    lib().orc_get_code(v, C.byref(s), C.byref(b))
    return s.value, b.value


def layout(color_type, hs, vs):
    L = Layout()
    rc = lib().orc_layout_init(C.byref(L), lib().orc_jpeg_color_type(color_type), hs, vs)
    if rc:
        raise OracleError(rc)
    return L


def block_counts(width, height, color_type, hs, vs, order):
    L = layout(color_type, hs, vs)
    per = (C.c_size_t * 4)()
    total = lib().orc_block_counts(width, height, C.byref(L), order, per)
    return total, list(per)[:L.ncomp]


def encode_blocks(pixels, width, height, color_type, hs, vs, quality=None, order=ORDER_MCU,
                  variant=FDCT_SCALAR, q=None):
    """pixels -> (nblocks, 64) int16 zigzag coefficients, as the reference would produce them."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
    if q is None:
        q = qtables(quality)
    if width == 0 or height == 0 or px.size < width * height * BPP[color_type]:
        total = 0
    else:
        total, _ = block_counts(width, height, color_type, hs, vs, order)
    out = np.empty((max(total, 1), 64), dtype=np.int16)
    rc = lib().orc_encode_blocks(px.ctypes.data, px.size, width, height, color_type, hs, vs, q, order,
                                 variant, out.ctypes.data)
    if rc:
        raise OracleError(rc)
    return out[:total]


def encode_blocks_avx2(pixels, width, height, color_type, hs, vs, quality=None, order=ORDER_MCU, q=None):
    """The AVX2 stand-in for the reference's `simd` feature (jpegenc_oracle_avx2.c): must equal
    encode_blocks(..., FDCT_SCALAR).  Returns None where it does not apply (layout or CPU)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
    if q is None:
        q = qtables(quality)
    total, _ = block_counts(width, height, color_type, hs, vs, order)
    out = np.empty((max(total, 1), 64), dtype=np.int16)
    fn = lib().orc_encode_blocks_avx2
    fn.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(QTable), C.c_int, C.c_void_p]
    rc = fn(px.ctypes.data, px.size, width, height, color_type, hs, vs, q, order, out.ctypes.data)
    if rc == 100:
        return None
    if rc:
        raise OracleError(rc)
    return out[:total]


def histogram(planar_blocks, width, height, color_type, hs, vs, progressive_scans=0):
    L = layout(color_type, hs, vs)
    per = (C.c_size_t * 4)()
    lib().orc_block_counts(width, height, C.byref(L), ORDER_PLANAR, per)
    blocks = np.ascontiguousarray(planar_blocks, dtype=np.int16)
    freq = np.zeros((2, 2, 257), dtype=np.uint32)
    lib().orc_histogram(blocks.ctypes.data, per, C.byref(L), progressive_scans, freq.ctypes.data)
    return freq


def huffman_optimized(freq):
    f = (C.c_uint32 * 257)(*[int(v) for v in freq])
    bits = (C.c_uint8 * 16)()
    vals = (C.c_uint8 * 256)()
    n = lib().orc_huffman_optimized(f, bits, vals)
    return list(bits), list(vals)[:n]


def huffman_lookup(bits, values):
    b = (C.c_uint8 * 16)(*bits)
    v = (C.c_uint8 * max(len(values), 1))(*values)
    size = (C.c_uint8 * 256)()
    code = (C.c_uint16 * 256)()
    lib().orc_huffman_lookup(b, v, len(values), size, code)
    return list(size), list(code)


def icc_segments(data):
    """Encoder::add_icc_profile chunking (src/encoder.rs:392-417) -> [(2, bytes), ...]."""
    max_chunk = 65535 - 2 - 12 - 2
    n = -(-len(data) // max_chunk)
    if n >= 255:
        raise OracleError(ERR_ICC_TOO_LARGE)
    return [(2, b"ICC_PROFILE\0" + bytes([i + 1, n]) + data[i * max_chunk:(i + 1) * max_chunk])
            for i in range(n)]


def exif_segment(data):
    """Encoder::add_exif_metadata (src/encoder.rs:426-435)."""
    return (1, b"Exif\0\0" + data)


def encode_jpeg(pixels, width, height, color_type, quality, sampling=None, progressive_scans=0,
                restart_interval=0, optimize=False, qpresets=(Q_DEFAULT, Q_DEFAULT),
                qcustoms=(None, None), density=None, app_segments=(), variant=FDCT_SCALAR):
    """Whole-file encode with Encoder::new(_, quality) defaults plus the listed setters."""
    cfg = Config()
    lib().orc_config_default(C.byref(cfg), quality)
    if sampling is not None:
        cfg.hs, cfg.vs = sampling
    cfg.progressive_scans = progressive_scans
    cfg.restart_interval = restart_interval
    cfg.optimize_huffman = 1 if optimize else 0
    cfg.fdct_variant = variant
    for i in range(2):
        cfg.qpreset[i] = qpresets[i]
        if qpresets[i] == Q_CUSTOM:
            for k in range(64):
                cfg.qcustom[i][k] = int(qcustoms[i][k])
    if density is not None:
        cfg.density_unit, cfg.density_x, cfg.density_y = density
    keep = []
    cfg.n_app = len(app_segments)
    for i, (nr, data) in enumerate(app_segments):
        buf = C.create_string_buffer(bytes(data), len(data))
        keep.append(buf)
        cfg.app_data[i] = C.cast(buf, C.c_void_p)
        cfg.app_len[i] = len(data)
        cfg.app_nr[i] = nr
    px = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
    cap = max(4096, width * height * BPP[color_type] * 2 + 65536 * (len(app_segments) + 2))
    out = np.empty(cap, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().orc_encode_jpeg(C.byref(cfg), px.ctypes.data, px.size, width, height, color_type,
                               out.ctypes.data, cap, C.byref(n))
    if rc:
        raise OracleError(rc)
    return out[:n.value].tobytes()
