"""Device-side unpackers for the pixel formats a camera / decoder pipeline hands over device-resident ("custom
pixel formats as device functors") - 16-bit packed RGB (JPEGENC_RGB565 / JPEGENC_BGR565), packed 4:2:2 (YUYV / UYVY), and
16-bit planar / semi-planar surfaces (P010, P016, planar 10-bit) through plane descriptors with pixel_stride 4 and `shift`.

The reference's extension point for such sources is a user ImageBuffer whose fill_buffers does the unpacking on the host
(image_buffer.rs:40-98).  Every test feeds the SAME data (a) through that extension point - jpegenc_encoder_encode_image with a
host fill_row that unpacks - and (b) unpacked on the host through the oracle's Rgb / Ycbcr encode, and requires the device
unpackers to give byte-identical files (and coefficient-identical blocks).  Nothing here reads /root/reference.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding(pkg):
    b = importlib.import_module("jpeg_encoder_amd.binding")
    if b.device_count() < 1:
        pytest.fail("no MI355X visible: the HIP path has no CPU fallback")
    return b


def unpack565(words, bgr=False):
    """(h, w) uint16 -> (h, w, 3) uint8 RGB: channels widened by bit replication (the header's definition)."""
    w = words.astype(np.uint32)
    hi, g6, lo = (w >> 11) & 31, (w >> 5) & 63, w & 31
    r5, b5 = (lo, hi) if bgr else (hi, lo)
    return np.stack([(r5 << 3) | (r5 >> 2), (g6 << 2) | (g6 >> 4), (b5 << 3) | (b5 >> 2)], axis=-1).astype(np.uint8)


def _encoder(binding, kw):
    e = binding.Encoder(kw["quality"])
    if "sampling" in kw:
        e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
    if kw.get("progressive_scans"):
        e.set_progressive_scans(kw["progressive_scans"])
    if kw.get("restart_interval"):
        e.set_restart_interval(kw["restart_interval"])
    if kw.get("optimize"):
        e.set_optimized_huffman_tables(True)
    if kw.get("variant"):
        e.set_fdct_variant(binding.FDCT_SIMD)
    return e


def _okw(kw):
    return {k: v for k, v in kw.items()}


@pytest.mark.parametrize("bgr", [False, True], ids=["rgb565", "bgr565"])
def test_all_65536_words_unpack_like_the_definition(binding, oracle, bgr):
    """Every 16-bit value once (256 x 256), 4:4:4 and 4:2:0, both FDCT variants: coefficients equal the oracle's on the
    host-unpacked RGB image; and as flat 8x8 blocks at quality 100 (DC = 8 * (Y - 128): a +-1 of any channel shows)."""
    ct = binding.BGR565 if bgr else binding.RGB565
    words = np.arange(65536, dtype=np.uint16).reshape(256, 256)
    rgb = unpack565(words, bgr)
    raw = words.view(np.uint8).reshape(256, 512)
    for hs, vs in ((1, 1), (2, 2), (2, 1), (1, 2)):
        for variant in (binding.FDCT_SCALAR, binding.FDCT_SIMD):
            for order in (binding.ORDER_MCU, binding.ORDER_PLANAR):
                got = binding.blocks_host(raw, 256, 256, ct, hs, vs, 100, order, variant)
                want = oracle.encode_blocks(rgb, 256, 256, oracle.RGB, hs, vs, 100, order, variant)
                assert np.array_equal(got, want), (hs, vs, variant, order)
    flat = np.repeat(np.repeat(words, 8, axis=0), 8, axis=1)
    got = binding.blocks_host(flat.view(np.uint8).reshape(2048, 4096), 2048, 2048, ct, 1, 1, 100, binding.ORDER_MCU)
    want = oracle.encode_blocks(unpack565(flat, bgr), 2048, 2048, oracle.RGB, 1, 1, 100, oracle.ORDER_MCU)
    assert np.array_equal(got, want) and not got[:, 1:].any()


@pytest.mark.parametrize("w,h", [(1, 1), (7, 9), (37, 21), (258, 128), (515, 301), (1920, 1080)])
@pytest.mark.parametrize("kw", [dict(quality=80), dict(quality=90, sampling=(2, 1), restart_interval=3), dict(quality=85, sampling=(2, 2), progressive_scans=4),
                                dict(quality=92, sampling=(1, 2), optimize=True), dict(quality=75, sampling=(2, 2), variant=1), dict(quality=100, sampling=(1, 1))],
                         ids=["444", "422-restart", "420-progressive", "440-optimised", "420-simd", "444-q100"])
def test_rgb565_files_match_the_host_unpacking_image_buffer(binding, oracle, w, h, kw):
    """Host pixels, device-resident pixels, the worker-pool batch and the device-resident batch of 16-bit packed frames: the same
    file as (a) the library's own ImageBuffer path with a host fill_row that unpacks the words and converts them with the
    reference's rgb_to_ycbcr, and (b) the oracle fed the host-unpacked RGB image.  Both entropy coders."""
    import torch
    rng = np.random.default_rng(w * 31 + h)
    words = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    if w >= 64:                                                         # (photo-like: a gradient under the noise)
        yy, xx = np.mgrid[0:h, 0:w]
        r5, g6, b5 = (xx * 31 // w), (yy * 63 // h), ((xx + yy) * 31 // (w + h))
        words = (((r5 << 11) | (g6 << 5) | b5) ^ (words & 0x0821)).astype(np.uint16)
    raw = np.ascontiguousarray(words).view(np.uint8).reshape(h, w * 2)
    okw = {k: v for k, v in kw.items() if k != "variant"}
    if kw.get("variant"):
        okw["variant"] = oracle.FDCT_SIMD
    for bgr in (False, True):
        ct = binding.BGR565 if bgr else binding.RGB565
        rgb = unpack565(words, bgr)
        want = oracle.encode_jpeg(rgb, w, h, oracle.RGB, **okw)
        ycc = np.array([[oracle.rgb_to_ycbcr(*map(int, p)) for p in row] for row in rgb], dtype=np.uint8) if w * h <= 800 else None
        for device_entropy in (True, False):
            e = _encoder(binding, kw)
            e.set_device_entropy(device_entropy)
            assert e.encode(raw, w, h, ct) == want, (bgr, device_entropy)
            if ycc is not None:                                         # the reference's own extension point, unpacking on the host
                assert e.encode_image(binding.J_YCBCR, w, h, lambda y: [ycc[y, :, 0], ycc[y, :, 1], ycc[y, :, 2]]) == want
            d = torch.from_numpy(np.stack([raw, raw[::-1].copy()])).cuda()
            assert e.encode_device(d[0].data_ptr(), w, h, ct) == want
            files = e.encode_batch_device(d.data_ptr(), w * h * 2, 2, w, h, ct)
            assert files[0] == want and files[1] == oracle.encode_jpeg(unpack565(words[::-1], bgr), w, h, oracle.RGB, **okw)
            assert e.encode_batch([raw] * 3, w, h, ct) == [want] * 3
            e.close()


@pytest.mark.parametrize("sampling", [(4, 1), (4, 2), (1, 4), (2, 4)])
def test_rgb565_sampling_factors_of_four(binding, oracle, sampling):
    """16-bit packed RGB at the sampling factors of 4 (sequential files, encoder.rs:558-559): coefficients in both block orders and
    whole files - host pixels, device-resident pixels, a batch - against the oracle fed the host-unpacked RGB image."""
    import torch
    for (w, h) in ((37, 21), (258, 128), (515, 301)):
        rng = np.random.default_rng(w + h)
        words = rng.integers(0, 65536, (h, w), dtype=np.uint16)
        raw = np.ascontiguousarray(words).view(np.uint8).reshape(h, w * 2)
        for bgr in (False, True):
            ct = binding.BGR565 if bgr else binding.RGB565
            rgb = unpack565(words, bgr)
            for order in (binding.ORDER_MCU, binding.ORDER_PLANAR):
                for variant in (binding.FDCT_SCALAR, binding.FDCT_SIMD):
                    assert np.array_equal(binding.blocks_host(raw, w, h, ct, sampling[0], sampling[1], 85, order, variant),
                                          oracle.encode_blocks(rgb, w, h, oracle.RGB, sampling[0], sampling[1], 85, order, variant)), (w, h, bgr, order, variant)
            for kw in (dict(quality=85, sampling=sampling), dict(quality=77, sampling=sampling, restart_interval=5, optimize=True)):
                want = oracle.encode_jpeg(rgb, w, h, oracle.RGB, **kw)
                e = _encoder(binding, kw)
                assert e.encode(raw, w, h, ct) == want
                d = torch.from_numpy(raw.copy()).cuda()
                assert e.encode_device(d.data_ptr(), w, h, ct) == want
                assert e.encode_batch([raw] * 2, w, h, ct) == [want] * 2
                e.close()


def test_rgb565_rejects_what_the_tuned_kernels_do_not_take(binding):
    raw = np.zeros((16, 32), dtype=np.uint8)
    with pytest.raises(binding.JpegEncError) as err:
        binding.Encoder(80).encode(raw[:, :30], 16, 16, binding.BGR565)       # 2 bytes per pixel: too short
    assert err.value.status == binding.ERR_BAD_IMAGE_DATA
    assert binding.lib().jpegenc_bytes_per_pixel(binding.RGB565) == 2 and binding.lib().jpegenc_bytes_per_pixel(binding.BGR565) == 2


def _rep(plane, sx, sy, w, h):
    return np.repeat(np.repeat(plane, sy, axis=0), sx, axis=1)[:h, :w]


def _smooth(a):
    return (a.astype(np.int32) // 4 + (np.add.outer(np.arange(a.shape[0]), np.arange(a.shape[1])) // 3) % 192).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(16, 8), (37, 21), (258, 128), (515, 301), (1920, 1080)])
@pytest.mark.parametrize("kw", [dict(quality=85, sampling=(2, 1)), dict(quality=92, sampling=(2, 1), restart_interval=4),
                                dict(quality=80, sampling=(2, 1), progressive_scans=4, optimize=True), dict(quality=77, sampling=(2, 1), variant=1)],
                         ids=["baseline", "restart", "progressive-optimised", "simd"])
def test_packed_422_yuyv_and_uyvy(binding, oracle, w, h, kw):
    """YUYV / UYVY (one buffer, Y every 2 bytes, Cb / Cr every 4): three plane descriptors over the same memory
    (jpegenc_packed_planes), sampling factor F_2_1, planes_subsampled.  Same file as the oracle fed the interleaved YCbCr image
    with each chroma sample repeated twice, and as the ImageBuffer path unpacking the rows on the host."""
    import torch
    rng = np.random.default_rng(w + 3 * h)
    cw = -(-w // 2)
    y = _smooth(rng.integers(0, 256, (h, 2 * cw), dtype=np.uint8))
    cb = _smooth(rng.integers(0, 256, (h, cw), dtype=np.uint8))
    cr = _smooth(rng.integers(0, 256, (h, cw), dtype=np.uint8))
    full = np.stack([y[:, :w], _rep(cb, 2, 1, w, h), _rep(cr, 2, 1, w, h)], axis=-1)
    okw = {k: v for k, v in kw.items() if k != "variant"}
    if kw.get("variant"):
        okw["variant"] = oracle.FDCT_SIMD
    want = oracle.encode_jpeg(full, w, h, oracle.YCBCR, **okw)
    pitch = 4 * cw + 12
    for fmt in (binding.SURFACE_YUYV, binding.SURFACE_UYVY):
        packed = np.zeros((h, pitch), dtype=np.uint8)
        quad = [y[:, 0::2], cb, y[:, 1::2], cr] if fmt == binding.SURFACE_YUYV else [cb, y[:, 0::2], cr, y[:, 1::2]]      # Y0 U Y1 V / U Y0 V Y1
        packed[:, :4 * cw] = np.stack(quad, axis=-1).reshape(h, 4 * cw)
        d = torch.from_numpy(packed).cuda()
        planes, sampling = binding.packed_planes(fmt, [d.data_ptr()], [pitch])
        assert sampling == binding.F_2_1 and [p[2] for p in planes] == [2, 4, 4]
        for device_entropy in (True, False):
            e = _encoder(binding, kw)
            e.set_device_entropy(device_entropy)
            assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=True) == want, (fmt, device_entropy)
            if w * h <= 40000:
                assert e.encode_image(binding.J_YCBCR, w, h, lambda r: [full[r, :, 0], full[r, :, 1], full[r, :, 2]]) == want
            # a pool of four such frames in shared launches (two distinct buffers)
            d2 = torch.from_numpy(packed[::-1].copy()).cuda()
            planes2, _ = binding.packed_planes(fmt, [d2.data_ptr()], [pitch])
            files = e.encode_planes_batch_device(binding.J_YCBCR, w, h, [planes, planes2, planes, planes2], planes_subsampled=True)
            want2 = oracle.encode_jpeg(full[::-1].copy(), w, h, oracle.YCBCR, **okw)
            assert files == [want, want2, want, want2]
            e.close()


@pytest.mark.parametrize("w,h", [(16, 8), (16, 16), (37, 21), (37, 22), (258, 128), (515, 301), (514, 300), (1920, 1080)])
@pytest.mark.parametrize("kw", [dict(quality=85, sampling=(2, 2)), dict(quality=92, sampling=(2, 2), restart_interval=4),
                                dict(quality=80, sampling=(2, 2), progressive_scans=4, optimize=True), dict(quality=77, sampling=(2, 2), variant=1),
                                dict(quality=88, sampling=(1, 2))],
                         ids=["420", "420-restart", "420-progressive-optimised", "420-simd", "440"])
def test_packed_422_coded_as_420(binding, oracle, w, h, kw):
    """planes_subsampled = 2: YUYV / UYVY frames (chroma at half the columns, EVERY row) coded at F_2_2 - the device takes every
    second chroma row, and the rows below an image whose height is no multiple of 16 repeat its LAST row (even heights: a row
    the decimation itself never takes).  Same file as the oracle fed the interleaved YCbCr image with each chroma sample repeated
    twice along its row, and as the ImageBuffer path unpacking the rows on the host (the reference's extension point,
    image_buffer.rs:86-98).  F_1_2 from the same surface: luma and chroma at full width, chroma rows halved - only where the chroma
    planes are full width, i.e. not from a packed 4:2:2 surface: there the test describes three planar planes of full width."""
    import torch
    rng = np.random.default_rng(7 * w + h)
    okw = {k: v for k, v in kw.items() if k != "variant"}
    if kw.get("variant"):
        okw["variant"] = oracle.FDCT_SIMD
    if kw["sampling"] == (1, 2):
        # (vertical decimation only: full-width planes, mode 2 == mode 0 for them - the kernel decimates the rows itself either way)
        planes_px = [_smooth(rng.integers(0, 256, (h, w), dtype=np.uint8)) for _ in range(3)]
        full = np.stack(planes_px, axis=-1)
        want = oracle.encode_jpeg(full, w, h, oracle.YCBCR, **okw)
        d = [torch.from_numpy(p.copy()).cuda() for p in planes_px]
        planes = [(t.data_ptr(), w, 1, 0) for t in d]
        e = _encoder(binding, kw)
        assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=binding.PLANES_SUBSAMPLED_H) == want
        assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=binding.PLANES_FULL) == want
        e.close()
        return
    cw = -(-w // 2)
    y = _smooth(rng.integers(0, 256, (h, 2 * cw), dtype=np.uint8))
    cb = _smooth(rng.integers(0, 256, (h, cw), dtype=np.uint8))
    cr = _smooth(rng.integers(0, 256, (h, cw), dtype=np.uint8))
    cb[-1] ^= 0x55                                                      # the last row differs from the one above it: the bottom edge shows which one is repeated
    cr[-1] ^= 0x2A
    full = np.stack([y[:, :w], _rep(cb, 2, 1, w, h), _rep(cr, 2, 1, w, h)], axis=-1)
    want = oracle.encode_jpeg(full, w, h, oracle.YCBCR, **okw)
    pitch = 4 * cw + 20
    for fmt in (binding.SURFACE_YUYV, binding.SURFACE_UYVY):
        packed = np.zeros((h, pitch), dtype=np.uint8)
        quad = [y[:, 0::2], cb, y[:, 1::2], cr] if fmt == binding.SURFACE_YUYV else [cb, y[:, 0::2], cr, y[:, 1::2]]
        packed[:, :4 * cw] = np.stack(quad, axis=-1).reshape(h, 4 * cw)
        d = torch.from_numpy(packed).cuda()
        planes, _ = binding.packed_planes(fmt, [d.data_ptr()], [pitch])
        for device_entropy in (True, False):
            e = _encoder(binding, kw)
            e.set_device_entropy(device_entropy)
            assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=binding.PLANES_SUBSAMPLED_H) == want, (fmt, device_entropy)
            if w * h <= 40000:
                assert e.encode_image(binding.J_YCBCR, w, h, lambda r: [full[r, :, 0], full[r, :, 1], full[r, :, 2]]) == want
            # a pool: two distinct buffers, the second with another pitch (its bottom-edge row lies elsewhere: a layout of its own)
            pitch2 = pitch + 16
            packed2 = np.zeros((h, pitch2), dtype=np.uint8)
            packed2[:, :4 * cw] = packed[::-1, :4 * cw]
            d2 = torch.from_numpy(packed2).cuda()
            planes2, _ = binding.packed_planes(fmt, [d2.data_ptr()], [pitch2])
            files = e.encode_planes_batch_device(binding.J_YCBCR, w, h, [planes, planes2, planes, planes2], planes_subsampled=binding.PLANES_SUBSAMPLED_H)
            want2 = oracle.encode_jpeg(full[::-1].copy(), w, h, oracle.YCBCR, **okw)
            assert files == [want, want2, want, want2]
            e.close()
    # what the call refuses
    with binding.Encoder(80) as e:
        e.set_sampling_factor(binding.F_4_1)
        with pytest.raises(binding.JpegEncError) as err:
            e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=binding.PLANES_SUBSAMPLED_H)
        assert err.value.status == binding.ERR_INVALID_ARGUMENT
        e.set_sampling_factor(binding.F_2_2)
        with pytest.raises(binding.JpegEncError):
            e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=3)


@pytest.mark.parametrize("w,h", [(16, 16), (37, 21), (258, 128), (1920, 1080)])
@pytest.mark.parametrize("fmt", ["p010", "p016", "i010", "nv21", "yv12"])
@pytest.mark.parametrize("kw", [dict(quality=88, sampling=(2, 2)), dict(quality=80, sampling=(2, 2), optimize=True), dict(quality=93, sampling=(2, 2), restart_interval=2, variant=1)],
                         ids=["baseline", "optimised", "restart-simd"])
def test_deep_and_swapped_420_surfaces(binding, oracle, w, h, fmt, kw):
    """P010 / P016 (16-bit words, value in the high bits, interleaved chroma: pixel_stride 4, shift 8), planar 10-bit with the
    value in the low bits (pixel_stride 2, shift 2), and the swapped-chroma layouts NV21 / YV12: the 8-bit sample is eight bits
    of the word, nothing is unpacked on the host.  Expected: the oracle on the 8-bit samples those bits spell."""
    import torch
    rng = np.random.default_rng(w * 5 + h + len(fmt))
    cw, ch = -(-w // 2), -(-h // 2)
    y8 = _smooth(rng.integers(0, 256, (h, w), dtype=np.uint8))
    cb8 = _smooth(rng.integers(0, 256, (ch, cw), dtype=np.uint8))
    cr8 = _smooth(rng.integers(0, 256, (ch, cw), dtype=np.uint8))
    full = np.stack([y8, _rep(cb8, 2, 2, w, h), _rep(cr8, 2, 2, w, h)], axis=-1)
    okw = {k: v for k, v in kw.items() if k != "variant"}
    if kw.get("variant"):
        okw["variant"] = oracle.FDCT_SIMD
    want = oracle.encode_jpeg(full, w, h, oracle.YCBCR, **okw)

    def deep(a, bits, msb):          # 8-bit samples -> 16-bit words whose eight sample bits are a, the rest noise
        low = rng.integers(0, 1 << (bits - 8), a.shape, dtype=np.uint16) if bits > 8 else 0
        v = (a.astype(np.uint16) << (bits - 8)) | low
        if msb:
            return (v << (16 - bits)) | (rng.integers(0, 1 << (16 - bits), a.shape, dtype=np.uint16) if bits < 16 else 0)
        return v | (rng.integers(0, 1 << (16 - bits), a.shape, dtype=np.uint16) << bits)      # garbage ABOVE the value must not leak
    keep = []
    if fmt in ("p010", "p016"):
        bits = 10 if fmt == "p010" else 16
        ypitch, cpitch = 2 * w + 6, 4 * cw + 8
        yb = np.zeros((h, ypitch), np.uint8); yb[:, :2 * w] = deep(y8, bits, True).view(np.uint8).reshape(h, 2 * w)
        uv = np.stack([deep(cb8, bits, True), deep(cr8, bits, True)], axis=-1)
        cbuf = np.zeros((ch, cpitch), np.uint8); cbuf[:, :4 * cw] = np.ascontiguousarray(uv).view(np.uint8).reshape(ch, 4 * cw)
        keep = [torch.from_numpy(yb).cuda(), torch.from_numpy(cbuf).cuda()]
        planes, sampling = binding.packed_planes(binding.SURFACE_P010 if fmt == "p010" else binding.SURFACE_P016, [t.data_ptr() for t in keep], [ypitch, cpitch])
        assert [(p[2], p[4]) for p in planes] == [(2, 8), (4, 8), (4, 8)]
    elif fmt == "i010":
        ypitch, cpitch = 2 * w + 2, 2 * cw + 10
        bufs = []
        for a, pitch in ((y8, ypitch), (cb8, cpitch), (cr8, cpitch)):
            b = np.zeros((a.shape[0], pitch), np.uint8); b[:, :2 * a.shape[1]] = deep(a, 10, False).view(np.uint8).reshape(a.shape[0], 2 * a.shape[1])
            bufs.append(b)
        keep = [torch.from_numpy(b).cuda() for b in bufs]
        planes, sampling = binding.packed_planes(binding.SURFACE_I010, [t.data_ptr() for t in keep], [ypitch, cpitch, cpitch])
        assert [(p[2], p[4]) for p in planes] == [(2, 2)] * 3
    elif fmt == "nv21":
        vu = np.ascontiguousarray(np.stack([cr8, cb8], axis=-1))
        keep = [torch.from_numpy(y8).cuda(), torch.from_numpy(vu).cuda()]
        planes, sampling = binding.packed_planes(binding.SURFACE_NV21, [t.data_ptr() for t in keep], [w, 2 * cw])
    else:
        keep = [torch.from_numpy(y8).cuda(), torch.from_numpy(cr8).cuda(), torch.from_numpy(cb8).cuda()]          # Y, V, U
        planes, sampling = binding.packed_planes(binding.SURFACE_YV12, [t.data_ptr() for t in keep], [w, cw, cw])
    assert sampling == binding.F_2_2
    for device_entropy in (True, False):
        e = _encoder(binding, kw)
        e.set_device_entropy(device_entropy)
        assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=True) == want, (fmt, device_entropy)
        files = e.encode_planes_batch_device(binding.J_YCBCR, w, h, [planes] * 3, planes_subsampled=True)
        assert files == [want] * 3
        e.close()


def test_plane_descriptor_validation(binding):
    import torch
    d = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = d.data_ptr()
    with binding.Encoder(80) as e:
        for planes, why in (([(p, 64, 3, 0)] * 3, "stride 3"), ([(p, 64, 1, 0, 8)] * 3, "shift on bytes"), ([(p + 1, 64, 2, 0, 8)] * 3, "odd 16-bit samples"),
                            ([(p, 64, 4, 0, 2)] * 3, "shift 2 with stride 4"), ([(p, 64, 2, 0, 9)] * 3, "shift 9")):
            with pytest.raises(binding.JpegEncError) as err:
                e.encode_planes_device(binding.J_YCBCR, 16, 16, planes)
            assert err.value.status == binding.ERR_INVALID_ARGUMENT, why
    with pytest.raises(binding.JpegEncError):
        binding.packed_planes(99, [p], [64])
