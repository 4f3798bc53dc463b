"""BASELINE config 3 as a BATCH, and the multi-device batch entry point, through the C ABI against the oracle
(byte-identical files).  Run with `pytest -m gpu` on an MI355X; the box has one GPU, so the multi-device path is
exercised with the same device listed twice (two independent worker sets) - the sharding rule, the per-shard
encoders and the result plumbing are the code under test; the 1 -> 8 scaling itself is the driver's SCALE run."""
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


@pytest.fixture(scope="module")
def c3(pkg, synth, oracle):
    """The 125 frames one GPU of eight gets from the 1000-frame batch, and the oracle's files for them."""
    batch = importlib.import_module("jpeg_encoder_amd.batch")
    pool = batch.FramePool(synth)
    want = {k: oracle.encode_jpeg(pool(k), batch.C3_W, batch.C3_H, oracle.RGB, batch.C3_QUALITY) for k in range(batch.POOL)}
    return batch, pool, want


def test_config3_batch_125_frames_byte_identical(binding, c3):
    """125 frames of 1920x1080 RGB q=80 4:2:0 through jpegenc_encoder_encode_batch_to_buffers (worker pool, one
    stream per worker, replayed launch sequences), every file compared with the oracle's."""
    batch, pool, want = c3
    mine = binding.shard_frames(batch.C3_FRAMES, 8, 3)                 # rank 3 of 8
    assert len(mine) == 125
    frames = [pool(k) for k in mine]
    with binding.Encoder(batch.C3_QUALITY) as enc:
        outs = [np.empty(2 << 20, dtype=np.uint8) for _ in frames]
        for _ in range(2):                                             # second pass: captured graphs replayed
            lens = enc.encode_batch_into(frames, batch.C3_W, batch.C3_H, binding.RGB, outs)
            for i, k in enumerate(mine):
                assert outs[i][:lens[i]].tobytes() == want[k % batch.POOL], f"frame {k}"


def test_run_sharded_batch_on_the_gpu(binding, c3):
    """bench.py's c3_batch leg (jpeg_encoder_amd.batch.run_sharded_batch) with the library as the encoder."""
    batch, pool, want = c3
    with binding.Encoder(batch.C3_QUALITY) as enc:
        outs = [np.empty(2 << 20, dtype=np.uint8) for _ in range(64)]

        def encode_frames(frames):
            lens = enc.encode_batch_into(frames, batch.C3_W, batch.C3_H, binding.RGB, outs)
            return [outs[i][:lens[i]] for i in range(len(frames))]
        result, files = batch.run_sharded_batch(binding, encode_frames, pool, 64, batch.C3_W, batch.C3_H, warmup_frames=8)
    assert result["frames"] == 64 and result["per_rank_frames"] == [64] and result["frames_per_s"] > 0
    for k, f in files.items():
        assert bytes(f) == want[k % batch.POOL]


def test_batch_frames_in_page_locked_memory_are_uploaded_in_place(binding, c3):
    """Frames from jpegenc_host_alloc, from a registered numpy array and from pageable memory in ONE batch (the workers
    stage only the pageable ones); single- and multi-device entry points; unregistering returns the range to pageable."""
    batch, pool, want = c3
    fb = batch.C3_W * batch.C3_H * 3
    hb = binding.HostBuffer(8 * fb)
    reg = np.empty(8 * fb + 4096, dtype=np.uint8)[37:37 + 8 * fb]        # an unaligned range of the caller's own memory
    binding.host_register(reg)
    try:
        frames = []
        for k in range(48):
            src = pool(k).reshape(-1)
            if k % 3 == 0:
                v = hb.array[(k % 8) * fb:(k % 8 + 1) * fb]; v[:] = pool(k % 8).reshape(-1); src = v
            elif k % 3 == 1:
                v = reg[(k % 8) * fb:(k % 8 + 1) * fb]; v[:] = pool(k % 8).reshape(-1); src = v
            frames.append(src)
        keys = [k % 8 if k % 3 != 2 else k % batch.POOL for k in range(48)]
        with binding.Encoder(batch.C3_QUALITY) as enc:
            outs = [np.empty(2 << 20, dtype=np.uint8) for _ in frames]
            for devices in (None, [0, 0]):
                lens = enc.encode_batch_into(frames, batch.C3_W, batch.C3_H, binding.RGB, outs, devices=devices)
                for i, key in enumerate(keys):
                    assert outs[i][:lens[i]].tobytes() == want[key], f"frame {i} devices {devices}"
    finally:
        binding.host_unregister(reg)
        hb.close()
    # a registration that ends inside the frame: treated as pageable (staged), same bytes
    half = np.empty(fb, dtype=np.uint8)
    half[:] = pool(3).reshape(-1)
    binding.host_register(half[:fb // 2])
    try:
        with binding.Encoder(batch.C3_QUALITY) as enc:
            outs = [np.empty(2 << 20, dtype=np.uint8) for _ in range(4)]
            lens = enc.encode_batch_into([half] * 4, batch.C3_W, batch.C3_H, binding.RGB, outs)
            assert all(outs[i][:lens[i]].tobytes() == want[3] for i in range(4))
    finally:
        binding.host_unregister(half[:fb // 2])
    with pytest.raises(binding.JpegEncError):
        binding.host_unregister(reg)                                      # not registered any more
    assert binding.lib().jpegenc_host_free(None) == 0


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_encode_batch_multi_shards(binding, c3, devices):
    """jpegenc_encoder_encode_batch_multi_to_buffers: frame k -> devices[k % n]; same files as the oracle whatever
    the device list; the handle keeps its shards across calls."""
    batch, pool, want = c3
    frames = [pool(k) for k in range(37)]
    with binding.Encoder(batch.C3_QUALITY) as enc:
        for _ in range(2):
            files = enc.encode_batch_multi_to_buffers(devices, frames, batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
            assert [f == want[k % batch.POOL] for k, f in enumerate(files)] == [True] * 37
        # another configuration through the same handle and shards: progressive + optimised (per-frame tables)
        enc.set_progressive(True)
        enc.set_optimized_huffman_tables(True)
        small = [np.ascontiguousarray(f[:120, :200]) for f in frames[:7]]
        files = enc.encode_batch_multi_to_buffers(devices, small, 200, 120, binding.RGB, 1 << 20)
    from oracle import pyoracle
    for f, px in zip(files, small):
        assert f == pyoracle.encode_jpeg(px, 200, 120, pyoracle.RGB, batch.C3_QUALITY, progressive_scans=4, optimize=True)


def test_encode_batch_multi_on_distinct_devices(binding, c3):
    """One process driving SEVERAL GPUs (skipped on the one-GPU boxes): every device of the node gets its own child encoder,
    worker set, streams, pinned staging and default code tables (jpegenc_encoder_encode_batch_multi, host_multi.cpp) - the
    125-frame shard size of config 3 per device, the files compared with the oracle's, twice (the second pass replays the
    captured launch sequences of every device), then with the worker threads bound to each device's NUMA node."""
    n_dev = min(binding.device_count(), 8)
    if n_dev < 2:
        pytest.skip("needs at least two GPUs in one process")
    batch, pool, want = c3
    devices = list(range(n_dev))
    frames = [pool(k) for k in range(125 * n_dev if n_dev <= 2 else 64 * n_dev)]
    with binding.Encoder(batch.C3_QUALITY) as enc:
        for bind in (False, False, True):
            enc.set_numa_bind(bind)
            files = enc.encode_batch_multi_to_buffers(devices, frames, batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
            assert [f == want[k % batch.POOL] for k, f in enumerate(files)] == [True] * len(frames)
        # a device list in another order, and with a device twice: frame k -> devices[k % n] whatever the list says
        for devs in (devices[::-1], devices + devices[:1]):
            files = enc.encode_batch_multi_to_buffers(devs, frames[:37], batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
            assert [f == want[k % batch.POOL] for k, f in enumerate(files)] == [True] * 37
    # every device on its own, from its own handle and thread, at the same time (the one-process-per-GPU shape inside one process)
    import threading
    errors = []

    def one(dev):
        try:
            with binding.Encoder(batch.C3_QUALITY, device=dev) as e:
                outs = [np.empty(2 << 20, dtype=np.uint8) for _ in range(32)]
                lens = e.encode_batch_into(frames[:32], batch.C3_W, batch.C3_H, binding.RGB, outs)
                for i in range(32):
                    assert outs[i][:lens[i]].tobytes() == want[i % batch.POOL], (dev, i)
        except Exception as exc:                                       # noqa: BLE001 - reported below
            errors.append((dev, repr(exc)))
    threads = [threading.Thread(target=one, args=(d,)) for d in devices]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_encode_batch_multi_sink_and_errors(binding, c3, synth):
    import ctypes as C
    batch, pool, want = c3
    frames = [pool(k) for k in range(5)]
    with binding.Encoder(batch.C3_QUALITY) as enc:
        # callback form: per-frame users, calls of one frame in order
        outs = [[] for _ in frames]

        def sink(user, ptr, n):
            outs[user or 0].append(C.string_at(ptr, n))
            return 0
        cb = binding.WRITE_FN(sink)
        ptrs = (C.c_void_p * 5)(*[f.ctypes.data for f in frames])
        users = (C.c_void_p * 5)(*range(5))
        devs = (C.c_int * 2)(0, 0)
        binding.check(binding.lib().jpegenc_encoder_encode_batch_multi(enc._h, devs, 2, ptrs, frames[0].size, 5, batch.C3_W, batch.C3_H,
                                                                       binding.RGB, cb, users))
        assert [b"".join(o) for o in outs] == [want[k] for k in range(5)]
        # an unknown device is reported before any work, like Encoder::encode's validation
        with pytest.raises(binding.JpegEncError) as e:
            enc.encode_batch_multi_to_buffers([0, 99], frames, batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
        assert e.value.status == binding.ERR_NO_DEVICE
        with pytest.raises(binding.JpegEncError) as e:
            enc.encode_batch_multi_to_buffers([], frames, batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
        assert e.value.status == binding.ERR_INVALID_ARGUMENT
        with pytest.raises(binding.JpegEncError) as e:                 # BadImageData before any device work
            enc.encode_batch_multi_to_buffers([0, 0], [f[:100] for f in frames], batch.C3_W, batch.C3_H, binding.RGB, 2 << 20)
        assert e.value.status == binding.ERR_BAD_IMAGE_DATA
        assert enc.encode_batch_multi_to_buffers([0, 0], [], batch.C3_W, batch.C3_H, binding.RGB, 16) == []


def test_batch_device_falls_back_for_frames_the_device_coder_declines(binding, oracle, synth):
    """A frame of more than ~2.45 M blocks (jpegenc_scan_max_bytes == 0): the device-resident batch entry point must
    encode it like jpegenc_encoder_encode_device does (host entropy coder) instead of failing - same bytes."""
    import torch
    w, h = 16384, 10000                                                 # luma: 2 560 000 blocks
    L = binding.layout(w, h, binding.LUMA, 1, 1, binding.ORDER_MCU)
    assert binding.scan_max_bytes(L, binding.baseline_scan()) == 0
    px = np.ascontiguousarray(np.tile(synth.test_img_gray(256, 250), (40, 64)))     # 10000 x 16384
    assert px.shape == (h, w)
    d = torch.from_numpy(np.stack([px, px[::-1].copy()])).cuda()
    with binding.Encoder(90) as enc:
        files = enc.encode_batch_device(d.data_ptr(), w * h, 2, w, h, binding.LUMA)
        one = enc.encode_device(d.data_ptr(), w, h, binding.LUMA)
    assert files[0] == one == oracle.encode_jpeg(px, w, h, oracle.LUMA, 90)
    assert files[1] == oracle.encode_jpeg(px[::-1].copy(), w, h, oracle.LUMA, 90)


@pytest.mark.parametrize("ct,hs,vs,w,h,restart", [
    (1, 2, 2, 258, 128, 0), (1, 1, 1, 258, 128, 0), (1, 2, 1, 515, 77, 0), (1, 1, 2, 77, 515, 7), (2, 2, 2, 333, 201, 0),
    (3, 2, 2, 1920, 1080, 0), (4, 1, 1, 640, 360, 40), (1, 2, 2, 37, 21, 1), (1, 2, 2, 8, 8, 0), (1, 2, 2, 3840, 2160, 240),
    (6, 1, 1, 200, 120, 0), (0, 1, 1, 200, 120, 0), (5, 2, 2, 200, 120, 5), (0, 1, 1, 1100, 700, 33), (5, 2, 1, 1030, 515, 0),
    (6, 2, 1, 700, 300, 4), (8, 1, 1, 1920, 1080, 0), (7, 1, 1, 515, 301, 2), (7, 2, 2, 515, 301, 0), (6, 2, 2, 333, 201, 3)],
    ids=["rgb420", "rgb444", "rgb422", "rgb440-rst7", "rgba420", "bgr-1080p", "bgra444-rst40", "rgb420-tiny-rst1", "rgb-one-mcu",
         "rgb-4k-rst240", "cmyk", "luma", "ycbcr420-rst5", "luma-large-rst33", "ycbcr422", "cmyk-2x1-rst4", "ycck-1080p",
         "cmyk-as-ycck-rst2", "cmyk-as-ycck-420-unfused", "cmyk-420-unfused"])
def test_pixels_scan_device_matches_the_two_kernel_path_and_the_oracle(binding, oracle, synth, ct, hs, vs, w, h, restart):
    """jpegenc_pixels_scan_device (ONE fused kernel from pixels to coded runs for the RGB family) against
    jpegenc_blocks_device + jpegenc_scan_device, and against the scan bytes inside the oracle's file."""
    import torch
    bpp = binding.BPP[ct]
    n = 3
    px = np.stack([synth.lcg_image(w, h, bpp, 5 + i) if i != 1 else
                   np.ascontiguousarray(np.broadcast_to((np.add.outer(np.arange(h), np.arange(w)) // 3 % 256).astype(np.uint8)[..., None], (h, w, bpp)))
                   for i in range(n)])
    d_px = torch.from_numpy(px).cuda()
    L = binding.layout(w, h, ct, hs, vs, binding.ORDER_MCU)
    scan = binding.baseline_scan(restart_interval=restart)
    cap, wsz = binding.scan_max_bytes(L, scan), binding.scan_workspace_size(L, scan, n)
    q = binding.qtables(85)
    nblk = int(L.total_blocks)
    d_co = torch.empty((n, nblk * 64), dtype=torch.int16, device="cuda")
    d_ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    outs = []
    fused = binding.pixels_scan_fused(w, h, ct, hs, vs)
    assert fused == (3 <= int(L.total_blocks) // int(L.mcus) <= 6)   # one kernel for 3 to 6 blocks per MCU (4-component 2x2 layouts have 7 or 10)
    for which in ("pixels", "two"):
        d_out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
        d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
        if which == "pixels":
            binding.pixels_scan_device(d_px.data_ptr(), w * h * bpp, n, w, h, ct, hs, vs, q, d_out.data_ptr(), cap, d_len.data_ptr(),
                                       d_ws.data_ptr(), wsz, restart_interval=restart,
                                       d_coeffs_ptr=None if fused else d_co.data_ptr())
        else:
            binding.blocks_device(d_px.data_ptr(), w * h * bpp, n, w, h, ct, hs, vs, q, binding.ORDER_MCU, binding.FDCT_SCALAR,
                                  d_co.data_ptr(), nblk)
            binding.scan_device(d_co.data_ptr(), nblk, n, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), wsz)
        torch.cuda.synchronize()
        outs.append([bytes(d_out[i, :int(d_len[i])].cpu().numpy()) for i in range(n)])
    assert outs[0] == outs[1]
    for i in range(n):                                                  # the scan is the tail of the oracle's file: ... SOS header | scan | EOI
        ref = oracle.encode_jpeg(px[i], w, h, ct, 85, sampling=(hs, vs), restart_interval=restart)
        assert ref.endswith(outs[0][i] + b"\xff\xd9") and len(outs[0][i]) > 0


@pytest.mark.parametrize("env", [{"JPEGENC_FUSED": "1"}, {"JPEGENC_FUSED": "0"}, {"JPEGENC_FUSED": "1", "JPEGENC_PACK_WINDOW_WORDS": "8"},
                                 {"JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES": "0", "JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES": "0"},
                                 {"JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES": "100000000", "JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES": "100000000"},
                                 {"JPEGENC_NO_FINISH": "1"}, {"JPEGENC_NO_DONE_FLAG": "1"}, {"JPEGENC_FORCE_FINISH_GAVE_UP": "1"},
                                 {"JPEGENC_FORCE_FINISH_GAVE_UP": "1", "JPEGENC_PACK_WINDOW_WORDS": "8"}],
                         ids=["fused", "two-kernels", "fused-tiny-window", "dma-both-ways", "zero-copy-both-ways-any-size",
                              "separate-push-and-stuff", "stream-wait", "finish-gave-up", "finish-gave-up-tiny-window"])
def test_encoder_with_and_without_the_fused_kernel(binding, tmp_path, env):
    """The Encoder codes its interleaved baseline scan of an RGB-family image straight from the pixels (one workgroup =
    one run of 64 MCUs; JPEGENC_FUSED=0 keeps block kernel + coder; the switches are read once per process, hence the
    child process): single frames (direct, captured and replayed launch sequences), restart intervals, 4-byte pixels,
    blocks longer than a lane's strip (noise at quality 100) and runs longer than the window (forced by
    JPEGENC_PACK_WINDOW_WORDS: the second-walk path), the worker-pool batch and the device-resident batch -
    byte-identical to the oracle's files every way.  Small frames are read from and coded into pinned host memory by
    the kernels themselves (no DMA nodes); two runs force the DMA path and the zero-copy path for every size.  A single frame of
    up to 512 runs without restart markers is finished by the kernel's own workgroups (finish_run.hip.h - the default in every
    run above): the last four runs keep k_push / k_stuff as separate launches, make the host wait for the stream instead of the
    kernel's flag, and pretend that a workgroup gave up waiting on every second frame (the frame is then coded again through
    the ordinary sequence)."""
    import os
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "fused_child.py"
    script.write_text(textwrap.dedent(f"""
        import importlib, sys
        sys.path.insert(0, {root!r})
        import numpy as np, torch
        import __graft_entry__ as ge
        ge.load_package()
        b = importlib.import_module("jpeg_encoder_amd.binding")
        synth = importlib.import_module("jpeg_encoder_amd.synth")
        from oracle import pyoracle as o
        cases = [(b.RGB, 3, 258, 128, dict(quality=80)), (b.RGB, 3, 258, 128, dict(quality=100)), (b.RGB, 3, 515, 77, dict(quality=85, sampling=(2, 1))),
                 (b.BGR, 3, 77, 515, dict(quality=70, sampling=(1, 2), restart_interval=3)), (b.RGBA, 4, 640, 360, dict(quality=90, sampling=(2, 2))),
                 (b.BGRA, 4, 333, 201, dict(quality=60, restart_interval=1)), (b.RGB, 3, 1920, 1080, dict(quality=80)),
                 (b.RGB, 3, 3840, 2160, dict(quality=90, sampling=(2, 2), restart_interval=240)),
                 (b.RGB, 3, 1920, 1080, dict(quality=85, variant=o.FDCT_SIMD)), (b.BGRA, 4, 515, 301, dict(quality=95, sampling=(1, 1), variant=o.FDCT_SIMD)),
                 (b.RGB, 3, 1030, 70, dict(quality=100, sampling=(1, 1))), (b.RGB, 3, 1000, 200, dict(quality=100, sampling=(2, 2), restart_interval=7)),
                 (b.RGB, 3, 96, 80, dict(quality=95, variant=o.FDCT_SIMD))]
        for i, (ct, bpp, w, h, kw) in enumerate(cases):
            px = synth.lcg_image(w, h, bpp, 9)
            if i < len(cases) - 3:                               # (the last three stay pure noise: blocks of up to ~1 700 bits)
                px = (px.astype(np.int16) // 4 + np.add.outer(np.arange(h), np.arange(w))[..., None] // 3).clip(0, 255).astype(np.uint8)
            e = b.Encoder(kw["quality"])
            if "sampling" in kw: e.set_sampling_factor(b.sampling_factor(*kw["sampling"]))
            if kw.get("restart_interval"): e.set_restart_interval(kw["restart_interval"])
            if kw.get("variant"): e.set_fdct_variant(b.FDCT_SIMD)
            want = o.encode_jpeg(px, w, h, ct, **kw)
            for _ in range(4):                                   # direct, direct, capture, replay
                assert e.encode(px, w, h, ct) == want, (ct, w, h, kw)
            d = torch.from_numpy(np.stack([px, px[::-1].copy(), px])).cuda()
            files = e.encode_batch_device(d.data_ptr(), w * h * bpp, 3, w, h, ct)
            assert files[0] == want and files[2] == want and files[1] == o.encode_jpeg(px[::-1].copy(), w, h, ct, **kw)
            assert e.encode_batch([px] * 5, w, h, ct) == [want] * 5
        print("FUSED-OK")
    """))
    r = subprocess.run([sys.executable, str(script)], env=dict(os.environ, JPEGENC_LIB=binding.DIAG_LIB_PATH, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "FUSED-OK" in r.stdout, r.stderr[-3000:]


@pytest.mark.parametrize("ct", [6, 7, 8], ids=["cmyk", "cmyk-as-ycck", "ycck"])
@pytest.mark.parametrize("hs,vs", [(4, 2), (2, 4), (4, 1), (1, 4)])
def test_four_component_layouts_with_factor_4_in_32_mcu_groups(binding, oracle, synth, ct, hs, vs):
    """4-component layouts with 4x2 / 2x4 sampling need 11 or 18 waves per 64 MCUs: the tuned kernels take them in
    groups of 32 MCUs (waves of a 1x1 component half full).  Sizes with several groups per MCU row, a ragged last
    group and both edges padded; both block orders; a two-frame batch on the device."""
    import torch
    for w, h in ((1500, 260), (333, 517)):
        px = synth.lcg_image(w, h, 4, 3 + ct + hs)
        for order in (0, 1):
            got = binding.blocks_host(px, w, h, ct, hs, vs, 88, order)
            want = oracle.encode_blocks(px, w, h, ct, hs, vs, 88, order)
            assert got.shape == want.shape and np.array_equal(got, want), (ct, hs, vs, w, h, order)
    w, h = 515, 301
    px = np.stack([synth.lcg_image(w, h, 4, 1), synth.lcg_image(w, h, 4, 2)])
    d = torch.from_numpy(px).cuda()
    L = binding.layout(w, h, ct, hs, vs, binding.ORDER_MCU)
    n = int(L.total_blocks)
    d_co = torch.zeros((2, n + 5, 64), dtype=torch.int16, device="cuda")
    binding.blocks_device(d.data_ptr(), w * h * 4, 2, w, h, ct, hs, vs, binding.qtables(70), binding.ORDER_MCU, binding.FDCT_SIMD,
                          d_co.data_ptr(), n + 5)
    torch.cuda.synchronize()
    for f in range(2):
        want = oracle.encode_blocks(px[f], w, h, ct, hs, vs, 70, 0, variant=oracle.FDCT_SIMD)
        assert np.array_equal(d_co[f, :n].cpu().numpy(), want) and not d_co[f, n:].any()


def _replicated(plane, sx, sy, w, h):
    return np.repeat(np.repeat(plane, sy, axis=0), sx, axis=1)[:h, :w]


@pytest.mark.parametrize("w,h", [(258, 128), (515, 301), (1920, 1080)])
@pytest.mark.parametrize("kw", [dict(quality=80), dict(quality=90, sampling=(2, 1)), dict(quality=75, sampling=(1, 2), restart_interval=5),
                                dict(quality=85, sampling=(2, 2), progressive_scans=4, optimize=True), dict(quality=95, sampling=(1, 1), optimize=True)],
                         ids=["420", "422", "440-restart", "420-progressive-optimised", "444-optimised"])
def test_encode_planes_device_yuv_surfaces(binding, oracle, synth, w, h, kw):
    """Device-resident planar sources (jpegenc_encoder_encode_planes_device): I420-style planes (chroma already
    decimated, padded pitches), NV12 (interleaved UV), and full-resolution planes - byte-identical to the oracle fed the
    equivalent interleaved YCbCr image (chroma repeated sx x sy times: what the ImageBuffer of such a surface delivers)."""
    import torch
    hs, vs = kw.get("sampling", (2, 2) if kw["quality"] < 90 else (1, 1))
    rng = np.random.default_rng(w * 7 + h)
    cw, ch = -(-w // hs), -(-h // vs)
    smooth = lambda a: (a.astype(np.int16) // 4 + np.add.outer(np.arange(a.shape[0]), np.arange(a.shape[1])) // 3).clip(0, 255).astype(np.uint8)
    y = smooth(rng.integers(0, 256, (h, w), dtype=np.uint8))
    cb = smooth(rng.integers(0, 256, (ch, cw), dtype=np.uint8))
    cr = smooth(rng.integers(0, 256, (ch, cw), dtype=np.uint8))
    full = np.stack([y, _replicated(cb, hs, vs, w, h), _replicated(cr, hs, vs, w, h)], axis=-1)
    want = oracle.encode_jpeg(full, w, h, oracle.YCBCR, **kw)

    def enc():
        e = binding.Encoder(kw["quality"])
        if "sampling" in kw:
            e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
        if kw.get("progressive_scans"):
            e.set_progressive_scans(kw["progressive_scans"])
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        if kw.get("optimize"):
            e.set_optimized_huffman_tables(True)
        return e
    # (1) I420-like: three planes, chroma decimated, every pitch padded (and odd for the chroma planes)
    ypitch, cpitch = w + 13, cw + 7
    d_y = torch.zeros((h, ypitch), dtype=torch.uint8, device="cuda"); d_y[:, :w] = torch.from_numpy(y).cuda()
    d_cb = torch.zeros((ch, cpitch), dtype=torch.uint8, device="cuda"); d_cb[:, :cw] = torch.from_numpy(cb).cuda()
    d_cr = torch.zeros((ch, cpitch), dtype=torch.uint8, device="cuda"); d_cr[:, :cw] = torch.from_numpy(cr).cuda()
    got = enc().encode_planes_device(binding.J_YCBCR, w, h, [(d_y.data_ptr(), ypitch, 1, 0), (d_cb.data_ptr(), cpitch, 1, 0), (d_cr.data_ptr(), cpitch, 1, 0)],
                                     planes_subsampled=True)
    assert got == want
    # (2) NV12-like: Y plane + one interleaved UV plane
    uv = np.stack([cb, cr], axis=-1)
    d_uv = torch.from_numpy(np.ascontiguousarray(uv)).cuda()
    got = enc().encode_planes_device(binding.J_YCBCR, w, h, [(d_y.data_ptr(), ypitch, 1, 0), (d_uv.data_ptr(), cw * 2, 2, 0), (d_uv.data_ptr() + 1, cw * 2, 2, 0)],
                                     planes_subsampled=True)
    assert got == want
    # (3) full-resolution planes (what fill_buffers delivers): the encoder decimates
    d_full = torch.from_numpy(np.ascontiguousarray(full.transpose(2, 0, 1))).cuda()
    got = enc().encode_planes_device(binding.J_YCBCR, w, h, [(d_full[c].data_ptr(), w, 1, 0) for c in range(3)])
    assert got == want
    # (4) the same from the interleaved image itself: three "planes" of pixel stride ... 3 is not a plane stride: rejected
    with pytest.raises(binding.JpegEncError) as err:
        enc().encode_planes_device(binding.J_YCBCR, w, h, [(d_y.data_ptr(), w * 3, 3, 0)] * 3)
    assert err.value.status == binding.ERR_INVALID_ARGUMENT


def test_planes_per_plane_launch_path_still_matches(binding):
    """The described planar sources take ONE launch (every wave on its own plane) and, for interleaved baseline scans, the
    one-kernel pixels -> bits path; sampling factors of 4 keep one block-kernel launch per plane.  That older path is
    forced here for every planes test (the switches are read once per process: a child pytest)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, JPEGENC_PLANES_PER_PLANE_LAUNCHES="1", JPEGENC_FUSED="0", JPEGENC_LIB=binding.DIAG_LIB_PATH)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-k", "encode_planes_device"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_encode_planes_device_luma_and_cmyk(binding, oracle, synth):
    import torch
    w, h = 333, 201
    g = synth.test_img_gray(w, h)
    d = torch.zeros((h, w + 3), dtype=torch.uint8, device="cuda"); d[:, :w] = torch.from_numpy(g).cuda()
    with binding.Encoder(85) as e:
        assert e.encode_planes_device(binding.J_LUMA, w, h, [(d.data_ptr(), w + 3, 1, 0)]) == oracle.encode_jpeg(g, w, h, oracle.LUMA, 85)
    cmyk = synth.test_img_cmyk(w, h)
    dp = torch.from_numpy(np.ascontiguousarray(cmyk.transpose(2, 0, 1))).cuda()
    for kw in (dict(quality=90), dict(quality=70, sampling=(2, 2), restart_interval=9), dict(quality=80, sampling=(2, 1), optimize=True)):
        e = binding.Encoder(kw["quality"])
        if "sampling" in kw:
            e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        if kw.get("optimize"):
            e.set_optimized_huffman_tables(True)
        # CmykImage inverts every channel (image_buffer.rs:247-256): planar CMYK with invert = 1 is ColorType::Cmyk
        got = e.encode_planes_device(binding.J_CMYK, w, h, [(dp[c].data_ptr(), w, 1, 1) for c in range(4)])
        assert got == oracle.encode_jpeg(cmyk, w, h, oracle.CMYK, **kw), kw
        # and YCCK planes as they are = ColorType::Ycck
        got = e.encode_planes_device(binding.J_YCCK, w, h, [(dp[c].data_ptr(), w, 1, 0) for c in range(4)])
        assert got == oracle.encode_jpeg(cmyk, w, h, oracle.YCCK, **kw), kw


@pytest.mark.parametrize("quality,hs,vs", [(100, 1, 1), (100, 2, 2), (97, 2, 2)])
def test_long_blocks_take_the_lds_second_walk(binding, oracle, synth, quality, hs, vs):
    """Noise at quality 97-100: blocks of up to ~1 700 bits - longer than a lane's 512-bit strip - and runs longer than the
    LDS window, i.e. the second walk into window + strips (and, for the longest runs, into the HBM slot), at a size with
    hundreds of groups: jpegenc_pixels_scan_device against jpegenc_blocks_device + jpegenc_scan_device and the oracle."""
    import torch
    w, h, n = 1920, 1080, 2
    rng = np.random.default_rng(quality + hs)
    px = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    px[1, :, : w // 2] = synth.test_img_rgb(w, h)[:, : w // 2]              # half of the second frame smooth: short and long runs side by side
    d_px = torch.from_numpy(px).cuda()
    L = binding.layout(w, h, binding.RGB, hs, vs, binding.ORDER_MCU)
    scan = binding.baseline_scan()
    cap, wsz = binding.scan_max_bytes(L, scan), binding.scan_workspace_size(L, scan, n)
    q = binding.qtables(quality)
    nblk = int(L.total_blocks)
    d_co = torch.empty((n, nblk * 64), dtype=torch.int16, device="cuda")
    d_ws = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    outs = []
    for which in ("one-kernel", "two"):
        d_out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
        d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
        if which == "one-kernel":
            binding.pixels_scan_device(d_px.data_ptr(), w * h * 3, n, w, h, binding.RGB, hs, vs, q, d_out.data_ptr(), cap, d_len.data_ptr(),
                                       d_ws.data_ptr(), wsz)
        else:
            binding.blocks_device(d_px.data_ptr(), w * h * 3, n, w, h, binding.RGB, hs, vs, q, binding.ORDER_MCU, binding.FDCT_SCALAR,
                                  d_co.data_ptr(), nblk)
            binding.scan_device(d_co.data_ptr(), nblk, n, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), wsz)
        torch.cuda.synchronize()
        outs.append([bytes(d_out[i, :int(d_len[i])].cpu().numpy()) for i in range(n)])
    assert outs[0] == outs[1]
    for i in range(n):
        ref = oracle.encode_jpeg(px[i], w, h, oracle.RGB, quality, sampling=(hs, vs))
        assert ref.endswith(outs[0][i] + b"\xff\xd9") and len(outs[0][i]) > 1_000_000


@pytest.mark.parametrize("kw", [dict(quality=80), dict(quality=90, sampling=(2, 1), restart_interval=4), dict(quality=85, sampling=(2, 2), progressive_scans=4),
                                dict(quality=88, sampling=(2, 2), optimize=True), dict(quality=70, sampling=(4, 1))],
                         ids=["420", "422-rst4", "420-progressive", "420-optimised-one-by-one", "411-one-by-one"])
def test_encode_planes_batch_device(binding, oracle, synth, kw):
    """jpegenc_encoder_encode_planes_batch_device: a pool of I420 surfaces anywhere in device memory (padded pitches) and a
    pool of NV12 surfaces, five frames per call sharing their launches (plane addresses through a device table) - each file
    byte-identical to the oracle fed the equivalent interleaved YCbCr image; a frame whose pitch differs from the others'
    stays in the shared launches (addresses AND pitches are per frame in the table), one whose sample stride differs sends
    the batch down the one-frame-at-a-time path - same bytes."""
    import torch
    w, h, n = 515, 301, 5
    hs, vs = kw.get("sampling", (2, 2) if kw["quality"] < 90 else (1, 1))
    cw, ch = -(-w // hs), -(-h // vs)
    rng = np.random.default_rng(11)
    smooth = lambda a: (a.astype(np.int16) // 4 + np.add.outer(np.arange(a.shape[0]), np.arange(a.shape[1])) // 3).clip(0, 255).astype(np.uint8)
    okw = dict(kw)
    e = binding.Encoder(kw["quality"])
    if "sampling" in kw:
        e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
    if kw.get("progressive_scans"):
        e.set_progressive_scans(kw["progressive_scans"])
    if kw.get("restart_interval"):
        e.set_restart_interval(kw["restart_interval"])
    if kw.get("optimize"):
        e.set_optimized_huffman_tables(True)
    ypitch, cpitch = w + 13, cw + 7
    keep, i420, nv12, want = [], [], [], []
    for f in range(n):
        y, cb, cr = (smooth(rng.integers(0, 256, s, dtype=np.uint8)) for s in ((h, w), (ch, cw), (ch, cw)))
        full = np.stack([y, _replicated(cb, hs, vs, w, h), _replicated(cr, hs, vs, w, h)], axis=-1)
        want.append(oracle.encode_jpeg(full, w, h, oracle.YCBCR, **okw))
        d_y = torch.zeros((h, ypitch), dtype=torch.uint8, device="cuda"); d_y[:, :w] = torch.from_numpy(y).cuda()
        d_cb = torch.zeros((ch, cpitch), dtype=torch.uint8, device="cuda"); d_cb[:, :cw] = torch.from_numpy(cb).cuda()
        d_cr = torch.zeros((ch, cpitch), dtype=torch.uint8, device="cuda"); d_cr[:, :cw] = torch.from_numpy(cr).cuda()
        d_uv = torch.from_numpy(np.ascontiguousarray(np.stack([cb, cr], axis=-1))).cuda()
        keep += [d_y, d_cb, d_cr, d_uv]
        i420.append([(d_y.data_ptr(), ypitch, 1, 0), (d_cb.data_ptr(), cpitch, 1, 0), (d_cr.data_ptr(), cpitch, 1, 0)])
        nv12.append([(d_y.data_ptr(), ypitch, 1, 0), (d_uv.data_ptr(), cw * 2, 2, 0), (d_uv.data_ptr() + 1, cw * 2, 2, 0)])
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, i420, planes_subsampled=True) == want
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, nv12, planes_subsampled=True) == want
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, i420[:1], planes_subsampled=True) == want[:1]
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, [], planes_subsampled=True) == []
    # one frame with another luma pitch and one with other chroma pitches: per-frame pitches in the table
    y3 = torch.zeros((h, w + 64), dtype=torch.uint8, device="cuda"); y3[:, :w] = keep[4 * 3][:, :w]
    cb1 = torch.zeros((ch, cw + 1), dtype=torch.uint8, device="cuda"); cb1[:, :cw] = keep[4 * 1 + 1][:, :cw]
    cr1 = torch.zeros((ch, cw + 33), dtype=torch.uint8, device="cuda"); cr1[:, :cw] = keep[4 * 1 + 2][:, :cw]
    odd = [list(fr) for fr in i420]
    odd[3][0] = (y3.data_ptr(), w + 64, 1, 0)
    odd[1][1] = (cb1.data_ptr(), cw + 1, 1, 0)
    odd[1][2] = (cr1.data_ptr(), cw + 33, 1, 0)
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, odd, planes_subsampled=True) == want
    # an NV12 surface in an I420 pool (sample stride differs): one frame at a time
    mixed = [list(fr) for fr in i420]
    mixed[2] = nv12[2]
    assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, mixed, planes_subsampled=True) == want
    with pytest.raises(binding.JpegEncError):
        e.encode_planes_batch_device(binding.J_YCBCR, w, h, [[(0, ypitch, 1, 0)] * 3] * 2, planes_subsampled=True)


def test_randomised_planar_sources(binding, oracle, synth):
    """Fuzz-style sweep over described planar sources: size, sampling factor (1, 2 and 4), subsampled or full-resolution
    planes, I420-like or NV12-like chroma, padded pitches, quality, scan mode, restart interval, FDCT build, one surface
    at a time and as a pool - every file byte-identical to the oracle fed the equivalent interleaved YCbCr image.
    (JPEGENC_FUZZ_TRIALS / JPEGENC_FUZZ_SEED: soak length.)"""
    import os
    import torch
    rng = np.random.default_rng(int(os.environ.get("JPEGENC_FUZZ_SEED", "4242")))
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (1, 4)]
    for trial in range(int(os.environ.get("JPEGENC_FUZZ_TRIALS", "40"))):
        w, h = int(rng.integers(1, 700)), int(rng.integers(1, 400))
        hs, vs = samplings[int(rng.integers(0, len(samplings)))]
        subsampled = bool(rng.integers(0, 2))
        nv12 = bool(rng.integers(0, 2)) and (subsampled or (hs < 4 and vs < 4))
        kw = dict(quality=int(rng.integers(1, 101)), sampling=(hs, vs))
        mode = int(rng.integers(0, 4))
        if mode == 1:
            kw["progressive_scans"] = int(rng.integers(2, 12))
        elif mode == 2:
            kw["optimize"] = True
        if rng.integers(0, 3) == 0:
            kw["restart_interval"] = int(rng.integers(1, 40))
        variant = oracle.FDCT_SIMD if rng.integers(0, 4) == 0 else oracle.FDCT_SCALAR
        nframes = int(rng.integers(1, 6))
        cw, ch = (-(-w // hs), -(-h // vs)) if subsampled else (w, h)
        e = binding.Encoder(kw["quality"])
        e.set_sampling_factor(binding.sampling_factor(hs, vs))
        if kw.get("progressive_scans"):
            e.set_progressive_scans(kw["progressive_scans"])
        if kw.get("optimize"):
            e.set_optimized_huffman_tables(True)
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        if variant == oracle.FDCT_SIMD:
            e.set_fdct_variant(binding.FDCT_SIMD)
        ypad, cpad = int(rng.integers(0, 40)), int(rng.integers(0, 9))
        mixed_pitches = bool(rng.integers(0, 2))                      # a pool whose surfaces differ in pitch (per-frame pitch table)
        nv12_allowed = subsampled or (hs < 4 and vs < 4)
        mixed_layouts = nv12_allowed and rng.integers(0, 4) == 0        # ... and in layout: NV12 surfaces among I420 ones (one set of launches per layout)
        frames, want, keep, full_list = [], [], [], []
        for f in range(nframes):
            if mixed_layouts:
                nv12 = bool(rng.integers(0, 2))
            if mixed_pitches and f:
                ypad, cpad = int(rng.integers(0, 40)), int(rng.integers(0, 9))
            noisy = trial % 3 != 0
            mk = (lambda s: rng.integers(0, 256, s, dtype=np.uint8)) if noisy else (lambda s: (np.add.outer(np.arange(s[0]), np.arange(s[1])) // 3 + f).astype(np.uint8))
            y, cb, cr = mk((h, w)), mk((ch, cw)), mk((ch, cw))
            full_list.append((y, cb, cr))
            full = np.stack([y, _replicated(cb, hs, vs, w, h) if subsampled else cb, _replicated(cr, hs, vs, w, h) if subsampled else cr], axis=-1)
            want.append(oracle.encode_jpeg(np.ascontiguousarray(full), w, h, oracle.YCBCR, variant=variant, **kw))
            d_y = torch.zeros((h, w + ypad), dtype=torch.uint8, device="cuda"); d_y[:, :w] = torch.from_numpy(y).cuda()
            if nv12:
                d_uv = torch.zeros((ch, 2 * cw + 2 * cpad), dtype=torch.uint8, device="cuda")
                d_uv[:, 0:2 * cw:2] = torch.from_numpy(cb).cuda(); d_uv[:, 1:2 * cw:2] = torch.from_numpy(cr).cuda()
                keep += [d_y, d_uv]
                frames.append([(d_y.data_ptr(), w + ypad, 1, 0), (d_uv.data_ptr(), 2 * cw + 2 * cpad, 2, 0), (d_uv.data_ptr() + 1, 2 * cw + 2 * cpad, 2, 0)])
            else:
                d_cb = torch.zeros((ch, cw + cpad), dtype=torch.uint8, device="cuda"); d_cb[:, :cw] = torch.from_numpy(cb).cuda()
                d_cr = torch.zeros((ch, cw + cpad), dtype=torch.uint8, device="cuda"); d_cr[:, :cw] = torch.from_numpy(cr).cuda()
                keep += [d_y, d_cb, d_cr]
                frames.append([(d_y.data_ptr(), w + ypad, 1, 0), (d_cb.data_ptr(), cw + cpad, 1, 0), (d_cr.data_ptr(), cw + cpad, 1, 0)])
        what = (trial, w, h, hs, vs, subsampled, nv12, kw, variant, nframes)
        if os.environ.get("JPEGENC_FUZZ_VERBOSE"):
            print(what, flush=True)
        assert e.encode_planes_device(binding.J_YCBCR, w, h, frames[0], planes_subsampled=subsampled) == want[0], what
        assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, frames, planes_subsampled=subsampled) == want, what
        # ---- the same pictures with every plane in a random packed / deep layout: samples 1, 2 or 4 bytes apart starting at any
        # byte of the group, or eight bits (any shift) of 16-bit little-endian words - one layout per component for the whole pool
        factor4 = hs == 4 or vs == 4
        if factor4 and not subsampled:
            continue                                                   # (strides above 1 are not decimated by 4 on the device)
        layout = []
        for c in range(3):
            stride = int(rng.choice([1, 2, 4]))
            deep = stride > 1 and bool(rng.integers(0, 2))
            shift = (8 if stride == 4 or factor4 else int(rng.choice([8, 8, 2, 4, 6, 1, 7]))) if deep else 0
            off = int(rng.integers(0, stride // 2)) * 2 if deep else int(rng.integers(0, stride))
            layout.append((stride, shift, off))
        frames2 = []
        for f in range(nframes):
            planes = []
            comps = [np.ascontiguousarray(a) for a in (full_list[f][0], full_list[f][1], full_list[f][2])]
            for c, (stride, shift, off) in enumerate(layout):
                a = comps[c]
                rows, cols = a.shape
                pitch = cols * stride + int(rng.integers(0, 5)) * 4 + (4 if stride > 1 else 0)
                buf = rng.integers(0, 256, (rows, pitch), dtype=np.uint8)
                if shift:
                    word = (a.astype(np.uint32) << shift) | rng.integers(0, 1 << shift, a.shape).astype(np.uint32)
                    word = (word | (rng.integers(0, 256, a.shape).astype(np.uint32) << (shift + 8))) & 0xFFFF
                    buf[:, off:off + cols * stride:stride] = (word & 0xFF).astype(np.uint8)
                    buf[:, off + 1:off + 1 + cols * stride:stride] = (word >> 8).astype(np.uint8)
                else:
                    buf[:, off:off + cols * stride:stride] = a
                t = torch.from_numpy(buf).cuda()
                keep.append(t)
                planes.append((t.data_ptr() + off, pitch, stride, 0, shift))
            frames2.append(planes)
        what2 = what + (layout,)
        assert e.encode_planes_device(binding.J_YCBCR, w, h, frames2[0], planes_subsampled=subsampled) == want[0], what2
        assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, frames2, planes_subsampled=subsampled) == want, what2


def test_planar_source_launch_sequence_is_replayed_correctly(binding, oracle, synth):
    """One encoder, the same I420 surface several times in a row: the single-scan launch sequence is captured on the second
    identical call and replayed from the third (hipGraph) - the plane descriptors are part of what it bakes in.  New pixels
    in the same surface must come out of a replay, a surface at another address or with NV12 chroma must not be served by
    the old sequence."""
    import torch
    w, h = 640, 360
    rng = np.random.default_rng(5)
    cw, ch = w // 2, h // 2

    def surface():
        y, cb, cr = (rng.integers(0, 256, s, dtype=np.uint8) // 3 + 60 for s in ((h, w), (ch, cw), (ch, cw)))
        full = np.stack([y, _replicated(cb, 2, 2, w, h), _replicated(cr, 2, 2, w, h)], axis=-1)
        return y, cb, cr, oracle.encode_jpeg(np.ascontiguousarray(full), w, h, oracle.YCBCR, 80, sampling=(2, 2))
    e = binding.Encoder(80)
    e.set_sampling_factor(binding.F_2_2)
    y, cb, cr, want = surface()
    d_y, d_cb, d_cr = (torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (y, cb, cr))
    planes = [(d_y.data_ptr(), w, 1, 0), (d_cb.data_ptr(), cw, 1, 0), (d_cr.data_ptr(), cw, 1, 0)]
    for _ in range(5):                                             # direct, direct (keys match: capture), replay, replay, replay
        assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=True) == want
    y2, cb2, cr2, want2 = surface()                                # new content, same surface: the replayed sequence reads it
    d_y.copy_(torch.from_numpy(np.ascontiguousarray(y2))); d_cb.copy_(torch.from_numpy(np.ascontiguousarray(cb2))); d_cr.copy_(torch.from_numpy(np.ascontiguousarray(cr2)))
    torch.cuda.synchronize()
    for _ in range(2):
        assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=True) == want2
    d_uv = torch.from_numpy(np.ascontiguousarray(np.stack([cb, cr], axis=-1))).cuda()          # another chroma layout and address
    d_y3 = torch.from_numpy(np.ascontiguousarray(y)).cuda()
    nv12 = [(d_y3.data_ptr(), w, 1, 0), (d_uv.data_ptr(), cw * 2, 2, 0), (d_uv.data_ptr() + 1, cw * 2, 2, 0)]
    for _ in range(4):
        assert e.encode_planes_device(binding.J_YCBCR, w, h, nv12, planes_subsampled=True) == want
    assert e.encode_planes_device(binding.J_YCBCR, w, h, planes, planes_subsampled=True) == want2


def test_dense_frames_are_routed_to_the_two_kernels_and_back(binding, oracle, synth):
    """A handle whose last scan of a frame size coded to more than ~390 bits per block (noise-like frames, very high qualities)
    takes block kernel + k_block_code for the next frame of that size instead of the pixels -> bits kernel, whose long-block
    second walk loses there, and returns to the one kernel when the content thins out (host_internal.h: DeviceCtx::dense_last_time,
    BatchBuffers::dense_bits_per_block).  Either way the bytes are the oracle's: dense and sparse frames alternate through one
    handle - single calls, the worker-pool batch, the device-resident batch in rounds of two."""
    import torch
    w, h = 1280, 720                                                   # above 1 MB of pixels (smaller frames always keep the one kernel)
    rng = np.random.default_rng(17)
    dense = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    sparse = synth.test_img_rgb(w, h)
    mixed = dense.copy()
    mixed[:, : w // 2] = sparse[:, : w // 2]
    for quality, sampling in ((96, (2, 2)), (98, (1, 1))):
        want = {id(a): oracle.encode_jpeg(a, w, h, oracle.RGB, quality, sampling=sampling) for a in (dense, sparse, mixed)}
        assert len(want[id(dense)]) * 8 > 390 * (w * h // 64) and len(want[id(sparse)]) * 8 < 200 * (w * h // 64)
        with binding.Encoder(quality) as e:
            e.set_sampling_factor(binding.sampling_factor(*sampling))
            seq = [dense, dense, dense, sparse, sparse, dense, mixed, sparse, dense, dense]
            for a in seq:
                assert e.encode(a, w, h, binding.RGB) == want[id(a)]
            assert e.encode_batch(seq * 2, w, h, binding.RGB) == [want[id(a)] for a in seq * 2]
            e.set_batch_round_frames(2)                                # rounds of two: the third round is routed by what the first brought back
            order = [dense, dense, dense, dense, dense, dense, sparse, sparse, sparse, sparse, sparse, sparse, dense, dense]
            d = torch.from_numpy(np.stack(order)).cuda()
            files = e.encode_batch_device(d.data_ptr(), w * h * 3, len(order), w, h, binding.RGB)
            assert files == [want[id(a)] for a in order]
            files = e.encode_batch_device(d.data_ptr(), w * h * 3, len(order), w, h, binding.RGB)      # (this call starts with what the last one learnt)
            assert files == [want[id(a)] for a in order]


def test_planes_pool_that_cannot_share_launches_goes_through_the_worker_pool(binding, oracle, synth):
    """A pool of surfaces that mixes sample strides (NV12 frames among I420 frames) - or asks for optimised tables - cannot share
    its launches: the frames go one per launch sequence through the pool of host workers, handed out in order.  Same files; and a
    sink that fails on frame 13 of 24 makes the call return JPEGENC_ERR_WRITE naming that frame with every frame before it delivered
    complete."""
    import ctypes as C
    import torch
    w, h, n = 640, 360, 24
    cw, ch = w // 2, h // 2
    rng = np.random.default_rng(23)
    smooth = lambda a: (a.astype(np.int16) // 4 + np.add.outer(np.arange(a.shape[0]), np.arange(a.shape[1])) // 3).clip(0, 255).astype(np.uint8)
    keep, frames, want = [], [], []
    for f in range(n):
        y, cb, cr = (smooth(rng.integers(0, 256, s, dtype=np.uint8)) for s in ((h, w), (ch, cw), (ch, cw)))
        full = np.stack([y, _replicated(cb, 2, 2, w, h), _replicated(cr, 2, 2, w, h)], axis=-1)
        want.append(oracle.encode_jpeg(full, w, h, oracle.YCBCR, 85, sampling=(2, 2)))
        d_y = torch.from_numpy(y).cuda()
        if f % 3 == 1:                                              # an NV12 surface among the I420 ones
            d_uv = torch.from_numpy(np.ascontiguousarray(np.stack([cb, cr], axis=-1))).cuda()
            keep += [d_y, d_uv]
            frames.append([(d_y.data_ptr(), w, 1, 0), (d_uv.data_ptr(), 2 * cw, 2, 0), (d_uv.data_ptr() + 1, 2 * cw, 2, 0)])
        else:
            d_cb, d_cr = torch.from_numpy(cb).cuda(), torch.from_numpy(cr).cuda()
            keep += [d_y, d_cb, d_cr]
            frames.append([(d_y.data_ptr(), w, 1, 0), (d_cb.data_ptr(), cw, 1, 0), (d_cr.data_ptr(), cw, 1, 0)])
    with binding.Encoder(85) as e:
        e.set_sampling_factor(binding.F_2_2)
        assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, frames, planes_subsampled=True) == want
        # the failing sink, through the C ABI
        arr = (binding.Plane * (4 * n))()
        for f, planes in enumerate(frames):
            for i, t in enumerate(planes):
                arr[4 * f + i] = binding._plane(t)
        got = [bytearray() for _ in range(n)]

        def sink(user, ptr, nbytes):
            k = user or 0
            if k == 13:
                return 1
            got[k] += C.string_at(ptr, nbytes)
            return 0
        cb_ = binding.WRITE_FN(sink)
        users = (C.c_void_p * n)(*range(n))
        fn = binding.lib().jpegenc_encoder_encode_planes_batch_device
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(binding.Plane), C.c_int, C.c_int, binding.WRITE_FN, C.POINTER(C.c_void_p)]
        rc = fn(e._h, binding.J_YCBCR, w, h, arr, n, 1, cb_, users)
        assert rc == binding.ERR_WRITE and b"frame 13" in binding.lib().jpegenc_last_error()
        assert [bytes(g) for g in got[:13]] == want[:13]
        assert all(bytes(g) in (b"", want[k]) for k, g in enumerate(got) if k > 13)      # later frames: delivered whole or not at all


@pytest.mark.parametrize("kw", [dict(quality=85, sampling=(2, 2), optimize=True), dict(quality=70, sampling=(2, 1), progressive_scans=5, optimize=True),
                                dict(quality=92, sampling=(1, 1), optimize=True, restart_interval=11)],
                         ids=["sequential", "progressive", "444-restart"])
def test_pools_with_per_frame_optimised_tables_share_their_launches(binding, oracle, synth, kw):
    """optimize_huffman_table gives every image its own tables (encoder.rs:1086-1200).  A pool of device-resident frames or
    described surfaces still shares its launches: the block kernel counts the symbols of every frame of a round, one host step
    builds the tables, the coder reads frame f's table set (EntropyParams: kLutPerFrame).  Frames with very different
    statistics (noise, a gradient, flat) in rounds of 3 of 11: every file equals the oracle's for THAT frame."""
    import torch
    w, h, n = 328, 200, 11
    hs, vs = kw["sampling"]
    rng = np.random.default_rng(5)
    frames, want, keep, surfaces = [], [], [], []
    for f in range(n):
        kind = f % 3
        px = (rng.integers(0, 256, (h, w, 3), dtype=np.uint8) if kind == 0 else
              synth.test_img_rgb(w, h) if kind == 1 else np.full((h, w, 3), 17 * f, dtype=np.uint8))
        px = np.ascontiguousarray(np.roll(px, 5 * f, axis=1))
        frames.append(px)
        want.append(oracle.encode_jpeg(px, w, h, oracle.YCBCR, **kw))
    d = torch.from_numpy(np.stack(frames)).cuda()

    def enc():
        e = binding.Encoder(kw["quality"])
        e.set_sampling_factor(binding.sampling_factor(hs, vs))
        if kw.get("progressive_scans"):
            e.set_progressive_scans(kw["progressive_scans"])
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        e.set_optimized_huffman_tables(True)
        e.set_batch_round_frames(3)
        return e
    with enc() as e:
        assert e.encode_batch_device(d.data_ptr(), w * h * 3, n, w, h, binding.YCBCR) == want
        # the same frames as full-resolution planes of a pool of described surfaces
        planar = torch.from_numpy(np.ascontiguousarray(np.stack(frames).transpose(0, 3, 1, 2))).cuda()
        pool = [[(planar[f, c].data_ptr(), w, 1, 0) for c in range(3)] for f in range(n)]
        assert e.encode_planes_batch_device(binding.J_YCBCR, w, h, pool) == want
        # and one at a time: same files
        assert e.encode_device(d.data_ptr() + 4 * w * h * 3, w, h, binding.YCBCR) == want[4]


def test_register_ahead_uploads(binding, oracle, synth):
    """jpegenc_encoder_set_batch_upload(REGISTER_AHEAD): one thread of the handle page-locks the batch's pageable frames ahead of the
    workers, which upload them where they lie.  Same files as the staged default for frames that are slices of ONE array (their size
    is no multiple of the page: neighbours share pages), given in reverse order, allocated one by one, partly or wholly page-locked by
    the caller already; a sink failure in the middle ends the call with ERR_WRITE and nothing stays page-locked behind the call."""
    import ctypes as C
    w, h, n = 1000, 701, 12                                     # 2 103 000 bytes per frame: above the 2 MB small-frame path, not a page multiple
    fb = w * h * 3
    block = np.empty(n * fb + 64, dtype=np.uint8)
    frames = [block[5 + i * fb: 5 + (i + 1) * fb] for i in range(n)]       # unaligned slices of one allocation
    for i, f in enumerate(frames):
        f[:] = synth.lcg_image(w, h, 3, 700 + i).reshape(-1)
    want = [oracle.encode_jpeg(f, w, h, oracle.RGB, 85) for f in frames]
    e = binding.Encoder(85)
    staged = e.encode_batch(frames, w, h, binding.RGB)
    assert staged == want
    e.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD)
    assert e.encode_batch(frames, w, h, binding.RGB) == want
    assert e.encode_batch(frames[::-1], w, h, binding.RGB) == want[::-1]
    singles = [np.ascontiguousarray(f.copy()) for f in frames]
    assert e.encode_batch(singles, w, h, binding.RGB) == want
    lib = binding.lib()
    lib.jpegenc_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.jpegenc_host_unregister.argtypes = [C.c_void_p]
    # the caller has page-locked frame 3 wholly and the first half of frame 7: left as they are (in place / staged)
    assert lib.jpegenc_host_register(singles[3].ctypes.data, fb) == 0
    assert lib.jpegenc_host_register(singles[7].ctypes.data, fb // 2) == 0
    try:
        assert e.encode_batch(singles, w, h, binding.RGB) == want
    finally:
        assert lib.jpegenc_host_unregister(singles[3].ctypes.data) == 0
        assert lib.jpegenc_host_unregister(singles[7].ctypes.data) == 0
    # nothing of ours is still locked: the caller can lock every frame itself now
    for f in singles:
        assert lib.jpegenc_host_register(f.ctypes.data, fb) == 0
        assert lib.jpegenc_host_unregister(f.ctypes.data) == 0
    # a failing sink
    seen = []

    def sink(user, ptr, nbytes):
        seen.append(user or 0)
        return 3 if (user or 0) == 5 else 0
    cb = binding.WRITE_FN(sink)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    users = (C.c_void_p * n)(*range(n))
    fn = lib.jpegenc_encoder_encode_batch
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, binding.WRITE_FN, C.POINTER(C.c_void_p)]
    assert fn(e._h, ptrs, fb, n, w, h, binding.RGB, cb, users) == binding.ERR_WRITE
    assert lib.jpegenc_host_register(block.ctypes.data, block.nbytes) == 0       # (whole array: fails if any page of it were still locked)
    assert lib.jpegenc_host_unregister(block.ctypes.data) == 0
    e.set_batch_upload(binding.UPLOAD_STAGED)
    assert e.encode_batch(frames, w, h, binding.RGB) == want


def test_register_ahead_repeated_and_returning_frames(binding, oracle, synth):
    """Register-ahead with batches that name the same memory more than once - [A] * N (the usual benchmark pattern), [A, B, A],
    frames smaller than a page apart: a registration is shared by every frame that lies in it and must stay until the last of them
    has uploaded (round 5 released it after the next frame).  Same files as the oracle, nothing left page-locked."""
    import ctypes as C
    w, h = 1000, 701
    fb = w * h * 3
    a = np.ascontiguousarray(synth.lcg_image(w, h, 3, 901).reshape(-1))
    b = np.ascontiguousarray(synth.lcg_image(w, h, 3, 902).reshape(-1))
    want_a, want_b = oracle.encode_jpeg(a, w, h, oracle.RGB, 85), oracle.encode_jpeg(b, w, h, oracle.RGB, 85)
    lib = binding.lib()
    lib.jpegenc_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.jpegenc_host_unregister.argtypes = [C.c_void_p]
    with binding.Encoder(85) as e:
        e.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD)
        for _ in range(3):
            assert e.encode_batch([a] * 40, w, h, binding.RGB) == [want_a] * 40
            assert e.encode_batch([a, b, a], w, h, binding.RGB) == [want_a, want_b, want_a]
            assert e.encode_batch([a, b] * 15 + [a], w, h, binding.RGB) == [want_a, want_b] * 15 + [want_a]
        # overlapping windows of one array, 1 000 bytes apart: every frame lies almost wholly inside its predecessors' pages
        block = np.empty(fb + 24 * 1000, dtype=np.uint8)
        block[:] = np.resize(a, block.size)
        windows = [block[i * 1000: i * 1000 + fb] for i in range(24)]
        want_w = [oracle.encode_jpeg(wd, w, h, oracle.RGB, 85) for wd in windows]
        assert e.encode_batch(windows, w, h, binding.RGB) == want_w
        assert e.encode_batch(windows[::-1], w, h, binding.RGB) == want_w[::-1]
        for arr in (a, b, block):                                   # nothing of ours is still locked
            assert lib.jpegenc_host_register(arr.ctypes.data, arr.nbytes) == 0
            assert lib.jpegenc_host_unregister(arr.ctypes.data) == 0


def test_set_batch_workers_caps_every_pool(binding, oracle, synth):
    """jpegenc_encoder_set_batch_workers: the handle's batch calls keep at most that many host threads busy, the caller's included -
    a fresh handle told 2 has had 2 workers after a host-fed batch (1: the caller's thread alone), whatever the CPU count says; the
    files do not depend on it; device-resident batches, thumbnails and per-frame optimised tables run under the same cap."""
    import torch
    w, h, n = 1280, 720, 12                                         # 2.76 MB per frame: the worker pool, not the staged thumbnail rounds
    frames = [np.ascontiguousarray(synth.lcg_image(w, h, 3, 40 + i)) for i in range(n)]
    want = [oracle.encode_jpeg(f, w, h, oracle.RGB, 80) for f in frames]
    for cap in (2, 1, 3):
        with binding.Encoder(80) as e:
            assert e.batch_workers() == 0
            e.set_batch_workers(cap)
            assert e.batch_workers() == cap
            assert e.encode_batch(frames, w, h, binding.RGB) == want
            assert len(e.batch_worker_info()) == cap
            # the device-resident batch (rounds, background assembly) and its per-frame optimised tables under the same budget
            d = torch.from_numpy(np.stack(frames)).cuda()
            assert e.encode_batch_device(d.data_ptr(), w * h * 3, n, w, h, binding.RGB) == want
            e.set_optimized_huffman_tables(True)
            want_opt = [oracle.encode_jpeg(f, w, h, oracle.RGB, 80, optimize=True) for f in frames[:4]]
            assert e.encode_batch_device(d.data_ptr(), w * h * 3, 4, w, h, binding.RGB) == want_opt
    # thumbnails: rounds staged by the handle's copier threads
    tw, th, tn = 160, 120, 40
    thumbs = [np.ascontiguousarray(synth.lcg_image(tw, th, 3, 300 + i)) for i in range(tn)]
    want_t = [oracle.encode_jpeg(f, tw, th, oracle.RGB, 80) for f in thumbs]
    with binding.Encoder(80) as e:
        e.set_batch_workers(1)
        assert e.encode_batch(thumbs, tw, th, binding.RGB) == want_t
        e.set_batch_workers(0)                                      # back to automatic: at most 4 workers with device entropy coding
        assert e.encode_batch(frames, w, h, binding.RGB) == want
        assert 1 <= len(e.batch_worker_info()) <= 4
    with pytest.raises(binding.JpegEncError):
        with binding.Encoder(80) as e:
            e.set_batch_workers(65)


def test_multi_device_children_inherit_the_batch_settings(binding, oracle, synth):
    """jpegenc_encoder_encode_batch_multi hands every per-device child what the parent was told about batches: the thread budget, the
    upload mode and the register-cache budget (round 5 copied only the configuration and the NUMA switch).  Device 0 listed twice."""
    w, h, n = 1280, 720, 10
    frames = [np.ascontiguousarray(synth.lcg_image(w, h, 3, 70 + i)) for i in range(n)]
    want = [oracle.encode_jpeg(f, w, h, oracle.RGB, 80) for f in frames]
    outs = [np.empty(2 << 20, dtype=np.uint8) for _ in frames]
    with binding.Encoder(80) as e:
        e.set_batch_workers(2)
        e.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD)
        e.set_register_cache(64 << 20)
        lens = e.encode_batch_into(frames, w, h, binding.RGB, outs, devices=[0, 0])
        assert [outs[i][:lens[i]].tobytes() for i in range(n)] == want
        info = e.batch_shard_info()
        assert len(info) == 2
        for child in info:
            assert child == {"device": 0, "batch_workers": 2, "upload_mode": binding.UPLOAD_REGISTER_AHEAD, "register_cache_bytes": 64 << 20, "pool_workers": 2}
        # changed on the parent: the children follow at the next call
        e.set_batch_workers(3)
        e.set_batch_upload(binding.UPLOAD_STAGED)
        e.set_register_cache(0)
        lens = e.encode_batch_into(frames, w, h, binding.RGB, outs, devices=[0, 0])
        assert [outs[i][:lens[i]].tobytes() for i in range(n)] == want
        for child in e.batch_shard_info():
            assert (child["batch_workers"], child["upload_mode"], child["register_cache_bytes"]) == (3, binding.UPLOAD_STAGED, 0)
            assert child["pool_workers"] == 3


def test_thumbnail_batch_names_the_failing_frame_in_the_callers_numbering(binding, synth):
    """A staged batch of thumbnails runs in rounds; a sink that fails in a round after the first is reported as "frame K" with K counted
    from the start of the batch, and every frame below K has been delivered whole."""
    import ctypes as C
    tw, th = 640, 360                                               # 691 200 bytes: 64 MB rounds hold 97 frames
    n, bad = 230, 201
    base = np.ascontiguousarray(synth.lcg_image(tw, th, 3, 5))
    frames = [base] * n
    got = [0] * n

    def sink(user, ptr, nbytes):
        k = user or 0
        if k == bad:
            return 1
        got[k] += nbytes
        return 0
    cb = binding.WRITE_FN(sink)
    lib = binding.lib()
    fn = lib.jpegenc_encoder_encode_batch
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, binding.WRITE_FN, C.POINTER(C.c_void_p)]
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in frames])
    users = (C.c_void_p * n)(*range(n))
    with binding.Encoder(80) as e:
        single = len(e.encode(base, tw, th, binding.RGB))
        assert fn(e._h, ptrs, base.size, n, tw, th, binding.RGB, cb, users) == binding.ERR_WRITE
        message = lib.jpegenc_last_error().decode()
        assert message.startswith(f"frame {bad}:"), message
        assert all(g == single for g in got[:bad])


@pytest.mark.parametrize("ct,hs,vs,w,h,restart", [(1, 2, 2, 1920, 1080, 0), (1, 1, 1, 515, 301, 7), (7, 2, 2, 515, 301, 0), (2, 2, 1, 1280, 720, 0)],
                         ids=["rgb420-1080p", "rgb444-rst7", "cmyk-as-ycck-420-unfused", "rgba422"])
def test_scan_lanes_alternate_behind_one_call_site(binding, oracle, synth, ct, hs, vs, w, h, restart):
    """jpegenc_scan_lanes: submits alternate between two internal lanes (stream + workspace each; coefficient scratch for layouts the
    one kernel does not take), ordered behind the producer's stream; join() orders a stream behind all of them.  Eleven submits of
    differing frame counts, every one into its own output: the same bytes and lengths as the oracle's files hold."""
    import torch
    bpp = binding.BPP[ct]
    per, calls = 3, 11
    frames = [synth.lcg_image(w, h, bpp, 60 + i) for i in range(4)]
    want = []
    for f in frames:
        ref = oracle.encode_jpeg(f, w, h, ct, 85, sampling=(hs, vs), restart_interval=restart)
        want.append(ref)
    L = binding.layout(w, h, ct, hs, vs, binding.ORDER_MCU)
    cap = binding.scan_max_bytes(L, binding.baseline_scan(restart_interval=restart))
    q = binding.qtables(85)
    producer = torch.cuda.Stream()
    consumer = torch.cuda.Stream()
    with binding.ScanLanes(w, h, ct, hs, vs, per, restart_interval=restart) as lanes:
        outs, lens, picks, pixels = [], [], [], []
        for c in range(calls):
            n = 1 + c % per
            pick = [(c + k) % len(frames) for k in range(n)]
            with torch.cuda.stream(producer):                        # the pixels are produced on the caller's stream: the lane must wait for them
                d_px = torch.from_numpy(np.stack([frames[k] for k in pick])).cuda(non_blocking=False).clone()
            d_out = torch.zeros((n, cap), dtype=torch.uint8, device="cuda")
            d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
            torch.cuda.current_stream().synchronize()              # (the zero fills above ran on the default stream)
            lanes.submit(d_px.data_ptr(), w * h * bpp, n, q, d_out.data_ptr(), cap, d_len.data_ptr(), producer.cuda_stream)
            outs.append(d_out); lens.append(d_len); picks.append(pick); pixels.append(d_px)
        with pytest.raises(binding.JpegEncError):
            lanes.submit(pixels[0].data_ptr(), w * h * bpp, per + 1, q, outs[0].data_ptr(), cap, lens[0].data_ptr(), producer.cuda_stream)
        lanes.join(consumer.cuda_stream)
        consumer.synchronize()
        for c in range(calls):
            for k, f in enumerate(picks[c]):
                got = bytes(outs[c][k, :int(lens[c][k])].cpu().numpy())
                assert len(got) > 0 and want[f].endswith(got + b"\xff\xd9"), f"submit {c} frame {k}"


def test_batch_workers_stage_ahead_with_mixed_and_reused_buffers(binding, oracle, synth):
    """A batch worker stages its next pageable frame while its current one is on the link: batches that mix pageable and page-locked
    frames, repeat a buffer, and reuse the same buffers with new content from call to call must deliver what the buffers hold when
    the call is made - with 1, 2 and 3 workers (a worker claims the frame after its current one)."""
    w, h = 1280, 720
    fb = w * h * 3
    imgs = [np.ascontiguousarray(synth.lcg_image(w, h, 3, 500 + i).reshape(-1)) for i in range(5)]
    want = [oracle.encode_jpeg(im, w, h, oracle.RGB, 80) for im in imgs]
    a, c = imgs[0].copy(), imgs[2].copy()
    locked = binding.HostBuffer(fb)
    try:
        locked.array[:] = imgs[1]
        for workers in (1, 2, 3):
            with binding.Encoder(80) as e:
                e.set_batch_workers(workers)
                a[:] = imgs[0]; c[:] = imgs[2]
                frames = [a, locked.array, a, c, locked.array, c, a]
                assert e.encode_batch(frames, w, h, binding.RGB) == [want[0], want[1], want[0], want[2], want[1], want[2], want[0]]
                # the same buffers, new content: nothing staged during the call before may be taken for it
                a[:] = imgs[3]; c[:] = imgs[4]
                assert e.encode_batch(frames, w, h, binding.RGB) == [want[3], want[1], want[3], want[4], want[1], want[4], want[3]]
                assert e.encode_batch([c, a], w, h, binding.RGB) == [want[4], want[3]]
    finally:
        locked.close()


def test_register_ahead_from_two_handles_over_the_same_memory(binding, oracle, synth):
    """Two batches that page-lock ahead at the same time over identical and neighbouring memory (the per-device children of a multi-device
    batch, device 0 listed twice: child d takes frames d, d + 2, ...): a frame that touches the other batch's registration is staged - it is
    not taken for the caller's page-locked memory, and nothing is uploaded from a registration its owner may release.  Same files."""
    w, h = 1000, 701
    fb = w * h * 3
    block = np.empty(13 * fb + 64, dtype=np.uint8)                   # slices of one array: every frame shares pages with both neighbours
    slices = [block[7 + i * fb: 7 + (i + 1) * fb] for i in range(13)]
    for i, s in enumerate(slices):
        s[:] = synth.lcg_image(w, h, 3, 950 + i).reshape(-1)
    want = [oracle.encode_jpeg(s, w, h, oracle.RGB, 85) for s in slices]
    outs = [np.empty(2 << 20, dtype=np.uint8) for _ in range(16)]
    with binding.Encoder(85) as e:
        e.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD)
        for _ in range(3):
            lens = e.encode_batch_into(slices, w, h, binding.RGB, outs, devices=[0, 0])
            assert [outs[i][:lens[i]].tobytes() for i in range(13)] == want
            same = [slices[0]] * 16                                   # both children name the same memory all the time
            lens = e.encode_batch_into(same, w, h, binding.RGB, outs, devices=[0, 0])
            assert all(outs[i][:lens[i]].tobytes() == want[0] for i in range(16))


def test_randomised_host_fed_batches(binding, oracle, synth):
    """Random host-fed batches through the worker pool: frame sizes on both sides of the staged-thumbnail limit, 1 .. 28 frames, thread
    budgets 0 .. 5, staged and register-ahead uploads, frames that are pageable, page-locked by the caller (wholly or only their first
    half), slices of one array a few bytes apart, or the same buffer several times - every file equal to the single-image call's
    (itself pinned to the oracle on the first trial of every geometry).  JPEGENC_FUZZ_TRIALS / JPEGENC_FUZZ_SEED: soak length."""
    import ctypes as C
    import os
    rng = np.random.default_rng(int(os.environ.get("JPEGENC_FUZZ_SEED", "7")))
    trials = int(os.environ.get("JPEGENC_BATCH_FUZZ_TRIALS", os.environ.get("JPEGENC_FUZZ_TRIALS", "24")))
    lib = binding.lib()
    lib.jpegenc_host_register.argtypes = [C.c_void_p, C.c_size_t]
    lib.jpegenc_host_unregister.argtypes = [C.c_void_p]
    geometries = [(1280, 720), (1000, 701), (640, 360), (1920, 1080), (333, 201)]
    checked = set()
    for trial in range(trials):
        w, h = geometries[int(rng.integers(len(geometries)))]
        fb = w * h * 3
        quality = int(rng.choice([50, 80, 90]))
        n = int(rng.integers(1, 29))
        distinct = int(rng.integers(1, min(n, 6) + 1))
        block = np.empty(distinct * (fb + 24) + 64, dtype=np.uint8)
        images = []
        for i in range(distinct):
            off = 3 + i * (fb + int(rng.integers(0, 24)))
            img = block[off:off + fb]
            img[:] = synth.lcg_image(w, h, 3, 1000 * trial + i).reshape(-1) if i % 2 else np.resize(synth.test_img_rgb(w, h).reshape(-1), fb)
            img[:16] = (trial * 7 + i) & 255
            images.append(img)
        pinned = binding.HostBuffer(fb)
        pinned.array[:] = images[0]
        half = np.empty(fb, dtype=np.uint8)
        half[:] = images[-1]
        registered_half = bool(rng.integers(2)) and not os.environ.get("JPEGENC_FUZZ_NO_HALF") and lib.jpegenc_host_register(half.ctypes.data, fb // 2) == 0
        try:
            with binding.Encoder(quality) as e:
                want = [e.encode(img, w, h, binding.RGB) for img in images]
                if (w, h, quality) not in checked:
                    checked.add((w, h, quality))
                    assert want[0] == oracle.encode_jpeg(images[0], w, h, oracle.RGB, quality)
                frames, expect = [], []
                for k in range(n):
                    which = int(rng.integers(distinct + 2))
                    if which == distinct:
                        frames.append(pinned.array); expect.append(want[0])
                    elif which == distinct + 1:
                        frames.append(half); expect.append(want[-1])
                    else:
                        frames.append(images[which]); expect.append(want[which])
                workers, ahead = int(rng.integers(0, 6)), bool(rng.integers(3) == 0) and not os.environ.get("JPEGENC_FUZZ_NO_RA")
                e.set_batch_workers(workers)
                e.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD if ahead else binding.UPLOAD_STAGED)
                if os.environ.get("JPEGENC_FUZZ_VERBOSE"):
                    import sys
                    print(f"trial {trial}: {w}x{h} q{quality} n={n} distinct={distinct} workers={workers} register_ahead={ahead} half_registered={registered_half} "
                          f"block@{block.ctypes.data:#x} half@{half.ctypes.data:#x}", file=sys.stderr, flush=True)
                for _ in range(2):
                    assert e.encode_batch(frames, w, h, binding.RGB) == expect, f"trial {trial}: {w}x{h} q{quality} n={n}"
        finally:
            if registered_half:
                assert lib.jpegenc_host_unregister(half.ctypes.data) == 0
            pinned.close()
