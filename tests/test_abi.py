"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads without a
GPU, exports every symbol include/jpegenc_mi355x.h declares, and its host-side logic (tables,
geometry, validation, configuration) matches the oracle.  No compute calls here."""
import os
import re

import numpy as np
import pytest


@pytest.fixture(scope="module")
def binding(pkg):
    import importlib
    b = importlib.import_module("jpeg_encoder_amd.binding")
    if not os.path.exists(b.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return b


def test_header_symbols_are_exported(binding):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "jpegenc_mi355x.h")).read()
    declared = set(re.findall(r"\b(jpegenc_[a-z0-9_]+)\s*\((?!\*)", header))   # not `type (*array)[2]` parameters
    declared -= {"jpegenc_write_fn", "jpegenc_fill_row_fn"}
    assert declared == set(binding.ABI_SYMBOLS)
    lib = binding.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.jpegenc_abi_version() == 2


def test_qtable_matches_oracle(binding, oracle):
    rng = np.random.default_rng(0)
    for quality in (1, 10, 49, 50, 75, 80, 90, 95, 100, 0, 255):
        for preset in range(9):
            got = binding.qtables(quality, (preset, preset))
            want = oracle.qtables(quality, (preset, preset))
            for i in range(2):
                assert list(got[i].table) == list(want[i].table)
                assert list(got[i].reciprocals) == list(want[i].recip)
                assert list(got[i].corrections) == list(want[i].corr)
    custom = [int(v) for v in rng.integers(0, 70000, 64) % 65536]
    got = binding.qtables(33, (binding.Q_CUSTOM, binding.Q_CUSTOM), (custom, custom))
    want = oracle.qtables(33, (oracle.Q_CUSTOM, oracle.Q_CUSTOM), (custom, custom))
    assert list(got[1].table) == list(want[1].table) and list(got[1].reciprocals) == list(want[1].recip)


def test_layout_matches_oracle(binding, oracle):
    for ct in range(9):
        for hs, vs in [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2), (1, 4), (2, 4)]:
            for w, h in [(1, 1), (8, 8), (9, 17), (258, 128), (37, 21), (1920, 1080)]:
                for order in (0, 1):
                    L = binding.layout(w, h, ct, hs, vs, order)
                    total, per = oracle.block_counts(w, h, ct, hs, vs, order)
                    assert L.total_blocks == total
                    assert list(L.blocks)[:L.num_components] == per


def test_survey_geometry(binding):
    """SURVEY.md §8 geometry table."""
    assert binding.layout(256, 256, binding.RGB, 1, 1, 0).total_blocks == 3072
    L = binding.layout(3840, 2160, binding.RGB, 2, 2, 0)
    assert (L.mcus, L.total_blocks) == (32400, 194400)
    L = binding.layout(1920, 1080, binding.RGB, 2, 2, 0)
    assert (L.mcus, L.total_blocks) == (8160, 48960)
    assert binding.layout(7680, 4320, binding.CMYK, 1, 1, 0).total_blocks == 2073600
    assert binding.layout(3840, 2160, binding.RGB, 1, 1, 1).total_blocks == 388800
    assert list(binding.layout(258, 128, binding.RGB, 2, 2, 0).blocks)[:3] == [17 * 8 * 4, 17 * 8, 17 * 8]
    assert list(binding.layout(258, 128, binding.RGB, 2, 2, 1).blocks)[:3] == [33 * 16, 17 * 8, 17 * 8]


def test_encoder_configuration_api(binding):
    """Encoder::new defaults and setters (src/encoder.rs:239-364, tests :1323-1331)."""
    e = binding.Encoder(100)
    assert e.sampling_factor() == binding.F_1_1
    assert binding.Encoder(89).sampling_factor() == binding.F_2_2
    assert binding.Encoder(90).sampling_factor() == binding.F_1_1
    assert e.progressive_scans() is None
    e.set_progressive(True)
    assert e.progressive_scans() == 4                 # test_set_progressive
    e.set_progressive(False)
    assert e.progressive_scans() is None
    e.set_progressive_scans(64)
    assert e.progressive_scans() == 64
    for bad in (0, 1, 65):
        with pytest.raises(binding.JpegEncError) as err:   # the reference panics here
            e.set_progressive_scans(bad)
        assert err.value.status == binding.ERR_INVALID_ARGUMENT
    assert e.restart_interval() is None
    e.set_restart_interval(32)
    assert e.restart_interval() == 32
    e.set_restart_interval(0)
    assert e.restart_interval() is None
    assert e.optimized_huffman_tables() is False
    e.set_optimized_huffman_tables(True)
    assert e.optimized_huffman_tables() is True
    assert e.density() == (binding.DENSITY_PIXEL_ASPECT_RATIO, 1, 1)
    e.set_density(binding.DENSITY_INCHES, 300, 300)
    assert e.density() == (binding.DENSITY_INCHES, 300, 300)
    assert e.quantization_tables() == (binding.Q_DEFAULT, binding.Q_DEFAULT)
    e.set_quantization_tables(binding.Q_FLAT, binding.Q_CUSTOM, chroma_custom=[3] * 64)
    assert e.quantization_tables() == (binding.Q_FLAT, binding.Q_CUSTOM)
    e.set_sampling_factor(0x80 | 0x22)                # R_4_2_0 alias
    assert e.sampling_factor() == 0xA2
    with pytest.raises(binding.JpegEncError):
        e.set_sampling_factor(0x44)


def test_encoder_segment_validation(binding):
    """add_app_segment / add_icc_profile limits (src/encoder.rs:374-417)."""
    e = binding.Encoder(90)
    for nr in (0, 16):
        with pytest.raises(binding.JpegEncError) as err:
            e.add_app_segment(nr, b"x")
        assert err.value.status == binding.ERR_INVALID_APP_SEGMENT
    with pytest.raises(binding.JpegEncError) as err:
        e.add_app_segment(1, b"x" * 65534)
    assert err.value.status == binding.ERR_APP_SEGMENT_TOO_LARGE
    e.add_app_segment(15, b"x" * 65533)
    with pytest.raises(binding.JpegEncError) as err:
        e.add_icc_profile(b"\0" * (65519 * 255))
    assert err.value.status == binding.ERR_ICC_TOO_LARGE
    e.add_icc_profile(b"\0" * (65519 * 3))
    e.add_exif_metadata(b"MM\0*")


def test_validation_precedes_device_work(binding, synth):
    """BadImageData / ZeroImageDimensions are reported without touching the GPU
    (src/encoder.rs:447-454, 521-526) — these must hold even on a box with no device."""
    px = synth.test_img_rgb()
    e = binding.Encoder(90)
    with pytest.raises(binding.JpegEncError) as err:
        e.encode(px.reshape(-1)[:-1], 258, 128, binding.RGB)
    assert err.value.status == binding.ERR_BAD_IMAGE_DATA
    assert "need at least 99072" in str(err.value)
    with pytest.raises(binding.JpegEncError) as err:
        e.encode(px, 0, 128, binding.RGB)
    assert err.value.status == binding.ERR_ZERO_IMAGE_DIMENSIONS
    with pytest.raises(binding.JpegEncError) as err:
        binding.blocks_host(px.reshape(-1)[:100], 258, 128, binding.RGB, 2, 2, 90)
    assert err.value.status == binding.ERR_BAD_IMAGE_DATA


def test_no_cpu_fallback(binding, synth):
    """Without a GPU every compute entry point must fail loudly, never compute on the host."""
    if binding.device_count() > 0:
        pytest.skip("a GPU is present")
    px = synth.test_img_rgb()
    with pytest.raises(binding.JpegEncError) as err:
        binding.blocks_host(px, 258, 128, binding.RGB, 2, 2, 90)
    assert err.value.status == binding.ERR_NO_DEVICE
    with pytest.raises(binding.JpegEncError) as err:
        binding.Encoder(90).encode(px, 258, 128, binding.RGB)
    assert err.value.status == binding.ERR_NO_DEVICE


def test_free_functions(binding, oracle):
    """rgb_to_ycbcr / cmyk_to_ycck re-exports (src/lib.rs:45-49)."""
    import ctypes as C
    out = (C.c_uint8 * 4)()
    rng = np.random.default_rng(5)
    for r, g, b, k in rng.integers(0, 256, (200, 4)):
        binding.lib().jpegenc_rgb_to_ycbcr(int(r), int(g), int(b), out)
        assert tuple(out)[:3] == oracle.rgb_to_ycbcr(int(r), int(g), int(b))
        binding.lib().jpegenc_cmyk_to_ycck(int(r), int(g), int(b), int(k), out)
        assert tuple(out) == oracle.cmyk_to_ycck(int(r), int(g), int(b), int(k))


def test_sampling_factor_from_factors(binding):
    """SamplingFactor::from_factors (encoder.rs:157-171) and the table test at :1302-1321."""
    f = binding.lib().jpegenc_sampling_factor_from_factors
    valid = {(1, 1), (1, 2), (1, 4), (2, 1), (2, 2), (2, 4), (4, 1), (4, 2)}
    for h in range(0, 6):
        for v in range(0, 6):
            want = (h << 4) | v if (h, v) in valid else -1
            assert f(h, v) == want
    assert f(2, 2) == binding.F_2_2 and f(4, 1) == binding.F_4_1


@pytest.mark.parametrize("kw", [
    dict(quality=90), dict(quality=75, sampling=(2, 2), restart_interval=5), dict(quality=60, sampling=(4, 1)),
    dict(quality=85, progressive_scans=4), dict(quality=92, progressive_scans=7, restart_interval=3, sampling=(2, 1)),
    dict(quality=80, optimize=True), dict(quality=70, progressive_scans=5, optimize=True, sampling=(2, 2)),
    dict(quality=88, progressive_scans=40)],
    ids=["baseline", "420-restart", "sequential-411", "progressive", "progressive-restart", "optimised", "progressive-optimised", "progressive-40"])
def test_host_half_from_coefficients_without_a_gpu(binding, oracle, synth, kw):
    """jpegenc_encoder_encode_coefficients: the host half of the encoder on its own (markers, Huffman table
    construction incl. the optimised tables' histogram, the host entropy coder, restart bookkeeping) fed with the
    oracle's coefficients - the library's host logic, byte for byte against the reference restatement, on CPU."""
    for ct, bpp in ((oracle.RGB, 3), (oracle.LUMA, 1), (oracle.CMYK, 4), (oracle.YCCK, 4)):
        w, h = 203, 117
        px = synth.lcg_image(w, h, bpp, 60 + ct)
        if ct == oracle.LUMA:
            px = (np.add.outer(np.arange(h), np.arange(w))[..., None] // 2).astype(np.uint8)       # smooth: long zero runs
        e = binding.Encoder(kw["quality"])
        hs, vs = kw.get("sampling", (2, 2) if kw["quality"] < 90 else (1, 1))
        if "sampling" in kw:
            e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
        if kw.get("progressive_scans"):
            e.set_progressive_scans(kw["progressive_scans"])
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        if kw.get("optimize"):
            e.set_optimized_huffman_tables(True)
        order = e.block_order()
        co = oracle.encode_blocks(px, w, h, ct, hs, vs, kw["quality"], order)
        assert e.encode_coefficients(co, w, h, ct) == oracle.encode_jpeg(px, w, h, ct, **kw), (ct, kw)
    with pytest.raises(binding.JpegEncError) as err:
        e.encode_coefficients(co[:-1], w, h, ct)
    assert err.value.status == binding.ERR_BAD_IMAGE_DATA


def _configured(binding, kw):
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


def test_reference_round_trip_cases_through_the_host_half(binding, oracle, synth):
    """The reference's own round-trip tests on its 258x128 gradient (src/lib.rs:200-472; the case table is shared
    with tests/test_oracle_files.py, which decodes the oracle's files with libjpeg): the library's host half must
    produce those very files, and they must decode within the reference's tolerance (check_result, lib.rs:160-186)."""
    from test_oracle_files import RGB_CASES, _check
    px = synth.test_img_rgb()
    for name, kw in sorted(RGB_CASES.items()):
        e = _configured(binding, kw)
        hs, vs = kw.get("sampling", (2, 2) if kw["quality"] < 90 else (1, 1))
        co = oracle.encode_blocks(px, 258, 128, oracle.RGB, hs, vs, kw["quality"], e.block_order())
        data = e.encode_coefficients(co, 258, 128, binding.RGB)
        assert data == oracle.encode_jpeg(px, 258, 128, oracle.RGB, **kw), name
        _check(data, px, "RGB")


def test_host_half_randomised_without_a_gpu(binding, oracle):
    """Randomised sweep of the host half on CPU (the GPU suite's sweep covers the whole pipeline): every ColorType,
    sampling factor, scan mode incl. up to 64 progressive scans, restart intervals, optimised tables."""
    import os
    rng = np.random.default_rng(int(os.environ.get("JPEGENC_FUZZ_SEED", "77")))
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2), (1, 4), (2, 4)]
    for trial in range(int(os.environ.get("JPEGENC_HOST_FUZZ_TRIALS", "150"))):
        ct = int(rng.integers(0, 9))
        w, h = int(rng.integers(1, 160)), int(rng.integers(1, 100))
        px = rng.integers(0, 256, (h, w, binding.BPP[ct]), dtype=np.uint8)
        if trial % 3 == 0:
            px = (np.add.outer(np.arange(h), np.arange(w))[..., None] // 3 + np.arange(binding.BPP[ct])).astype(np.uint8)
        hs, vs = samplings[int(rng.integers(0, 8))]
        kw = dict(quality=int(rng.integers(1, 101)), sampling=(hs, vs))
        mode = int(rng.integers(0, 4))
        if mode == 1:
            kw["progressive_scans"] = int(rng.integers(2, 65))
        elif mode == 2:
            kw["optimize"] = True
        elif mode == 3:
            kw["progressive_scans"] = int(rng.integers(2, 8))
            kw["optimize"] = True
        if rng.integers(0, 3) == 0:
            kw["restart_interval"] = int(rng.integers(1, 40))
        e = _configured(binding, kw)
        if ct == oracle.LUMA:
            hs = vs = 1                                     # sampling is ignored for Luma (encoder.rs:574-576)
        co = oracle.encode_blocks(px, w, h, ct, hs, vs, kw["quality"], e.block_order())
        assert e.encode_coefficients(co, w, h, ct) == oracle.encode_jpeg(px, w, h, ct, **kw), (trial, ct, w, h, kw)


def test_host_half_metadata_and_custom_tables_without_a_gpu(binding, oracle, synth):
    """Density, APPn, chunked ICC profile, Exif (encoder.rs:374-435; writer.rs:216-239) and custom / preset
    quantisation tables in the DQT, through the host half on CPU."""
    w, h = 130, 70
    px = synth.lcg_image(w, h, 3, 5)
    icc = bytes(i % 251 for i in range(150 * 1024))                                   # three ICC chunks
    e = binding.Encoder(95)
    e.set_density(1, 300, 150)
    e.add_app_segment(15, b"HOHOHO\0")
    e.add_icc_profile(icc)
    e.add_exif_metadata(b"II*\0")
    cust = [int(v) for v in (np.arange(64) * 3 + 2)]
    e.set_quantization_tables(binding.Q_CUSTOM, 4, cust, None)                        # custom luma, preset 4 chroma
    segs = [(15, b"HOHOHO\0")] + oracle.icc_segments(icc) + [oracle.exif_segment(b"II*\0")]
    q = oracle.qtables(95, presets=(oracle.Q_CUSTOM, 4), customs=(cust, None))
    want = oracle.encode_jpeg(px, w, h, oracle.RGB, 95, density=(1, 300, 150), app_segments=segs,
                              qpresets=(oracle.Q_CUSTOM, 4), qcustoms=(cust, None))
    co = oracle.encode_blocks(px, w, h, oracle.RGB, 1, 1, None, e.block_order(), q=q)
    assert e.encode_coefficients(co, w, h, binding.RGB) == want


def _build_example(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "encode_ppm"
    libdir = os.path.join(root, "jpeg-encoder_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "encode_ppm.c"), "-o", str(exe), "-L" + libdir, "-ljpegenc_mi355x",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def _build_batch_example(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "batch_device"
    libdir = os.path.join(root, "jpeg-encoder_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "batch_device.c"), "-o", str(exe), "-L" + libdir, "-ljpegenc_mi355x", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_batch_example_builds_against_the_header(binding, tmp_path):
    """examples/batch_device.c (device-resident frames -> files, plain C + the HIP runtime's C API) compiles warning-free against the
    header; without arguments it prints its usage."""
    import os
    import subprocess
    if not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no ROCm headers on this host")
    exe = _build_batch_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_c_example_builds_against_the_header(binding, tmp_path):
    """examples/encode_ppm.c: plain C against include/jpegenc_mi355x.h and the shared library (no torch in
    the process); without arguments it prints its usage."""
    import subprocess
    exe = _build_example(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def _build_cpp_example(tmp_path):
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "encode_cpp"
    libdir = os.path.join(root, "jpeg-encoder_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "encode_cpp.cpp"), "-o", str(exe), "-L" + libdir, "-ljpegenc_mi355x",
                    "-Wl,-rpath," + libdir], check=True)
    return exe


def test_cpp_face_builds_and_fails_loudly_without_a_gpu(binding, tmp_path):
    """include/jpegenc_mi355x.hpp (the reference's Encoder API over the C ABI, header-only) compiles warning-free;
    without a GPU the example ends in EncodingError{NoDevice} - there is no CPU fallback to fall into."""
    import subprocess
    import torch
    exe = _build_cpp_example(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the example for real")
    r = subprocess.run([str(exe), str(tmp_path / "some.jpeg")], capture_output=True, text=True)
    assert r.returncode == 2 and "no usable gfx950 device" in r.stderr, (r.returncode, r.stderr)


def test_integration_notes_show_every_entry_point():
    """INTEGRATION.md's Rust `extern "C"` block binds everything include/jpegenc_mi355x.h declares."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "jpegenc_mi355x.h")).read()
    notes = open(os.path.join(root, "INTEGRATION.md")).read()
    names = sorted(set(re.findall(r"\b(jpegenc_[a-z0-9_]+)\s*\((?!\*)", header)))
    assert len(names) > 40
    missing = [n for n in names if n not in notes]
    assert not missing, missing


def _k1_code_sizes(freq):
    """Figure K.1 of T.81 as huffman.rs:103-157 runs it (an independent reading: plain Python, no library)."""
    freq = list(freq)
    others, size = [-1] * 257, [0] * 257
    while True:
        v1 = v2 = None
        least = 1 << 62
        for i, f in enumerate(freq):
            if f and f <= least:
                least, v1 = f, i
        if v1 is None:
            break
        least = 1 << 62
        for i, f in enumerate(freq):
            if f and f <= least and i != v1:
                least, v2 = f, i
        if v2 is None:
            break
        freq[v1] += freq[v2]
        freq[v2] = 0
        size[v1] += 1
        while others[v1] >= 0:
            v1 = others[v1]
            size[v1] += 1
        others[v1] = v2
        size[v2] += 1
        while others[v2] >= 0:
            v2 = others[v2]
            size[v2] += 1
    return size


def _fibonacci_ac_blocks(nsym):
    """Blocks (zig-zag i16, DC 0) whose AC symbol histogram grows like the Fibonacci numbers over `nsym` (run, size)
    symbols, and that histogram.  Symbols: run 0 for the 15 most frequent, then run 1, then run 2; blocks hold one
    run class each (a symbol takes run + 1 slots), end in a non-zero coefficient (no EOB) and are topped up with
    the most frequent symbol (0, 1), whose exact count does not matter to the tree's depth."""
    fib = [1, 1]
    while len(fib) < nsym:
        fib.append(fib[-1] + fib[-2])
    symbols = [(r, s) for r in range(3) for s in range(1, 16)][:nsym]
    counts = dict(zip(symbols, reversed(fib)))
    fillers = placed01 = 0
    classes = []
    for r in (2, 1, 0):
        want = {s: c for (rr, s), c in counts.items() if rr == r}
        if r == 0:
            want[1] = placed01 = max(want[1] - fillers, 1)
        if not want:
            continue
        vals = np.concatenate([np.full(c, 1 << (s - 1), dtype=np.int16) for s, c in sorted(want.items())])
        per = 63 // (r + 1)
        n = -(-len(vals) // per)
        blk = np.zeros((n, 64), dtype=np.int16)
        full = np.zeros(n * per, dtype=np.int16)
        full[:len(vals)] = vals
        blk[:, r + 1::r + 1][:, :per] = full.reshape(n, per)
        last_used = (len(vals) - (n - 1) * per) * (r + 1)                 # zig-zag index of the last real symbol of the last block
        blk[n - 1, last_used + 1:] = 1                                  # top up: (0, 1) symbols
        fillers += 63 - last_used
        if per * (r + 1) < 63:                                          # run 1: slot 63 of every block
            fillers += int(np.count_nonzero(blk[:n - 1, 63] == 0))
            blk[:n - 1, 63] = 1
        classes.append(blk)
    hist = [0] * 257
    for (r, s), c in counts.items():
        hist[(r << 4) | s] = c
    hist[0x01] = placed01 + fillers
    hist[256] = 1
    return np.concatenate(classes[::-1]), hist


def test_optimised_table_with_a_code_longer_than_32_bits_is_an_error(binding, oracle):
    """HuffmanTable::new_optimized indexes `bits: [u8; 33]` with the code size (huffman.rs:161-165) and panics when
    Figure K.1 produces a size above 32 - an AC histogram that grows like the Fibonacci numbers over 36 symbols does.
    The drop-in must report an error there (it used to write past a stack array and emit a corrupt DHT)."""
    # the construction, checked small: its designed histogram is what the oracle counts on its blocks
    small, hist = _fibonacci_ac_blocks(20)
    got = oracle.histogram(small, len(small) * 8, 8, oracle.LUMA, 1, 1)[0, 1]
    assert [int(v) for v in got] == hist
    assert max(_k1_code_sizes(hist)) <= 32
    e = binding.Encoder(90)
    e.set_optimized_huffman_tables(True)
    jpg = e.encode_coefficients(small, len(small) * 8, 8, binding.LUMA)       # 20 symbols: a valid optimised file ...
    bits, vals = oracle.huffman_optimized(hist)                         # ... whose AC DHT is the oracle's table for that histogram
    assert bytes([0xFF, 0xC4]) + (2 + 1 + 16 + len(vals)).to_bytes(2, "big") + bytes([0x10]) + bytes(bits) + bytes(vals) in jpg
    blocks, hist = _fibonacci_ac_blocks(36)
    assert max(_k1_code_sizes(hist)) > 32, "the construction must reach a 33-bit code"
    cols = 8000
    rows = -(-len(blocks) // cols)
    blocks = np.concatenate([blocks, np.repeat(blocks[:1], cols * rows - len(blocks), axis=0)])
    w, h = cols * 8, rows * 8
    assert w <= 65535
    with pytest.raises(binding.JpegEncError) as err:
        e.encode_coefficients(blocks, w, h, binding.LUMA)
    assert err.value.status == binding.ERR_INVALID_ARGUMENT and "32 bits" in str(err.value)
    e.set_optimized_huffman_tables(False)                               # the same frame with fixed tables is fine
    assert e.encode_coefficients(blocks, w, h, binding.LUMA)[:2] == b"\xff\xd8"


def test_frames_the_device_entropy_coder_declines(binding):
    """The decision behind the per-frame fall-back of jpegenc_encoder_encode_batch_device: jpegenc_scan_max_bytes /
    _workspace_size are 0 from about 2.45 M blocks (32-bit bit offsets), non-zero below."""
    def cap(w, h, ct, hs, vs):
        L = binding.layout(w, h, ct, hs, vs, binding.ORDER_MCU)
        s = binding.baseline_scan()
        return binding.scan_max_bytes(L, s), binding.scan_workspace_size(L, s, 1)
    assert cap(10000, 8000, binding.RGB, 1, 1) == (0, 0)
    assert cap(16384, 16384, binding.RGB, 2, 2) == (0, 0)
    assert cap(65535, 65535, binding.RGB, 2, 2) == (0, 0)
    assert cap(16384, 10000, binding.LUMA, 1, 1) == (0, 0)
    ok = cap(3840, 2160, binding.RGB, 2, 2)
    assert ok[0] > 0 and ok[1] > 0
    ok = cap(7680, 4320, binding.CMYK, 1, 1)
    assert ok[0] > 0 and ok[1] > 0


def test_which_layouts_take_the_one_kernel_path(binding):
    """jpegenc_pixels_scan_fused (no GPU needed): the pixels -> bits kernel takes every ColorType whose MCU has at most 6
    blocks (and at least 3) at sampling factors 1 and 2 - what the Encoder's interleaved baseline scan runs through - and declines the rest
    (4-component 2x2 layouts: 7 or 10 blocks per MCU; frames the device coder declines)."""
    b = binding
    yes = [(b.RGB, 1, 1), (b.RGB, 2, 1), (b.RGB, 1, 2), (b.RGB, 2, 2), (b.RGBA, 2, 2), (b.BGR, 2, 1), (b.BGRA, 1, 1),
           (b.YCBCR, 2, 2), (b.YCBCR, 1, 1), (b.CMYK, 1, 1), (b.CMYK, 2, 1), (b.YCCK, 1, 1), (b.YCCK, 2, 1), (b.CMYK_AS_YCCK, 1, 1), (b.CMYK, 1, 2)]
    no = [(b.CMYK, 2, 2), (b.YCCK, 2, 2), (b.CMYK_AS_YCCK, 2, 2)]
    assert not b.pixels_scan_fused(640, 360, b.LUMA, 1, 1)      # one block per MCU: a one-wave workgroup per 64 blocks does not pay
    for ct, hs, vs in yes:
        assert b.pixels_scan_fused(640, 360, ct, hs, vs), (ct, hs, vs)
    for ct, hs, vs in no:
        L = b.layout(640, 360, ct, hs, vs, b.ORDER_MCU)
        assert int(L.total_blocks) // int(L.mcus) > 6
        assert not b.pixels_scan_fused(640, 360, ct, hs, vs), (ct, hs, vs)
    # (the device coder's 32-bit bit offsets end at about 2.45 M blocks: 96 Mpixel 4:2:0 frames pass, 128 Mpixel ones do not)
    assert b.pixels_scan_fused(12000, 8000, b.RGB, 2, 2) and not b.pixels_scan_fused(16000, 8000, b.RGB, 2, 2)


def test_shipping_library_reads_no_diagnostic_switches(pkg):
    """The shipping library's behaviour must not depend on the caller's environment: the diagnostic switches
    (JPEGENC_FUSED, JPEGENC_PACK_WINDOW_WORDS, ... - INTEGRATION.md section 6) are compiled into the -DJPEGENC_DIAG build
    only (csrc/diag_env.h).  What is left in libjpegenc_mi355x.so: JPEGENC_TRACE and JPEGENC_NUMA_BIND."""
    import importlib
    import re
    b = importlib.import_module("jpeg_encoder_amd.binding")
    default = os.path.join(os.path.dirname(b.DIAG_LIB_PATH), "libjpegenc_mi355x.so")
    names = lambda path: set(m.decode() for m in re.findall(rb"JPEGENC_[A-Z0-9_]{3,}", open(path, "rb").read()))
    assert names(default) <= {"JPEGENC_TRACE", "JPEGENC_NUMA_BIND"}, sorted(names(default))
    assert os.path.exists(b.DIAG_LIB_PATH), "build.sh builds the diagnostic library beside the shipping one"
    diag = names(b.DIAG_LIB_PATH)
    assert {"JPEGENC_FUSED", "JPEGENC_PACK_WINDOW_WORDS", "JPEGENC_SCANS_ONE_BY_ONE", "JPEGENC_TRACE"} <= diag
    # same exported entry points
    import subprocess
    syms = lambda path: set(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", path], text=True).splitlines()
                            if " T " in l and "jpegenc_" in l)
    assert syms(default) == syms(b.DIAG_LIB_PATH)


def test_host_copy_is_a_plain_copy_that_needs_no_device(binding):
    """jpegenc_host_copy: the batch workers' staging copy (streaming stores above 256 KB), exported for callers that fill their own
    page-locked pools - every length and alignment copies exactly, nothing beyond the range is touched."""
    import numpy as np
    rng = np.random.default_rng(3)
    for n in (0, 1, 31, 4096, (256 << 10) - 1, (256 << 10) + 77, 3 * 1920 * 1080):
        for so, do in ((0, 0), (1, 3), (13, 32)):
            src = rng.integers(0, 256, n + 64, dtype=np.uint8)
            dst = np.full(n + 128, 0xA5, dtype=np.uint8)
            assert binding.lib().jpegenc_host_copy(dst.ctypes.data + do + 32, src.ctypes.data + so, n) == binding.OK
            assert np.array_equal(dst[do + 32:do + 32 + n], src[so:so + n])
            assert (dst[:do + 32] == 0xA5).all() and (dst[do + 32 + n:] == 0xA5).all()
    assert binding.lib().jpegenc_host_copy(None, None, 16) == binding.ERR_INVALID_ARGUMENT
