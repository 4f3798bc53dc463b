"""Deterministic synthetic inputs (no files, no network).

These reproduce the image generators the reference uses in its own tests and benches so that
parity tests here read like the reference's:
  * gradient images   — src/lib.rs:81-147 (create_test_img_rgb/rgba/gray/cmyk)
  * LCG noise         — src/avx2/ycbcr.rs:183-203 (SimpleRng, seed 42, low byte of the state)
  * Criterion pattern — criterion/benches/encode.rs:6-55 (create_bench_img), any size
"""
import numpy as np


def _clamped_xy(width, height):
    y, x = np.mgrid[0:height, 0:width]
    return np.minimum(x, 255), y


def test_img_rgb(width=258, height=128):
    """src/lib.rs:81-98 — width 258 gives an odd MCU count per row."""
    x, y = _clamped_xy(width, height)
    return np.stack([x, y * 2, (x + y * 2) // 2], axis=-1).astype(np.uint8)


def test_img_rgba(width=258, height=128):
    """src/lib.rs:100-118"""
    x, y = _clamped_xy(width, height)
    return np.stack([x, y * 2, (x + y * 2) // 2, x], axis=-1).astype(np.uint8)


def test_img_gray(width=258, height=128):
    """src/lib.rs:120-134 — luma of the RGB gradient."""
    rgb = test_img_rgb(width, height).astype(np.int64)
    yy = (19595 * rgb[..., 0] + 38470 * rgb[..., 1] + 7471 * rgb[..., 2] + 0x7FFF) >> 16
    return yy.astype(np.uint8)


def test_img_cmyk(width=258, height=192):
    """src/lib.rs:136-153"""
    x, y = _clamped_xy(width, height)
    return np.stack([x, y * 3 // 2, (x + y * 3 // 2) // 2, 255 - (x + y) // 2], axis=-1).astype(np.uint8)


def lcg_bytes(n, seed=42):
    """state = state*6364136223846793005 + 1 (mod 2^64); byte = state & 0xFF.

    The low byte of that state only depends on the low byte of the previous state, so the stream
    is 256-periodic; generate one period and tile it.
    """
    period = np.empty(256, dtype=np.uint8)
    s = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(256):
        s = (s * 6364136223846793005 + 1) & 0xFFFFFFFFFFFFFFFF
        period[i] = s & 0xFF
    reps = -(-n // 256)
    return np.tile(period, reps)[:n].copy()


def lcg_image(width, height, channels=3, seed=42):
    return lcg_bytes(width * height * channels, seed).reshape(height, width, channels)


def criterion_pattern(width=2000, height=1800):
    """criterion/benches/encode.rs:6-55 scaled to any size (RGB)."""
    y, x = np.mgrid[0:height, 0:width].astype(np.int64)
    p = x * y
    img = np.stack([x % 256, x % 256, p % 256], axis=-1).astype(np.uint8)
    rules = [(29, (96, 96, 255)), (27, (255, 96, 96)), (25, (96, 255, 96)), (23, (0, 255, 0)),
             (21, (0, 0, 255)), (19, (255, 0, 0)), (17, (255, 255, 255)), (13, (0, 0, 0))]
    for mod, colour in rules:          # later entries have priority (first match wins upstream)
        img[p % mod == 0] = colour
    return img


def noise_image(width, height, channels=3, seed=0):
    """Full-entropy bytes (numpy PCG64) for kernel benchmarking at large sizes."""
    return np.random.default_rng(seed).integers(0, 256, (height, width, channels), dtype=np.uint8)
