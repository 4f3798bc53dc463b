"""BASELINE config 3: a 1000-frame batch of 1920x1080 RGB q=80 4:2:0 frames sharded frame-wise over the
GPUs of a node, one process (rank) per GPU.

JPEG frames are independent and all distinct (frame k seeded 42 + k): frame k belongs to shard k % world (`jpegenc_shard_frames`, the same C
function the library's own multi-device batch uses), every rank encodes its own frames from pageable host
memory to complete JPEG files in host buffers, and NO pixel or coefficient ever crosses ranks.  The only
exchange is bookkeeping - frame counts, the slowest rank's wall time, per-frame digests - through two small
`torch.distributed` tensor all-reduces, which run the same over RCCL (bench.py --gpus N) and gloo (the CPU
test, tests/test_batch_gloo.py, which injects the per-frame encoder and keeps everything else).
"""
import hashlib
import time

import numpy as np

C3_W, C3_H, C3_QUALITY, C3_FRAMES = 1920, 1080, 80, 1000      # BASELINE.json configs[2]; q=80 -> default F_2_2 (encoder.rs:256-260)


def photo_like_frame(synth, k, w=C3_W, h=C3_H):
    """Frame k of the batch (numpy form): the reference's test gradient (lib.rs:81-98) scaled to the frame size, shifted by
    16 k columns and carrying +-6 of noise seeded 42 + k (frame k seeded 42+k) - entropy-codes like a
    photograph.  All frames of a batch are distinct."""
    g = np.roll(synth.test_img_rgb(w, h), (16 * k) % w, axis=1).astype(np.int16)
    rng = np.random.default_rng(42 + k)
    return np.clip(g + rng.integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8)


class ShardFrames:
    """make_frame(k) for the frames of ONE rank's shard: every frame of the batch is distinct (gradient shifted by 16 k
    columns + noise seeded 42 + k), a rank only ever materialises the frames it owns (6.2 GB / N for config 3), and they
    stay resident in pageable host memory.  With a torch device the frames are generated there (seconds for a thousand
    1080p frames instead of most of a minute in numpy) and copied to the host; the two generators draw different noise, so
    a digest is comparable between runs of the same kind only."""

    def __init__(self, synth, w=C3_W, h=C3_H, torch=None, device=None):
        self.synth, self.w, self.h, self.torch, self.device, self.cache = synth, w, h, torch, device, {}
        self._base = None

    def _device_frame(self, k):
        torch = self.torch
        if self._base is None:
            self._base = torch.from_numpy(self.synth.test_img_rgb(self.w, self.h)).to(self.device).to(torch.int16)
            self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(42 + k)
        g = torch.roll(self._base, shifts=(16 * k) % self.w, dims=1)
        noise = torch.randint(-6, 7, g.shape, dtype=torch.int16, device=self.device, generator=self._gen)
        return np.ascontiguousarray(torch.clamp(g + noise, 0, 255).to(torch.uint8).cpu().numpy())

    def materialise(self, ks):
        for k in ks:
            self(k)
        self._base = None                                          # (device memory is not needed any more)

    def __call__(self, k):
        if k not in self.cache:
            self.cache[k] = self._device_frame(k) if self.torch is not None and self.device is not None else \
                np.ascontiguousarray(photo_like_frame(self.synth, k, self.w, self.h))
        return self.cache[k]


POOL = 25        # tests/test_gpu_batch_multi.py: a cycling pool, so that the oracle has 25 files to produce instead of 1000


class FramePool:
    """make_frame(k) = distinct frame k % POOL: for the parity tests, which need the oracle's file of every distinct frame.
    bench.py's c3_batch leg uses ShardFrames (every frame distinct)."""

    def __init__(self, synth, w=C3_W, h=C3_H):
        self.synth, self.w, self.h, self.cache = synth, w, h, {}

    def __call__(self, k):
        key = k % POOL
        if key not in self.cache:
            self.cache[key] = np.ascontiguousarray(photo_like_frame(self.synth, key, self.w, self.h))
        return self.cache[key]


def per_rank_table(dist, values, world, rank, device=None, force=False):
    """[world][len(values)] with row `rank` = values on every rank: one SUM all-reduce of a table every rank writes one row of
    (the same bookkeeping exchange run_sharded_batch uses; no pixels, no files)."""
    table = np.zeros((world, len(values)), dtype=np.float64)
    table[rank] = values
    if _dist_ready(dist, force):
        import torch
        t = torch.from_numpy(table).to(device) if device is not None else torch.from_numpy(table)
        dist.all_reduce(t)
        table = t.cpu().numpy()
    return table


def _dist_ready(dist, force=False):
    return dist is not None and dist.is_available() and dist.is_initialized() and (force or dist.get_world_size() > 1)


def run_sharded_batch(binding, encode_frames, make_frame, num_frames, width, height, world=1, rank=0, dist=None,
                      warmup_frames=0, digests=True, device=None, force_collectives=False):
    """One rank's share of a frame-sharded batch + the bookkeeping exchange.

    encode_frames(list of HxWx3 uint8 arrays) -> list of bytes : this rank's encoder (bench.py: the library's
        batch API on the rank's GPU; the gloo test: anything, e.g. the oracle).
    make_frame(k) -> pixels of frame k of the batch.
    device: where the bookkeeping tensors live (the rank's GPU under RCCL, None = CPU under gloo).
    force_collectives: run the barrier and the all-reduces even with a single rank (a self-test of the RCCL path).
    Returns a dict every rank agrees on: frames per rank, wall time of the slowest rank, aggregate rates, and -
    with digests - a checksum of the per-frame checksums in frame order (independent of world size).
    The exchange is two small tensor all-reduces (no pixels, no coefficients, no files)."""
    mine = binding.shard_frames(num_frames, world, rank)
    # A rank that fails - materialising its shard, in the warm-up, in the timed pass - must still take part in the barrier and
    # the all-reduces below, or the other ranks wait for it until the collective times out: the failure travels in the table
    # and every rank raises together afterwards.
    failure, frames, files, seconds = None, [], [], -1.0
    try:
        frames = [make_frame(k) for k in mine]
        if warmup_frames and frames:
            encode_frames(frames[:warmup_frames])                  # buffers, page faults, clocks
    except Exception as exc:                                       # noqa: BLE001 - reported after the collectives
        failure = exc
    if _dist_ready(dist, force_collectives):
        dist.barrier()
    if failure is None:
        try:
            t0 = time.perf_counter()
            files = encode_frames(frames) if frames else []
            seconds = time.perf_counter() - t0
            if len(files) != len(mine):
                raise RuntimeError(f"rank {rank}: {len(files)} files for {len(mine)} frames")
        except Exception as exc:                                   # noqa: BLE001
            failure, files = exc, []
    # per-rank slots (frames, seconds, bytes, failed) and per-frame (owner count, 60 bits of SHA-256): every entry is written by
    # exactly one rank, so a SUM all-reduce assembles the table on every rank
    stats = np.zeros((world, 4), dtype=np.float64)
    stats[rank] = (len(mine), seconds, sum(len(f) for f in files), 0.0 if failure is None else 1.0)
    table = np.zeros((num_frames, 2), dtype=np.int64)
    for k, f in zip(mine, files):
        table[k, 0] = 1
        if digests:
            table[k, 1] = int.from_bytes(hashlib.sha256(f).digest()[:8], "big") >> 4
    if _dist_ready(dist, force_collectives):
        import torch
        t_stats = torch.from_numpy(stats).to(device) if device is not None else torch.from_numpy(stats)
        t_table = torch.from_numpy(table).to(device) if device is not None else torch.from_numpy(table)
        dist.all_reduce(t_stats)
        dist.all_reduce(t_table)
        stats, table = t_stats.cpu().numpy(), t_table.cpu().numpy()
    if failure is not None:
        raise failure
    if stats[:, 3].any():
        raise RuntimeError(f"rank(s) {[int(r) for r in np.flatnonzero(stats[:, 3])]} failed their shard of the batch")
    if not np.array_equal(table[:, 0], np.ones(num_frames, dtype=np.int64)):
        bad = np.flatnonzero(table[:, 0] != 1)
        raise RuntimeError(f"sharding lost or duplicated frames, e.g. frame {int(bad[0])} encoded {int(table[bad[0], 0])} times")
    total = int(stats[:, 0].sum())
    if total != num_frames:
        raise RuntimeError(f"sharding lost frames: {total} of {num_frames}")
    slowest = float(stats[:, 1].max())
    out = {"frames": total, "per_rank_frames": [int(v) for v in stats[:, 0]], "seconds": round(slowest, 6),
           "per_rank_seconds": [round(float(v), 6) for v in stats[:, 1]],
           "frames_per_s": round(total / slowest, 1) if slowest > 0 else None,
           "Mpixels_per_s": round(total * width * height / slowest / 1e6, 1) if slowest > 0 else None,
           "jpeg_bytes_per_frame": int(stats[:, 2].sum() / max(total, 1))}
    if digests:
        out["digest"] = hashlib.sha256(b"".join(int(v).to_bytes(8, "big") for v in table[:, 1])).hexdigest()[:16]
    return out, dict(zip(mine, files))
