"""N>1 path on CPU: a world_size-2 gloo job drives jpeg_encoder_amd.batch.run_sharded_batch - the SAME function
bench.py's `c3_batch` leg runs on every rank - with the per-rank encoder injected (the oracle stands in for the
device encoder here: test infrastructure only).  Sharding rule, frame bookkeeping, the MAX-over-ranks timing and
the checksum of checksums are the code under test; the rule itself is the library's C function
jpegenc_shard_frames, which jpegenc_encoder_encode_batch_multi uses as well."""
import hashlib
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, W, H, Q = 7, 96, 64, 80

WORKER = textwrap.dedent("""
    import importlib, json, os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    batch = importlib.import_module("jpeg_encoder_amd.batch")
    binding = importlib.import_module("jpeg_encoder_amd.binding")
    synth = importlib.import_module("jpeg_encoder_amd.synth")
    from oracle import pyoracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, W, H, Q = {n}, {w}, {h}, {q}
    make = lambda k: synth.lcg_image(W, H, 3, 42 + k)
    enc = lambda frames: [pyoracle.encode_jpeg(px, W, H, pyoracle.RGB, Q) for px in frames]
    result, mine = batch.run_sharded_batch(binding, enc, make, N, W, H, world, rank, dist)
    assert sorted(mine) == binding.shard_frames(N, world, rank)
    dist.barrier()
    print("RESULT " + json.dumps(result))
    dist.destroy_process_group()
""")


def _digest_of(files):
    """The checksum of checksums run_sharded_batch reports: SHA-256 over 60 bits of every file's SHA-256, in frame order."""
    return hashlib.sha256(b"".join((int.from_bytes(hashlib.sha256(f).digest()[:8], "big") >> 4).to_bytes(8, "big")
                                   for f in files)).hexdigest()[:16]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_frames_partition(pkg):
    import importlib
    b = importlib.import_module("jpeg_encoder_amd.binding")
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 4, 8):
            shards = [b.shard_frames(n, world, r) for r in range(world)]
            assert sorted(k for s in shards for k in s) == list(range(n))
            assert max(map(len, shards)) - min(map(len, shards)) <= 1
    assert b.shard_frames(1000, 8, 3)[:3] == [3, 11, 19]           # frame k -> GPU k mod 8
    assert len(b.shard_frames(1000, 8, 7)) == 125
    for bad in ((-1, 2, 0), (5, 0, 0), (5, 2, 2), (5, 2, -1)):
        try:
            b.shard_frames(*bad)
        except b.JpegEncError as exc:
            assert exc.status == b.ERR_INVALID_ARGUMENT
        else:
            raise AssertionError(bad)


def test_single_process_batch(pkg, oracle, synth):
    """world = 1: the same function without torch.distributed."""
    import importlib
    batch = importlib.import_module("jpeg_encoder_amd.batch")
    b = importlib.import_module("jpeg_encoder_amd.binding")
    make = lambda k: synth.lcg_image(W, H, 3, 42 + k)
    enc = lambda frames: [oracle.encode_jpeg(px, W, H, oracle.RGB, Q) for px in frames]
    result, files = batch.run_sharded_batch(b, enc, make, N, W, H)
    assert result["frames"] == N and result["per_rank_frames"] == [N] and sorted(files) == list(range(N))
    assert result["digest"] == _digest_of([enc([make(k)])[0] for k in range(N)])


def test_world_size_2_gloo(tmp_path, oracle, synth):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n=N, w=W, h=H, q=Q))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-2000:]
    results = [json.loads([l for l in out.splitlines() if l.startswith("RESULT ")][0][len("RESULT "):]) for out, _ in outs]
    assert results[0] == results[1]                                   # every rank agrees on the bookkeeping
    r = results[0]
    assert r["frames"] == N and r["per_rank_frames"] == [4, 3] and len(r["per_rank_seconds"]) == 2
    assert r["seconds"] == max(r["per_rank_seconds"])                 # MAX over ranks
    # checksum of checksums == a single-process encode of every frame, whatever the world size
    assert r["digest"] == _digest_of([oracle.encode_jpeg(synth.lcg_image(W, H, 3, 42 + k), W, H, oracle.RGB, Q) for k in range(N)])


def test_a_failing_rank_does_not_leave_the_others_in_a_collective(tmp_path):
    """One rank's encoder raises in the timed pass: the failure travels through the bookkeeping all-reduce and BOTH ranks raise
    promptly - no rank is left waiting in a barrier or an all-reduce until the collective times out."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, n=N, w=W, h=H, q=Q).replace(
        "enc = lambda frames:", "enc = (lambda frames: (_ for _ in ()).throw(RuntimeError('rank 1 breaks'))) if rank == 1 else lambda frames:"))
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]               # (a hang would run into the 30-minute gloo timeout: caught here)
    assert procs[0].returncode != 0 and procs[1].returncode != 0
    assert "failed their shard" in outs[0][1] and "rank 1 breaks" in outs[1][1]
