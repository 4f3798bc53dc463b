"""N>1 path on CPU: world_size-2 gloo job.  The per-frame encoder is injected, so the sharding /
manifest logic that bench.py --gpus N and the batch config (C3) rely on is exercised without a GPU
(the oracle stands in for the device encoder here — test infrastructure only)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import importlib, json, os, sys
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import __graft_entry__ as ge
    ge.load_package()
    sharding = importlib.import_module("jpeg_encoder_amd.sharding")
    synth = importlib.import_module("jpeg_encoder_amd.synth")
    from oracle import pyoracle
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    N, W, H = 7, 96, 64
    make = lambda k: synth.lcg_image(W, H, 3, 42 + k)
    enc = lambda px: pyoracle.encode_jpeg(px, W, H, pyoracle.RGB, 80)
    local = sharding.encode_shard(N, make, enc, world, rank)
    assert sorted(local) == list(sharding.frames_for_rank(N, world, rank))
    manifest = sharding.gather_manifest(local, dist)
    dist.barrier()
    if rank == 0:
        print("MANIFEST " + json.dumps({{str(k): v for k, v in sorted(manifest.items())}}))
    dist.destroy_process_group()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_frames_for_rank_partition(pkg):
    import importlib
    sharding = importlib.import_module("jpeg_encoder_amd.sharding")
    for n in (0, 1, 7, 1000):
        for world in (1, 2, 4, 8):
            seen = sorted(k for r in range(world) for k in sharding.frames_for_rank(n, world, r))
            assert seen == list(range(n))
            sizes = [len(sharding.frames_for_rank(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(sharding.frames_for_rank(1000, 8, 3))[:3] == [3, 11, 19]      # frame k -> GPU k mod 8


def test_world_size_2_gloo(tmp_path, oracle, synth):
    import hashlib
    import json
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
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
    line = [l for l in outs[0][0].splitlines() if l.startswith("MANIFEST ")][0]
    manifest = json.loads(line[len("MANIFEST "):])
    assert sorted(int(k) for k in manifest) == list(range(7))
    for k in range(7):                      # same bytes as a single-process encode of every frame
        want = oracle.encode_jpeg(synth.lcg_image(96, 64, 3, 42 + k), 96, 64, oracle.RGB, 80)
        assert manifest[str(k)] == [len(want), hashlib.sha256(want).hexdigest()[:16]]
