"""Frame-level sharding of a batch across ranks (one process per GPU).

JPEG frames are independent: frame k goes to rank k mod world_size, every rank
encodes its own frames, and NO pixel or coefficient ever crosses ranks.  The only exchange is
bookkeeping (per-frame byte counts / digests) through `all_gather_object`, which works the same
over RCCL (GPU job) and gloo (CPU test).
"""
import hashlib


def frames_for_rank(num_frames, world_size, rank):
    """Indices of the frames rank `rank` encodes: k with k % world_size == rank."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank/world_size")
    return range(rank, num_frames, world_size)


def encode_shard(num_frames, make_frame, encode_fn, world_size=1, rank=0):
    """Encode this rank's frames.  make_frame(k) -> pixels, encode_fn(pixels) -> bytes."""
    return {k: encode_fn(make_frame(k)) for k in frames_for_rank(num_frames, world_size, rank)}


def gather_manifest(local, dist=None):
    """Every rank learns {frame: (size, sha256[:16])} for the whole batch.  `dist` is
    torch.distributed (initialised) or None for a single process."""
    mine = {k: (len(v), hashlib.sha256(v).hexdigest()[:16]) for k, v in local.items()}
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mine
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, mine)
    merged = {}
    for part in parts:
        for k, v in part.items():
            if k in merged:
                raise RuntimeError(f"frame {k} encoded by two ranks")
            merged[k] = v
    return merged
