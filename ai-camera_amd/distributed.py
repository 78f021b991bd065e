"""Multi-GPU layout: independent video streams shard one per GPU, one process per
GPU, no data-path collective.  The only exchange step is the OPTIONAL cross-camera ReID gallery
all-gather of BASELINE.json configs[4] (not in the reference; README.md:210 lists it as future
work): every K frames each rank contributes a fixed-shape shard of its confirmed tracks'
latest embeddings; RCCL (backend "nccl") over xGMI on GPUs, gloo on CPU for the tests.

The message is tiny (T_max x D fp32 = 256 KiB per rank) and therefore latency-bound: one
all_gather_into_tensor of the packed shard per exchange -- never per track, never per frame."""
from __future__ import annotations

import os

import numpy as np

GALLERY_T_MAX = 128


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1-process defaults)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, local_rank, world = dist_env()
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return dist


def stream_seed(base_seed: int, rank: int) -> int:
    """configs[3]: rank r processes the synthetic stream with seed base+r (independent cameras)."""
    return base_seed + rank


def shard_streams(n_streams: int, world: int, rank: int):
    """Stream ids owned by `rank` when n_streams cameras are spread over `world` GPUs."""
    return list(range(rank, n_streams, world))


def pack_gallery_shard(track_ids, embeddings, dim, t_max=GALLERY_T_MAX):
    """fp32 [t_max, 2 + dim]: column 0 = valid flag, column 1 = track id, rest = embedding."""
    shard = np.zeros((t_max, 2 + dim), np.float32)
    n = min(len(track_ids), t_max)
    if n:
        shard[:n, 0] = 1.0
        shard[:n, 1] = np.asarray(track_ids[:n], np.float32)
        shard[:n, 2:] = np.asarray(embeddings[:n], np.float32)
    return shard


def unpack_gallery(gathered, world):
    """-> list over ranks of (track_ids int32 [k], embeddings fp32 [k, dim])."""
    out = []
    g = np.asarray(gathered).reshape(world, -1, gathered.shape[-1])
    for r in range(world):
        valid = g[r, :, 0] > 0.5
        out.append((g[r, valid, 1].astype(np.int32), g[r, valid, 2:]))
    return out


def all_gather_gallery(shard: np.ndarray, device=None):
    """One collective per exchange. Returns the [world, t_max, 2+dim] array on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return shard[None]
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(shard))
    if dist.get_backend() == "nccl":
        t = t.cuda(device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t)          # ranks concatenated along dim 0
    return out.cpu().numpy().reshape((world,) + tuple(t.shape))


def cross_camera_matches(gathered_per_rank, my_rank, max_cosine_distance=0.2):
    """For each local confirmed track: the closest track of any OTHER camera within the cosine
    threshold (annotation only; per-stream association is untouched). Returns {local_id: (rank, id, d)}."""
    from .core.matching import cosine_distance   # HIP cosine kernel (aic_appearance_cost)
    ids, emb = gathered_per_rank[my_rank]
    res = {}
    if not len(ids):
        return res
    for r, (oids, oemb) in enumerate(gathered_per_rank):
        if r == my_rank or not len(oids):
            continue
        d = cosine_distance(emb, oemb)
        j = d.argmin(1)
        for i, tid in enumerate(ids):
            dist_ij = float(d[i, j[i]])
            if dist_ij <= max_cosine_distance and (tid not in res or dist_ij < res[tid][2]):
                res[int(tid)] = (r, int(oids[j[i]]), dist_ij)
    return res


def reduce_max_time(seconds: float) -> float:
    """MAX over ranks of a wall-clock interval (the bench contract)."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_sum(value: float) -> float:
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
