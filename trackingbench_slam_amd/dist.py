"""Multi-GPU sharding of the tracking hot path: one process per GPU, frames shard embarrassingly
 -- no collective inside the data path -- and ONE exchange step at the end of a batch: the
per-frame track records (keypoints, descriptors, matches, pose) are gathered to rank 0 over RCCL
(`torch.distributed` backend "nccl" on ROCm) or gloo (CPU tests).

The reference has no distributed code at all; this module is the MI355X-side design.
"""
import os

import torch
import torch.distributed as dist


def shard_range(total, world_size, rank):
    """Contiguous block of frame indices [lo, hi) owned by `rank` (B/ngpu frames each, remainder spread)."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns
    (rank, world_size, local_rank). Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


_GATHER_BUFFERS = {}  # (name, shape, dtype, device, world) -> list of receive tensors, reused from batch to batch


def gather_tracks(records, dst=0, group=None, concat=True):
    """Gather fixed-shape per-rank track tensors to `dst`.

    records: dict name -> tensor [F_local, ...] (same shape and dtype on every rank: rows are padded to the
    plan's keypoint capacity, the true lengths travel in the *_counts tensors).  Returns on `dst` a dict
    name -> tensor [world * F_local, ...] in rank order (= global frame order for contiguous shards), and
    None elsewhere.  With one process it returns the records unchanged.  concat=False returns the per-rank
    parts instead (a list per name, receive buffers that the next call reuses): the steady-state form, one
    exchange step per batch with no allocation and no extra copy on the receiving rank."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(records)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = {} if rank == dst else None
    for name in sorted(records):
        t = records[name].contiguous()
        if rank == dst:
            key = (name, tuple(t.shape), t.dtype, str(t.device), world)
            parts = _GATHER_BUFFERS.get(key)
            if parts is None:
                parts = [torch.empty_like(t) for _ in range(world)]
                _GATHER_BUFFERS[key] = parts
            dist.gather(t, gather_list=parts, dst=dst, group=group)
            out[name] = torch.cat(parts, 0) if concat else parts
        else:
            dist.gather(t, gather_list=None, dst=dst, group=group)
    return out


def gather_tracks_async(records, dst=0, group=None, slot=0):
    """The exchange step as a background operation: like gather_tracks(..., concat=False), but returns
    (parts or None, handles) at once; call wait_tracks(handles) before the record tensors are written again (RCCL runs
    the collective on its own stream after the work already queued on the current one; wait() orders the current
    stream behind it, it does not block the host). `slot` picks the receive-buffer set on `dst`: alternate it when two
    exchanges are in flight."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(records), []
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = {} if rank == dst else None
    handles = []
    for name in sorted(records):
        t = records[name].contiguous()
        if rank == dst:
            key = (name, tuple(t.shape), t.dtype, str(t.device), world, slot)
            parts = _GATHER_BUFFERS.get(key)
            if parts is None:
                parts = [torch.empty_like(t) for _ in range(world)]
                _GATHER_BUFFERS[key] = parts
            handles.append(dist.gather(t, gather_list=parts, dst=dst, group=group, async_op=True))
            out[name] = parts
        else:
            handles.append(dist.gather(t, gather_list=None, dst=dst, group=group, async_op=True))
    return out, handles


def wait_tracks(handles):
    for h in handles:
        h.wait()


def pipeline_records(p):
    """The track records of a TrackingPipeline batch as a dict of device tensors."""
    return {
        "kps": p.trk_kps, "desc": p.trk_desc, "kp_counts": p.trk_counts,
        "matches": p.matches, "match_counts": p.match_counts,
        "pose": p.Tout, "n_inliers": p.n_inliers,
    }
