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


# ---- packed exchange (round 3): rank 0 receives live rows, not capacity
_PACKED = {"kps": "kp_counts", "desc": "kp_counts", "matches": "match_counts"}   # variable-length record -> its per-frame counts


def pack_records(records, packer=None):
    """Move the live rows of the variable-length records to the front, frame after frame: records[name] is [F, cap, ...] with
    counts[f] live rows per frame -> a [F * cap, ...] tensor whose first sum(counts) rows are the live ones in frame order
    (the rows behind them are unspecified). Fixed-shape records pass through. Returns (packed dict, totals int64 [2] on the
    records' device: live keypoint rows, live match rows). Device-side torch ops only (stable sort of a 0/1 key)."""
    out = dict(records)
    totals = []
    order_cache = {}
    for name, cname in _PACKED.items():
        if name not in records:
            continue
        t, cnt = records[name], records[cname]
        F, cap = t.shape[0], t.shape[1]
        if packer is not None:      # the library's compaction kernel (tb_pack_rows_dev): one launch per record, no sort
            out[name] = packer(name, t, cnt)
            continue
        if cname not in order_cache:
            dead = (torch.arange(cap, device=t.device)[None, :] >= cnt[:, None].to(torch.int64)).reshape(F * cap)
            order_cache[cname] = torch.sort(dead.to(torch.uint8), stable=True).indices   # live rows first, original order kept
        out[name] = t.reshape((F * cap,) + tuple(t.shape[2:])).index_select(0, order_cache[cname])
    for cname in ("kp_counts", "match_counts"):
        totals.append(records[cname].to(torch.int64).sum() if cname in records else torch.zeros((), dtype=torch.int64, device=next(iter(records.values())).device))
    return out, torch.stack(totals)


def unpack_records(packed, counts_by_name):
    """Inverse of pack_records on the receiving side, for checks: list of per-frame row blocks."""
    res = {}
    for name, cname in _PACKED.items():
        if name in packed:
            cnt = counts_by_name[cname].tolist()
            off, rows = 0, []
            for c in cnt:
                rows.append(packed[name][off:off + c])
                off += c
            res[name] = rows
    return res


def gather_tracks_packed(records, dst=0, group=None, slot=0, async_op=True, packer=None):
    """The exchange step with compacted payloads (SURVEY 8e's first option: counts, then live records). Every rank packs its
    variable-length records (pack_records); one tiny all-reduce (MAX) tells everybody the largest live row counts of the
    batch; the SAME gather collective as gather_tracks then moves only that many rows per rank -- equal shapes, as RCCL's
    gather wants, but sized by the data (ranks hold the same workload, so the largest rank is within a few rows of the
    others) instead of by the plan's capacity. Reading the two maxima on the host waits for the chain that wrote the
    records; bench.py calls this right after step(), which has joined the BA partitions by then, so the extractor chain is
    long done. Returns (parts or None, handles, bytes this rank contributed, rows = (kp_rows, match_rows) per rank sent)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dict(records), [], 0, (0, 0)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    packed, totals = pack_records(records, packer)
    mx = totals.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    kp_rows, m_rows = (int(v) for v in mx.cpu().tolist())
    rows = {"kp_counts": max(kp_rows, 1), "match_counts": max(m_rows, 1)}
    out = {} if rank == dst else None
    handles, nbytes = [], 0
    for name in sorted(packed):
        t = packed[name]
        if name in _PACKED:
            t = t[:rows[_PACKED[name]]]
        t = t.contiguous()
        nbytes += t.numel() * t.element_size()
        if rank == dst:
            full = packed[name]
            key = ("packed", name, tuple(full.shape), full.dtype, str(full.device), world, slot)
            bufs = _GATHER_BUFFERS.get(key)
            if bufs is None:
                bufs = [torch.empty_like(full) for _ in range(world)]
                _GATHER_BUFFERS[key] = bufs
            parts = [b[:t.shape[0]] for b in bufs] if name in _PACKED else bufs
            h = dist.gather(t, gather_list=parts, dst=dst, group=group, async_op=async_op)
            out[name] = parts
        else:
            h = dist.gather(t, gather_list=None, dst=dst, group=group, async_op=async_op)
        if async_op:
            handles.append(h)
    return out, handles, nbytes, (kp_rows, m_rows)


def pipeline_records(p):
    """The track records of a TrackingPipeline batch as a dict of device tensors."""
    return {
        "kps": p.trk_kps, "desc": p.trk_desc, "kp_counts": p.trk_counts,
        "matches": p.matches, "match_counts": p.match_counts,
        "pose": p.Tout, "n_inliers": p.n_inliers,
    }
