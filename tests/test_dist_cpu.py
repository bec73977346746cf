"""CPU tests of the multi-process path: world_size-2 gloo gather of track records and frame sharding."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from trackingbench_slam_amd import dist as tbd
    r, w, _ = tbd.init_from_env("gloo")
    F, cap = 3, 5
    lo, hi = tbd.shard_range(2 * F, w, r)
    rec = {
        "kps": torch.arange(F * cap * 7, dtype=torch.float32).reshape(F, cap, 7) + 1000 * r,
        "desc": (torch.arange(F * cap * 32) % 251).to(torch.uint8).reshape(F, cap, 32) + r,
        "kp_counts": torch.tensor([lo, lo + 1, lo + 2], dtype=torch.int32),
        "pose": torch.full((F, 16), float(r)),
    }
    out = tbd.gather_tracks(rec, dst=0)
    if r == 0:
        ok = out["kps"].shape == (w * F, cap, 7) and torch.equal(out["kps"][F:], rec["kps"] + 1000) \
            and torch.equal(out["desc"][F:], rec["desc"] + 1) \
            and out["kp_counts"].tolist() == [0, 1, 2, 3, 4, 5] and out["pose"][F:].eq(1).all().item()
        # steady-state form (what bench.py calls every step): per-rank parts in reused receive buffers
        a = tbd.gather_tracks(rec, dst=0, concat=False)
        first = a["kps"][1].data_ptr()
        rec2 = {k: v + 1 for k, v in rec.items()}
        b = tbd.gather_tracks(rec2, dst=0, concat=False)
        ok = ok and len(b["kps"]) == w and b["kps"][1].data_ptr() == first and torch.equal(b["kps"][1], rec["kps"] + 1001) \
            and torch.equal(b["desc"][0], rec2["desc"])
        # background form: two exchanges in flight on alternating buffer sets, then waited for in order
        o1, h1 = tbd.gather_tracks_async(rec, dst=0, slot=0)
        o2, h2 = tbd.gather_tracks_async(rec2, dst=0, slot=1)
        tbd.wait_tracks(h1); tbd.wait_tracks(h2)
        ok = ok and torch.equal(o1["kps"][1], rec["kps"] + 1000) and torch.equal(o2["kps"][1], rec["kps"] + 1001) \
            and torch.equal(o1["pose"][0], rec["pose"]) and o2["kp_counts"][1].tolist() == [4, 5, 6]
        q.put(bool(ok))
    else:
        tbd.gather_tracks(rec, dst=0, concat=False)
        rec2 = {k: v + 1 for k, v in rec.items()}
        tbd.gather_tracks(rec2, dst=0, concat=False)
        _, h1 = tbd.gather_tracks_async(rec, dst=0, slot=0)
        _, h2 = tbd.gather_tracks_async(rec2, dst=0, slot=1)
        tbd.wait_tracks(h1); tbd.wait_tracks(h2)
        q.put(out is None)
    torch.distributed.destroy_process_group()


def _worker_packed(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from trackingbench_slam_amd import dist as tbd
    r, w, _ = tbd.init_from_env("gloo")
    F, cap = 4, 9
    g = torch.Generator().manual_seed(5 + r)
    kc = torch.tensor([[9, 0, 4, 7], [3, 9, 1, 0]][r], dtype=torch.int32)
    mc = torch.tensor([[2, 0, 0, 1], [0, 5, 0, 0]][r], dtype=torch.int32)
    rec = {"kps": torch.rand((F, cap, 7), generator=g), "desc": torch.randint(0, 255, (F, cap, 32), generator=g, dtype=torch.uint8),
           "kp_counts": kc, "matches": torch.randint(0, 1000, (F, cap, 4), generator=g, dtype=torch.int32), "match_counts": mc,
           "pose": torch.full((F, 16), float(r)), "n_inliers": torch.arange(F, dtype=torch.int32) + 10 * r}
    ok = True
    for slot in (0, 1):      # twice: the receive buffers of a slot are reused, a smaller batch must not see stale rows
        if slot == 1:
            rec = dict(rec); rec["kp_counts"] = torch.clamp(kc - 1, min=0); rec["match_counts"] = torch.clamp(mc - 1, min=0)
        parts, handles, nbytes, rows = tbd.gather_tracks_packed(rec, dst=0, slot=0)
        tbd.wait_tracks(handles)
        tot = int(rec["kp_counts"].sum())
        # the payload is sized by the LARGEST rank's live rows, not by F * cap
        ok = ok and rows[0] == max(int(c.sum()) for c in ([torch.tensor([9, 0, 4, 7]), torch.tensor([3, 9, 1, 0])] if slot == 0 else
                                                           [torch.tensor([8, 0, 3, 6]), torch.tensor([2, 8, 0, 0])]))
        ok = ok and nbytes < sum(v.numel() * v.element_size() for v in rec.values())
        if r == 0:
            for src in range(w):
                cnts = {"kp_counts": parts["kp_counts"][src], "match_counts": parts["match_counts"][src]}
                un = tbd.unpack_records({k: parts[k][src] for k in ("kps", "desc", "matches")}, cnts)
                if src == 0:   # rank 0's own records: every frame's live rows, in order
                    for f in range(F):
                        ok = ok and torch.equal(un["kps"][f], rec["kps"][f, :int(rec["kp_counts"][f])])
                        ok = ok and torch.equal(un["desc"][f], rec["desc"][f, :int(rec["kp_counts"][f])])
                        ok = ok and torch.equal(un["matches"][f], rec["matches"][f, :int(rec["match_counts"][f])])
                ok = ok and bool(parts["pose"][src].eq(float(src)).all()) and parts["n_inliers"][src].tolist() == [10 * src + i for i in range(F)]
                ok = ok and sum(len(x) for x in un["kps"]) == int(cnts["kp_counts"].sum())
            # rank 1's rows, regenerated here from its seed
            g1 = torch.Generator().manual_seed(6)
            k1 = torch.rand((F, cap, 7), generator=g1)
            c1 = parts["kp_counts"][1].tolist()
            un1 = tbd.unpack_records({"kps": parts["kps"][1]}, {"kp_counts": parts["kp_counts"][1]})
            for f in range(F):
                ok = ok and torch.equal(un1["kps"][f], k1[f, :c1[f]])
    q.put(bool(ok))
    torch.distributed.destroy_process_group()


def test_gather_tracks_packed_gloo_world2():
    """The compacted exchange: live rows only, sized by the largest rank (counts first, then one gather per record)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_packed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert res == [True, True]


def test_gather_tracks_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert res == [True, True]


def test_shard_range_partitions():
    from trackingbench_slam_amd import dist as tbd
    for total in (0, 1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [tbd.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert tbd.gather_tracks({"a": torch.zeros(2)})["a"].shape == (2,)
