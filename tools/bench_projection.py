#!/usr/bin/env python3
"""Measurement for the widened rows (SURVEY 8f rows 1 + 3): device-resident lookup grids + batched
Matcher::searchByProjection(F1, F2). One step = P frame pairs (2000 keys in the current frame, 2000 map points in the
reference frame, the bench's image size); prints one JSON line in the style of bench.py.

    python tools/bench_projection.py [--pairs 256] [--steps 20] [--warmup 3]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from trackingbench_slam_amd import capi, synth  # noqa: E402
from trackingbench_slam_amd.projection import BatchedProjection  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--keys", type=int, default=2000)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()
    torch.cuda.set_device(0)
    ctx = capi.Context(0)  # own stream; torch.cuda.synchronize() below is device-wide
    distinct = [synth.projection_case(100 + i, n1=args.keys, nmp=args.keys, width=1280, height=720) for i in range(8)]
    cases = [distinct[i % len(distinct)] for i in range(args.pairs)]
    bp = BatchedProjection(ctx, cases, torch.device("cuda", 0), nratio=8.0)
    for _ in range(args.warmup):
        bp.run()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bp.run()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    nmatch = int(bp.out_counts.sum().item())
    # candidates the window search touches: what the kernel must read at least (28 B key + 32 B descriptor each) + the
    # query side (36 B map point + 32 B descriptor + 28 B key) + the match records it writes
    name, (calls, ms) = max(prof.items(), key=lambda kv: kv[1][1])
    import oracle
    c = distinct[0]
    t1 = time.perf_counter(); done = 0
    while time.perf_counter() - t1 < args.cpu_seconds:
        c = distinct[done % len(distinct)]
        oracle.search_by_projection(c["Tcw"], c["cam"], c["width"], c["height"], c["k1"], c["d1"], c["taken1"], c["k2"], c["mp"],
                                    c["mp_desc"], c["sf"], 8.0)
        done += 1
    cpu_el = time.perf_counter() - t1
    out = {"metric": "frame pairs/sec (lookup grid + searchByProjection, device resident)", "value": round(args.pairs * args.steps / el, 1),
           "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
           "higher_is_better": True, "dtype": "f32 (projection) + u8 (Hamming)", "data": "synthetic",
           "config": {"workload": "%d pairs/step, %d keys x %d map points per pair, 1280x720, nRatio 8, TH_HIGH 100, rotation check" %
                      (args.pairs, args.keys, args.keys), "matches_per_step": nmatch},
           "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
           "dominant_kernel": name,
           "cpu_baseline": {"value": round(done / cpu_el, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                            "sample": "%d pairs in %.1f s on 1 host thread (grid build + search)" % (done, cpu_el)}}
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
