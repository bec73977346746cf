#!/usr/bin/env python3
"""Measurement for the optical-flow matcher (first part): pyramidal LK tracking of the left frame's
keys into the right image, device resident. One step = P stereo pairs (1280x720, 2000 points each, window 21, 4
pyramid levels: the call of matcher.cpp:744); prints one JSON line in the style of bench.py.

    python tools/bench_flow.py [--pairs 64] [--steps 10] [--warmup 2]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from trackingbench_slam_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--points", type=int, default=2000)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--stereo", action="store_true",
                    help="time LocalBA::AddMapPointsByStereo instead (CLAHE + tracker + frame test + RANSAC + depths, batched)")
    args = ap.parse_args()
    import oracle
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0)  # own stream; torch.cuda.synchronize() below is device-wide
    W, H, n, P = args.width, args.height, args.points, args.pairs
    distinct = []
    for i in range(min(4, P)):
        L, R = synth.frame(200 + i, W, H, stereo=True)
        lv, sf = oracle.pyramid(L, 8, 0.8)
        k, _, _ = oracle.orb_extract(lv, sf, n, 80, 30)
        xy = np.stack([k["x"], k["y"]], 1).astype(np.float32)
        pts = np.resize(xy, (n, 2)) if len(xy) else np.zeros((n, 2), np.float32)   # exactly n points (repeats if fewer)
        distinct.append((L, R, pts))
    Ls = torch.from_numpy(np.stack([distinct[i % len(distinct)][0] for i in range(P)])).to(dev)
    Rs = torch.from_numpy(np.stack([distinct[i % len(distinct)][1] for i in range(P)])).to(dev)
    pts = torch.from_numpy(np.stack([distinct[i % len(distinct)][2] for i in range(P)])).to(dev)
    out = torch.zeros_like(pts)
    status = torch.zeros((P, n), dtype=torch.uint8, device=dev)
    err = torch.zeros((P, n), dtype=torch.float32, device=dev)

    depth = torch.zeros((P, n), dtype=torch.float32, device=dev)
    cam = oracle.camera(718.856, 718.856, W / 2, H / 2, W, H)
    import ctypes as C
    camr = np.ascontiguousarray(cam, capi.CAMERA)

    def step():
        if args.stereo:   # the keys of the LEFT (current) image tracked into the equalised RIGHT (stereo) image
            ctx.check(capi.lib().tb_add_map_points_by_stereo_batch_dev(
                ctx._h, P, C.c_void_p(Rs.data_ptr()), C.c_void_p(Ls.data_ptr()), W, H, W, C.c_size_t(W * H), camr.ctypes.data_as(C.c_void_p),
                C.c_void_p(pts.data_ptr()), None, n, C.c_float(386.1448), C.c_void_p(out.data_ptr()), C.c_void_p(status.data_ptr()),
                C.c_void_p(depth.data_ptr())))
        else:
            ctx.optical_flow_pyr_lk_batch_dev(P, Ls.data_ptr(), Rs.data_ptr(), W, H, W, W * H, pts.data_ptr(), 0, n, out.data_ptr(),
                                              status.data_ptr(), err.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile_enable(False)
    # parity spot check of the last step against the oracle (bit for bit)
    L, R, p0 = distinct[0]
    if args.stereo:
        od = oracle.add_map_points_by_stereo(R, L, cam, p0, 386.1448)
        ok = bool(np.array_equal(depth[0].cpu().numpy().view(np.uint32), od.view(np.uint32)))
    else:
        on, os_, oe, _ = oracle.optical_flow_pyr_lk(L, R, p0)
        ok = bool(np.array_equal(out[0].cpu().numpy().view(np.uint32), on.view(np.uint32)) and np.array_equal(status[0].cpu().numpy(), os_))
    t1 = time.perf_counter(); done = 0
    while time.perf_counter() - t1 < args.cpu_seconds:
        L, R, p0 = distinct[done % len(distinct)]
        if args.stereo:
            oracle.add_map_points_by_stereo(R, L, cam, p0, 386.1448)
        else:
            oracle.optical_flow_pyr_lk(L, R, p0)
        done += 1
    cpu_el = time.perf_counter() - t1
    name, (calls, ms) = max(prof.items(), key=lambda kv: kv[1][1])
    res = {"metric": ("stereo pairs/sec (AddMapPointsByStereo: CLAHE + LK + RANSAC F + depths, %d keys per pair, device resident)" if args.stereo
                      else "stereo pairs/sec (pyramidal LK, %d points per pair, device resident)") % n, "value": round(P * args.steps / el, 1),
           "unit": "pairs/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
           "higher_is_better": True, "dtype": "u8 / int32 fixed point (window) + int64 sums + f32 (2x2 solve)", "data": "synthetic",
           "config": {"workload": "%d pairs/step, %dx%d, %d points, window 21, 4 levels, 30 iterations / eps 0.01" % (P, W, H, n),
                      "tracked_fraction": round(float(status.float().mean().item()), 4), "matches_oracle_bit_for_bit": ok},
           "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
           "dominant_kernel": name, "dominant_avg_launch_ms": round(ms / max(calls, 1), 5),
           "cpu_baseline": {"value": round(done / cpu_el, 2), "unit": "pairs/s", "cores": 1, "kind": "port",
                            "sample": "%d pairs in %.1f s on 1 host thread (pyramids + tracking)" % (done, cpu_el)}}
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
