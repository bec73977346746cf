#!/usr/bin/env python3
"""bench.py -- frames/sec end-to-end of the tracking hot path on N MI355X (driver contract).

Workload (BASELINE.json configs[2], the one `metric` / north_star is quoted on): 1280x720 stereo frames,
8-level pyramid x 0.8, 2000 ORB keypoints per image (FAST 80/30), searchByBF left<->right (ratio 10,
minTh 30), motion-only pose optimisation per frame, and one 10-keyframe / 5000-point local BA window per
frame.  One "step" = one pass of the path over a resident batch of F stereo frames per GPU; inputs are
synthetic (trackingbench_slam_amd/synth.py) and already in HBM when the timed region starts.

N > 1: launched by torch.distributed.run, one rank per GPU; frames shard across ranks with no data-path
collective (weak scaling); the per-batch track records are gathered to rank 0 over RCCL inside the step.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trackingbench_slam_amd import dist as tbd  # noqa: E402
from trackingbench_slam_amd import synth  # noqa: E402
from trackingbench_slam_amd.pipeline import KITTI_K, TrackingPipeline  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured achievable
F64_MFMA_PEAK_TFLOPS = 78.6  # FP64 matrix peak = half the 157.3 TFLOP/s FP32 matrix peak of the guide's chip table


def level_pixels(w, h, nlevels, scale):
    sf = np.float32(1.0)
    px = []
    for i in range(nlevels):
        if i:
            sf = np.float32(sf * np.float32(scale))
        px.append((w if i == 0 else int(np.float32(w) * sf)) * (h if i == 0 else int(np.float32(h) * sf)))
    return px


def algorithmic_bytes(kernel, w, h, nlevels, scale, target, nimg, npairs):
    """SURVEY.md 8(d) per-unit algorithmic bytes x units of one step (see DESIGN.md 'Roofline accounting')."""
    px = level_pixels(w, h, nlevels, scale)
    spx = sum(px)
    per_img = {
        "k_resize": sum(px[l - 1] + px[l] for l in range(1, nlevels)),          # pyramid row
        "k_fast_cells": spx,                                                    # FAST detect row: 1 read / px
        "k_describe": 2 * spx + target * 2 * 961 + target * 60,                 # blur + orient/describe rows (fused)
        "k_octree": 0,                                                          # not HBM-bound (no 8d row)
    }
    if kernel in per_img:
        return per_img[kernel] * nimg
    if kernel == "k_bf_nn":
        return 2 * target * 32 * npairs                                         # (N1+N2)*32 B per pair
    return 0


def schur_roofs(ba_pts, nwin, ba_kf, free_edges, nfixed=2):
    """k_ba_schur, ALGORITHMIC work of one launch (one LM trial of `nwin` windows), DESIGN.md 'Roofline accounting':
    flops = lower triangle of the np x np Schur block plus the reduced right-hand side, K = 3 densified columns per
    point (the kernel executes more: 16x16 tiles pad the triangle, and every Hpl block is rebuilt from its
    observation instead of being fetched); bytes = one 16 B record per free-keyframe edge, one 96 B record per
    point, the per-workgroup partial blocks written."""
    np_ = 6 * (ba_kf - nfixed)
    flops = 2.0 * (np_ * (np_ + 1) / 2 + np_) * 3 * ba_pts * nwin
    R = (np_ + 15) // 16
    nchunks = (ba_pts + 3) // 4
    G = min(max(1024 // max(nwin, 1), 1), max((nchunks + 3) // 4, 1))     # ba_dims() in k_ba.hip
    out = G * (R * (R + 1) // 2 * 256 + np_) * 8
    nbytes = free_edges * 16 + nwin * (ba_pts * 96 + out)
    return flops, nbytes


def pmc_traffic(kernel, units, geometry=None):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/, collected with 64
    images-pairs / BA windows per launch, the guide's gfx950 FETCH_SIZE correction applied), scaled to this run's
    units per launch; None if absent. The passes are per workload: 1280x720 / 10-KF windows by default, and one
    file per other measured geometry (`*_hbm_traffic_pmc_<W>x<H>.json`)."""
    path = os.path.join(ROOT, "profiles", "r01_hbm_traffic_pmc.json")
    if geometry is not None and tuple(geometry) != (1280, 720):
        import glob
        hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic_pmc_%dx%d.json" % tuple(geometry))))
        if not hits:
            return None
        path = hits[-1]
    try:
        with open(path) as f:
            rec = json.load(f)["kernels"]
        key = kernel if kernel in rec else {"k_resize": "k_resize_lds"}.get(kernel, kernel)
        return int(rec[key]["hbm_bytes_per_launch"] * units / 64.0)
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(args, seconds=20.0):
    """CPU restatement of the reference path (oracle/, kind "port", 1 thread) on a bounded sample of the same
    workload: whole stereo frames end to end until ~`seconds` of CPU time."""
    import oracle
    K = KITTI_K
    inputs = []
    for i in range(4):  # input generation is not part of the path: prepared before the clock starts
        L, R = synth.frame(i, args.width, args.height, stereo=True)
        ba = None if args.no_ba else synth.ba_problem(i, args.ba_kf, args.ba_pts, K)
        inputs.append((L, R, synth.pose_problem(i, args.target + 100, K), ba))
    done, t0 = 0, time.perf_counter()
    while True:
        L, R, (_, Ti, obs), ba = inputs[done % len(inputs)]
        lvL, sf = oracle.pyramid(L, args.levels, args.scale)
        lvR, _ = oracle.pyramid(R, args.levels, args.scale)
        k1, d1, _ = oracle.orb_extract(lvL, sf, args.target, args.init_th, args.min_th)
        k2, d2, _ = oracle.orb_extract(lvR, sf, args.target, args.init_th, args.min_th)
        m = oracle.search_by_bf(d1, d2, 10.0, 30.0)
        oracle.pose_opt(K, Ti, obs[:max(len(m), 0)])
        if ba is not None:
            Pt, Pi, Xt, Xi, bo = ba
            oracle.local_ba(K, Pi, 2, Xi, bo, args.ba_iters)
        done += 1
        el = time.perf_counter() - t0
        if el >= seconds or done >= 64:
            break
    return {"value": done / el, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d stereo frames end-to-end (synthetic %dx%d, same stages incl. local BA: %s) in %.1f s on 1 host "
                      "thread" % (done, args.width, args.height, "no" if args.no_ba else "yes", el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=512, help="stereo frames per GPU per step")
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--scale", type=float, default=0.8)
    ap.add_argument("--target", type=int, default=2000)
    ap.add_argument("--init-th", type=float, default=80.0)
    ap.add_argument("--min-th", type=float, default=30.0)
    ap.add_argument("--ba-kf", type=int, default=10)
    ap.add_argument("--ba-pts", type=int, default=5000)
    ap.add_argument("--ba-iters", type=int, default=10)
    ap.add_argument("--no-ba", action="store_true", help="diagnostic only: drop the local-BA stage")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--ba-split", type=int, default=0,
                    help="partitions of the BA windows, one stream + host thread each (0 = by batch size: 1 up to 64 frames, else 3)")
    ap.add_argument("--distinct", type=int, default=8, help="distinct synthetic stereo pairs tiled over the batch")
    args = ap.parse_args()

    rank, world, local = tbd.init_from_env("nccl")
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    dev = local if world > 1 else 0
    torch.cuda.set_device(dev)

    pipe = TrackingPipeline(args.width, args.height, args.levels, args.scale, args.target, args.init_th, args.min_th,
                            frames=args.frames, device=dev, with_ba=not args.no_ba, ba_kf=args.ba_kf, ba_pts=args.ba_pts,
                            ba_iters=args.ba_iters, seed=rank, ba_split=args.ba_split or (1 if args.frames <= 64 else 3))
    pipe.set_synthetic(distinct=args.distinct, first=rank * args.frames)

    state = {"sync_only": False}
    pending = []  # exchange steps in flight: the pipeline alternates between two record sets, so at most two

    def one_step():
        if len(pending) == 2:                     # the set this batch writes was sent two batches ago
            tbd.wait_tracks(pending.pop(0))
        pipe.step()
        if world > 1:
            if not state["sync_only"]:
                try:
                    _, handles = tbd.gather_tracks_async(tbd.pipeline_records(pipe), dst=0, slot=pipe._cur)
                    pending.append(handles)
                    return
                except (RuntimeError, TypeError, ValueError) as e:  # a backend without background gathers
                    print("warning: background track gather unavailable (%s); gathering in line" % e, file=sys.stderr)
                    state["sync_only"] = True
            tbd.gather_tracks(tbd.pipeline_records(pipe), dst=0, concat=False)

    def fence():
        while pending:
            tbd.wait_tracks(pending.pop(0))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    pipe.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    el = time.perf_counter() - t0
    prof = pipe.profile_report()
    pipe.profile_enable(False)
    t = torch.tensor([el], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())

    # after the timed region: each chain alone on an otherwise idle GPU. During the timed region the extractor chain and
    # the BA partitions share the chip, which stretches the HIP-event duration of every kernel by the time it waits for
    # CU slots (short kernels up to 10x): sums of those durations do not say which kernel needs the most GPU time.
    iso = {}          # kernel -> isolated ms per launch
    iso_step = {}     # kernel -> isolated ms per step (all partitions)
    if rank == 0:
        pipe.ctx.profile_report()
        pipe.ctx.profile_enable(True)
        pipe.extract_chain()
        torch.cuda.synchronize()
        for k, (c, ms) in pipe.ctx.profile_report().items():
            iso[k] = ms / max(c, 1)
            iso_step[k] = ms
        pipe.ctx.profile_enable(False)
        if pipe.bas:
            ba0, st0, cx0 = pipe.bas[0]
            cx0.profile_report()
            cx0.profile_enable(True)
            with torch.cuda.stream(st0):
                ba0.run()
            torch.cuda.synchronize()
            scale = sum(b.W for b, _, _ in pipe.bas) / float(ba0.W)
            for k, (c, ms) in cx0.profile_report().items():
                iso[k] = ms / max(c, 1)
                iso_step[k] = ms * scale
            cx0.profile_enable(False)

    if rank == 0:
        frames_total = args.frames * world * args.steps
        nimg, npairs = 2 * args.frames, args.frames
        # dominant kernel of the timed region (HIP events on the kernels' stream, tb_profile_*)
        # dominant kernel = the one that needs the most GPU time per step when its chain runs alone
        name = max(iso_step, key=iso_step.get) if iso_step else max(prof, key=lambda k: prof[k][1])
        calls, tot_ms = prof.get(name, (0, 0.0))
        abytes = algorithmic_bytes(name, args.width, args.height, args.levels, args.scale, args.target, nimg, npairs)
        launches_per_step = max(calls // max(args.steps, 1), 1)
        avg_ms_per_step = tot_ms / max(args.steps, 1)
        kern_ms = {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())}
        if name == "k_ba_schur":
            # one launch = one LM trial of one partition of the windows; both roofs are reported, `bound` is the
            # nearer one (the kernel rebuilds every Hpl block from its 16-byte record AND multiplies it on the FP64 matrix cores)
            nwin = pipe.bas[0][0].W
            free_edges = sum(b.free_edges for b, _, _ in pipe.bas) / max(len(pipe.bas), 1)
            fl, nb = schur_roofs(args.ba_pts, nwin, args.ba_kf, free_edges)
            launch_s = tot_ms / max(calls, 1) / 1e3
            tf, gbs = fl / 1e12 / launch_s, nb / 1e9 / launch_s
            common = {"kernel": name, "traffic": pmc_traffic(name, nwin, (args.width, args.height)), "launches_per_step": launches_per_step,
                      "avg_launch_ms": round(1e3 * launch_s, 5), "algorithmic_bytes_per_launch": int(nb),
                      "algorithmic_flops_per_launch": fl, "windows_per_launch": nwin,
                      "hbm_frac": round(gbs / HBM_PEAK_GBS, 5), "mfma_f64_frac": round(tf / F64_MFMA_PEAK_TFLOPS, 5),
                      "kernels_ms_per_step": kern_ms, "isolated_kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(iso_step.items())}}
            if iso.get(name):
                il = iso[name] / 1e3
                common["isolated"] = {"note": "same launch with its chain alone on the GPU (after the timed region)",
                                      "avg_launch_ms": round(iso[name], 5), "hbm_frac": round(nb / 1e9 / il / HBM_PEAK_GBS, 5),
                                      "mfma_f64_frac": round(fl / 1e12 / il / F64_MFMA_PEAK_TFLOPS, 5)}
            if gbs / HBM_PEAK_GBS >= tf / F64_MFMA_PEAK_TFLOPS:
                roofline = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(gbs / HBM_PEAK_GBS, 5), **common}
            else:
                roofline = {"bound": "mfma", "achieved": round(tf, 3), "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / F64_MFMA_PEAK_TFLOPS, 5), **common}
        else:
            if name in ("k_ba_schur_pairs", "k_ba_solve_big") and pipe.bas:
                # large windows (more than 10 free keyframes), one launch = one LM trial of one partition: bytes per step =
                # bytes per launch x launches. A pair item reads its 8-byte index pair, two 16-byte edge records and the
                # 96-byte point record; the lower triangle of the reduced system [np + 1][np] is written once by the
                # Schur kernel and read + written once by the solve.
                np_ = 6 * (args.ba_kf - 2)
                nwin_all = sum(b.W for b, _, _ in pipe.bas)
                per_trial = {"k_ba_schur_pairs": sum(b.pair_items for b, _, _ in pipe.bas) * (8 + 2 * 16 + 96) + nwin_all * (np_ + 1) * np_ * 4,
                             "k_ba_solve_big": nwin_all * (np_ + 1) * np_ * 8}[name]
                abytes = per_trial * max(calls // max(args.steps, 1), 1) // max(len(pipe.bas), 1)
            achieved = (abytes / 1e9) / (avg_ms_per_step / 1e3) if avg_ms_per_step > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 5),
                        "traffic": pmc_traffic(name, pipe.bas[0][0].W if name.startswith("k_ba_") and pipe.bas else args.frames,
                                               (args.width, args.height)),
                        "launches_per_step": launches_per_step, "avg_launch_ms": round(tot_ms / max(calls, 1), 5),
                        "algorithmic_bytes_per_step": abytes, "kernels_ms_per_step": kern_ms,
                        "isolated_kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(iso_step.items())}}
            if iso_step.get(name):
                roofline["isolated"] = {"note": "same kernel with its chain alone on the GPU (after the timed region)",
                                        "ms_per_step": round(iso_step[name], 4),
                                        "hbm_frac": round(abytes / 1e9 / (iso_step[name] / 1e3) / HBM_PEAK_GBS, 5)}
        out = {
            "metric": "frames/sec end-to-end (extract+match+local-BA)", "value": round(frames_total / el, 2),
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8 (pyramid/FAST/ORB/Hamming) + f64 (pose-opt/BA)", "data": "synthetic",
            "config": {"workload": "%dx%d stereo, %d-level pyramid x%.1f, %d kpts/image (FAST %g/%g), searchByBF L<->R, "
                                   "pose-opt%s" % (args.width, args.height, args.levels, args.scale, args.target,
                                                   args.init_th, args.min_th,
                                                   "" if args.no_ba else ", %d-KF/%d-pt local BA x%d iters" %
                                                   (args.ba_kf, args.ba_pts, args.ba_iters)),
                       "frames_per_gpu_per_step": args.frames, "distinct_synthetic_pairs": min(args.distinct, args.frames),
                       "parallelism": "frames sharded x%d, RCCL gather of tracks" % world},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(out))
    pipe.close()
    if world > 1:
        dist.barrier()  # rank 0 is the last to get here (isolated passes, CPU baseline): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
