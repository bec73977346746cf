#!/usr/bin/env python3
"""bench.py -- frames/sec end-to-end of the tracking hot path on N MI355X (driver contract).

Workload (BASELINE.json configs[2], the one `metric` / north_star is quoted on): 1280x720 stereo frames,
8-level pyramid x 0.8, 2000 ORB keypoints per image (FAST 80/30), searchByBF left<->right (ratio 10,
minTh 30), motion-only pose optimisation per frame, and one 10-keyframe / 5000-point local BA window per
frame.  One "step" = one pass of the path over a resident batch of F stereo frames per GPU; inputs are
synthetic (trackingbench_slam_amd/synth.py) and already in HBM when the timed region starts.  The stages are a
COMPOSITION of independent synthetic inputs: pose-opt consumes a seeded synthetic problem truncated to #matches
rows, local BA consumes seeded synthetic windows -- neither is derived from the extracted tracks (`config` says so).

N > 1: one rank per GPU. Either the driver launches the ranks (torch.distributed.run, WORLD_SIZE set) or
`python bench.py --gpus N` launches them itself: the parent process spawns `torch.distributed.run` as a CHILD before
it makes any GPU call (it never initialises HIP and never re-execs itself) and exits with the child's code. Frames
shard across ranks with no data-path collective (weak scaling); the per-batch track records are gathered to rank 0
over RCCL inside the step.  Rank 0 prints ONE JSON line; a rank count different from --gpus is an error (rc 2).
"""
import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from trackingbench_slam_amd import dist as tbd  # noqa: E402
from trackingbench_slam_amd import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the guide's achievable figure
F64_MFMA_PEAK_TFLOPS = 78.6  # FP64 matrix peak = half the 157.3 TFLOP/s FP32 matrix peak of the guide's chip table
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)  # hard-coded in the reference, LocalBA.cpp:356-359
KITTI_BF = 386.1448                                # fx * baseline of that camera (what AddMapPointsByStereo takes as bf)
EXTRACTOR_KERNELS = ("k_resize", "k_fast_cells", "k_octree", "k_describe")


# ------------------------------------------------------------------ accounting (tests/test_bench_accounting.py)
def level_pixels(w, h, nlevels, scale):
    sf = np.float32(1.0)
    px = []
    for i in range(nlevels):
        if i:
            sf = np.float32(sf * np.float32(scale))
        px.append((w if i == 0 else int(np.float32(w) * sf)) * (h if i == 0 else int(np.float32(h) * sf)))
    return px


def algorithmic_bytes(kernel, w, h, nlevels, scale, target, nimg, npairs):
    """SURVEY.md 8(d) per-unit algorithmic bytes x units of one step."""
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


def extractor_bytes_per_image(w, h, nlevels, scale, target):
    """SURVEY 8(d) 'extractor total' row: pyramid + FAST + blur + orient/describe = 15.43 MB @1280x720, N=2000."""
    return sum(algorithmic_bytes(k, w, h, nlevels, scale, target, 1, 1) for k in EXTRACTOR_KERNELS)


def schur_flops_sparse(sum_k2):
    """SURVEY 8(d) multi-KF BA row: sum over points of k^2 * 216 flop per window and LM trial, k = the point's
    observations by FREE keyframes (the ones that enter the Schur complement); sum_k2 = that sum over the launch."""
    return 216.0 * float(sum_k2)


def schur_roofs(ba_pts, nwin, ba_kf, free_edges, nfixed=2, num_cu=256):
    """k_ba_schur, one launch (one LM trial of `nwin` windows). DENSE flop count: the lower triangle of the np x np reduced
    system plus the rhs for 3 densified columns per point -- what the 16x16x4 MFMA tiles of rounds 1-2 executed, kept for
    comparison (the round-3 kernel's executed count comes from the windows' visibility patterns: BatchedLocalBA
    .mfma_flops_executed). bytes = one 16 B record per free-keyframe edge, one 96 B record per point, the partial systems
    written (one per Schur wavefront, or per workgroup for small batches: ba_dims() in k_ba.hip)."""
    np_ = 6 * (ba_kf - nfixed)
    flops = 2.0 * (np_ * (np_ + 1) / 2 + np_) * 3 * ba_pts * nwin
    slots, cap = 2 * num_cu, min(32, max((ba_pts // 64 + 3) // 4, 1))
    gbase = min(max(slots // max(nwin, 1), 1), cap)
    gextra = min(slots - gbase * nwin, nwin) if gbase < cap and slots > gbase * nwin else 0
    if gbase > 4:      # small batches: one partial system per workgroup
        nparts = gbase * nwin + gextra
    else:              # one per wavefront, the round's wavefronts dealt to the windows one by one
        total = 4 * max(slots, nwin)
        vbase = min(total // max(nwin, 1), 4 * cap)
        nparts = vbase * nwin + (total % max(nwin, 1) if vbase < 4 * cap else 0)
    out = nparts * (np_ * (np_ + 1) // 2 + np_) * 8
    nbytes = free_edges * 16 + nwin * ba_pts * 96 + out
    return flops, nbytes


def _pmc_files(geometry):
    import glob
    pats = ["*_hbm_traffic_pmc.json"] if geometry is None or tuple(geometry) == (1280, 720) else \
        ["*_hbm_traffic_pmc_%dx%d.json" % tuple(geometry)]
    hits = []
    for p in pats:
        hits += glob.glob(os.path.join(ROOT, "profiles", p))
    return sorted(hits)


def pmc_traffic(kernel, units, geometry=None, detail=False):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate passes, the guide's gfx950 correction applied by tools/pmc_traffic.py); None if absent.
    A pass records the units (images / BA windows) one launch processed (`units_per_launch`, 64 in round-1 files); the
    figure is used as measured when that equals this run's `units`, scaled linearly -- and flagged -- otherwise.
    The passes are per workload: 1280x720 / 10-KF windows by default, one file per other measured geometry."""
    for path in reversed(_pmc_files(geometry)):
        try:
            with open(path) as f:
                doc = json.load(f)
            rec = doc["kernels"]
            key = kernel
            aliases = {"k_resize": ("k_resize_lds",), "k_fast_cells": ("k_fast_blocks",), "k_ba_schur": ("k_ba_schur_g", "k_ba_schur_c")}
            for alias in (kernel,) + aliases.get(kernel, ()):
                if alias in rec:     # the profile carries the kernel's function name, tb_profile_* its stage name
                    key = alias
            if key not in rec:
                continue
            # round-1 files carry no units: they were taken at 64 stereo frames (128 images) / 64 windows per launch
            upl = doc.get("units_per_launch") or {"images": 128, "windows": 64, "pairs": 64}
            kind = "windows" if kernel.startswith("k_ba_") else "images" if kernel in EXTRACTOR_KERNELS else "pairs"
            base = float(upl.get(kind) or 64)
            val = int(rec[key]["hbm_bytes_per_launch"] * units / base)
            if detail:
                return {"bytes": val, "file": os.path.relpath(path, ROOT), "measured_at_units": base,
                        "scaled": bool(abs(base - units) > 1e-9)}
            return val
        except (OSError, KeyError, ValueError):
            continue
    return None


# ------------------------------------------------------------------ host-side baselines
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_frame(inp, args):
    import oracle
    L, R, ba = inp
    lvL, sf = oracle.pyramid(L, args.levels, args.scale)
    lvR, _ = oracle.pyramid(R, args.levels, args.scale)
    k1, d1, _ = oracle.orb_extract(lvL, sf, args.target, args.init_th, args.min_th)
    k2, d2, _ = oracle.orb_extract(lvR, sf, args.target, args.init_th, args.min_th)
    m = oracle.search_by_bf(d1, d2, 10.0, 30.0)
    obs = oracle.stereo_tracks_to_obs(k1, k2, m, KITTI_K, KITTI_BF, oracle.scale_factors(args.levels, args.scale)[3])
    oracle.pose_opt(KITTI_K, np.eye(4, dtype=np.float32), obs)
    if ba is not None:
        Pt, Pi, Xt, Xi, bo = ba
        oracle.local_ba(KITTI_K, Pi, 2, Xi, bo, args.ba_iters)
    return 1


def cpu_baseline(args, seconds=20.0):
    """CPU restatement of the reference path (oracle/, kind "port") on a bounded sample of the same workload: whole
    stereo frames end to end. Two legs of ~seconds/2 each: one host thread (the reference's own code is
    single-threaded), then frame-parallel on every host core this process may use (the oracle is C++ behind ctypes,
    which releases the GIL). `value` is the frame-parallel figure, the single-thread one sits beside it."""
    from concurrent.futures import ThreadPoolExecutor
    inputs = []
    for i in range(4):  # input generation is not part of the path: prepared before the clock starts
        L, R = synth.frame(i, args.width, args.height, stereo=True)
        ba = None if args.no_ba else synth.ba_problem(i, args.ba_kf, args.ba_pts, KITTI_K)
        inputs.append((L, R, ba))
    budget = max(seconds / 2.0, 1.0)
    done, t0 = 0, time.perf_counter()
    while True:
        _cpu_frame(inputs[done % len(inputs)], args)
        done += 1
        el1 = time.perf_counter() - t0
        if el1 >= budget or done >= 64:
            break
    single = done / el1
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    nthr = max(1, min(ncores, 16))          # a one-GPU box's CPU share is 16 cores however many the host shows
    deadline = time.perf_counter() + budget

    def worker(t):
        n = 0
        while True:
            _cpu_frame(inputs[(t + n) % len(inputs)], args)
            n += 1
            if time.perf_counter() >= deadline:
                return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=nthr) as pool:
        total = sum(pool.map(worker, range(nthr)))
    eln = time.perf_counter() - t0
    multi = total / eln
    return {"value": multi, "unit": "frames/s", "cores": nthr, "kind": "port", "cpu_model": cpu_model(), "host_cores": ncores,
            "single_thread": {"value": single, "cores": 1,
                              "sample": "%d stereo frames in %.1f s" % (done, el1)},
            "sample": "%d stereo frames end-to-end (synthetic %dx%d, same stages incl. local BA: %s), frame-parallel on %d "
                      "host threads in %.1f s; CPU restatement of the reference path, not the reference itself"
                      % (total, args.width, args.height, "no" if args.no_ba else "yes", nthr, eln)}


def measure_copy_bandwidth(dev, nbytes=1 << 30, reps=10):
    """Device-to-device copy of `nbytes` (read + write = 2 x nbytes of HBM traffic per copy) with the library's own
    16-byte-per-lane kernel (tb_measure_copy_seconds; HIP events on the context's stream): the MEASURED streaming bandwidth
    of this GPU, comparable with the hardware guide's 6.29 TB/s float4 copy, reported beside the 8 TB/s spec peak.
    Round 2 timed torch.Tensor.copy_ here (5.2 TB/s), which flattered every "fraction of the measured copy" by ~20 %."""
    from trackingbench_slam_amd import capi
    dev = torch.device("cuda", dev) if isinstance(dev, int) else dev
    st = torch.cuda.Stream(device=dev)
    ctx = capi.Context(dev.index if dev.index is not None else 0, stream=st.cuda_stream)
    with torch.cuda.stream(st):
        a = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        b = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        a.fill_(1)
    st.synchronize()
    sec = ctx.measure_copy_seconds(a.data_ptr(), b.data_ptr(), nbytes, reps)
    ctx.close()
    del a, b
    return 2.0 * nbytes / 1e9 / sec


# ------------------------------------------------------------------ launcher (python bench.py --gpus N, WORLD_SIZE unset)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(ngpus, argv):
    """Spawn `torch.distributed.run` with `ngpus` ranks of this script as a child process and return its exit code.
    Called before anything in this process touches the GPU (import torch does not initialise HIP); the parent only
    waits. The children see WORLD_SIZE and take the rank path of main()."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------ CPU stand-in for the launcher / exchange test (gloo)
class StubPipeline:
    """No-GPU stand-in with the record interface of TrackingPipeline, for tests/test_bench_launcher.py: step() fills
    the record set of this batch with values that encode (rank, step, frame) so that rank 0 can check what the exchange
    step delivered. Selected by --stub; its JSON line carries "stub": true and is not a measurement."""

    def __init__(self, frames, rank, cap=6):
        self.F, self.rank, self.cap, self.nstep = int(frames), rank, cap, 0
        F = self.F
        self._sets = [dict(matches=torch.zeros((F, cap, 4), dtype=torch.int32), match_counts=torch.zeros(F, dtype=torch.int32),
                           Tout=torch.zeros((F, 16)), n_inliers=torch.zeros(F, dtype=torch.int32),
                           trk_kps=torch.zeros((F, cap, 7)), trk_desc=torch.zeros((F, cap, 32), dtype=torch.uint8),
                           trk_counts=torch.zeros(F, dtype=torch.int32)) for _ in range(2)]
        self._cur = 0
        self.bas = []

    matches = property(lambda s: s._sets[s._cur]["matches"])
    match_counts = property(lambda s: s._sets[s._cur]["match_counts"])
    Tout = property(lambda s: s._sets[s._cur]["Tout"])
    n_inliers = property(lambda s: s._sets[s._cur]["n_inliers"])
    trk_kps = property(lambda s: s._sets[s._cur]["trk_kps"])
    trk_desc = property(lambda s: s._sets[s._cur]["trk_desc"])
    trk_counts = property(lambda s: s._sets[s._cur]["trk_counts"])

    @staticmethod
    def expected(rank, step, F):
        """(pose value, kp_counts) rank `rank` publishes at step `step`."""
        return float(1000 * rank + step), [(rank * F + f + step) % 7 for f in range(F)]

    def stream_ctx(self):
        return contextlib.nullcontext()

    def drain(self):
        pass

    def step(self):
        self._cur ^= 1
        self.nstep += 1
        val, cnt = self.expected(self.rank, self.nstep, self.F)
        s = self._sets[self._cur]
        s["Tout"].fill_(val)
        s["trk_counts"].copy_(torch.tensor(cnt, dtype=torch.int32))
        s["trk_kps"].fill_(val + 0.5)
        s["trk_desc"].fill_((self.rank * 16 + self.nstep) % 256)
        s["matches"].fill_(self.rank * 100 + self.nstep)
        s["match_counts"].fill_(min(self.nstep, self.cap))
        s["n_inliers"].fill_(self.rank)

    def close(self):
        pass


def parse_args(argv=None):
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
    ap.add_argument("--region-events", choices=("auto", "dominant", "on", "off"), default="auto",
                    help="HIP events inside the timed region: 'dominant' times the roofline's kernel only (k_ba_schur, 30 launches "
                         "per step), 'on' every kernel (~400 launches per step: two event records each cost ~2 %% of the step), "
                         "'off' none. auto: dominant above 64 frames per GPU; below, off (the event records are a third of a "
                         "~5 us kernel and keep the local-BA call from replaying its HIP graph). The per-kernel tables "
                         "come from the isolated passes right after the region either way")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--ba-split", type=int, default=0,
                    help="partitions of the BA windows, one stream + host thread each (0 = by batch size: 1 up to 64 frames, else 3)")
    ap.add_argument("--ba-lag", action="store_true",
                    help="let the BA windows of batch s run beside the extraction of batch s + 1 instead of joining the partitions at "
                         "the end of every step (measured: 20.42 against 20.44 ms per step -- the step is bound by the chip's total "
                         "work, not by the tail of a partition's chain; the timed region ends with everything complete either way)")
    ap.add_argument("--distinct", type=int, default=64, help="distinct synthetic stereo pairs tiled over the batch")
    ap.add_argument("--ba-distinct", type=int, default=32, help="distinct synthetic BA windows tiled over the batch")
    ap.add_argument("--exchange", choices=("packed", "padded"), default="packed",
                    help="the track gather at the end of a batch (N > 1): packed = live rows only, sized by the largest rank of "
                         "the batch (counts first, then one gather per record); padded = whole capacity-sized record tensors")
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)   # gloo: CPU test of the launcher / exchange
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)  # StubPipeline, no GPU: not a measurement
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        print("error: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: start the ranks ourselves, as a child, before any GPU call in this process
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else argv)

    rank, world, local = tbd.init_from_env(args.backend)
    if world != args.gpus:
        print("error: --gpus %d but %d rank(s) joined (WORLD_SIZE)" % (args.gpus, world), file=sys.stderr)
        if world > 1:
            dist.destroy_process_group()
        return 2
    gpu = not args.stub
    dev = local if world > 1 else 0

    if gpu:
        from trackingbench_slam_amd.pipeline import TrackingPipeline
        torch.cuda.set_device(dev)
        pipe = TrackingPipeline(args.width, args.height, args.levels, args.scale, args.target, args.init_th, args.min_th,
                                frames=args.frames, device=dev, with_ba=not args.no_ba, ba_kf=args.ba_kf, ba_pts=args.ba_pts,
                                ba_iters=args.ba_iters, seed=rank, ba_split=args.ba_split or (1 if args.frames <= 64 else 3),
                                ba_distinct=args.ba_distinct, ba_lag=args.ba_lag)
        pipe.set_synthetic(distinct=args.distinct, first=rank * args.frames)
    else:
        pipe = StubPipeline(args.frames, rank)

    state = {"sync_only": False, "xbytes": 0, "xsteps": 0, "xrows": (0, 0)}
    pending = []  # exchange steps in flight: the pipeline alternates between two record sets, so at most two
    received = []  # stub mode: (step, parts) rank 0 got

    def one_step():
        # every torch-side operation of the exchange is issued on the chain's own stream: the collective then starts
        # behind the kernels that write the records, and wait_tracks() orders the NEXT writer of a record set behind it
        with pipe.stream_ctx():
            if len(pending) == 2:                 # the set this batch writes was sent two batches ago
                tbd.wait_tracks(pending.pop(0))
        pipe.step()
        if world > 1:
            with pipe.stream_ctx():
                if not state["sync_only"]:
                    try:
                        if args.exchange == "packed":
                            parts, handles, nb, rows = tbd.gather_tracks_packed(tbd.pipeline_records(pipe), dst=0, slot=pipe._cur,
                                                                                packer=getattr(pipe, "pack_rows", None))
                            state["xbytes"] += nb; state["xsteps"] += 1; state["xrows"] = rows
                        else:
                            recs = tbd.pipeline_records(pipe)
                            parts, handles = tbd.gather_tracks_async(recs, dst=0, slot=pipe._cur)
                            state["xbytes"] += sum(v.numel() * v.element_size() for v in recs.values()); state["xsteps"] += 1
                        pending.append(handles)
                        if not gpu and rank == 0:
                            received.append((pipe.nstep, parts))
                        return
                    except (RuntimeError, TypeError, ValueError) as e:  # a backend without background gathers
                        print("warning: background track gather unavailable (%s); gathering in line" % e, file=sys.stderr)
                        state["sync_only"] = True
                parts = tbd.gather_tracks(tbd.pipeline_records(pipe), dst=0, concat=False)
                if not gpu and rank == 0:
                    received.append((pipe.nstep, parts))

    def fence():
        pipe.drain()                              # lagged BA partitions: their host drivers have queued everything
        with pipe.stream_ctx():
            while pending:
                tbd.wait_tracks(pending.pop(0))
        if world > 1:
            dist.barrier()
        if gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    fence()
    mode = args.region_events if args.region_events != "auto" else ("dominant" if args.frames > 64 else "off")
    if mode == "dominant" and args.no_ba:
        mode = "on"          # without the BA stage the roofline's kernel is an extractor kernel: time them all
    region_events = gpu and mode != "off"
    if region_events:
        # the kernel that needs the most GPU time per step on every configuration measured so far: the Schur complement
        # (windows of more than 10 free keyframes: its block-pair form); report() falls back to the isolated pass if not
        pipe.profile_enable(True, only=("k_ba_schur_pairs" if args.ba_kf - 2 > 10 else "k_ba_schur") if mode == "dominant" else None)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    fence()
    el = time.perf_counter() - t0
    prof = {}
    if region_events:
        prof = pipe.profile_report()
        pipe.profile_enable(False)
    t = torch.tensor([el], dtype=torch.float64, device="cuda" if gpu else "cpu")
    joined = torch.ones(1, dtype=torch.int64, device="cuda" if gpu else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(joined, op=dist.ReduceOp.SUM)
    el = float(t.item())
    n_joined = int(joined.item())

    if not gpu:
        rc = 0
        if rank == 0:
            ok = True
            if world > 1:
                # the last two exchanges are still intact in their receive-buffer sets: check what every rank sent
                for step_no, parts in received[-2:]:
                    for r in range(world):
                        val, cnt = StubPipeline.expected(r, step_no, args.frames)
                        ok = ok and bool(parts["pose"][r].eq(val).all()) and parts["kp_counts"][r].tolist() == cnt \
                            and bool(parts["kps"][r].eq(val + 0.5).all()) and bool(parts["matches"][r].eq(r * 100 + step_no).all())
            print(json.dumps({"stub": True, "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
                              "frames_per_rank": args.frames, "gather_check": "ok" if ok else "MISMATCH",
                              "exchanges": len(received), "exchange": args.exchange,
                              "bytes_per_rank_per_step": state["xbytes"] // max(state["xsteps"], 1)}))
            rc = 0 if ok and n_joined == args.gpus else 3
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return rc

    out = None
    if rank == 0:
        out = report(args, pipe, prof, el, world, n_joined, dev)
        if world > 1:
            per_rank = state["xbytes"] // max(state["xsteps"], 1)
            out["config"]["exchange"] = {"mode": args.exchange if not state["sync_only"] else "in line (padded)",
                                         "bytes_per_rank_per_step": per_rank, "bytes_into_rank0_per_step": per_rank * (world - 1),
                                         "rows_per_rank_last_step": {"keypoints": state["xrows"][0], "matches": state["xrows"][1]}}
        print(json.dumps(out))
        sys.stdout.flush()
    pipe.close()
    if world > 1:
        dist.barrier()  # rank 0 is the last to get here (isolated passes, CPU baseline): leave together
        dist.destroy_process_group()
    return 0 if n_joined == args.gpus else 2


def isolated_passes(pipe, reps=3):
    """After the timed region: each chain ALONE on an otherwise idle GPU, HIP events around every launch. During the timed
    region the extractor chain and the BA partitions share the chip, which stretches the event duration of every kernel by
    the time it waits for CU slots (short kernels up to 10x), so sums of those durations do not say which kernel needs
    the most GPU time. Returns (per-launch ms, per-step ms) by kernel."""
    iso, iso_step = {}, {}
    pipe.ctx.profile_report()
    pipe.ctx.profile_enable(True)
    for _ in range(reps):
        pipe.extract_chain()
    torch.cuda.synchronize()
    for k, (c, ms) in pipe.ctx.profile_report().items():
        iso[k] = ms / max(c, 1)
        iso_step[k] = ms / reps
    pipe.ctx.profile_enable(False)
    if pipe.bas:
        ba0, st0, cx0 = pipe.bas[0]
        cx0.set_concurrency(1)          # this pass has the GPU to itself: its grids take the whole chip, as a lone caller's would
        with torch.cuda.stream(st0):
            ba0.run()                   # untimed: the first call at the new grid shape
        torch.cuda.synchronize()
        cx0.profile_report()
        cx0.profile_enable(True)
        with torch.cuda.stream(st0):
            for _ in range(reps):
                ba0.run()
        torch.cuda.synchronize()
        cx0.set_concurrency(len(pipe.bas))
        scale = sum(b.W for b, _, _ in pipe.bas) / float(ba0.W)
        for k, (c, ms) in cx0.profile_report().items():
            iso[k] = ms / max(c, 1)
            iso_step[k] = ms * scale / reps
        cx0.profile_enable(False)
    return iso, iso_step


def report(args, pipe, prof, el, world, n_joined, dev):
    nimg, npairs = 2 * args.frames, args.frames
    geometry = (args.width, args.height)
    copy_gbs = measure_copy_bandwidth(dev)
    iso, iso_step = isolated_passes(pipe)
    ms_per_step = 1e3 * el / args.steps
    kern_ms = {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())}

    # ---- the north-star figure: the extractor against the HBM roof (SURVEY 8d 'extractor total' row)
    ext_bytes = extractor_bytes_per_image(args.width, args.height, args.levels, args.scale, args.target) * nimg
    per_kernel = {}
    for k in EXTRACTOR_KERNELS:
        ab = algorithmic_bytes(k, args.width, args.height, args.levels, args.scale, args.target, nimg, npairs)
        launches = max(prof.get(k, (0, 0.0))[0] // max(args.steps, 1), 1)
        tr = pmc_traffic(k, nimg, geometry, detail=True)
        row = {"isolated_ms_per_step": round(iso_step.get(k, 0.0), 4), "launches_per_step": launches,
               "algorithmic_bytes_per_step": int(ab)}
        if ab and iso_step.get(k):
            gbs = ab / 1e9 / (iso_step[k] / 1e3)
            row.update({"achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)})
        if tr is not None:
            row.update({"traffic_bytes_per_step": tr["bytes"] * launches, "traffic_file": tr["file"], "traffic_scaled": tr["scaled"]})
            if ab:
                row["traffic_over_algorithmic"] = round(tr["bytes"] * launches / ab, 3)
        per_kernel[k] = row
    chain_ms = sum(iso_step.get(k, 0.0) for k in EXTRACTOR_KERNELS)
    ext = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "images_per_step": nimg,
           "algorithmic_bytes_per_image": int(ext_bytes // nimg), "algorithmic_bytes_per_step": int(ext_bytes),
           "isolated_chain_ms_per_step": round(chain_ms, 4),
           "achieved": round(ext_bytes / 1e9 / (chain_ms / 1e3), 1) if chain_ms else None,
           "frac": round(ext_bytes / 1e9 / (chain_ms / 1e3) / HBM_PEAK_GBS, 4) if chain_ms else None,
           "in_step_share": {"note": "same bytes over the whole timed step (BA, matcher and pose-opt overlap with it)",
                             "achieved": round(ext_bytes / 1e9 / (ms_per_step / 1e3), 1),
                             "frac": round(ext_bytes / 1e9 / (ms_per_step / 1e3) / HBM_PEAK_GBS, 4)},
           "frac_of_measured_copy_bw": round(ext_bytes / 1e9 / (chain_ms / 1e3) / copy_gbs, 4) if chain_ms else None,
           "kernels": per_kernel}

    # ---- the dominant kernel = the one that needs the most GPU time per step when its chain runs alone
    name = max(iso_step, key=iso_step.get) if iso_step else max(prof, key=lambda k: prof[k][1])
    calls, tot_ms = prof.get(name, (0, 0.0))
    launches_per_step = max(calls // max(args.steps, 1), 1)
    in_region_launch_ms = tot_ms / max(calls, 1)
    events_note = ("HIP events on the kernel's stream over the timed region; other streams share the GPU then, "
                   "so it includes waiting for CU slots -- `isolated` is the same launch with its chain alone")
    if name not in prof:   # --region-events off (small batches), or another kernel than the one timed in the region dominates:
        # the isolated pass right after the region is the only timing of this kernel
        in_region_launch_ms = iso.get(name, 0.0)
        launches_per_step = max(int(round(iso_step.get(name, 0.0) / in_region_launch_ms)), 1) if in_region_launch_ms else 1
        events_note = ("this kernel was not timed inside the region (see --region-events); this is the isolated "
                       "pass taken right after it, HIP events on the kernel's stream, its chain alone on the GPU")
    common = {"kernel": name, "launches_per_step": launches_per_step,
              "avg_launch_ms": round(in_region_launch_ms, 5),
              "avg_launch_ms_note": events_note,
              "hbm_copy_measured_GBs": round(copy_gbs, 1), "extractor": ext,
              "kernels_ms_per_step_in_region": kern_ms,
              "kernels_ms_per_step_in_region_note": "HIP-event durations of the kernels timed inside the region (--region-events: "
                                                    "'dominant' = the roofline's kernel only, 'on' = all); waiting for CU slots "
                                                    "beside the other streams included",
              "isolated_kernels_ms_per_step": {k: round(v, 4) for k, v in sorted(iso_step.items())}}

    schur = None
    if pipe.bas and "k_ba_schur" in iso:
        nwin = pipe.bas[0][0].W
        free_edges = pipe.bas[0][0].free_edges
        fl_dense, nb = schur_roofs(args.ba_pts, nwin, args.ba_kf, free_edges)
        fl = schur_flops_sparse(pipe.bas[0][0].sum_k2_free)
        c_s, ms_s = prof.get("k_ba_schur", (0, 0.0))
        il = iso["k_ba_schur"] / 1e3
        ls = (ms_s / max(c_s, 1)) / 1e3 or (il if "k_ba_schur" not in prof else None)   # not timed in the region: the isolated launch stands in
        fl_exec = float(pipe.bas[0][0].mfma_flops_executed)
        tr = pmc_traffic("k_ba_schur", nwin, geometry, detail=True)
        schur = {"kernel": "k_ba_schur", "bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "windows_per_launch": nwin,
                 "algorithmic_flops_per_launch": fl, "flops_rule": "SURVEY 8(d): sum over points of k_free^2 x 216 per window and trial",
                 "algorithmic_bytes_per_launch": int(nb),
                 "avg_launch_ms": round(1e3 * ls, 5) if ls else None,
                 "achieved": round(fl / 1e12 / ls, 3) if ls else None,
                 "frac": round(fl / 1e12 / ls / F64_MFMA_PEAK_TFLOPS, 5) if ls else None,
                 "hbm_frac": round(nb / 1e9 / ls / HBM_PEAK_GBS, 5) if ls else None,
                 "isolated": {"avg_launch_ms": round(iso["k_ba_schur"], 5),
                              "achieved": round(fl / 1e12 / il, 3), "frac": round(fl / 1e12 / il / F64_MFMA_PEAK_TFLOPS, 5),
                              "hbm_frac": round(nb / 1e9 / il / HBM_PEAK_GBS, 5)},
                 "executed": {"note": "flops the kernel's v_mfma_f64_4x4x4 instructions execute (groups of one visibility pattern: "
                                      "k-steps x instructions of the pattern x 512; padding rows / columns and mirrored blocks "
                                      "included) -- rounds 1-2 executed the dense count below, 2.97 x the algorithmic one",
                              "flops_per_launch": fl_exec, "over_algorithmic": round(fl_exec / fl, 3) if fl else None,
                              "frac": round(fl_exec / 1e12 / ls / F64_MFMA_PEAK_TFLOPS, 5) if ls else None,
                              "isolated_frac": round(fl_exec / 1e12 / il / F64_MFMA_PEAK_TFLOPS, 5)},
                 "dense": {"note": "dense lower triangle + rhs per point (what the 16x16x4 tiles of rounds 1-2 computed, zeros included)",
                           "flops_per_launch": fl_dense},
                 "traffic": tr["bytes"] if tr else None, "traffic_file": tr["file"] if tr else None,
                 "traffic_scaled": tr["scaled"] if tr else None}
        common["ba_schur"] = schur

    if name == "k_ba_schur" and schur is not None:
        roofline = {"bound": "mfma", "achieved": schur["achieved"], "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": schur["frac"], "traffic": schur["traffic"], **common}
    else:
        abytes = algorithmic_bytes(name, args.width, args.height, args.levels, args.scale, args.target, nimg, npairs)
        if name in ("k_ba_schur_pairs", "k_ba_solve_big") and pipe.bas:
            # large windows (more than 10 free keyframes), one launch = one LM trial of one partition: bytes per step =
            # bytes per launch x launches. A pair item reads its 8-byte index pair, two 16-byte edge records and the
            # 96-byte point record; the lower triangle of the reduced system [np + 1][np] is written once by the
            # Schur kernel and read + written once by the solve.
            np_ = 6 * (args.ba_kf - 2)
            nwin_all = sum(b.W for b, _, _ in pipe.bas)
            per_trial = {"k_ba_schur_pairs": sum(b.pair_items for b, _, _ in pipe.bas) * (8 + 2 * 16 + 96) + nwin_all * (np_ + 1) * np_ * 4,
                         "k_ba_solve_big": nwin_all * (np_ + 1) * np_ * 8}[name]
            abytes = per_trial * launches_per_step // max(len(pipe.bas), 1)
        if name == "k_ba_solve" and pipe.bas:
            # small batches: the one-wavefront-per-window solve is the longest kernel (a latency chain, not a bandwidth one).
            # Its algorithmic bytes: the partial reduced systems the Schur wavefronts / workgroups wrote, read once.
            nwin = pipe.bas[0][0].W
            _, nb_all = schur_roofs(args.ba_pts, nwin, args.ba_kf, pipe.bas[0][0].free_edges)
            abytes = (nb_all - pipe.bas[0][0].free_edges * 16 - nwin * args.ba_pts * 96) * launches_per_step
        per_launch = abytes / launches_per_step
        achieved = per_launch / 1e9 / (in_region_launch_ms / 1e3) if in_region_launch_ms > 0 else 0.0
        units = pipe.bas[0][0].W if name.startswith("k_ba_") and pipe.bas else nimg if name in EXTRACTOR_KERNELS else npairs
        tr = pmc_traffic(name, units, geometry, detail=True)
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": tr["bytes"] if tr else None,
                    "traffic_file": tr["file"] if tr else None, "traffic_scaled": tr["scaled"] if tr else None,
                    "algorithmic_bytes_per_launch": int(per_launch), **common}
        if name == "k_ba_solve":
            roofline["note"] = ("small batch: the longest kernel is a latency chain (assembly of the partial systems, then a 48-column "
                                "factorisation on ONE wavefront per window), not a bandwidth-bound one; the HBM fraction says how far from "
                                "any roof this shape is. `ba_schur` carries the Schur kernel's figures")
        if iso.get(name):
            roofline["isolated"] = {"avg_launch_ms": round(iso[name], 5),
                                    "achieved": round(per_launch / 1e9 / (iso[name] / 1e3), 2),
                                    "frac": round(per_launch / 1e9 / (iso[name] / 1e3) / HBM_PEAK_GBS, 5)}

    frames_total = args.frames * world * args.steps
    out = {
        "metric": "frames/sec end-to-end (extract+match+local-BA)", "value": round(frames_total / el, 2),
        "unit": "frames/s", "n_gpus": n_joined, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8 (pyramid/FAST/ORB/Hamming) + f64 (pose-opt/BA)", "data": "synthetic",
        "config": {"workload": "%dx%d stereo, %d-level pyramid x%.1f, %d kpts/image (FAST %g/%g), searchByBF L<->R, "
                               "pose-opt%s" % (args.width, args.height, args.levels, args.scale, args.target,
                                               args.init_th, args.min_th,
                                               "" if args.no_ba else ", %d-KF/%d-pt local BA x%d iters" %
                                               (args.ba_kf, args.ba_pts, args.ba_iters)),
                   "composition": "extract -> searchByBF L<->R -> stereo depth of every matched key (bf / |dx|, LocalBA.cpp:60-64) -> "
                                  "PoseOptimization from the identity on those tracks (map point = left key at its depth, observed "
                                  "at the right key's pixel: it finds the right camera). Local BA runs seeded synthetic 10-keyframe "
                                  "windows -- the reference has no multi-keyframe map to build them from (SURVEY D1)",
                   "frames_per_gpu_per_step": args.frames, "distinct_synthetic_pairs": min(args.distinct, args.frames),
                   "distinct_ba_windows": 0 if args.no_ba else min(args.ba_distinct, args.frames),
                   "ba_partitions": len(pipe.bas),
                   "ba_schedule": ("lagged: the BA windows of batch s run beside the extraction of batch s + 1 (a local-mapping thread one "
                                   "batch behind tracking); every batch's windows are complete inside the timed region"
                                   if getattr(pipe, "ba_lag", False) else "joined at the end of every step"),
                   "parallelism": "frames sharded x%d, RCCL gather of tracks" % world},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
    return out


if __name__ == "__main__":
    sys.exit(main())
