"""Host-side driver of the batched tracking hot path on one GPU.

One `TrackingPipeline.step()` = one pass of the path over a resident batch of F stereo frames:

    pyramid (2F images) -> ORB extract (2F) -> searchByBF left<->right (F pairs) -> stereo depth of the matched keys ->
    motion-only pose optimisation on those tracks (F problems) -> [multi-keyframe local BA, F windows] -> track records

Everything stays in HBM between stages; torch supplies device memory, the stream and (in dist.py) the
RCCL gather.  All compute goes through the C ABI of libtb_hip.so -- there is no torch or CPU fallback.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import capi, synth

KITTI_K = (718.856, 718.856, 607.1928, 185.2157)  # hard-coded in the reference, LocalBA.cpp:356-359
KITTI_BF = 386.1448                                # fx * baseline of the same camera (test/test_vo.cpp:716: AddMapPointsByStereo(..., 386.1448, 718.856))


class TrackingPipeline:
    def __init__(self, width=1280, height=720, nlevels=8, scale=0.8, target=2000, init_th=80.0, min_th=30.0,
                 frames=16, bf_ratio=10.0, bf_min_th=30.0, device=0, with_ba=True, ba_kf=10, ba_pts=5000, ba_iters=10,
                 seed=0, ba_split=3, ba_distinct=4, ba_lag=False):
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.F = int(frames)
        self.width, self.height, self.nlevels, self.scale = width, height, nlevels, scale
        self.target, self.init_th, self.min_th = target, init_th, min_th
        self.bf_ratio, self.bf_min_th = bf_ratio, bf_min_th
        # The extractor -> matcher -> pose-opt chain gets a torch stream of its own and the context runs on THAT handle:
        # torch's default stream has handle 0, which tb_set_stream() reads as "the context's own stream" -- torch-side
        # work (zero_(), the RCCL gather) would then be unordered against the kernels. Everything torch does for this
        # chain is issued under `with torch.cuda.stream(self.main)`.
        self.main = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("TB_MAIN_PRIO", "0")))   # A/B hook
        self.ctx = capi.Context(device, stream=self.main.cuda_stream)
        self.ex = capi.Extractor(self.ctx, width, height, nlevels, scale, 2 * self.F, target)
        self.kps_ptr, self.desc_ptr, self.counts_ptr, self.kp_cap = self.ex.results_dev()
        F, cap = self.F, self.kp_cap
        self.images = None
        self._pack_bufs = {}
        with torch.cuda.stream(self.main):   # every tensor of the chain is allocated, filled and uploaded ON the chain's stream
            self._alloc_chain(F, cap, seed)
        self._init_ba(with_ba, ba_lag, ba_split, ba_kf, ba_pts, ba_iters, ba_distinct, seed, device)

    def _alloc_chain(self, F, cap, seed):
        # Track records (what a batch hands on: keypoints, descriptors, matches, poses) exist twice: step() alternates
        # between the sets, so the exchange step of one batch (dist.gather_tracks_async) can still read its set while
        # the next batch is being computed into the other one.
        self._sets = [dict(matches=torch.zeros((F, cap, 4), dtype=torch.int32, device=self.dev),  # tb_match records
                           match_counts=torch.zeros(F, dtype=torch.int32, device=self.dev),
                           Tout=torch.zeros((F, 16), dtype=torch.float32, device=self.dev),
                           n_inliers=torch.zeros(F, dtype=torch.int32, device=self.dev),
                           trk_kps=torch.zeros((F, cap, 7), dtype=torch.float32, device=self.dev),
                           trk_desc=torch.zeros((F, cap, 32), dtype=torch.uint8, device=self.dev),
                           trk_counts=torch.zeros(F, dtype=torch.int32, device=self.dev)) for _ in range(2)]
        self._cur = 0
        # pose-opt inputs come from the tracks (tb_stereo_tracks_to_obs_batch_dev): per left <-> right match the left key's
        # stereo depth (LocalBA.cpp:60-64) back-projected to a map point, observed at the right key's pixel; the optimisation
        # starts at the identity (as test/test_vo.cpp:305-355 does) and finds the right camera's pose
        self.K = np.ascontiguousarray(KITTI_K, np.float64)
        self.Kf = np.ascontiguousarray(KITTI_K, np.float32)
        self.inv_sigma2 = np.ascontiguousarray(capi.scale_factors(self.nlevels, self.scale)[3], np.float32)
        self.obs_pitch = cap
        self.obs = torch.zeros((F, cap, 6), dtype=torch.float32, device=self.dev)
        self.obs_counts = torch.zeros(F, dtype=torch.int32, device=self.dev)
        self.Tin = torch.eye(4, dtype=torch.float32, device=self.dev).reshape(1, 16).repeat(F, 1).contiguous()
        self.outlier = torch.zeros((F, cap), dtype=torch.uint8, device=self.dev)
        self.pose_stats = torch.zeros((F, 8), dtype=torch.float64, device=self.dev)

    def _init_ba(self, with_ba, ba_lag, ba_split, ba_kf, ba_pts, ba_iters, ba_distinct, seed, device):
        F = self.F
        # Local BA runs on its own HIP stream and context: its LM rounds are a chain of small, latency-bound
        # kernels (64-block solves, one-block decisions) that leave most CUs idle, while the extractor kernels
        # are throughput bound -- the two overlap on the chip instead of queueing behind each other.
        self.with_ba = with_ba
        self.ba_lag = bool(ba_lag) and with_ba
        self._futures = []
        self.bas = []          # (BatchedLocalBA, torch stream, context) per partition of the windows
        self._step_done = None
        self._pool = None
        if with_ba:
            from concurrent.futures import ThreadPoolExecutor
            from .ba import BatchedLocalBA
            nsplit = max(1, min(int(ba_split), F))
            bounds = [F * i // nsplit for i in range(nsplit + 1)]
            for i in range(nsplit):
                # HIGH priority for the BA streams: a partition's chain of small latency-bound launches is the step's critical
                # path (it is still running when the extractor chain has finished); with equal priorities its workgroups queue
                # behind the extractor's half-million (measured, 512 frames: 21.66 -> 21.14 ms per step; TB_BA_PRIO=0 restores)
                st = torch.cuda.Stream(device=self.dev, priority=int(os.environ.get("TB_BA_PRIO", "-1")))
                cx = capi.Context(device, stream=st.cuda_stream)
                # the partitions run side by side: each sizes its grids for its share of the GPU (TB_BA_PEERS: A/B hook)
                cx.set_concurrency(int(os.environ.get("TB_BA_PEERS", nsplit)))
                self.bas.append((BatchedLocalBA(cx, bounds[i + 1] - bounds[i], ba_kf, ba_pts, ba_iters, seed * 16 + i, self.dev,
                                                distinct=max(1, -(-int(ba_distinct) // nsplit)), stream=st), st, cx))
            # each partition's driver blocks on its own stream once per call (LM termination is data dependent):
            # one host thread per partition keeps the partitions' kernel chains in flight together
            self._pool = ThreadPoolExecutor(max_workers=nsplit)

    # the record set the last (or running) step writes
    matches = property(lambda self: self._sets[self._cur]["matches"])
    match_counts = property(lambda self: self._sets[self._cur]["match_counts"])
    Tout = property(lambda self: self._sets[self._cur]["Tout"])
    n_inliers = property(lambda self: self._sets[self._cur]["n_inliers"])
    trk_kps = property(lambda self: self._sets[self._cur]["trk_kps"])
    trk_desc = property(lambda self: self._sets[self._cur]["trk_desc"])
    trk_counts = property(lambda self: self._sets[self._cur]["trk_counts"])

    def pack_rows(self, name, t, counts):
        """dist.pack_records' packer: tb_pack_rows_dev on the chain's stream; one reused destination per record and set."""
        key = (name, self._cur)
        buf = self._pack_bufs.get(key)
        if buf is None or buf.shape != (t.shape[0] * t.shape[1],) + tuple(t.shape[2:]):
            with torch.cuda.stream(self.main):
                buf = torch.empty((t.shape[0] * t.shape[1],) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            self._pack_bufs[key] = buf
        row_bytes = t.element_size() * int(np.prod(t.shape[2:]))
        self.ctx.check(capi.lib().tb_pack_rows_dev(self.ctx._h, C.c_void_p(t.data_ptr()), row_bytes, t.shape[1], C.c_void_p(counts.data_ptr()),
                                                   t.shape[0], C.c_void_p(buf.data_ptr()), None))
        return buf

    def stream_ctx(self):
        """Context manager that makes the chain's stream torch's current stream: torch-side work on the records (the RCCL
        gather, wait() of its handles, host copies) is issued inside it so that it is ordered against the kernels."""
        return torch.cuda.stream(self.main)

    def close(self):
        self.drain()
        torch.cuda.synchronize(self.dev)
        if self._pool is not None:
            self._pool.shutdown(wait=True)
        self.ex.close()
        self.ctx.close()
        for _, _, cx in self.bas:
            cx.close()

    def profile_enable(self, on=True, only=None):
        self.ctx.profile_enable(on, only)
        for _, _, cx in self.bas:
            cx.profile_enable(on, only)

    def profile_report(self):
        rep = dict(self.ctx.profile_report())
        for _, _, cx in self.bas:
            for k, (calls, ms) in cx.profile_report().items():
                c0, m0 = rep.get(k, (0, 0.0))
                rep[k] = (c0 + calls, m0 + ms)
        return rep

    def _run_ba(self, ba, st):
        torch.cuda.set_device(self.dev)  # the current device is per host thread
        ba.run()                         # resets its state and queues the LM rounds on `st` (the partition's context stream)

    # ---- inputs
    def set_stereo_frames(self, left, right):
        """left/right: uint8 arrays [F, H, W] (host). Kept resident in HBM: images [0,F) = left, [F,2F) = right."""
        left = np.ascontiguousarray(left, np.uint8); right = np.ascontiguousarray(right, np.uint8)
        assert left.shape == (self.F, self.height, self.width) and right.shape == left.shape
        # a step may still be reading the previous frames: join it, then upload on the chain's stream (the caching allocator
        # hands a freed block to the next allocation of the SAME stream only in stream order)
        self.drain()
        self.main.synchronize()
        with torch.cuda.stream(self.main):
            self.images = torch.from_numpy(np.concatenate([left, right], 0)).to(self.dev, non_blocking=False)
        self.ex.set_images_dev(self.images.data_ptr(), 2 * self.F, self.width, self.width * self.height)

    def set_synthetic(self, distinct=8, first=0):
        """Seeded synthetic stereo frames (synth.frame); `distinct` different pairs tiled over the batch."""
        pairs = [synth.frame(first + i, self.width, self.height, stereo=True) for i in range(min(distinct, self.F))]
        L = np.stack([pairs[i % len(pairs)][0] for i in range(self.F)])
        R = np.stack([pairs[i % len(pairs)][1] for i in range(self.F)])
        self.set_stereo_frames(L, R)
        return L, R

    # ---- one pass of the hot path
    def step(self):
        F, ex, ctx, L = self.F, self.ex, self.ctx, capi.lib()
        main = self.main
        self._cur ^= 1                           # this batch's records go to the other set
        if self.ba_lag:
            # Software-pipelined form (a local-mapping thread that runs one batch behind tracking): this batch's extractor
            # chain is queued at once; the BA partitions are handed their next batch as soon as their host drivers have
            # returned from the previous one (a driver blocks on its stream once per call: LM termination is data
            # dependent). step() does not wait for them -- the windows of batch s run beside the extraction of batch s + 1,
            # so the latency-bound tail of a partition's LM chain no longer leaves the chip to itself at the end of every
            # step. drain() joins everything; nothing reads a BA buffer between a step and the next without it.
            self.extract_chain()
            for fu in self._futures:
                fu.result()
            self._futures = [self._pool.submit(self._run_ba, ba, st) for ba, st, _ in self.bas]
            return
        futures = []
        for ba, st, _ in self.bas:
            # the BA windows run beside the extractor chain on their own streams (and host threads)
            if self._step_done is not None:
                st.wait_event(self._step_done)   # the previous step's consumers of the BA buffers are done
            futures.append(self._pool.submit(self._run_ba, ba, st))
        self.extract_chain()
        for fu in futures:
            fu.result()
        for _, st, _ in self.bas:
            main.wait_stream(st)                 # the step is complete when every stream is
        if self.bas:
            self._step_done = torch.cuda.Event()
            self._step_done.record(main)

    def drain(self):
        """Join the BA partitions' host drivers and order the main stream behind their streams (ba_lag: the windows of the
        last batch may still be running, or not even be queued completely, when step() returns)."""
        for fu in self._futures:
            fu.result()
        self._futures = []
        if self.ba_lag:
            for _, st, _ in self.bas:
                self.main.wait_stream(st)

    def extract_chain(self):
        """pyramid -> ORB -> searchByBF -> pose-opt -> track records of the resident batch, on the context's stream."""
        F, ex, ctx, L = self.F, self.ex, self.ctx, capi.lib()
        ex.build_pyramid(2 * F)
        ex.orb(2 * F, self.target, self.init_th, self.min_th)
        pitch = self.kp_cap * 32
        ctx.check(L.tb_search_by_bf_batch_dev(ctx._h, F, C.c_void_p(self.desc_ptr), C.c_void_p(self.counts_ptr),
                                              C.c_void_p(self.desc_ptr + F * pitch), C.c_void_p(self.counts_ptr + 4 * F),
                                              C.c_size_t(pitch), C.c_float(self.bf_ratio), C.c_float(self.bf_min_th),
                                              C.c_void_p(self.matches.data_ptr()), self.kp_cap,
                                              C.c_void_p(self.match_counts.data_ptr())))
        with torch.cuda.stream(self.main):
            self.outlier.zero_()                 # ordered on the chain's stream, between the matcher and pose-opt
        kp_bytes = self.kp_cap * 28
        ctx.check(L.tb_stereo_tracks_to_obs_batch_dev(ctx._h, F, C.c_void_p(self.kps_ptr), C.c_void_p(self.kps_ptr + F * kp_bytes),
                                                      self.kp_cap, C.c_void_p(self.matches.data_ptr()),
                                                      C.c_void_p(self.match_counts.data_ptr()), self.kp_cap,
                                                      self.Kf.ctypes.data_as(C.c_void_p), C.c_float(KITTI_BF),
                                                      self.inv_sigma2.ctypes.data_as(C.c_void_p), len(self.inv_sigma2),
                                                      C.c_void_p(self.obs.data_ptr()), self.obs_pitch,
                                                      C.c_void_p(self.obs_counts.data_ptr())))
        ctx.check(L.tb_pose_opt_batch_dev(ctx._h, F, self.K.ctypes.data_as(C.c_void_p), C.c_void_p(self.Tin.data_ptr()),
                                          C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.obs_counts.data_ptr()),
                                          self.obs_pitch, C.c_void_p(self.outlier.data_ptr()), C.c_void_p(self.Tout.data_ptr()),
                                          C.c_void_p(self.n_inliers.data_ptr()), C.c_void_p(self.pose_stats.data_ptr())))
        ex.copy_results_dev(F, self.trk_kps.data_ptr(), self.trk_desc.data_ptr(), self.trk_counts.data_ptr(), self.kp_cap)

    # ---- outputs (host copies, for tests)
    def frame_results(self, f):
        """(kps_left, desc_left, kps_right, desc_right, matches, Tcw, n_inliers, outlier) of frame f."""
        self.drain()
        torch.cuda.synchronize(self.dev)
        kl, dl = self.ex.results(f, self.kp_cap)
        kr, dr = self.ex.results(self.F + f, self.kp_cap)
        nm = int(self.match_counts[f].item())
        m = self.matches[f, :nm].cpu().numpy().view(capi.MATCH).reshape(-1)
        return (kl, dl, kr, dr, m, self.Tout[f].cpu().numpy().reshape(4, 4), int(self.n_inliers[f].item()),
                self.outlier[f].cpu().numpy())
