"""Batched multi-keyframe local BA windows for the pipeline (north-star extension, no reference
counterpart: SURVEY.md D1 / a17).  One seeded synthetic window per stereo frame (synth.ba_problem),
resident in HBM; run() re-optimises every window from its initial state through tb_local_ba_batch_dev."""
import ctypes as C

import numpy as np
import torch

from . import capi, synth
KITTI_K = (718.856, 718.856, 607.1928, 185.2157)


def _pattern_sorted(prob, nfixed):
    """The same window with its points renumbered by ascending visibility mask (stable), observations regrouped."""
    Pt, Pi, Xt, Xi, o = prob
    npt = len(Xi)
    free = o["kf"] >= nfixed
    mask = np.zeros(npt, np.int64)
    np.bitwise_or.at(mask, o["pt"][free], np.int64(1) << (o["kf"][free] - nfixed).astype(np.int64))
    perm = np.argsort(mask, kind="stable")          # new index r holds old point perm[r]
    rank = np.empty(npt, np.int64); rank[perm] = np.arange(npt)
    o2 = o.copy()
    o2["pt"] = rank[o["pt"]]
    o2 = o2[np.argsort(o2["pt"], kind="stable")]
    return Pt, Pi, Xt[perm], Xi[perm], o2


class BatchedLocalBA:
    def __init__(self, ctx, nwindows, nkf=10, npt=5000, iters=10, seed=0, device=None, nfixed=2, distinct=4, stream=None, presort=False):
        self.ctx, self.W, self.nkf, self.npt, self.iters, self.nfixed = ctx, int(nwindows), int(nkf), int(npt), int(iters), nfixed
        probs = [synth.ba_problem(seed * 100 + i, nkf, npt, KITTI_K) for i in range(min(distinct, self.W))]
        if presort:   # measurement aid: points renumbered in visibility-pattern order (what k_ba_groups' ranks would be)
            probs = [_pattern_sorted(p, nfixed) for p in probs]
        self.obs_pitch = max(len(p[4]) for p in probs)
        obs = np.zeros((self.W, self.obs_pitch), capi.BA_OBS)
        cnt = np.zeros(self.W, np.int32)
        poses = np.zeros((self.W, nkf, 16), np.float32)
        pts = np.zeros((self.W, npt, 3), np.float32)
        for w in range(self.W):
            Pt, Pi, Xt, Xi, o = probs[w % len(probs)]
            obs[w, :len(o)] = o
            cnt[w] = len(o)
            poses[w] = Pi.reshape(nkf, 16)
            pts[w] = Xi
        self.host = dict(obs=obs, counts=cnt, poses=poses, pts=pts)
        # edges with a free keyframe enter the Schur complement (bench.py's roofline accounting)
        self.free_edges = int(sum(int((obs[w, :cnt[w]]["kf"] >= nfixed).sum()) for w in range(self.W)))
        self.edges = int(cnt.sum())
        # (edge, edge) items of the block-pair lists the large-window Schur kernel walks (more than 10 free keyframes)
        self.pair_items = 0
        self.sum_k2_free = 0   # sum over points of (observations by free keyframes)^2: SURVEY 8(d)'s sparse Schur flop count / 216
        # what the pattern-compact Schur kernel EXECUTES per LM trial (k_ba.hip: groups of one visibility pattern, at most
        # 21 points / 64 edges; ceil(3 points / 4) k-steps of NACC(k) v_mfma_f64_4x4x4 instructions, 512 flop each; patterns of
        # more than five keyframes take the direct vector form, 2 x 36 flop per tile column and block pair)
        self.mfma_flops_executed = 0
        nacc = {1: 2, 2: 3, 3: 5, 4: 9, 5: 10}
        nfree = nkf - nfixed
        for w in range(self.W):
            o = obs[w, :cnt[w]]
            free = o["kf"] >= nfixed
            e = np.bincount(o["pt"][free], minlength=npt).astype(np.int64)
            self.pair_items += int((e * (e + 1) // 2).sum())
            self.sum_k2_free += int((e * e).sum())
            if nfree <= 10:
                mask = np.zeros(npt, np.int64)
                np.bitwise_or.at(mask, o["pt"][free], np.int64(1) << (o["kf"][free] - nfixed).astype(np.int64))
                pats, pc = np.unique(mask[mask > 0], return_counts=True)
                for m, c in zip(pats.tolist(), pc.tolist()):
                    k = bin(m).count("1")
                    cap = min(64 // k, 21)
                    full, rest = divmod(c, cap)
                    ksteps = full * ((3 * cap + 3) // 4) + ((3 * rest + 3) // 4 if rest else 0)
                    if k <= 5:
                        self.mfma_flops_executed += ksteps * nacc[k] * 512
                    else:
                        self.mfma_flops_executed += c * 3 * (k * (k + 1) // 2) * 72
        dev = device
        self.obs = torch.from_numpy(obs.view(np.uint8).reshape(self.W, self.obs_pitch, capi.BA_OBS.itemsize)).to(dev)
        self.counts = torch.from_numpy(cnt).to(dev)
        self.poses0 = torch.from_numpy(poses).to(dev)
        self.pts0 = torch.from_numpy(pts).to(dev)
        self.poses = torch.empty_like(self.poses0)
        self.pts = torch.empty_like(self.pts0)
        self.stats = torch.zeros((self.W, 8), dtype=torch.float64, device=dev)
        self.K = np.ascontiguousarray(KITTI_K, np.float64)
        # `stream`: the torch stream whose handle the context runs on (the pipeline passes it). Without one the reset
        # copies go to torch's current stream and are waited for on the host before the kernels are queued -- the
        # context's own stream is not ordered against torch's streams.
        self.stream = stream

    def run(self):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                self.poses.copy_(self.poses0)
                self.pts.copy_(self.pts0)
        else:
            self.poses.copy_(self.poses0)
            self.pts.copy_(self.pts0)
            torch.cuda.current_stream(self.poses.device).synchronize()
        self.ctx.check(capi.lib().tb_local_ba_batch_dev(
            self.ctx._h, self.W, self.K.ctypes.data_as(C.c_void_p), self.nkf, self.nfixed, C.c_void_p(self.poses.data_ptr()),
            self.npt, C.c_void_p(self.pts.data_ptr()), C.c_void_p(self.obs.data_ptr()), C.c_void_p(self.counts.data_ptr()),
            self.obs_pitch, self.iters, C.c_void_p(self.stats.data_ptr())))
