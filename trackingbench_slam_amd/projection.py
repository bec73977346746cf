"""Batched, device-resident projection matching (SURVEY 8f rows 1 + 3): Frame::AssignFeaturesToGrid for a batch of
current frames and Matcher::searchByProjection(F1, F2) for every (current frame, reference frame) pair, all in HBM
through tb_frame_grid_batch_dev / tb_search_by_projection_batch_dev.  torch supplies the device memory."""
import ctypes as C

import numpy as np
import torch

from . import capi, synth

GRID_CELLS = 120 * 36


class BatchedProjection:
    def __init__(self, ctx, cases, device=None, nratio=8.0, th_high=100, histo_len=30, check_orientation=True):
        """cases: list of dicts as synth.projection_case returns them (same camera / image size / scale factors)."""
        self.ctx, self.P = ctx, len(cases)
        dev = device if device is not None else torch.device("cuda", 0)
        c0 = cases[0]
        self.cam = np.ascontiguousarray(c0["cam"], capi.CAMERA)
        self.width, self.height = int(c0["width"]), int(c0["height"])
        self.sf = np.ascontiguousarray(c0["sf"], np.float32)
        self.nratio, self.th_high, self.histo_len, self.check = float(nratio), int(th_high), int(histo_len), int(check_orientation)
        self.pitch1 = max(max(len(c["k1"]) for c in cases), 1)
        self.pitch2 = max(max(len(c["k2"]) for c in cases), 1)
        P, p1, p2 = self.P, self.pitch1, self.pitch2
        k1 = np.zeros((P, p1), capi.KEYPOINT); d1 = np.zeros((P, p1, 32), np.uint8); tk = np.zeros((P, p1), np.uint8)
        k2 = np.zeros((P, p2), capi.KEYPOINT); mp = np.zeros((P, p2), capi.MAPPOINT); md = np.zeros((P, p2, 32), np.uint8)
        mp["bad"] = 1
        n1 = np.zeros(P, np.int32); n2 = np.zeros(P, np.int32); T = np.zeros((P, 16), np.float32)
        for i, c in enumerate(cases):
            a, b = len(c["k1"]), len(c["k2"])
            k1[i, :a] = c["k1"]; d1[i, :a] = c["d1"]; tk[i, :a] = c["taken1"]
            k2[i, :b] = c["k2"]; mp[i, :b] = c["mp"]; md[i, :b] = c["mp_desc"]
            n1[i], n2[i] = a, b
            T[i] = np.asarray(c["Tcw"], np.float32).reshape(16)

        def dev_bytes(a):
            return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).to(dev)
        self.k1, self.d1, self.tk, self.k2, self.mp, self.md = map(dev_bytes, (k1, d1, tk, k2, mp, md))
        self.n1 = torch.from_numpy(n1).to(dev); self.n2 = torch.from_numpy(n2).to(dev); self.T = torch.from_numpy(T).to(dev)
        self.cell_start = torch.zeros((P, GRID_CELLS + 1), dtype=torch.int32, device=dev)
        self.cell_items = torch.zeros((P, p1), dtype=torch.int32, device=dev)
        self.cap = p2
        self.out = torch.zeros((P, self.cap, 4), dtype=torch.int32, device=dev)
        self.out_counts = torch.zeros(P, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(P, dtype=torch.int32, device=dev)

    def build_grids(self):
        L = capi.lib()
        self.ctx.check(L.tb_frame_grid_batch_dev(self.ctx._h, self.P, C.c_void_p(self.k1.data_ptr()), C.c_void_p(self.n1.data_ptr()),
                                                 self.pitch1, self.width, self.height, C.c_void_p(self.cell_start.data_ptr()),
                                                 C.c_void_p(self.cell_items.data_ptr())))

    def search(self):
        L = capi.lib()
        self.ctx.check(L.tb_search_by_projection_batch_dev(
            self.ctx._h, self.P, C.c_void_p(self.T.data_ptr()), self.cam.ctypes.data_as(C.c_void_p), self.width, self.height,
            C.c_void_p(self.k1.data_ptr()), C.c_void_p(self.d1.data_ptr()), C.c_void_p(self.tk.data_ptr()), C.c_void_p(self.n1.data_ptr()),
            self.pitch1, C.c_void_p(self.cell_start.data_ptr()), C.c_void_p(self.cell_items.data_ptr()),
            C.c_void_p(self.k2.data_ptr()), C.c_void_p(self.mp.data_ptr()), C.c_void_p(self.md.data_ptr()), C.c_void_p(self.n2.data_ptr()),
            self.pitch2, self.sf.ctypes.data_as(C.c_void_p), len(self.sf), C.c_float(self.nratio), self.th_high, self.histo_len,
            self.check, C.c_void_p(self.out.data_ptr()), self.cap, C.c_void_p(self.out_counts.data_ptr()),
            C.c_void_p(self.flags.data_ptr())))

    def run(self):
        self.build_grids()
        self.search()

    def matches(self, p):
        """Host copy of pair p's match list (synchronises)."""
        n = min(int(self.out_counts[p].item()), self.cap)
        return self.out[p, :n].cpu().numpy().view(capi.MATCH).reshape(-1)
