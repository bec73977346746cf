#!/usr/bin/env python3
"""Driver for rocprofv3 passes and quick timing of the local-BA kernels: W windows of the bench size (10 keyframes, 5000
points, 10 LM iterations), a few runs, per-kernel HIP-event times. Usage: python tools/prof_ba.py [windows] [reps] [distinct] [presort|-] [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from trackingbench_slam_amd import capi  # noqa: E402
from trackingbench_slam_amd.ba import BatchedLocalBA  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 8
presort = len(sys.argv) > 4 and sys.argv[4] == "presort"
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
torch.cuda.set_device(0)
ctx = capi.Context(0)  # own stream; BatchedLocalBA waits for its reset copies on the host
ba = BatchedLocalBA(ctx, W, 10, 5000, iters, 0, torch.device("cuda", 0), distinct=distinct, presort=presort)
ba.run()
torch.cuda.synchronize()
ctx.profile_enable(True)
for _ in range(reps):
    ba.run()
torch.cuda.synchronize()
rep = ctx.profile_report()
for k, (c, ms) in sorted(rep.items()):
    print("%-16s %4d launches  %9.4f ms / run" % (k, c // reps, ms / reps))
print("chain %.4f ms / run (%d windows)" % (sum(ms for _, ms in rep.values()) / reps, W))
print("done", ba.stats[0].tolist())

import ctypes as C
if hasattr(capi.lib(), "tb_debug_ba_times"):
    buf = (C.c_ulonglong * 24)()
    capi.lib().tb_debug_ba_times(buf, 1)
    ba.run(); torch.cuda.synchronize()
    capi.lib().tb_debug_ba_times(buf, 1)
    ngr = max(buf[6], 1)
    names = {0: "stage + linearise", 1: "preload issue", 2: "block product", 3: "compact -> dense", 5: "between groups"}
    tot = sum(buf[i] for i in names)
    for i, nm in names.items():
        print("  %-18s %8.0f clk/group  %5.1f%%" % (nm, buf[i] / ngr, 100.0 * buf[i] / max(tot, 1)))
    print("  groups sampled %d, points per group %.2f" % (ngr, buf[7] / ngr))
    nwg = max(buf[10], 1)
    ns = max(buf[15], 1)
    print("  k_ba_solve per launch and window: assembly %.0f clk, factorisation %.0f, substitutions %.0f, state update %.0f" % (
        buf[11] / ns, buf[12] / ns, buf[13] / ns, buf[14] / ns))
    print("  per sampled wavefront: prologue %.0f clk, first loads + loop tail %.0f, epilogue %.0f, groups %.1f (%.0f clk)" % (
        buf[4] / nwg, buf[8] / nwg, buf[9] / nwg, ngr / nwg, tot / nwg))
