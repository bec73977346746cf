#!/usr/bin/env python3
"""Small driver for rocprofv3 passes and quick timing: runs the extractor stages (pyramid + ORB) of the bench workload a
few times on a resident batch and prints the per-kernel HIP-event times. Usage: python tools/prof_extract.py [frames] [reps] [distinct]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from trackingbench_slam_amd.pipeline import TrackingPipeline  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 8
p = TrackingPipeline(1280, 720, 8, 0.8, 2000, 80.0, 30.0, frames=F, with_ba=False)
p.set_synthetic(distinct=min(distinct, F))
for _ in range(2):
    p.ex.build_pyramid(2 * F)
    p.ex.orb(2 * F, 2000, 80.0, 30.0)
torch.cuda.synchronize()
p.ctx.profile_enable(True)
for _ in range(reps):
    p.ex.build_pyramid(2 * F)
    p.ex.orb(2 * F, 2000, 80.0, 30.0)
torch.cuda.synchronize()
rep = p.ctx.profile_report()
for k, (c, ms) in sorted(rep.items()):
    print("%-16s %4d launches  %9.4f ms / rep" % (k, c, ms / reps))
print("chain %.4f ms / rep (%d images)" % (sum(ms for _, ms in rep.values()) / reps, 2 * F))
print("done", p.ex.counts(2 * F)[:4])
# debug builds (make EXTRA=-DFB_TIMING): per-stage shader clocks of the FAST kernel
import ctypes as C
from trackingbench_slam_amd import capi
if hasattr(capi.lib(), "tb_debug_fast_times"):
    buf = (C.c_ulonglong * 16)()
    capi.lib().tb_debug_fast_times(buf, 1)
    p.ex.orb(2 * F, 2000, 80.0, 30.0)
    torch.cuda.synchronize()
    capi.lib().tb_debug_fast_times(buf, 1)
    nb = max(buf[8], 1)
    names = ["stage0 load", "stage1 cardinal", "stage2a expand", "stage2b score", "stage3 nms", "retry", "emit"]
    tot = sum(buf[i] for i in range(7))
    for i, nm in enumerate(names):
        print("  %-16s %8.0f clk/block  %5.1f%%" % (nm, buf[i] / nb, 100.0 * buf[i] / max(tot, 1)))
    print("  dense blocks (all, not sampled): %d; max records %d, max pixels %d, max corners %d among them" % (buf[14], buf[13], buf[15], buf[7]))
    print("  blocks %d, per block: records %.1f, pixels %.1f, corners %.1f, retry corners %.1f, retry cells %.3f"
          % (nb, buf[9] / nb, buf[10] / nb, buf[11] / nb, buf[12] / nb, buf[13] / nb))

if hasattr(capi.lib(), "tb_debug_octree_times"):
    buf = (C.c_ulonglong * 64)()
    capi.lib().tb_debug_octree_times(buf, 1)
    p.ex.orb(2 * F, 2000, 80.0, 30.0)
    torch.cuda.synchronize()
    capi.lib().tb_debug_octree_times(buf, 1)
    for l in range(8):
        nb = max(buf[16 + l], 1)
        print("  octree level %d: %8.0f clk/block, %6.0f candidates/block, %d blocks" % (l, buf[l] / nb, buf[32 + l] / nb, buf[16 + l]))
