#!/usr/bin/env python3
"""Small driver for rocprofv3 passes: runs the extractor stages (pyramid + ORB) of the bench workload a few
times on a resident batch. Usage: python tools/prof_extract.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from trackingbench_slam_amd.pipeline import TrackingPipeline  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
p = TrackingPipeline(1280, 720, 8, 0.8, 2000, 80.0, 30.0, frames=F, with_ba=False)
p.set_synthetic(distinct=min(8, F))
for _ in range(reps):
    p.ex.build_pyramid(2 * F)
    p.ex.orb(2 * F, 2000, 80.0, 30.0)
torch.cuda.synchronize()
print("done", p.ex.counts(2 * F)[:4])
