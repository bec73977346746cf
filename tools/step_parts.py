#!/usr/bin/env python3
"""What bounds a step of the bench pipeline: the whole step, the BA partitions alone (no extractor chain), the extractor chain
alone (no BA), and the BA partitions one after the other -- wall time per step, no per-kernel events.
Usage: python tools/step_parts.py [frames] [ba-split] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from trackingbench_slam_amd.pipeline import TrackingPipeline  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
split = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
pipe = TrackingPipeline(1280, 720, 8, 0.8, 2000, 80.0, 30.0, frames=F, device=0, with_ba=True, ba_kf=10, ba_pts=5000, ba_iters=10,
                        seed=0, ba_split=split, ba_distinct=32)
pipe.set_synthetic(distinct=64, first=0)


def timed(fn, n=steps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


full = timed(pipe.step)
chain = pipe.extract_chain
ext = timed(chain)
pipe.extract_chain = lambda: None
ba_only = timed(pipe.step)
pipe.extract_chain = chain


def serial():
    for ba, st, _ in pipe.bas:
        pipe._run_ba(ba, st)


for _, _, cx in pipe.bas:
    cx.set_concurrency(1)
ba_serial = timed(serial)
for _, _, cx in pipe.bas:
    cx.set_concurrency(len(pipe.bas))
print("frames %d, BA partitions %d: step %.3f ms | extractor chain alone %.3f | BA partitions alone (concurrent) %.3f | BA partitions "
      "one after the other, full-size grids %.3f" % (F, split, full, ext, ba_only, ba_serial))
pipe.close()
