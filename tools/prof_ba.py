#!/usr/bin/env python3
"""Driver for rocprofv3 passes over the local-BA kernels: W windows of the bench size, a few runs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from trackingbench_slam_amd import capi  # noqa: E402
from trackingbench_slam_amd.ba import BatchedLocalBA  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.cuda.set_device(0)
ctx = capi.Context(0)  # own stream; torch.cuda.synchronize() below is device-wide
ba = BatchedLocalBA(ctx, W, 10, 5000, 10, 0, torch.device("cuda", 0))
for _ in range(reps):
    ba.run()
torch.cuda.synchronize()
print("done", ba.stats[0].tolist())
