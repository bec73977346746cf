import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from trackingbench_slam_amd import capi
from trackingbench_slam_amd.ba import BatchedLocalBA
torch.cuda.set_device(0)
ctx = capi.Context(0)
ba = BatchedLocalBA(ctx, 171, 10, 5000, 10, 0, torch.device("cuda", 0), distinct=16)
for _ in range(2): ba.run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): ba.run()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 5 * 1e3
ctx.profile_enable(True)
for _ in range(3): ba.run()
torch.cuda.synchronize()
rep = ctx.profile_report()
print("wall %.3f ms per run; sum of kernel durations %.3f ms; launches %d" % (wall, sum(ms for _, ms in rep.values()) / 3, sum(c for c, _ in rep.values()) // 3))
