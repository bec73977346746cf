import sys, os
sys.path.insert(0, os.getcwd())
import torch
from trackingbench_slam_amd.pipeline import TrackingPipeline
F=32
p = TrackingPipeline(1280, 720, 8, 0.8, 2000, 80.0, 30.0, frames=F, with_ba=False)
p.set_synthetic(distinct=8)
for th in [(80,30),(250,250),(120,100),(60,60),(30,30),(10,10)]:
    for _ in range(2):
        p.ex.build_pyramid(2*F); p.ex.orb(2*F, 2000, th[0], th[1])
    torch.cuda.synchronize()
    p.ctx.profile_enable(True)
    for _ in range(5):
        p.ex.orb(2*F, 2000, th[0], th[1])
    r = p.ctx.profile_report(); p.ctx.profile_enable(False)
    print(th, {k: round(v[1]/v[0],4) for k,v in r.items()}, int(p.ex.counts(2)[0]))
