"""dev-only: cProfile of the pipeline step (where does the host time go?)"""
import cProfile, pstats, io, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from kinectpy_amd.pipeline import PipelineParams, SensorShardPipeline
xy, depth_h, rgb_h, inits, truth, _ = bench.make_group(0, 1, 4, 2)
depth = torch.as_tensor(depth_h).cuda(); rgb = torch.as_tensor(rgb_h).cuda()
pipe = SensorShardPipeline(xy, 4, inits, PipelineParams())
for k in range(5): pipe.step(depth[k % 2], rgb[k % 2])
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for k in range(100): pipe.step(depth[k % 2], rgb[k % 2])
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
