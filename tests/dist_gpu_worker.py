"""Worker of test_two_ranks_share_one_gpu (tests/test_parity_gpu.py): one rank of a 2-rank gloo group on one GPU.  Runs two frames
through SensorGroupPipeline + FrameStream + exchange and writes what it ended up with to OUT_DIR/rank<r>.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kinectpy_amd import parallel  # noqa: E402
from kinectpy_amd.pipeline import FrameStream, PipelineParams, SensorGroupPipeline  # noqa: E402

rank, world, local = parallel.init_distributed()
out_dir = os.environ["OUT_DIR"]
spg = 2
xy, depth_h, rgb_h, inits, truth, to_global = bench.make_group(rank, world, spg, 2)
depth = torch.as_tensor(depth_h).cuda()
rgb = torch.as_tensor(rgb_h).cuda()
pipe = SensorGroupPipeline(xy, inits, PipelineParams(), cloud_capacity=spg * 48 * 1024)
frames = FrameStream(pipe, 2)
got = []
for f in range(3):
    if frames.full():
        p, c, Ts = frames.pop()
        got.append((p.clone(), c.clone(), Ts) + tuple(pipe.exchange(p, c, Ts, to_global)))
    frames.submit(depth[f % 2], rgb[f % 2])
while frames.pending:
    p, c, Ts = frames.pop()
    got.append((p.clone(), c.clone(), Ts) + tuple(pipe.exchange(p, c, Ts, to_global)))
frames.close()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), to_global=to_global,
         **{f"own_p{i}": g[0].cpu().numpy() for i, g in enumerate(got)},
         **{f"own_T{i}": g[2] for i, g in enumerate(got)},
         **{f"all_p{i}": g[3].cpu().numpy() for i, g in enumerate(got)},
         **{f"all_c{i}": g[4].cpu().numpy() for i, g in enumerate(got)},
         **{f"all_T{i}": g[5].cpu().numpy() for i, g in enumerate(got)},
         **{f"counts{i}": np.array(g[6]) for i, g in enumerate(got)})
parallel.barrier()
