"""Worker of test_sensor_partition_equals_single_process_oracle (tests/test_parity_gpu.py): one rank of a gloo group whose
ranks share the one GPU of the box.  Runs the north-star partition (kinectpy_amd.pipeline.SensorShardPipeline on the HIP
kernels) for N_SENSORS sensors, serially and with two frames in flight (one communicator per slot), and writes what it
ended up with to OUT_DIR/rank<r>.npz."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import parallel  # noqa: E402
from kinectpy_amd.pipeline import FrameStream, PipelineParams, SensorShardPipeline  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

rank, world, local = parallel.init_distributed()
out_dir, S, mode = os.environ["OUT_DIR"], int(os.environ["N_SENSORS"]), os.environ.get("FUSED_FILTER", "sharded")
mine = parallel.shard_sensors(S, rank, world)
xy, depth_h, rgb_h, inits, _ = synth.sensor_ring(S, 2, sensors=mine)
depth, rgb = torch.as_tensor(depth_h).cuda(), torch.as_tensor(rgb_h).cuda()
save = {}
pipe = SensorShardPipeline(xy, S, inits, PipelineParams(), fused_filter=mode)
for f in range(2):
    p, c, Ts = pipe.step(depth[f], rgb[f])
    save[f"T{f}"] = Ts
    if p is not None:
        save[f"p{f}"], save[f"c{f}"] = p.cpu().numpy(), c.cpu().numpy()
# two frames in flight: one pipeline (and one communicator) per slot
groups = [parallel.new_group() for _ in range(2)]
for g in groups:
    parallel.warm(g, depth.device)
fs = FrameStream([SensorShardPipeline(xy, S, inits, PipelineParams(), group=g, fused_filter=mode) for g in groups])
got = []
for k in range(4):
    if fs.full():
        got.append(fs.pop())
    fs.submit(depth[k % 2], rgb[k % 2])
while fs.pending:
    got.append(fs.pop())
fs.close()
for k, (p, c, Ts) in enumerate(got):
    save[f"sT{k}"] = Ts
    if p is not None:
        save[f"sp{k}"], save[f"sc{k}"] = p.cpu().numpy(), c.cpu().numpy()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **save)
parallel.barrier()
