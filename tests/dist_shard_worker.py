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
from kinectpy_amd.pipeline import FrameStream, NativeShardPipeline, PipelineParams, SensorShardPipeline  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

rank, world, local = parallel.init_distributed()
out_dir, S, mode = os.environ["OUT_DIR"], int(os.environ["N_SENSORS"]), os.environ.get("FUSED_FILTER", "sharded")
mine = parallel.shard_sensors(S, rank, world)
xy, depth_h, rgb_h, inits, _ = synth.sensor_ring(S, 2, sensors=mine)
depth, rgb = torch.as_tensor(depth_h).cuda(), torch.as_tensor(rgb_h).cuda()
save = {}
native = os.environ.get("NATIVE_LOOP") == "1"       # the frame loop and its collectives inside the library (kpx_frame_step_sharded)
if native:
    # the native loop keeps rank 0's result only in "rank0" mode: empty clouds elsewhere
    def make(group=None):
        return NativeShardPipeline(xy, S, inits, PipelineParams(), comm=parallel.NativeComm.staged(group), fused_filter=mode)
    keep = lambda p: p is not None and not (mode == "rank0" and rank != 0)
else:
    def make(group=None):
        return SensorShardPipeline(xy, S, inits, PipelineParams(), group=group, fused_filter=mode)
    keep = lambda p: p is not None
pipe = make()
for f in range(2):
    p, c, Ts = pipe.step(depth[f], rgb[f]) if not (native and f == 1) else pipe.step(torch.as_tensor(depth_h[f]).pin_memory(), torch.as_tensor(rgb_h[f]).pin_memory())
    save[f"T{f}"] = Ts
    if keep(p):
        save[f"p{f}"], save[f"c{f}"] = p.cpu().numpy(), c.cpu().numpy()
# two frames in flight: one pipeline (and one communicator) per slot
groups = [parallel.new_group() for _ in range(2)]
for g in groups:
    parallel.warm(g, depth.device)
fs = FrameStream([make(g) for g in groups])
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
    if keep(p):
        save[f"sp{k}"], save[f"sc{k}"] = p.cpu().numpy(), c.cpu().numpy()
np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **save)
parallel.barrier()
