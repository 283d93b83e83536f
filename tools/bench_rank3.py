"""Only the sampler / normaliser rows of tools/bench_kernels.py (quick iteration on kpx_norm.hip)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402
from tools.bench_kernels import report, timed  # noqa: E402


def main():
    dev = torch.device("cuda")
    host = synth.filter_cloud(260_000)
    fused = torch.as_tensor(host).to(dev)
    ms, _ = timed(lambda: ops.sample_points(fused, 4096, 7))
    report("select_points_randomly 260k -> 4096", ms, n=260_000)
    rng = np.random.default_rng(0)
    for B in (1, 32, 256):
        xb = torch.as_tensor(np.stack([host[rng.choice(len(host), 4096, replace=False)] for _ in range(B)]).astype(np.float64)).to(dev)
        ms, (obb, _) = timed(lambda: ops.obb_batch(xb, check=False))
        v = obb[:, 15].cpu().numpy()
        report(f"obb batch {B} x 4096 f64", ms, ms_per_cloud=round(ms / B, 4), hull_vertices_mean=float(v.mean()))
    ms, (obb, _) = timed(lambda: ops.obb_batch(fused, check=False), reps=3, warm=1)
    report("obb one 260k-point cloud f32", ms, hull_vertices=float(obb[0, 15]))
    s = rng.normal(size=(4096, 3)); s /= np.linalg.norm(s, axis=1)[:, None]
    ms, (obb, _) = timed(lambda: ops.obb_batch(torch.as_tensor(s).to(dev), check=False), reps=3, warm=1)
    report("obb sphere 4096 (every point a vertex)", ms, hull_vertices=float(obb[0, 15]))


if __name__ == "__main__":
    main()
