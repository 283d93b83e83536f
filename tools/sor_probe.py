"""remove_statistical_outlier at the sizes the bench rows use: config 3's cloud after voxel_down_sample(10) with (20, 2.0), the floor
chain's (50, 0.30) and filter_outliers' defaults (200, 3.0) on a fused 4-sensor cloud; KPX_SOR_CELL=0/1 in child processes, keep lists
compared.    python tools/sor_probe.py"""
import os
import subprocess
import sys

code = r'''
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from kinectpy_amd import ops
from kinectpy_amd.utils import synth
dev = torch.device("cuda")
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).to(dev)
v1 = ops.voxel_downsample(c3, 10.0)[0]
fv = ops.voxel_downsample(torch.as_tensor(synth.frame_cloud()).to(dev), 10.0)[0]
fr = ops.voxel_downsample(torch.as_tensor(synth.frame_cloud()).to(dev), 35.0)[0]
out = {}
for name, cloud, k, r in (("config3 259k (20, 2.0)", v1, 20, 2.0), ("config3 259k (50, 0.30)", v1, 50, 0.30), ("fused 221k (200, 3.0)", fv, 200, 3.0),
                          ("frame-sized (20, 2.0)", fr, 20, 2.0), ("raw 1M (20, 2.0)", c3, 20, 2.0)):
    for _ in range(2):
        keep, stats, _ = ops.sor(cloud, k, r)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); keep, stats, _ = ops.sor(cloud, k, r); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    st = stats.cpu().numpy() if hasattr(stats, 'cpu') else np.asarray(stats)
    out[name] = (float(np.median(ts)), int(cloud.shape[0]), keep.cpu().numpy(), st)
    print(f"{os.environ.get('KPX_SOR_CELL', '1')} {name:28s} n {cloud.shape[0]:8d} kept {keep.shape[0]:8d}  {np.median(ts):8.3f} ms  {cloud.shape[0] / np.median(ts) / 1e3:8.1f} Mqueries/s  stats {st[:3]}")
np.savez(sys.argv[1], **{k.replace(" ", "_").replace(",", "").replace("(", "").replace(")", "").replace(".", "p"): v[2] for k, v in out.items()})
'''
outs = []
for flag in (("1",) if os.environ.get("KPX_PROBE_ONLY") else ("1", "0")):
    f = f"/tmp/sor_probe_{flag}.npz"
    r = subprocess.run([sys.executable, "-c", code, f], env={**os.environ, "KPX_SOR_CELL": flag}, capture_output=True, text=True)
    print(r.stdout, r.stderr[-1500:] if r.returncode else "")
    outs.append(f)
import numpy as np
if len(outs) < 2:
    sys.exit(0)
a, b = np.load(outs[0]), np.load(outs[1])
for k in a.files:
    print("keep lists equal:", k, np.array_equal(a[k], b[k]))
