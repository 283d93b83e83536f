"""where the cell kernel's queries and cycles go (variant library libkinectpx_sorstats.so built with -DKPX_SOR_CELL_STATS):
    KPX_LIBRARY=$PWD/kinectpy_amd/libkinectpx_sorstats.so python tools/sor_stats.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops, _lib
from kinectpy_amd.utils import synth
dev = torch.device("cuda")
L = _lib.load()
c3 = torch.as_tensor(synth.filter_cloud(1_000_000)).to(dev)
v1 = ops.voxel_downsample(c3, 10.0)[0]
fv = ops.voxel_downsample(torch.as_tensor(synth.frame_cloud()).to(dev), 10.0)[0]
names = ["settled r1", "settled r2", "fb m>cap r1", "fb m>cap r2", "fb uncovered", "fb ties", "sum m", "cells staged", "clk stage", "clk dist+cover", "clk k-th word", "clk select+sum", "clk wave total", "levels / steps"]
out = np.zeros(16, dtype=np.uint64)
for name, cloud, k, r in (("config3 259k k=20", v1, 20, 2.0), ("config3 k=50", v1, 50, 0.3), ("fused 221k k=200", fv, 200, 3.0)):
    ops.sor(cloud, k, r); torch.cuda.synchronize()
    L.kpx_debug_sor_stats(out.ctypes.data_as(C.c_void_p))
    ops.sor(cloud, k, r); torch.cuda.synchronize()
    L.kpx_debug_sor_stats(out.ctypes.data_as(C.c_void_p))
    n = cloud.shape[0]
    print(name, "n", n)
    for i, nm in enumerate(names):
        v = int(out[i])
        extra = ""
        if i < 6: extra = f"{100.0 * v / n:6.2f} % of queries"
        if i == 6: extra = f"{v / max(1, int(out[7])):6.1f} candidates per staged block"
        if 8 <= i <= 12: extra = f"{100.0 * v / max(1, int(out[12])):6.2f} % of wave time; {v * 0.01 / max(1, n) :8.4f} us per query"
        if i == 13: extra = f"{v / max(1, int(out[0]) + int(out[1])):6.2f} per settled query (wave trips / settled queries)"
        print(f"   {nm:16s} {v:14d}  {extra}")
