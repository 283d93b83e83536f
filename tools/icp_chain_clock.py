"""Where an iteration of the one-launch ICP chain goes (KPX_ICP_CHAIN_STAMPS=1): the bench's longest registration alone, 30 iterations;
per iteration, microseconds from the publication of its record.    python tools/icp_chain_clock.py"""
import os, sys
os.environ["KPX_ICP_CHAIN_STAMPS"] = "1"
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops
from kinectpy_amd.pipeline import PipelineParams
from kinectpy_amd.utils import synth
P = PipelineParams()
xy, depth_h, rgb_h, inits, _ = synth.sensor_ring(4, 1)
depth = torch.as_tensor(depth_h[0]).cuda()
fp, _, _, fcnt = ops.depth_to_cloud(depth, xy, None, 4, False, False, sync=False)
fk = ops._count(fcnt)
downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(4)], P.reg_voxel)]
tn = ops.estimate_normals(downs[0], 2.0 * P.reg_voxel, P.normals_nn)
for _ in range(5):
    ops.icp_batch([downs[3]], downs[0], P.icp_max_dist, [inits[2]], P.icp_mode, tn, 30)
torch.cuda.synchronize()
ops.prof_icp_chain()
ops.icp_batch([downs[3]], downs[0], P.icp_max_dist, [inits[2]], P.icp_mode, tn, 30)
torch.cuda.synchronize()
s = ops.prof_icp_chain().astype(np.int64)
names = ["b0 seen", "b0 prepared", "b0 swept", "b0 added", "b0 ticket", "last seen", "last ticket", "totals read", "update done", "published", "first seen", "last added"]
print("iteration: us since the previous record's publication (k = 0: since the first block saw record 0); step = publication to publication")
prev_pub = None
for k in range(31):
    r = s[k]
    if r[9] == 0:
        break
    base = prev_pub if prev_pub is not None else r[10]
    f = lambda v: (v - base) / 100.0
    print(f"k {k:2d}  first seen {f(r[10]):6.2f}  last seen {f(r[5]):6.2f} | b0: seen {f(r[0]):6.2f} prepared {f(r[1]):6.2f} swept {f(r[2]):6.2f} added {f(r[3]):6.2f} ticket {f(r[4]):6.2f} | "
          f"last added {f(r[11]):6.2f}  last ticket {f(r[6]):6.2f} | winner: totals {f(r[7]):6.2f} update {f(r[8]):6.2f} published {f(r[9]):6.2f}")
    sw = int(r[13])
    print(f"       b0 wave 0: rows searched {bin(int(r[14]) & 0xFFFF).count('1'):2d} certified {bin((int(r[14]) >> 16) & 0xFFFF).count('1'):2d}  tiles multiplied {sw & 0xFFFF}  "
          f"tile-box trips {(sw >> 16) & 0xFFFF}  operand trips {(sw >> 32) & 0xFFFF}  groups kept {(sw >> 48) & 0xFFFF}")
    if r[16]:
        t0 = int(r[16]); g = lambda i: (int(r[16 + i]) - t0) / 100.0 if r[16 + i] else float("nan")
        print(f"       its sweep (us from entry): bounds published {g(1):5.2f}  groups tested {g(2):5.2f}  tile boxes asked {g(3):5.2f}  tiles listed {g(4):5.2f}  "
              f"operands asked {g(5):5.2f}  multiplied {g(6):5.2f}  bounds tightened {g(7):5.2f}  loop left {g(8):5.2f}  exit {g(9):5.2f}")
    if r[32]:
        t0 = int(r[32]); g = lambda i: (int(r[32 + i]) - t0) / 100.0 if r[32 + i] else float("nan")
        print(f"       the update (us from entry): fitness / rmse / test {g(1):5.2f}  system set up {g(2):5.2f}  eliminated {g(3):5.2f}  angles, U {g(4):5.2f}  "
              f"motion bound {g(5):5.2f}  T = U T {g(6):5.2f}  reach {g(7):5.2f}")
    print(f"       searched: {int(r[26])} waves, {int(r[27])} rows ({int(r[28])} without a partner, {int(r[29])} without a certificate)")
    prev_pub = r[9]
