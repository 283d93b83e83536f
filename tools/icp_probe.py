"""Where does an ICP iteration's time go?  The bench's own registrations (31k x 31k down-sampled views, point to plane), run
  alone (one registration, nothing else on the device), as the step runs them (3 side by side on the lanes), and 2 frames x 3.
Prints per-launch kernel durations (HIP event pairs, stride 1) and host wall times.

    python tools/icp_probe.py [reps]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kinectpy_amd import ops  # noqa: E402
from kinectpy_amd.pipeline import PipelineParams  # noqa: E402
from kinectpy_amd.utils import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
USE_PROF = "--noprof" not in sys.argv          # --noprof: no event pairs around the launches (for rocprofv3 traces: the pairs widen the gaps)
P = PipelineParams()
xy, depth_h, rgb_h, inits, _ = synth.sensor_ring(4, 1)
depth = torch.as_tensor(depth_h[0]).cuda()
fp, _, _, fcnt = ops.depth_to_cloud(depth, xy, None, 4, False, False, sync=False)
fk = ops._count(fcnt)
downs = [d[0] for d in ops.voxel_downsample_batch([fp[i, :fk[i]] for i in range(4)], P.reg_voxel)]
tn = ops.estimate_normals(downs[0], 2.0 * P.reg_voxel, P.normals_nn)
print("clouds", [int(d.shape[0]) for d in downs])


def run(count):
    return ops.icp_batch(downs[1:1 + count], downs[0], P.icp_max_dist, inits[:count], P.icp_mode, tn, P.icp_max_iteration)


def loop(fn, n_threads):
    """reps calls of fn per thread, each thread on its own stream -> wall ms per call"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = [None]
    if n_threads == 1:
        for _ in range(reps):
            last[0] = fn()
    else:
        def worker():
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(reps):
                    fn()
                torch.cuda.current_stream().synchronize()
        ths = [threading.Thread(target=worker) for _ in range(n_threads)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, last[0]


def measure(label, fn, n_threads=1):
    for _ in range(3):
        fn()
    wall_free, r = loop(fn, n_threads)                     # no event pairs around the launches
    if not USE_PROF:
        print(f"{label:46s} wall {wall_free:7.3f} ms/call")
        return
    ops.prof_stride(1)
    ops.prof_begin(1 << 16)
    ops.prof_icp_phases()
    wall, r = loop(fn, n_threads)
    pr = ops.prof_end()
    ph = ops.prof_icp_phases()
    print("   last sweep launch, per-block phases (us):", {k: (round(v, 2) if not isinstance(v, dict) else {a: round(b, 2) for a, b in v.items()}) for k, v in ph.items()})
    if "--waves" in sys.argv:
        wv = ops.prof_icp_waves()
        us = wv["sweep_us"]
        if len(us):
            order = np.argsort(us)
            print(f"   waves {len(us)}: sweep us  p10 {np.percentile(us, 10):.2f}  p50 {np.percentile(us, 50):.2f}  p90 {np.percentile(us, 90):.2f}  p99 {np.percentile(us, 99):.2f}  max {us.max():.2f};"
                  f"  start spread {(wv['start'].max() - wv['start'].min()) * 0.01:.2f} us")
            for name in ("tiles", "box_trips", "mul_trips", "groups_kept"):
                v = wv[name]
                print(f"      {name:12s} mean {v.mean():6.2f}  p50 {np.percentile(v, 50):5.1f}  p90 {np.percentile(v, 90):5.1f}  p99 {np.percentile(v, 99):5.1f}  max {v.max():4d}   corr with sweep us {np.corrcoef(v, us)[0, 1]:+.2f}")
            print("      slowest 12 waves (us, tiles, box_trips, mul_trips, groups_kept, sampled rows with partner):",
                  [(round(float(us[i]), 1), int(wv["tiles"][i]), int(wv["box_trips"][i]), int(wv["mul_trips"][i]), int(wv["groups_kept"][i]), int(wv["with_partner"][i])) for i in order[-12:]])
            # what sharing the work INSIDE a block could give at best: a block lasts as long as its slowest wave; with its four waves'
            # work dealt out evenly it would last their mean
            nb = len(us) // 4
            if nb:
                blk = us[: nb * 4].reshape(nb, 4)
                bmax, bmean = blk.max(1), blk.mean(1)
                print(f"      blocks {nb}: slowest wave of a block  mean {bmax.mean():.2f}  p99 {np.percentile(bmax, 99):.2f}  max {bmax.max():.2f};"
                      f"  mean of a block's four waves  mean {bmean.mean():.2f}  p99 {np.percentile(bmean, 99):.2f}  max {bmean.max():.2f}")
            # least-squares cost model: sweep us ~ a + b box_trips + c mul_trips
            A = np.stack([np.ones_like(us), wv["box_trips"], wv["mul_trips"]], 1).astype(float)
            coef = np.linalg.lstsq(A, us, rcond=None)[0]
            print(f"      model: sweep us = {coef[0]:.2f} + {coef[1]:.2f} per tile-box fetch + {coef[2]:.2f} per operand fetch (4 tiles)")
    ms, cnt, work = pr["nn_local"]
    its = [x["iterations"] for x in r] if r else None
    print(f"{label:46s} wall {wall_free:7.3f} ms/call free, {wall:7.3f} with event pairs;  icp_iter launches/call {cnt / reps / n_threads:6.1f}  "
          f"avg {ms / max(cnt, 1) * 1e3:6.2f} us  flop/launch {work / max(cnt, 1):.3g}  iterations {its}")


measure("1 registration alone", lambda: run(1))
if "--single" in sys.argv:                       # counter passes: launches of ONE registration only
    sys.exit(0)
measure("3 registrations side by side (one frame)", lambda: run(3))
measure("2 frames in flight x 3 registrations", lambda: run(3), n_threads=2)
