"""Where does ONE frame's time go?  Reads a `rocprofv3 --kernel-trace --output-format csv` trace of `bench.py --overlap 1`, cuts it
into frames at the first extraction kernel and prints, for the median frame: every dispatch in order with its start offset,
duration and the idle gap in front of it, then totals per kernel (busy time, gaps in front, launches).

    python tools/frame_timeline.py <dir with *_kernel_trace.csv> [--list]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
if not rows:
    sys.exit("no kernel trace found")
nm = "Kernel_Name" if "Kernel_Name" in rows[0] else "kernel_name"
sc = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "start_timestamp"
ec = "End_Timestamp" if "End_Timestamp" in rows[0] else "end_timestamp"
ev = sorted((int(r[sc]), int(r[ec]), r[nm]) for r in rows)


def short(n):
    n = n.replace("kpx::", "").replace("void ", "")
    if "rocprim" in n:
        for k in ("onesweep", "histogram", "scan", "block_sort", "merge", "partition", "reduce", "transform", "select"):
            if k in n:
                return "rocprim:" + k
        return "rocprim:other"
    return n.split("(")[0][:44]


starts = [i for i, e in enumerate(ev) if "depth_onepass_vec_kernel<false" in e[2] or "depth_onepass_vec_kernel<0" in e[2]]
frames = [ev[a:b] for a, b in zip(starts[:-1], starts[1:])]
frames = [f for f in frames if len(f) > 20]
if not frames:
    sys.exit("no frames found")
spans = sorted(range(len(frames)), key=lambda i: frames[i][-1][1] - frames[i][0][0])
fr = frames[spans[len(spans) // 2]]
t0 = fr[0][0]
period = sorted(b[0][0] - a[0][0] for a, b in zip(frames[:-1], frames[1:]))[len(frames) // 2] / 1e3
print(f"{len(frames)} frames; median frame: {len(fr)} dispatches, first start -> last end {(fr[-1][1] - t0) / 1e3:.1f} us, frame period {period:.1f} us")
# averages over ALL frames (the frames of a run differ in their iteration counts)
tot_busy = sum((e - s_) for fr_ in frames for s_, e, _ in fr_) / len(frames) / 1e3
icp = [(e - s_) / 1e3 for fr_ in frames for s_, e, n in fr_ if "icp_iter" in n]
span_icp = []
for fr_ in frames:
    ii = [(s_, e) for s_, e, n in fr_ if "icp_iter" in n]
    if ii:
        span_icp.append((ii[-1][1] - ii[0][0]) / 1e3)
print(f"all frames: period {(frames[-1][0][0] - frames[0][0][0]) / (len(frames) - 1) / 1e3:.1f} us, busy {tot_busy:.1f} us, ICP launches {len(icp) / len(frames):.1f} per frame, "
      f"ICP busy {sum(icp) / len(frames):.1f} us, ICP chain first start -> last end {sum(span_icp) / max(len(span_icp), 1):.1f} us, sweeps >= 10 us: {sum(1 for v in icp if v >= 10) / len(frames):.1f} per frame "
      f"(avg {sum(v for v in icp if v >= 10) / max(1, sum(1 for v in icp if v >= 10)):.1f} us)")
busy = defaultdict(float)
gap = defaultdict(float)
cnt = defaultdict(int)
prev_end = t0
for s, e, n in fr:
    k = short(n)
    busy[k] += (e - s) / 1e3
    gap[k] += max(0, s - prev_end) / 1e3
    cnt[k] += 1
    if "--list" in sys.argv:
        print(f"  +{(s - t0) / 1e3:8.1f} us  gap {max(0, s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {k}")
    prev_end = max(prev_end, e)
tb, tg = sum(busy.values()), sum(gap.values())
print(f"busy {tb:.1f} us, idle in front of dispatches {tg:.1f} us, idle after the last dispatch until the next frame {period - (fr[-1][1] - t0) / 1e3:.1f} us")
for k in sorted(busy, key=lambda k: -(busy[k] + gap[k])):
    print(f"  {k:46s} x{cnt[k]:3d}  busy {busy[k]:7.1f}  gaps in front {gap[k]:7.1f}")
