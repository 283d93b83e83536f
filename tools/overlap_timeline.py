"""Where the time of a run with several frames in flight goes, from a rocprofv3 --kernel-trace directory:
     python tools/overlap_timeline.py DIR [window_ms=40]
   Over the LAST `window_ms` of kernel activity before the final 5 % of the trace: wall, union of kernel intervals (device busy), sum of
   durations (average concurrency), per queue: busy share, dispatches, mean idle gap in front of a dispatch; the kernels with the largest
   total idle time in front of them (host round trips, cross-stream waits); totals per kernel name."""
import csv, glob, os, sys, collections
d = sys.argv[1]
win_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")) for r in rows)
t_end = ev[int(len(ev) * 0.95)][0]
t0 = t_end - int(win_ms * 1e6)
w = [e for e in ev if e[0] >= t0 and e[1] <= t_end]
wall = (t_end - t0) / 1e3
busy, cur_s, cur_e = 0, None, None
for s, e, *_ in w:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
if cur_e is not None: busy += cur_e - cur_s
tot = sum(e - s for s, e, *_ in w)
print(f"window {wall:.0f} us: {len(w)} dispatches, device busy (union) {busy / 1e3:.0f} us = {busy / 1e3 / wall:.2f}, sum of durations {tot / 1e3:.0f} us = concurrency {tot / max(busy, 1):.2f} while busy")
byq = collections.defaultdict(list)
for e in w: byq[(e[3], e[4])].append(e)
gap_by_kernel = collections.defaultdict(lambda: [0, 0.0])
for q, lst in sorted(byq.items()):
    lst.sort()
    b = sum(e[1] - e[0] for e in lst)
    gaps = [max(0, lst[i][0] - lst[i - 1][1]) for i in range(1, len(lst))]
    for i in range(1, len(lst)):
        g = gap_by_kernel[lst[i][2][:70]]; g[0] += 1; g[1] += max(0, lst[i][0] - lst[i - 1][1])
    print(f"  queue/stream {q}: {len(lst):5d} dispatches, busy {b / 1e3 / wall:.2f}, mean gap {sum(gaps) / max(1, len(gaps)) / 1e3:.1f} us, gaps > 20 us: {sum(1 for g in gaps if g > 20000)} (sum {sum(g for g in gaps if g > 20000) / 1e3:.0f} us)")
print("idle time in front of kernels (same queue), top 12:")
for name, (c, g) in sorted(gap_by_kernel.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {g / 1e3:9.0f} us over {c:5d} dispatches ({g / 1e3 / max(c, 1):6.1f} us each)  {name}")
byk = collections.defaultdict(lambda: [0, 0])
for s, e, n, *_ in w: byk[n[:70]][0] += 1; byk[n[:70]][1] += e - s
print("kernel time in the window, top 14:")
for name, (c, t) in sorted(byk.items(), key=lambda x: -x[1][1])[:14]:
    print(f"  {t / 1e3:9.0f} us {t / tot * 100:5.1f}%  {c:5d} x {t / c / 1e3:6.1f} us  {name}")
# ---- one frame of one stream under load: every dispatch with the idle gap in front of it (argument 3 = "list") -------------------
if len(sys.argv) > 3 and sys.argv[3] == "list":
    q = sorted(byq.items(), key=lambda x: -len(x[1]))[0][1]
    starts = [i for i, e in enumerate(q) if "depth_onepass_vec_kernel<false" in e[2] or "depth_onepass_vec_kernel<0" in e[2]]
    frames = [q[a:b] for a, b in zip(starts[:-1], starts[1:]) if b - a > 20]
    frames.sort(key=lambda f: f[-1][1] - f[0][0])
    fr = frames[len(frames) // 2]
    print(f"\nmedian frame of one stream under load: {len(fr)} dispatches, first start -> last end {(fr[-1][1] - fr[0][0]) / 1e3:.0f} us")
    prev = fr[0][0]
    for s, e, n, *_ in fr:
        nm = n.replace("kpx::", "").replace("void ", "").split("(")[0][:60]
        print(f"  +{(s - fr[0][0]) / 1e3:8.1f} us  gap {max(0, s - prev) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {nm}")
        prev = max(prev, e)
