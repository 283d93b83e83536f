"""Condense a `rocprofv3 --kernel-trace --output-format csv` trace: per-kernel count / average duration, and for a chain kernel
(default icp_iter_kernel) the idle gap between consecutive launches on the same queue.

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [kernel substring]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "icp_iter_kernel"
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
if not rows:
    sys.exit("no kernel trace found")
name_col = "Kernel_Name" if "Kernel_Name" in rows[0] else "kernel_name"
s_col = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "start_timestamp"
e_col = "End_Timestamp" if "End_Timestamp" in rows[0] else "end_timestamp"
q_col = next((c for c in ("Queue_Id", "queue_id", "Stream_Id", "stream_id") if c in rows[0]), None)
per = defaultdict(list)
for r in rows:
    per[r[name_col]].append((int(r[s_col]), int(r[e_col]), r.get(q_col, "0")))
t_lo = min(int(r[s_col]) for r in rows)
t_hi = max(int(r[e_col]) for r in rows)
print(f"{len(rows)} dispatches over {(t_hi - t_lo) / 1e6:.3f} ms")
tot = sorted(((sum(e - s for s, e, _ in v), k) for k, v in per.items()), reverse=True)
for t, k in tot[:25]:
    v = per[k]
    print(f"{t / 1e3:10.1f} us total  {len(v):6d} x  {t / len(v) / 1e3:8.2f} us   {k[:110]}")
chain = [x for k, v in per.items() if key in k for x in v]
byq = defaultdict(list)
for s, e, q in chain:
    byq[q].append((s, e))
gaps = []
for q, v in byq.items():
    v.sort()
    for (s0, e0), (s1, e1) in zip(v, v[1:]):
        g = s1 - e0
        if 0 <= g < 40000:              # consecutive launches of one chain (a longer pause = the next call)
            gaps.append(g)
if gaps:
    gaps.sort()
    print(f"{key}: {len(chain)} launches on {len(byq)} queues; gap end->next start: median {gaps[len(gaps) // 2] / 1e3:.2f} us, "
          f"p10 {gaps[len(gaps) // 10] / 1e3:.2f}, p90 {gaps[len(gaps) * 9 // 10] / 1e3:.2f}, mean {sum(gaps) / len(gaps) / 1e3:.2f}")
