"""python tools/hip_api_gaps.py DIR: from a rocprofv3 --hip-runtime-trace csv, per host thread the API calls around the largest idle gaps
(time between the end of one runtime call and the start of the next on the same thread) -- what a frame thread does between two frames"""
import csv, glob, os, sys, collections
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*hip_api_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    by[r["Thread_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]))
for tid, ev in sorted(by.items(), key=lambda x: -len(x[1]))[:6]:
    ev.sort()
    n = len(ev)
    mid = ev[n // 2:]                                   # steady state
    launches = sum(1 for e in mid if "Launch" in e[2])
    print(f"thread {tid}: {n} calls, second half: {len(mid)} calls, {launches} launches, mean call {sum(e[1]-e[0] for e in mid)/len(mid)/1e3:.1f} us")
    hist = collections.Counter()
    tot = collections.Counter()
    for e in mid:
        hist[e[2]] += 1; tot[e[2]] += e[1] - e[0]
    for name, t in tot.most_common(8):
        print(f"     {name:40s} x{hist[name]:6d}  total {t/1e3:9.0f} us  mean {t/hist[name]/1e3:7.1f} us")
    gaps = sorted(((mid[i][0] - mid[i-1][1], i) for i in range(1, len(mid))), reverse=True)[:3]
    for g, i in gaps:
        print(f"   gap {g/1e3:.0f} us before call {i}:")
        for e in mid[max(0, i-4): i+4]:
            print(f"      +{(e[0]-mid[i][0])/1e3:9.1f} us  dur {(e[1]-e[0])/1e3:7.1f}  {e[2]}")
