"""durations of and gaps between the launches of icp_iter_batch_kernel in a rocprofv3 kernel trace (last complete chain of 31)"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
it = [(s, e) for s, e, n in ev if "icp_iter_batch_kernel" in n]
chain = it[-31:]
print("launch: duration us / gap in front us")
print(" ".join(f"{(e - s) / 1e3:.1f}/{(s - chain[i - 1][1]) / 1e3 if i else 0:.1f}" for i, (s, e) in enumerate(chain)))
print("chain first start -> last end %.1f us, sum of durations %.1f us" % ((chain[-1][1] - chain[0][0]) / 1e3, sum(e - s for s, e in chain) / 1e3))
