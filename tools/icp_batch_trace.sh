#!/bin/bash
# tools/icp_batch_trace.sh NAME [registrations]: per-launch durations / gaps of the iteration kernel in the LAST batch call of
# tools/icp_batch_trace.py, under rocprofv3 --kernel-trace (whichever form KPX_ICP_ROWS selects) -> gpurun_out/NAME_launches.txt
export TMPDIR=/tmp
ROOT=$PWD
name=$1
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/kt_b
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_b -o t -- python3 "$ROOT/tools/icp_batch_trace.py" ${2:-3} > /tmp/kt_b.log 2>&1 || { tail -5 /tmp/kt_b.log; exit 1; }
python3 - /tmp/kt_b > $ROOT/gpurun_out/${name}_launches.txt <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
it = [(s, e) for s, e, n in ev if "icp_rows_kernel" in n or "icp_iter_batch_kernel" in n]
# the last chain: launches after the last icp_batch_init_kernel
init = [s for s, e, n in ev if "icp_batch_init_kernel" in n]
chain = [x for x in it if x[0] > init[-1]]
print("launches", len(chain), "duration us / gap in front us")
print(" ".join(f"{(e - s) / 1e3:.1f}/{(s - chain[i - 1][1]) / 1e3 if i else 0:.1f}" for i, (s, e) in enumerate(chain)))
print("chain first start -> last end %.1f us, sum of durations %.1f us" % ((chain[-1][1] - chain[0][0]) / 1e3, sum(e - s for s, e in chain) / 1e3))
PY
tail -2 /tmp/kt_b.log
cat $ROOT/gpurun_out/${name}_launches.txt
