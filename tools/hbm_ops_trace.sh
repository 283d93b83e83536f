#!/bin/bash
# kernel-level durations of the roofline_targets leg (bench.roofline_targets) under rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/hot
cat > /tmp/hot_run.py <<PY
import sys, json
sys.path.insert(0, "$ROOT")
import torch, bench
from kinectpy_amd import ops
rows = bench.roofline_targets(torch, ops, quick=False)
for r in rows:
    print(json.dumps(r))
PY
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hot -o t -- python3 /tmp/hot_run.py > $ROOT/gpurun_out/hbm_ops_rows.txt 2>/tmp/hot.err || { tail -5 /tmp/hot.err; exit 1; }
f=$(find /tmp/hot -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/hbm_ops_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:70]:
    print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
cat $ROOT/gpurun_out/hbm_ops_kernels.txt
