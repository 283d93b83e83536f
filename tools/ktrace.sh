#!/bin/bash
# per-kernel durations of one Python script under rocprofv3 --kernel-trace --stats:
#   tools/ktrace.sh OUT_NAME script.py [args...]   ->   gpurun_out/OUT_NAME_kernels.txt
export TMPDIR=/tmp
ROOT=$PWD
name=$1; shift
mkdir -p $ROOT/gpurun_out/$(dirname $name)
cd /tmp && rm -rf /tmp/kt_run
timeout -k 10 ${KTRACE_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_run -o t -- python3 "$ROOT/$1" "${@:2}" > /tmp/kt_run.log 2>&1 || { tail -5 /tmp/kt_run.log; exit 1; }
f=$(find /tmp/kt_run -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $ROOT/gpurun_out/${name}_kernels.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:60]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
grep -v "^$" /tmp/kt_run.log | tail -5
cat $ROOT/gpurun_out/${name}_kernels.txt
