#!/bin/bash
# rocprofv3 kernel-trace durations of the ICP kernels (no event pairs, no phase clock), per library variant
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
out=$ROOT/gpurun_out/exp_kernel_trace.txt
: > $out
cd /tmp
for v in "$@"; do
  lib=$ROOT/kinectpy_amd/libkinectpx_$v.so
  [ "$v" = "cur" ] && lib=$ROOT/kinectpy_amd/libkinectpx.so
  export KPX_LIBRARY=$lib
  rm -rf /tmp/kt_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -o t -- python3 $ROOT/tools/icp_probe.py 20 --noprof $EXTRA > /tmp/kt_$v.log 2>&1 || { tail -5 /tmp/kt_$v.log; exit 1; }
  echo "== $v" >> $out
  grep wall /tmp/kt_$v.log >> $out
  f=$(find /tmp/kt_$v -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:8]:
    print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.2f} us min {float(r['MinNs'])/1e3:7.2f} max {float(r['MaxNs'])/1e3:7.2f} share {float(r['TotalDurationNs'])/tot*100:5.1f}%")
PY
done
cat $out
