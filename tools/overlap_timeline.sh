#!/bin/bash
# tools/overlap_timeline.sh NAME [bench args]: the bench's frame loop under rocprofv3 --kernel-trace -> gpurun_out/NAME_overlap.txt
export TMPDIR=/tmp
ROOT=$PWD
name=$1; shift
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/kt_o
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_o -o t -- python3 "$ROOT/bench.py" --no-targets --cpu-budget-s 0 --no-stages --spread-blocks 0 --steps 300 --warmup 20 "$@" > /tmp/kt_o.log 2>&1 || { tail -5 /tmp/kt_o.log; exit 1; }
python3 $ROOT/tools/overlap_timeline.py /tmp/kt_o 40 list > $ROOT/gpurun_out/${name}_overlap.txt
grep -o '"value": [0-9.]*' /tmp/kt_o.log | head -1
cat $ROOT/gpurun_out/${name}_overlap.txt
