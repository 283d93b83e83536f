#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/ft
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft -o t -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-targets --cpu-budget-s 0 --spread-blocks 0 --overlap 1 > /tmp/ft.log 2>&1 || { tail -5 /tmp/ft.log; exit 1; }
tail -1 /tmp/ft.log | cut -c1-200
python3 $ROOT/tools/frame_timeline.py /tmp/ft --list > $ROOT/gpurun_out/frame_timeline.txt
grep -v "^  +" $ROOT/gpurun_out/frame_timeline.txt
