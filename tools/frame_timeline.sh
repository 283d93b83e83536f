#!/bin/bash
# one frame at a time under rocprofv3 --kernel-trace: per-kernel busy time and gaps of the median frame (tools/frame_timeline.py);
# extra environment (A/B switches) is inherited.  Usage: tools/frame_timeline.sh [tag]
export TMPDIR=/tmp
ROOT=$PWD
tag=${1:-cur}
mkdir -p gpurun_out
cd /tmp && rm -rf /tmp/ft_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ft_$tag -o t -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-targets --cpu-budget-s 0 --spread-blocks 0 --overlap 1 > /tmp/ft_$tag.log 2>&1 || { tail -5 /tmp/ft_$tag.log; exit 1; }
python3 $ROOT/tools/frame_timeline.py /tmp/ft_$tag --list > $ROOT/gpurun_out/frame_timeline_$tag.txt
grep -v "^  +" $ROOT/gpurun_out/frame_timeline_$tag.txt | head -9
echo "sweep durations in launch order (us):"
grep "icp_iter_batch_kernel" $ROOT/gpurun_out/frame_timeline_$tag.txt | awk '{printf "%s ", $7} END {print ""}'
