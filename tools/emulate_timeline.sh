#!/bin/bash
# tools/emulate_timeline.sh NAME RANK [frames in flight]: rank RANK of the 8-GPU sensor partition alone on this GPU (bench.py
# --emulate-world 8, peers replayed) under rocprofv3 --kernel-trace -> gpurun_out/NAME_overlap.txt (tools/overlap_timeline.py)
export TMPDIR=/tmp
ROOT=$PWD
name=$1; rank=${2:-1}; depth=${3:-4}
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/kt_e
KPX_EMULATE_ONLY_RANKS=1 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_e -o t -- python3 "$ROOT/bench.py" --emulate-world 8 --sensors 8 --steps 300 --emulate-ranks $rank --overlap $depth --fused-filter ${4:-sharded} > /tmp/kt_e.log 2>&1 || { tail -5 /tmp/kt_e.log; exit 1; }
python3 $ROOT/tools/overlap_timeline.py /tmp/kt_e 30 list > $ROOT/gpurun_out/${name}_overlap.txt
tail -1 /tmp/kt_e.log | cut -c1-600
head -40 $ROOT/gpurun_out/${name}_overlap.txt
