#!/bin/bash
# Same-box A/B runs (boxes differ by +-15 %: only comparisons inside one call mean anything).  One script for every experiment kept
# under profiles/rNN/exp_*.txt:
#
#   tools/ab.sh env  NAME VALUE_A VALUE_B [...]     one environment switch (KPX_ICP_LIGHT_SKIP 1 0, KPX_ICP_SPLIT 2 1 0, KPX_RADIX 1 0,
#                                                   KPX_FRAME_ZORDER 1 0, GPU_MAX_HW_QUEUES 4 8, ...): bench at 4 frames in flight and at 1
#   tools/ab.sh lib  cur NAME [...]                 library variants built by tools/build_variant.sh: ICP probe + bench line
#   tools/ab.sh flag "--overlap 2" "--overlap 4" [...]   bench.py argument sets (frames in flight, --python-step, ...)
#   tools/ab.sh procs                               1 process x 1|2 frames against 2 and 3 processes (GIL / runtime locks vs the device)
# Output: gpurun_out/exp_<mode>_<first argument>.txt (copy what you keep into profiles/rNN/).
mkdir -p gpurun_out
mode=$1; shift
tag=$(echo "$1" | tr -c 'A-Za-z0-9_\n' '_')
out=gpurun_out/exp_${mode}_${tag}.txt
: > $out
B="python bench.py --no-targets --cpu-budget-s 0"
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("   value", d["value"], "ms", d["ms_per_step"], "pinned", (d.get("from_pinned_host") or {}).get("value"), "spread", d.get("spread"))'
bench4() { timeout -k 10 300 $B --steps 300 --warmup 20 "$@" 2>/dev/null | python -c "$line" >> $out || exit 1; }
bench1() { timeout -k 10 300 $B --steps 200 --warmup 20 --overlap 1 --spread-blocks 0 "$@" 2>/dev/null | python -c "$line" >> $out || exit 1; }
case $mode in
env)
  name=$1; shift
  for rep in 1 2; do for v in "$@"; do
    echo "== $name=$v (4 frames in flight, then 1)" >> $out
    export $name=$v; bench4; bench1; unset $name
  done; done ;;
lib)
  for v in "$@"; do
    lib=$PWD/kinectpy_amd/libkinectpx_$v.so; [ "$v" = "cur" ] && lib=$PWD/kinectpy_amd/libkinectpx.so
    echo "== $v" >> $out
    KPX_LIBRARY=$lib timeout -k 10 200 python tools/icp_probe.py 20 --waves 2>&1 | grep -E "wall|waves [0-9]|model" >> $out || exit 1
    export KPX_LIBRARY=$lib; for rep in 1 2; do bench4; done; unset KPX_LIBRARY
  done ;;
flag)
  for rep in 1 2; do for f in "$@"; do
    echo "== bench.py $f" >> $out
    timeout -k 10 300 $B --steps 300 --warmup 20 $f 2>/dev/null | python -c "$line" >> $out || exit 1
  done; done ;;
procs)
  Q="--steps 150 --warmup 10 --spread-blocks 0"
  one() { timeout -k 10 300 $B $Q "$@" 2>/dev/null | python -c "$line"; }
  echo "== 1 process, 1 frame in flight" >> $out; one --overlap 1 >> $out
  echo "== 1 process, 2 frames in flight" >> $out; one --overlap 2 >> $out
  echo "== 2 processes, 1 frame each" >> $out; (one --overlap 1 >> $out) & (one --overlap 1 >> $out); wait
  echo "== 2 processes, 2 frames each" >> $out; (one --overlap 2 >> $out) & (one --overlap 2 >> $out); wait
  echo "== 3 processes, 1 frame each" >> $out; (one --overlap 1 >> $out) & (one --overlap 1 >> $out) & (one --overlap 1 >> $out); wait ;;
*) echo "usage: tools/ab.sh env|lib|flag|procs ..."; exit 2 ;;
esac
cat $out
