#!/bin/bash
# same-box A/B of one environment switch:  tools/exp_env_ab.sh NAME VALUE_A VALUE_B   (bench at 4 frames in flight and 1)
mkdir -p gpurun_out
name=$1; a=$2; b=$3
out=gpurun_out/exp_${name}.txt
: > $out
for rep in 1 2; do
for v in $a $b; do
  echo "== $name=$v" >> $out
  env $name=$v timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-targets --cpu-budget-s 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   bench value', d['value'], 'ms', d['ms_per_step'], 'spread', d.get('spread'))" >> $out || exit 1
  env $name=$v timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-targets --cpu-budget-s 1 --overlap 1 --spread-blocks 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   overlap 1: value', d['value'], 'ms', d['ms_per_step'])" >> $out || exit 1
done
done
cat $out
