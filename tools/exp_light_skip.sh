#!/bin/bash
# same-box A/B: LightSkip on / off (KPX_ICP_LIGHT_SKIP), four frames in flight and one at a time
mkdir -p gpurun_out
out=gpurun_out/exp_light_skip.txt
: > $out
for rep in 1 2; do
for m in 1 0; do
  echo "== KPX_ICP_LIGHT_SKIP=$m" >> $out
  KPX_ICP_LIGHT_SKIP=$m timeout -k 10 200 python tools/icp_probe.py 20 --noprof 2>&1 | grep -E "wall" >> $out || exit 1
  KPX_ICP_LIGHT_SKIP=$m timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-targets --cpu-budget-s 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   bench value', d['value'], 'ms', d['ms_per_step'], 'spread', d.get('spread'))" >> $out || exit 1
  KPX_ICP_LIGHT_SKIP=$m timeout -k 10 300 python bench.py --steps 200 --warmup 20 --no-targets --cpu-budget-s 1 --overlap 1 --spread-blocks 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   overlap 1: value', d['value'], 'ms', d['ms_per_step'])" >> $out || exit 1
done
done
cat $out
