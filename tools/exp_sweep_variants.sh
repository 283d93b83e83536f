#!/bin/bash
# same-box A/B of libkinectpx variants (tools/build_variant.sh): ICP probe walls + the bench line's value
mkdir -p gpurun_out
out=gpurun_out/exp_sweep_variants.txt
: > $out
for v in "$@"; do
  lib=$PWD/kinectpy_amd/libkinectpx_$v.so
  [ "$v" = "cur" ] && lib=$PWD/kinectpy_amd/libkinectpx.so
  echo "== $v" >> $out
  KPX_LIBRARY=$lib timeout -k 10 200 python tools/icp_probe.py 20 --waves 2>&1 | grep -E "wall|waves [0-9]|model" >> $out || exit 1
  for rep in 1 2; do
    KPX_LIBRARY=$lib timeout -k 10 300 python bench.py --steps 300 --warmup 20 --no-targets --cpu-budget-s 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('   bench value', d['value'], 'ms', d['ms_per_step'], 'spread', d.get('spread'))" >> $out || exit 1
  done
done
cat $out
