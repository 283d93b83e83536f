#!/bin/bash
# A/B of the one-launch-per-iteration ICP batch (KPX_ICP_FUSE=1, default) against the two-kernel iteration, on one box
for r in 1 2; do for f in 0 1; do for o in 1 2; do
  KPX_ICP_FUSE=$f timeout -k 10 200 python3 bench.py --steps 150 --warmup 10 --overlap $o --cpu-budget-s 0 2>/dev/null > /tmp/ab.json
  python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('fuse', $f, 'overlap', $o, d['value'], d['ms_per_step'])"
done; done; done
