#!/bin/bash
# effect of the number of hardware queues HIP multiplexes its streams onto (GPU_MAX_HW_QUEUES, default 4)
for r in 1 2; do for q in 4 8 16; do for o in 1 2; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 200 python3 bench.py --steps 150 --warmup 10 --overlap $o --cpu-budget-s 0 2>/dev/null > /tmp/ab.json
  python3 -c "import json; d=json.load(open('/tmp/ab.json')); print('queues', $q, 'overlap', $o, d['value'], d['ms_per_step'])"
done; done; done
