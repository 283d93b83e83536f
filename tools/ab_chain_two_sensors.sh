#!/bin/bash
B="python bench.py --no-targets --cpu-budget-s 0 --no-stages --sensors 2"
line='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("   value", d["value"], "ms", d["ms_per_step"], "pinned", (d.get("from_pinned_host") or {}).get("value"), "spread", (d.get("spread") or {}).get("median"))'
for rep in 1 2; do for v in 1 0; do
  echo "== two-sensor rig (one registration of ~31k rows per frame), KPX_ICP_CHAIN=$v: four frames in flight, then one"
  KPX_ICP_CHAIN=$v timeout -k 10 300 $B --steps 300 --warmup 20 2>/dev/null | python -c "$line"
  KPX_ICP_CHAIN=$v timeout -k 10 300 $B --steps 200 --warmup 20 --overlap 1 --spread-blocks 0 2>/dev/null | python -c "$line"
done; done
