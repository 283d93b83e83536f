BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("value", d["value"], "median", s["median"], "max", s["max"], "min", s["min"])'
for rep in 1 2 3; do for cfg in "4 3" "8 4" "8 5" "4 6" "12 4" "12 6"; do set -- $cfg
echo -n "GPU_MAX_HW_QUEUES=$1 overlap $2  "; GPU_MAX_HW_QUEUES=$1 $BENCH --overlap $2 2>/dev/null | python -c "$P"
done; done
