BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("value", d["value"], "median", s["median"], "max", s["max"], "min", s["min"])'
for q in 4 8 16; do for ov in 3 4 6; do
echo -n "GPU_MAX_HW_QUEUES=$q overlap $ov  "; GPU_MAX_HW_QUEUES=$q $BENCH --overlap $ov 2>/dev/null | python -c "$P"
done; done
