BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("value", d["value"], "median", s["median"], "max", s["max"], "min", s["min"])'
for rep in 1 2; do for ov in 1 2 3 4; do
echo -n "native  overlap $ov  "; $BENCH --overlap $ov 2>/dev/null | python -c "$P"
echo -n "python  overlap $ov  "; $BENCH --overlap $ov --python-step 2>/dev/null | python -c "$P"
done; done
