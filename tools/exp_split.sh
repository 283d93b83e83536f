BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("value", d["value"], "median", s["median"], "max", s["max"], "min", s["min"])'
for rep in 1 2 3; do for v in 0 1; do
echo -n "KPX_ICP_SPLIT=$v  "; KPX_ICP_SPLIT=$v $BENCH 2>/dev/null | python -c "$P"
done; done
echo -n "KPX_ICP_SPLIT=0 overlap 1 "; KPX_ICP_SPLIT=0 $BENCH --overlap 1 2>/dev/null | python -c "$P"
echo -n "KPX_ICP_SPLIT=1 overlap 1 "; KPX_ICP_SPLIT=1 $BENCH --overlap 1 2>/dev/null | python -c "$P"
