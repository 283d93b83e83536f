BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("value", d["value"], "median", s["median"], "max", s["max"], "min", s["min"])'
for rep in 1 2 3; do for w in 4 2 1; do
L=$PWD/kinectpy_amd/libkinectpx.so; [ $w != 4 ] && L=$PWD/kinectpy_amd/libkinectpx_w$w.so
echo -n "waves/block=$w  "; KPX_LIBRARY=$L $BENCH 2>/dev/null | python -c "$P"
done; done
for w in 4 2 1; do L=$PWD/kinectpy_amd/libkinectpx.so; [ $w != 4 ] && L=$PWD/kinectpy_amd/libkinectpx_w$w.so
echo -n "waves/block=$w overlap 1 "; KPX_LIBRARY=$L $BENCH --overlap 1 2>/dev/null | python -c "$P"; done
