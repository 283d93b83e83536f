# A/B of an environment switch inside ONE box: bash tools/exp_ab.sh VAR A B [bench args]; prints the median / min over 8 blocks of 20 steps
VAR=$1; A=$2; B=$3; shift 3
BENCH="python bench.py --steps 50 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 8 $@"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d["spread"]; print("median", s["median"], "max", s["max"], "min", s["min"])'
for rep in 1 2 3; do for v in $A $B; do echo -n "$VAR=$v  "; env $VAR=$v $BENCH 2>/dev/null | python -c "$P"; done; done
