B="python bench.py --steps 150 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 3"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["spread"]["median"])'
for ov in 2 3 4; do for si in 0 0.0001; do echo "== overlap $ov switch-interval $si"; $B --overlap $ov --switch-interval $si 2>/dev/null | python -c "$P"; done; done
echo "== KPX_ICP_BATCH_LAUNCH=0 overlap 2"; KPX_ICP_BATCH_LAUNCH=0 $B --overlap 2 2>/dev/null | python -c "$P"
echo "== KPX_ICP_BATCH_LAUNCH=0 overlap 3"; KPX_ICP_BATCH_LAUNCH=0 $B --overlap 3 2>/dev/null | python -c "$P"
