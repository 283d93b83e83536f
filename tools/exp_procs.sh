# one process, one frame at a time / two frames in flight / two processes with one frame each / two processes with two frames each
B="python bench.py --steps 150 --warmup 10 --no-targets --cpu-budget-s 0 --spread-blocks 0"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
echo "== 1 process, overlap 1"; $B --overlap 1 2>/dev/null | python -c "$P"
echo "== 1 process, overlap 2"; $B --overlap 2 2>/dev/null | python -c "$P"
echo "== 2 processes, overlap 1 each"; ($B --overlap 1 2>/dev/null | python -c "$P") & ($B --overlap 1 2>/dev/null | python -c "$P"); wait
echo "== 2 processes, overlap 2 each"; ($B --overlap 2 2>/dev/null | python -c "$P") & ($B --overlap 2 2>/dev/null | python -c "$P"); wait
echo "== 3 processes, overlap 1 each"; ($B --overlap 1 2>/dev/null | python -c "$P") & ($B --overlap 1 2>/dev/null | python -c "$P") & ($B --overlap 1 2>/dev/null | python -c "$P"); wait
