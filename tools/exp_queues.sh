python tools/icp_probe.py 20 > gpurun_out/probe4.log 2>&1; grep -A0 "phases" gpurun_out/probe4.log | head -2
for q in 4 8 16; do for ov in 2 3 4; do
echo "== GPU_MAX_HW_QUEUES=$q overlap=$ov"; GPU_MAX_HW_QUEUES=$q python bench.py --steps 100 --warmup 10 --overlap $ov --no-targets --cpu-budget-s 0 --spread-blocks 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['spread'], d['config']['priming_steps'])"
done; done
