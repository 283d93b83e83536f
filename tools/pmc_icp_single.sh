#!/bin/bash
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out/r04c; mkdir -p $OUT
cd /tmp
KPX_ICP_CHAIN=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch1" -o b -- python3 "$ROOT/tools/icp_probe.py" 3 --noprof --single > /dev/null 2>> "$OUT/err.txt"
KPX_ICP_CHAIN=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write1" -o b -- python3 "$ROOT/tools/icp_probe.py" 3 --noprof --single > /dev/null 2>> "$OUT/err.txt"
cd $ROOT
python3 tools/pmc_summary.py "$OUT/pmc_icp_single.csv" "$OUT/pmc_fetch1" "$OUT/pmc_write1"
rm -rf "$OUT/pmc_fetch1" "$OUT/pmc_write1"
grep icp_iter "$OUT/pmc_icp_single.csv"
