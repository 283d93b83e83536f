#!/bin/bash
# The filter / extract / selection probes kept under profiles/rNN (round 5), separate from refresh_profiles.sh so that each fits one
# gpurun call:  bash tools/refresh_profiles_filters.sh gpurun_out/r05
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof}")
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
touch "$OUT/bench_n1.err"
# round 5: the filter at the bench rows' sizes (block / wave forms against the wave-per-query passes alone, keep lists compared), the
# batch extract rows and the selections with their A/B switches
timeout -k 10 200 python3 tools/sor_probe.py 2>> "$OUT/bench_n1.err" | grep -v "^ *$" > "$OUT/sor_probe.txt"
( timeout -k 10 100 python3 tools/bench_extract_masked.py; KPX_MEDIAN_FRAME=0 KPX_ONEPASS_BATCH=0 timeout -k 10 100 python3 tools/bench_extract_masked.py ) 2>> "$OUT/bench_n1.err" | grep -v amdgpu > "$OUT/extract_batch_forms.txt"
cd /tmp
for kk in 20 200; do
    timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sor$kk" -o s -- python3 "$ROOT/tools/sor_one.py" $kk > /dev/null 2>> "$OUT/bench_n1.err"
    python3 "$ROOT/tools/pmc_valu.py" "$OUT/pmc_sor_k$kk.csv" "$OUT/pmc_sor$kk" sor_
    rm -rf "$OUT/pmc_sor$kk"
done
cd "$ROOT"
bash tools/ktrace.sh sor_k200 tools/sor_one.py 200 > /dev/null 2>> "$OUT/bench_n1.err" && cp gpurun_out/sor_k200_kernels.txt "$OUT/sor_k200_kernels.txt"
bash tools/ktrace.sh sor_k20 tools/sor_one.py 20 > /dev/null 2>> "$OUT/bench_n1.err" && cp gpurun_out/sor_k20_kernels.txt "$OUT/sor_k20_kernels.txt"
timeout -k 10 200 python3 tools/bench_select.py 2>> "$OUT/bench_n1.err" | grep -v amdgpu > "$OUT/selections.txt"
ls -la "$OUT"
