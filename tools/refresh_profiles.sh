#!/bin/bash
# Re-creates the files kept under profiles/rNN on a 1-GPU MI355X box:  bash tools/refresh_profiles.sh gpurun_out/r02
# (run from the repository root; copy the results into profiles/rNN afterwards)
set -e -o pipefail
OUT=$(realpath -m "${1:-gpurun_out/prof}")
ROOT=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
Q="--no-targets --cpu-budget-s 0 --spread-blocks 0"
timeout -k 10 500 python3 bench.py --workload reference-stream > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
KPX_NN_ENGINE=dense timeout -k 10 300 python3 bench.py $Q > "$OUT/bench_n1_dense_engine.json" 2>> "$OUT/bench_n1.err"
timeout -k 10 300 python3 bench.py $Q --overlap 1 > "$OUT/bench_n1_overlap1.json" 2>> "$OUT/bench_n1.err"
timeout -k 10 300 python3 tools/bench_kernels.py > "$OUT/kernels.json" 2>> "$OUT/bench_n1.err"
KPX_ICP_CHAIN=0 timeout -k 10 120 python3 tools/icp_probe.py 20 --waves > "$OUT/icp_probe.txt" 2>> "$OUT/bench_n1.err"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --steps 40 --warmup 5 $Q > "$OUT/bench_n1_under_rocprof.json" 2>> "$OUT/bench_n1.err"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o b -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 $Q --overlap 1 > /dev/null 2>> "$OUT/bench_n1.err"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o b -- python3 "$ROOT/bench.py" --steps 5 --warmup 1 $Q --overlap 1 > /dev/null 2>> "$OUT/bench_n1.err"
# the dominant kernel with ONE registration per launch (what bench.py's quiet timing measures): traffic per launch
# (KPX_ICP_CHAIN=0: the launch-per-iteration form, the one the roofline row is about; a single registration would otherwise run as one chain)
KPX_ICP_CHAIN=0 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch1" -o b -- python3 "$ROOT/tools/icp_probe.py" 3 --noprof --single > /dev/null 2>> "$OUT/bench_n1.err"
KPX_ICP_CHAIN=0 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write1" -o b -- python3 "$ROOT/tools/icp_probe.py" 3 --noprof --single > /dev/null 2>> "$OUT/bench_n1.err"
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -o m -- python3 "$ROOT/tools/mfma_probe.py" > /dev/null 2>> "$OUT/bench_n1.err"
cd "$ROOT"
python3 tools/pmc_mfma.py "$OUT/pmc_mfma.csv" "$OUT/pmc_mfma"
python3 tools/pmc_summary.py "$OUT/pmc_summary.csv" "$OUT/pmc_fetch" "$OUT/pmc_write"
python3 tools/pmc_summary.py "$OUT/pmc_icp_single.csv" "$OUT/pmc_fetch1" "$OUT/pmc_write1"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/bench_n1_kernel_stats.csv" \;
find "$OUT/stats" -name "*domain_stats.csv" -exec cp {} "$OUT/bench_n1_domain_stats.csv" \;
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_mfma" "$OUT/pmc_fetch1" "$OUT/pmc_write1"
cd "$ROOT"
bash tools/hbm_ops_trace.sh > /dev/null 2>> "$OUT/bench_n1.err" && cp gpurun_out/hbm_ops_kernels.txt "$OUT/hbm_ops_kernels.txt"
bash tools/frame_timeline.sh overlap1 > /dev/null 2>> "$OUT/bench_n1.err" && cp gpurun_out/frame_timeline_overlap1.txt "$OUT/frame_timeline_overlap1.txt"
timeout -k 10 120 python3 tools/nn_dense_probe.py > "$OUT/nn_dense_probe.txt" 2>> "$OUT/bench_n1.err"
timeout -k 10 120 python3 tools/voxel_probe.py > "$OUT/voxel_probe.txt" 2>> "$OUT/bench_n1.err"
ls -la "$OUT"
