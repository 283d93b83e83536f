#!/bin/bash
# instruction counters of the ICP chain kernel (one registration, tools/icp_chain_one.py): per wave and iteration
#   tools/pmc_chain.sh   ->  gpurun_out/pmc_chain.txt
export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out
cd /tmp && rm -rf /tmp/pc1 /tmp/pc2
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace --output-format csv -d /tmp/pc1 -o a -- python3 "$ROOT/tools/icp_chain_one.py" > /tmp/pc1.log 2>&1 || { tail -5 /tmp/pc1.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F64 --kernel-trace --output-format csv -d /tmp/pc2 -o a -- python3 "$ROOT/tools/icp_chain_one.py" > /tmp/pc2.log 2>&1 || { tail -5 /tmp/pc2.log; exit 1; }
python3 - > $ROOT/gpurun_out/pmc_chain.txt <<'PY'
import csv, glob, collections
for d in ("/tmp/pc1", "/tmp/pc2"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"][:40]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(name, r["Counter_Name"])] += 1
    for name, c in acc.items():
        if "icp_chain" in name or "icp_iter_batch" in name:
            for k, v in c.items():
                print(f"{name:42s} {k:32s} total {v:14.0f} per dispatch {v / calls[(name, k)]:14.0f}")
PY
cat $ROOT/gpurun_out/pmc_chain.txt
