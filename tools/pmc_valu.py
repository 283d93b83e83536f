"""Condense a `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace` pass into one row per kernel:
    python tools/pmc_valu.py OUT.csv DIR [name-filter]
valu_issue_frac = SQ_INSTS_VALU x 4 cycles (a 64-wide wave instruction on a 16-lane SIMD) / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the
share of the dispatch's SIMD-cycles in which a vector instruction could have been issuing -- the instruction roof of the kNN
selection kernels, which are neither HBM- nor MFMA-bound.  Counters come back summed over the XCDs / SIMDs (see pmc_mfma.py)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out, d = sys.argv[1], sys.argv[2]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace(",", ";")
        if flt and flt not in name:
            continue
        a = acc[name][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "SQ_INSTS_VALU_avg", "SQ_INSTS_LDS_avg", "SQ_WAVES_avg", "GRBM_GUI_ACTIVE_avg", "valu_issue_frac"])
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", [0, 0])[1]):
        v = c.get("SQ_INSTS_VALU", [0, 0.0])
        if v[0] == 0:
            continue
        n = v[0]
        lds = c.get("SQ_INSTS_LDS", [1, 0.0])[1] / n
        waves = c.get("SQ_WAVES", [1, 0.0])[1] / n
        gui = c.get("GRBM_GUI_ACTIVE", [1, 0.0])[1] / n
        w.writerow([k, n, round(v[1] / n, 1), round(lds, 1), round(waves, 1), round(gui, 1), round(v[1] / n * 4.0 / (gui / 8.0 * 1024), 4) if gui else ""])
