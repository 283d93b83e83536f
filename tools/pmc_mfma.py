"""Condense a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace` pass into one row per MFMA kernel:
    python tools/pmc_mfma.py OUT.csv DIR
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the share of SIMD-cycles of the dispatch in which the
matrix pipe was busy.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (feature_nn_kernel: 22.3M for a 1.2 ms dispatch = 8 x 2.8M
cycles), SQ_VALU_MFMA_BUSY_CYCLES summed over all SIMDs; raw counters are kept beside the ratio (ROCm 7.2 has no gfx950
derived-counter definitions, MI355X_MICROARCH.md)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out, d = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace(",", ";")
        a = acc[name][row["Counter_Name"]]
        a[0] += 1
        a[1] += float(row["Counter_Value"])
with open(out, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "dispatches", "SQ_VALU_MFMA_BUSY_CYCLES_avg", "SQ_BUSY_CYCLES_avg", "GRBM_GUI_ACTIVE_avg", "mfma_busy_frac"])
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0])[1]):
        m = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])
        if m[1] <= 0:
            continue
        n = m[0]
        busy = c.get("SQ_BUSY_CYCLES", [1, 0.0])[1] / n
        gui = c.get("GRBM_GUI_ACTIVE", [1, 0.0])[1] / n
        w.writerow([k, n, round(m[1] / n, 1), round(busy, 1), round(gui, 1), round(m[1] / n / (gui / 8.0 * 1024), 4) if gui else ""])
