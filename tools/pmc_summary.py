"""Condense rocprofv3 counter passes into profiles/rNN/pmc_summary.csv.

    python tools/pmc_summary.py OUT.csv FETCH_DIR WRITE_DIR

FETCH_DIR / WRITE_DIR hold the `*_counter_collection.csv` of two separate runs
(`rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and `--pmc WRITE_SIZE --kernel-trace ...`).  Values stay raw (KB per
dispatch as the counter reports them); the gfx950 corrections of MI355X_MICROARCH.md are applied by the reader.
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)                    # drop the argument list
    return name.replace(",", ";")


def collect(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                a = acc[row["Counter_Name"]][short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    rows = []
    for d in dirs:
        for counter, kernels in collect(d).items():
            top = sorted(kernels.items(), key=lambda kv: -kv[1][1])[:14]
            rows += [(counter, k, n, round(tot / n, 2), round(tot, 1)) for k, (n, tot) in top]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["counter", "kernel", "dispatches", "avg_KB_per_dispatch_raw", "total_KB_raw"])
        w.writerows(rows)


if __name__ == "__main__":
    main()
