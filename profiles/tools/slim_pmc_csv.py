#!/usr/bin/env python
"""Keep the columns of a rocprofv3 counter_collection.csv that the summaries use (kernel, counter, value, grid) and only the
hot-path kernels: the raw file of a cfg3 / cfg5 pass is tens of megabytes.  python slim_pmc_csv.py in.csv out.csv"""
import csv
import sys

KEEP = ("Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size")
with open(sys.argv[1]) as src, open(sys.argv[2], "w", newline="") as dst:
    rd = csv.DictReader(src)
    cols = [c for c in KEEP if c in rd.fieldnames]
    wr = csv.DictWriter(dst, cols)
    wr.writeheader()
    for row in rd:
        if "al::k_" in row["Kernel_Name"]:
            wr.writerow({c: row[c] for c in cols})
