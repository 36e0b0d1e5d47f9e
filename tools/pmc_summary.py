#!/usr/bin/env python3
"""Median per launch of every counter in the `*counter_collection.csv` files below the given
rocprofv3 output directories, for the library's own kernels (names starting with k_):

    python3 tools/pmc_summary.py gpurun_out/r02_pmc_rd gpurun_out/r02_pmc_wr ... > profiles/<name>.txt
"""
import csv
import glob
import os
import statistics
import sys
from collections import OrderedDict

rows = OrderedDict()
for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if name.startswith("k_"):
                rows.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (name, counter), vals in rows.items():
    print("%-28s %-36s n=%3d median=%14.0f" % (name, counter, len(vals), statistics.median(vals)))
