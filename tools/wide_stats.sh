#!/bin/bash
# usage: tools/wide_stats.sh <kind> <n>: per-kernel times of the device-wide flat float Sum under rocprofv3
export TMPDIR=/tmp
d=gpurun_out/wide_$1_$2
rm -rf $d
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/probe_flat1.py $1 $2 > $d.txt 2>&1 || exit $?
python3 - $(ls $d/*/*kernel_stats.csv) <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_wide" in r["Name"]:
        print("   %-40s calls %4s avg %8.1f us" % (r["Name"].split("(")[0].replace("void ","")[:40], r["Calls"], float(r["AverageNs"])/1e3))
PY
