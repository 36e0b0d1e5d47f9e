#!/bin/bash
# usage: tools/run_variants.sh <probe.py> <variant .so files...>; prints the top kernel lines of rocprofv3 per variant
export TMPDIR=/tmp
probe=$1; shift
for so in "$@"; do
  name=$(basename $so .so)
  rm -rf gpurun_out/var_$name
  MRPDR_LIB=$so timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/var_$name -- python3 $probe > gpurun_out/var_$name.txt 2>&1; rc=$?; if [ $rc -gt 1 ]; then exit $rc; fi
  echo "== $name: $(grep -h encode gpurun_out/var_$name.txt | tail -1)"
  f=$(ls gpurun_out/var_$name/*/*kernel_stats.csv)
  python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].split("(")[0].replace("void ","").startswith(("k_tile","k_size","k_scan","k_copy","k_walk","k_count")) and int(r["Calls"])>10:
        print("   %-34s calls %4s avg %8.1f us" % (r["Name"].split("(")[0][:34], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
