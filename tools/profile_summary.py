#!/usr/bin/env python3
"""Turns rocprofv3 output directories into the summaries kept under profiles/.

  python3 tools/profile_summary.py --tag r01 --stats gpurun_out/r01_stats \
          --rd gpurun_out/r01_rd --wr gpurun_out/r01_wr

--stats : directory of `rocprofv3 --kernel-trace --stats -- python3 bench.py ...`
--rd/--wr : directories of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs.
rocprofv3 reports both counters in KB per dispatch.  On gfx950 FETCH_SIZE counts half of a wide
coalesced read stream, so HBM read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact.
"""
import argparse
import csv
import glob
import json
import os
import shutil
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    return name.split("(")[0].strip()


def pmc_rows(d, counter):
    rows = OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter or not short(r["Kernel_Name"]).replace("void ", "").startswith("k_"):
                continue
            rows.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats")
    ap.add_argument("--rd")
    ap.add_argument("--wr")
    ap.add_argument("--cmd", default="python3 bench.py --steps 5 --warmup 2 --no-cpu --no-extras")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if a.stats:
        f = sorted(glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True))
        shutil.copy(f[-1], os.path.join(out, "%s_bench_c3_kernel_stats.csv" % a.tag))
    if a.rd and a.wr:
        rd, wr = pmc_rows(a.rd, "FETCH_SIZE"), pmc_rows(a.wr, "WRITE_SIZE")
        lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs), " + a.cmd,
                 "values are KB per dispatch as rocprofv3 reports them; on gfx950 FETCH_SIZE counts half of a "
                 "wide coalesced read stream, so HBM read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact",
                 "", "%-12s %-48s %5s %14s %14s %14s" % ("counter", "kernel", "n", "avg_KB", "min_KB", "max_KB")]
        for cname, rows in (("FETCH_SIZE", rd), ("WRITE_SIZE", wr)):
            for k, v in rows.items():
                lines.append("%-12s %-48s %5d %14.1f %14.1f %14.1f" % (cname, k, len(v), sum(v) / len(v), min(v), max(v)))
        open(os.path.join(out, "%s_bench_c3_pmc_summary.txt" % a.tag), "w").write("\n".join(lines) + "\n")

        def per_launch(key):
            # the full-size launches only (warm-up of other shapes would be smaller)
            r = [x for x in rd[key] if x > 0.5 * max(rd[key])]
            w = [x for x in wr[key] if x > 0.5 * max(wr[key])]
            return int(2 * 1024 * sum(r) / len(r)), int(1024 * sum(w) / len(w))
        dec = [k for k in rd if "k_copy<4, true, false" in k][0]
        enc = [k for k in rd if "k_copy<4, false, true" in k][0]
        dr, dw = per_launch(dec)
        er, ew = per_launch(enc)
        json.dump({"_how": "profiles/%s_bench_c3_pmc_summary.txt: per launch, HBM bytes = 2 x FETCH_SIZE x 1024 "
                           "+ WRITE_SIZE x 1024 (gfx950 FETCH_SIZE correction)" % a.tag,
                   "k_copy_decode_swap4_hbm_bytes_per_launch": dr + dw,
                   "k_copy_decode_swap4_read": dr, "k_copy_decode_swap4_write": dw,
                   "k_copy_encode4_hbm_bytes_per_launch": er + ew,
                   "k_copy_encode4_read": er, "k_copy_encode4_write": ew,
                   "algorithmic_bytes_per_launch": 2 * 4096 * 65536 * 4},
                  open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
