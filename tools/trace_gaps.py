"""Reads a rocprofv3 kernel trace (csv) of tools/probe_c2_graph.py and prints, for the eager phase and
the graph-replay phase, the average duration of every kernel of a wave and the average gap in front
of it (end of the kernel before -> its start)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# phase markers: the three fill kernels with grid sizes that nothing else uses are the last three 'fill' launches
fills = [i for i, r in enumerate(rows) if "FillFunctor<int>" in r["Kernel_Name"]]
a, b, c = fills[-3], fills[-2], fills[-1]
for name, lo, hi in (("eager", a + 1, b), ("graph replay", b + 1, c)):
    ph = rows[lo:hi]
    per_kernel = {}
    order = []
    prev_end = None
    for r in ph:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if k not in per_kernel:
            per_kernel[k] = [0, 0.0, 0.0]
            order.append(k)
        per_kernel[k][0] += 1
        per_kernel[k][1] += e - s
        if prev_end is not None:
            per_kernel[k][2] += s - prev_end
        prev_end = e
    total = (int(ph[-1]["End_Timestamp"]) - int(ph[0]["Start_Timestamp"])) / 1e3
    waves = per_kernel[order[0]][0]
    print("%s: %d kernels, %d waves, %.2f us per wave on the device's clock" % (name, len(ph), waves, total / waves))
    for k in order:
        cnt, dur, gap = per_kernel[k]
        print("   %-60s x%-4d  runs %6.2f us   gap in front %6.2f us" % (k, cnt, dur / cnt / 1e3, gap / cnt / 1e3))
