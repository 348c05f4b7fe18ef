"""Summarises the passes of tools/pmc_compare.sh: per-wavefront means of every SQ counter for each variant, side by side."""
import csv, glob, os, sys

out, kernel = sys.argv[1], sys.argv[2]
variants = sys.argv[3:]
table = {}
for v in variants:
    for path in glob.glob(os.path.join(out, v, "p*", "**", "*counter_collection.csv"), recursive=True):
        sums, counts = {}, {}
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                c = r["Counter_Name"]
                sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"]); counts[c] = counts.get(c, 0) + 1
        for c in sums:
            table.setdefault(c, {})[v] = sums[c] / counts[c]
waves = table.get("SQ_WAVES", {})
print("%-34s" % "counter (per wavefront)" + "".join("%16s" % v for v in variants))
for c in sorted(table):
    row = "%-34s" % c
    for v in variants:
        x = table[c].get(v)
        row += "%16s" % ("-" if x is None else ("%.0f" % x if c == "SQ_WAVES" else "%.1f" % (x / waves.get(v, 1.0))))
    print(row)
