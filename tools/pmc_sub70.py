"""Summarises tools/pmc_sub70.sh: per-wavefront means of the SQ counters of each profiled instantiation, side by side (only the decoder's own dispatches)."""
import csv, glob, os, sys
out = sys.argv[1]
cases = [tuple(line.split("|")[i].strip() for i in (0, 2)) for line in open(sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_sub70_cases.txt"))
         if line.strip() and not line.startswith("#")]
table = {}
for name, kernel in cases:
    for path in glob.glob(os.path.join(out, name, "p*", "**", "*counter_collection.csv"), recursive=True):
        sums, counts = {}, {}
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                c = r["Counter_Name"]
                sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"]); counts[c] = counts.get(c, 0) + 1
        for c in sums:
            table.setdefault(c, {})[name] = sums[c] / counts[c]
waves = table.get("SQ_WAVES", {})
print("%-30s" % "counter (per wavefront)" + "".join("%21s" % n for n, _ in cases))
for c in sorted(table):
    row = "%-30s" % c
    for n, _ in cases:
        x = table[c].get(n)
        row += "%21s" % ("-" if x is None else ("%.0f" % x if c == "SQ_WAVES" else "%.1f" % (x / waves.get(n, 1.0))))
    print(row)
