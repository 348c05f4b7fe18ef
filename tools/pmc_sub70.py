"""Summarises tools/pmc_sub70.sh: per-wavefront means of the SQ counters of each profiled instantiation, side by side (only the decoder's own dispatches)."""
import csv, glob, os, sys
out = sys.argv[1]
cases = [("rle_8_1_8_8", "run_length_encoding::decompress<8u, 1u, 8u, 8u>"), ("rle_4_1_1_4", "run_length_encoding::decompress<4u, 1u, 1u, 4u>"),
         ("ib_4_1_2", "incidence_bitmaps::decompress<4u, 1u, 2u, true>"), ("ib_4_4_2", "incidence_bitmaps::decompress<4u, 4u, 2u, true>"),
         ("dzbv_4_2_8", "variable_width::decompress<4u, 2u, 8u>"), ("dzbv_4_4_8", "variable_width::decompress<4u, 4u, 8u>"),
         ("model_patched_short", "model::patched::decompress<4u, short"), ("model_patched_int", "model::patched::decompress<4u, int")]
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
