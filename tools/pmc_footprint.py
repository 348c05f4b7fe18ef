"""Summarises tools/pmc_footprint.sh: per-launch means of every counter by column size, and the ratios that matter."""
import csv, glob, os, sys

out, kernel = sys.argv[1], sys.argv[2]
sizes = sys.argv[3:]
table = {}
for s in sizes:
    for path in glob.glob(os.path.join(out, s, "p*", "**", "*counter_collection.csv"), recursive=True):
        sums, counts = {}, {}
        for r in csv.DictReader(open(path)):
            if kernel in r["Kernel_Name"]:
                c = r["Counter_Name"]
                sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"]); counts[c] = counts.get(c, 0) + 1
        for c in sums:
            table.setdefault(c, {})[s] = sums[c] / counts[c]
print("%-36s" % "counter (per launch)" + "".join("%18s" % ("2^" + s) for s in sizes))
for c in sorted(table):
    print("%-36s" % c + "".join("%18s" % ("-" if table[c].get(s) is None else "%.4g" % table[c][s]) for s in sizes))


def ratio(name, a, b, scale=1.0):
    if a in table and b in table:
        print("%-36s" % name + "".join("%18s" % ("-" if not table[b].get(s) else "%.4g" % (scale * table[a].get(s, 0.0) / table[b][s])) for s in sizes))


ratio("UTCL1 miss / request", "TCP_UTCL1_TRANSLATION_MISS", "TCP_UTCL1_REQUEST")
ratio("UTCL2 busy / GUI active", "GRBM_UTCL2_BUSY", "GRBM_GUI_ACTIVE")
ratio("EA read latency (cycles)", "TCC_EA0_RDREQ_LEVEL", "TCC_EA0_RDREQ")
ratio("EA write latency (cycles)", "TCC_EA0_WRREQ_LEVEL", "TCC_EA0_WRREQ")
ratio("TCP->TCC read latency (cycles)", "TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_READ_REQ")
ratio("TCP->TCC write latency (cycles)", "TCP_TCC_WRITE_REQ_LATENCY", "TCP_TCC_WRITE_REQ")
ratio("wave cycles / wave (quad-cycles)", "SQ_WAVE_CYCLES", "SQ_WAVES")
ratio("waiting share of wave cycles", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")
