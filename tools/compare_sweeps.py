"""python tools/compare_sweeps.py FILE A B — per instantiation, the % of peak of variants A and B in the output of tools/sweep_variants.sh (development aid)"""
import collections, re, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    m = re.match(r'(\S+) (\S+?)::kernel_t<(.*)>\s+([\d.]+) ms\s+([\d.]+) GB/s\s+([\d.]+)%', l)
    if m:
        d[m.group(2).replace("decompression::", "") + '<' + m.group(3)[:70] + '>'][m.group(1)].append(float(m.group(6)))
a, b = sys.argv[2], sys.argv[3]
for k, v in d.items():
    if v[a] and v[b]:
        print('%-95s %s %s   %s %s   %+.1f' % (k, a, v[a], b, v[b], sum(v[b]) / len(v[b]) - sum(v[a]) / len(v[a])))
