"""The REGRET of the wrappers' launch heuristics: over a grid of column shapes, the rate of the path each wrapper's typed launch() picks against the
best of the paths it could have been forced onto (the environment switches of run_anchors.hpp / the Delta and DZB-variable wrappers, which are read
once per process: every forced configuration is a child process decoding the whole grid).

    python tools/regret.py [SCHEME,...|all] [log2n,...=26,28]          the driver: prints chosen / best / regret per shape, and the worst
    python tools/regret.py --worker SCHEME LOG2N                       (a child: one line per shape — % of the 8 TB/s HBM peak)

Schemes: run_length_encoding, run_position_encoding_relative, run_position_encoding_absolute (segment lengths 64 ... 2^20 incl. 2052 / 4100 / 8200, 3000 / 10 000 and the whole tiles between 8192 and 65 536 x mean
runs 2, 4, 32, 200), delta (<4,8,2>: segment lengths 1024 ... 2^22 and one segment), discard_zero_bytes_variable (segment lengths 64 ... 2^20).
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))

# (REGRET_SEGMENTS / REGRET_MEANS / REGRET_PATHS in the environment: another grid, or only some of the forced paths — e.g. where a rule's thresholds lie)
RUN_SEGMENTS = tuple(int(x) for x in os.environ["REGRET_SEGMENTS"].split(",")) if "REGRET_SEGMENTS" in os.environ else (64, 1024, 2052, 3000, 4096, 4100, 8200, 10000, 12288, 16384, 24576, 32768, 65536, 1 << 20)
RUN_MEANS = tuple(int(x) for x in os.environ["REGRET_MEANS"].split(",")) if "REGRET_MEANS" in os.environ else (2, 4, 32, 200)
SHAPES = {
    "run_length_encoding": [(L, m) for L in RUN_SEGMENTS for m in RUN_MEANS],
    "run_position_encoding_relative": [(L, m) for L in RUN_SEGMENTS if L <= 65536 for m in RUN_MEANS],
    "run_position_encoding_absolute": [(L, m) for L in RUN_SEGMENTS for m in RUN_MEANS],
    "delta": [(L, 0) for L in (1024, 4096, 8192, 16384, 65536, 1 << 18, 1 << 20, 1 << 22, 0)],   # 0: the column is one segment
    "discard_zero_bytes_variable": [(L, 0) for L in (64, 1000, 1024, 4096, 16384, 65536, 1 << 20)],
}
RUN_CONFIGURATIONS = {
    "chosen": {},
    "segments as they are": {"GIDDY_RUN_FINER_ANCHORS_FROM": "0", "GIDDY_RUN_FUSED_PREPASS": "0"},
    "three-launch pre-pass": {"GIDDY_RUN_FINE_FORCE": "1", "GIDDY_RUN_FINE_ONE_LAUNCH": "0", "GIDDY_RUN_FUSED_PREPASS": "0"},
    "one-launch pre-pass": {"GIDDY_RUN_FINE_FORCE": "1", "GIDDY_RUN_FINE_ONE_LAUNCH": "2", "GIDDY_RUN_FUSED_PREPASS": "0"},
    "pre-pass in the decoder": {"GIDDY_RUN_FINE_FORCE": "1", "GIDDY_RUN_FINE_ONE_LAUNCH": "2", "GIDDY_RUN_FUSED_FROM": "1"},
    "groups of segments": {"GIDDY_RUN_FUSED_FROM": "1", "GIDDY_RUN_GROUP_ALL": "1"},
}
CONFIGURATIONS = {
    "run_length_encoding": RUN_CONFIGURATIONS, "run_position_encoding_relative": RUN_CONFIGURATIONS, "run_position_encoding_absolute": RUN_CONFIGURATIONS,
    "delta": {"chosen": {}, "chained tiles wherever they apply": {"GIDDY_DELTA_CHAINED_FROM": "1"}, "never chained tiles": {"GIDDY_DELTA_CHAINED_FROM": str(1 << 62)}},
    "discard_zero_bytes_variable": {"chosen": {}, "segments as they are": {"GIDDY_DZBV_FINE_ANCHORS": "0"}},
}


def worker(scheme, log2n):
    import torch
    import workloads as w
    device = torch.device("cuda", 0)
    stream = torch.cuda.current_stream().cuda_stream
    n = 1 << log2n
    for L, mean in SHAPES[scheme]:
        try:
            # THREE columns of different seeds decoded in turn: what a column's compressed arrays cost when they come from HBM, not from the Infinity Cache
            # (one column repeated flatters every path that reads its run arrays twice: the groups of segments at C4 are 86 % warm and 70 % cold)
            works = []
            for seed in range(0, 1000 * int(os.environ.get("REGRET_COLUMNS", "3")), 1000):   # (REGRET_COLUMNS: more of them, where three columns' run arrays still fit the Infinity Cache)
                w.SEED_OFFSET = seed
                if scheme == "run_length_encoding": work = w.RunLength(n, device, segment_length=L, mean_run=mean)
                elif scheme.startswith("run_position_encoding"): work = w.RunPosition(n, device, segment_length=L, mean_run=mean, relative=scheme.endswith("relative"))
                elif scheme == "delta": work = w.DeltaPatched(n, device, segment_length=L or n)
                else: work = w.DiscardZeroBytesVariable(n, device, segment_length=L)
                works.append(work)
            def run():
                for work in works: work.launch_fields(work.output, work.column_fields(), stream, work.config)
            for _ in range(2): run()
            torch.cuda.synchronize()
            times = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(2): run()
                b.record(); torch.cuda.synchronize()
                times.append(a.elapsed_time(b) / (2 * len(works)))
            ms = sorted(times)[len(times) // 2]
            print("SHAPE %d %d %.2f" % (L, mean, sum(work.algorithmic_bytes for work in works) / len(works) / ms / 1e6 / 80), flush=True)
            del works, work
        except Exception as problem:
            print("SHAPE %d %d nan  # %s" % (L, mean, str(problem).splitlines()[0][:100]), flush=True)
        torch.cuda.empty_cache()


def driver(schemes, sizes):
    worst = []
    for scheme in schemes:
        for log2n in sizes:
            results = {}
            wanted = os.environ["REGRET_PATHS"].split(",") if "REGRET_PATHS" in os.environ else list(CONFIGURATIONS[scheme])
            for label, switches in CONFIGURATIONS[scheme].items():
                if label not in wanted: continue
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", scheme, str(log2n)], env=dict(os.environ, **switches),
                                     capture_output=True, text=True, timeout=3000).stdout
                for line in out.splitlines():
                    if line.startswith("SHAPE "):
                        L, mean, pct = line.split()[1:4]
                        results.setdefault((int(L), int(mean)), {})[label] = float(pct)
            labels = [label for label in CONFIGURATIONS[scheme] if label in wanted]
            print("# %s, 2^%d outputs — %% of the HBM peak (algorithmic bytes; three columns of different seeds in turn: inputs from HBM) by path: %s" % (scheme, log2n, " | ".join(labels)))
            for (L, mean), by in sorted(results.items()):
                rates = [by.get(label, float("nan")) for label in labels]
                valid = [r for r in rates if r == r]
                best = max(valid) if valid else float("nan")
                chosen = by.get("chosen", float("nan"))
                regret = best - chosen
                best_label = labels[rates.index(best)] if valid else "-"
                print("%-32s 2^%d  L = %-8s mean run %-4s  %s   chosen %5.1f  best %5.1f (%s)  regret %4.1f" % (
                    scheme, log2n, L if L else "column", mean if mean else "-", " ".join("%5.1f" % r for r in rates), chosen, best, best_label, regret), flush=True)
                worst.append((regret, scheme, log2n, L, mean, best_label))
    worst.sort(reverse=True)
    print("# the largest regrets (points of the HBM peak between the path taken and the best forced one):")
    for regret, scheme, log2n, L, mean, label in worst[:15]:
        print("#   %4.1f  %s 2^%d L = %s mean run %s: '%s' would have been better" % (regret, scheme, log2n, L if L else "column", mean if mean else "-", label))
    over = [x for x in worst if x[0] > 3.0]
    print("# shapes with a regret above 3 points: %d of %d" % (len(over), len(worst)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]))
    else:
        names = [a for a in sys.argv[1:] if not a[0].isdigit()]
        schemes = list(SHAPES) if not names or names[0] == "all" else names[0].split(",")
        sizes = [int(x) for a in sys.argv[1:] if a[0].isdigit() for x in a.split(",")] or [26, 28]
        driver(schemes, sizes)
