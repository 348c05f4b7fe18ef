"""Turns the raw rocprofv3 outputs of tools/profile_round.sh (gpurun_out/prof/) into the committed
summaries under profiles/:  <round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the default
bench command, our kernels only), <round>_pmc_hbm.md / hbm_traffic.json (HBM bytes per launch from
separate --pmc FETCH_SIZE / WRITE_SIZE passes, corrected with the calibration kernels as
MI355X_MICROARCH.md §HBM prescribes: units are KiB; FETCH_SIZE counts half of a streaming read on gfx950)."""
import csv
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_sources

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "prof")
OUT = os.path.join(ROOT, "profiles")
ROUND = sys.argv[1] if len(sys.argv) > 1 else "r01"

KERNELS = {  # workload -> (substring of the kernel name, algorithmic bytes per launch at 2^30 (2^28 for DZB-V))
    "dictionary": "dictionary::decompress",
    "frame_of_reference": "frame_of_reference::decompress",
    "delta_patched": "delta::decompress",
    "run_length_encoding": "run_length_encoding::decompress",
    "run_position_encoding_relative": "run_position_encoding::decompress",
    "run_position_encoding_absolute": "run_position_encoding::decompress",
    "discard_zero_bytes_variable": "variable_width::decompress",
    "discard_zero_bytes_variable_3bit": "variable_width::decompress",
    "discard_zero_bytes_fixed": "fixed_width::decompress",
    "model": "model::decompress",
    "incidence_bitmaps": "incidence_bitmaps::decompress",
    "delta_fused_patches": "delta::patched::decompress",
    "dictionary_fused_patches": "dictionary::patched::decompress",
    "frame_of_reference_fused_patches": "frame_of_reference::patched::decompress",
    "delta_fused_compressed_patches": "delta::patched::decompress",
    "model_fused_patches": "model::patched::decompress",
    "discard_zero_bytes_fixed_fused_patches": "fixed_width::patched::decompress",
}


# decoder launches per step: variable-width discard-zero-bytes decodes a 2 GiB column as two consecutive launches of ~1 GiB of output
# (kernel_wrappers/decompression/discard_zero_bytes_variable.hpp); the traffic is reported per STEP, as bench.py's algorithmic bytes are
def launches_per_step(workload, elements):
    """as the wrapper cuts it: GIDDY_DZBV_LAUNCH_BYTES (1 GiB) of 8-byte output per launch, rounded to the nearest number of pieces — derived from the
    column's size, not a constant (round-3 advisor: the constant 2 was only right at 2^28 elements)"""
    if not workload.startswith("discard_zero_bytes_variable") or not elements:
        return 1
    return max(1, (8 * int(elements) + (1 << 29)) >> 30)


def details_of(stdout_path, details_path):
    """the whole record of a bench run: its --details file (since round 6 the final stdout line is the compact one), else the last stdout line"""
    try:
        return open(details_path).read().strip().splitlines()[-1]
    except OSError:
        return open(stdout_path).read().strip().splitlines()[-1]


def timed_region(durations, steps):
    """The dispatches of a kernel that belong to a timed region of the bench command: of the FULL-SIZE column — at least half the kernel's longest
    dispatch, which leaves out the thousands of 2^18-element columns `small_columns` launches afterwards and the pre-passes' tiny launches (round 5 took
    "the last N dispatches" and reported those: 3.8 us for FOR) — and of those the last `steps` (the headline runs 20 timed steps, the others 10; DZB-variable
    2^28 x 8 bytes is two launches per step; a run decoder's entry ends with 3 x 10 launches of three columns in turn)."""
    if not durations:
        return []
    longest = max(durations)
    full_size = [x for x in durations if 2 * x >= longest]
    return full_size[-steps:]


# which of a scheme's kernels a workload's step launches, where the scheme's name alone does not say (template arguments as rocprofv3 prints them)
ROW_FILTERS = {
    "run_position_encoding_relative": lambda name: ", true," in name.split("(")[0],
    "run_position_encoding_absolute": lambda name: ", false," in name.split("(")[0],
    "delta_patched": lambda name: "patched" not in name,
}


def roofline_from_the_trace(durations_by_kernel, bench):
    """<round>_roofline_from_rocprof.md: every workload of the bench line under rocprofv3 — algorithmic bytes per step (the line's) over the mean duration of
    the kernel's timed-region dispatches in the KERNEL TRACE, beside the fraction the line itself reports from HIP events.  Where consecutive workloads launch
    the same kernel, the kernel's full-size dispatches are dealt out to them in order by the number of steps each says it launched (`step_runs`)."""
    entries = [("dictionary", bench["roofline"], None, bench.get("step_runs"))] + \
              [(n, e["roofline"], e.get("elements"), e.get("step_runs")) for n, e in bench.get("schemes", {}).items()]
    out = ["# Roofline fractions recomputed from the rocprofv3 kernel trace (%s)\n" % ROUND,
           "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sharded none --extras none --schemes coverage` (tools/profile_round.sh);",
           "timed region = the last 10 (headline: 20) steps' dispatches of the workload's decoder kernel, of those that last at least half the kernel's longest dispatch;",
           "fraction = algorithmic bytes per step / (launches per step x their mean duration) / 8 TB/s.  `events` = the same fraction as bench.py reports it from HIP events in the same run.",
           "(A run decoder's figure is the cold one in both columns: three columns of different seeds decoded in turn.)\n",
           "| workload | kernel | launches/step | timed-region mean ns | algorithmic bytes/step | frac (trace) | frac (events) | trace/events |", "|---|---|---|---|---|---|---|---|"]
    records = []
    # the decoder kernel of each workload: of the rows of its scheme's namespace, the one the run spent most time in (not a pre-pass)
    def head(name):   # the kernel's name and template arguments, without its parameter list
        name = name.replace("void ", "")
        return name[: name.index(">(") + 1] if ">(" in name else name.split("(")[0]
    chosen = {}
    for workload, roofline, elements, runs in entries:
        scheme = KERNELS.get("dictionary" if workload == "dictionary_zipf" else "delta_patched" if workload == "delta_compressed_patches" else workload)
        if scheme is None:
            continue
        namespace, patched = scheme.rsplit("::", 1)[0] + "::", "::patched::" in scheme
        accept = ROW_FILTERS.get(workload, lambda name: True)
        rows = [k for k in durations_by_kernel if not k.startswith("headline alone: ") and ("::" + namespace) in head(k).split("<")[0]
                and ("::patched::" in head(k).split("<")[0]) == patched and "decode_pieces" not in k and "_pieces<" not in k and accept(head(k))]
        if rows:
            chosen[workload] = max(rows, key=lambda k: sum(durations_by_kernel[k]))
    for workload, roofline, elements, runs in entries:
        kernel = chosen.get(workload)
        if kernel is None:
            continue
        per_step = launches_per_step(workload, elements)
        steps = 20 if workload == "dictionary" else 10
        durations = durations_by_kernel[kernel]
        longest = max(durations)
        full_size = [x for x in durations if 2 * x >= longest]
        sharing = [(w, r) for w, _, _, r in entries if chosen.get(w) == kernel]
        if len(sharing) > 1:
            # (a workload launches its step a few more times than it counts — its constructor's launch that resolves the configuration, a verification's —: up to four
            # steps per workload are allowed for and placed at the BEGINNING of each workload's stretch, where they are; the timed region is a stretch's END)
            counted = None if any(r is None for _, r in sharing) else sum(r for _, r in sharing) * per_step
            extra = None if counted is None else len(full_size) - counted
            if extra is None or extra < 0 or extra > 4 * per_step * len(sharing):
                out.append("| %s | `%s` | %d | — | %d | — | %.4f | (shared kernel: %d full-size dispatches, the workloads' steps do not add up to it) |" % (
                    workload, head(kernel).replace("cuda::kernels::", ""), per_step, roofline["algorithmic_bytes_per_launch"], roofline["frac"], len(full_size)))
                continue
            before = 0
            for position, (w, r) in enumerate(sharing):
                uncounted = extra // len(sharing) + (1 if position < extra % len(sharing) else 0)
                if w == workload:
                    full_size = full_size[before: before + uncounted + r * per_step]
                    break
                before += uncounted + r * per_step
        timed = full_size[-steps * per_step:]
        mean_ns = sum(timed) / max(len(timed), 1)
        frac = roofline["algorithmic_bytes_per_launch"] / (per_step * mean_ns * 1e-9) / 8e12
        records.append({"workload": workload, "kernel": kernel, "launches_per_step": per_step, "timed_calls": len(timed), "mean_ns": mean_ns,
                        "algorithmic_bytes_per_step": roofline["algorithmic_bytes_per_launch"], "frac_trace": frac, "frac_events": roofline["frac"]})
        out.append("| %s | `%s` | %d | %.1f | %d | %.4f | %.4f | %.3f |" % (workload, head(kernel).replace("cuda::kernels::", ""), per_step, mean_ns,
                                                                           roofline["algorithmic_bytes_per_launch"], frac, roofline["frac"], frac / roofline["frac"]))
    with open(os.path.join(OUT, ROUND + "_roofline_from_rocprof.md"), "w") as f:
        f.write("\n".join(out) + "\n")
    print("\n".join(out))
    return records


def records_of_the_passes():
    """{workload: WORKLOAD_RECORD of tools/bench_one.py} out of the --pmc passes' logs"""
    found = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for workload in KERNELS:
            try:
                for line in open(os.path.join(RAW, "pmc_%s_%s.log" % (counter, workload))):
                    if line.startswith("WORKLOAD_RECORD "):
                        found[workload] = json.loads(line[len("WORKLOAD_RECORD "):])
            except (OSError, ValueError):
                pass
    return found


def mean_counter(path, kernel_substring):
    values = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel_substring in r["Kernel_Name"]]
    return (sum(values) / len(values), len(values)) if values else (None, 0)


def main():
    os.makedirs(OUT, exist_ok=True)
    # ---- kernel stats ------------------------------------------------------------------------------
    stats = list(csv.DictReader(open(os.path.join(RAW, "trace", "bench_kernel_stats.csv"))))
    ours = [r for r in stats if "cuda::kernels::" in r["Name"]]
    # the stats average includes the cold first launches (verification + warm-up); the timed region is the
    # last `steps` dispatches of each kernel — average those from the kernel trace as well
    trace = list(csv.DictReader(open(os.path.join(RAW, "trace", "bench_kernel_trace.csv"))))
    durations_by_kernel = {}
    for r in trace:
        durations_by_kernel.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    # the headline kernel's timed region from the run of the headline workload alone, where there is one (tools/profile_round.sh)
    try:
        for r in csv.DictReader(open(os.path.join(RAW, "trace_headline", "bench_kernel_trace.csv"))):
            if "dictionary::decompress" in r["Kernel_Name"]:
                durations_by_kernel.setdefault("headline alone: " + r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    except OSError:
        pass
    bench_line = details_of(os.path.join(RAW, "bench_under_rocprof.json"), os.path.join(RAW, "bench_under_rocprof.details.json"))
    with open(os.path.join(OUT, ROUND + "_bench_under_rocprof.json"), "w") as f:
        f.write(bench_line + "\n")
    records = roofline_from_the_trace(durations_by_kernel, json.loads(bench_line))
    by_kernel = {}
    for record in records:
        by_kernel.setdefault(record["kernel"], []).append(record)
    # One row per kernel of the stats — and, for a decoder kernel, one row per WORKLOAD that launches it, with THAT workload's timed region (its last 10 / 20 steps'
    # full-size dispatches, told apart from the other workloads' by the steps each launched), its algorithmic bytes per step and the fraction of the 8 TB/s peak they give:
    # every roofline fraction of the bench line can be recomputed from this file alone.
    with open(os.path.join(OUT, ROUND + "_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "Workload", "LaunchesPerStep", "TimedRegionCalls", "TimedRegionAverageNs",
                    "AlgorithmicBytesPerStep", "FracOfPeakFromTrace", "FracOfPeakByEvents"])
        for r in ours:
            common = [r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]]
            if r["Name"] in by_kernel:
                for record in by_kernel[r["Name"]]:
                    w.writerow(common + [record["workload"], record["launches_per_step"], record["timed_calls"], "%.1f" % record["mean_ns"],
                                         record["algorithmic_bytes_per_step"], "%.4f" % record["frac_trace"], "%.4f" % record["frac_events"]])
            else:
                d = durations_by_kernel.get(r["Name"], [])
                timed = timed_region(d, 10)
                w.writerow(common + ["", "", len(timed), "%.1f" % (sum(timed) / max(len(timed), 1)), "", "", ""])

    # ---- calibration ---------------------------------------------------------------------------------
    known_kib = (1 << 30) / 1024.0
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        for r in csv.DictReader(open(os.path.join(RAW, "pmc_" + counter, "calibrate", "cal_counter_collection.csv"))):
            if "calibrate_" in r["Kernel_Name"]:
                name = r["Kernel_Name"].replace("void ", "")
                name = name[: name.index(">(") + 1] if ">(" in name else name
                cal.setdefault(counter, {})[name] = float(r["Counter_Value"])
    read_factor = known_kib / [v for k, v in cal["FETCH_SIZE"].items() if "calibrate_read" in k and "(4)" in k][0]
    write_factor = known_kib / [v for k, v in cal["WRITE_SIZE"].items() if "calibrate_write" in k and "(4)" in k][0]

    # ---- per-kernel traffic -----------------------------------------------------------------------------
    bench = json.loads(bench_line)
    algorithmic = {"dictionary": bench["roofline"]["algorithmic_bytes_per_launch"]}
    durations = {"dictionary": bench["roofline"]["mean_launch_ms"]}
    elements = {}
    for name, entry in bench.get("schemes", {}).items():
        algorithmic[name] = entry["roofline"]["algorithmic_bytes_per_launch"]
        durations[name] = entry["roofline"]["mean_launch_ms"]
        elements[name] = entry.get("elements")
    elements.setdefault("discard_zero_bytes_variable_3bit", elements.get("discard_zero_bytes_variable"))
    try:   # workloads outside the default bench line: algorithmic bytes from the committed coverage bench
        coverage = json.load(open(os.path.join(OUT, ROUND + "_bench.json")))
        for name, entry in coverage.get("schemes", {}).items():
            algorithmic.setdefault(name, entry["roofline"]["algorithmic_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        pass
    for name, record in records_of_the_passes().items():   # every workload of the passes says its own algorithmic bytes
        algorithmic.setdefault(name, record["algorithmic_bytes_per_launch"])
        elements.setdefault(name, record["elements"])
    traffic, lines = {}, []
    for workload, kernel in KERNELS.items():
        try:
            fetch, nf = mean_counter(os.path.join(RAW, "pmc_FETCH_SIZE", workload, "pmc_counter_collection.csv"), kernel)
            write, nw = mean_counter(os.path.join(RAW, "pmc_WRITE_SIZE", workload, "pmc_counter_collection.csv"), kernel)
        except OSError:
            continue
        if fetch is None or write is None:
            continue
        per_step = launches_per_step(workload, elements.get(workload))
        read_bytes = fetch * 1024 * read_factor * per_step
        write_bytes = write * 1024 * write_factor * per_step
        total = read_bytes + write_bytes
        traffic[workload] = {"kernel": kernel, "fetch_size_kib_raw": fetch, "write_size_kib_raw": write,
                             "hbm_read_bytes": int(read_bytes), "hbm_write_bytes": int(write_bytes),
                             "hbm_bytes_per_launch": int(total), "algorithmic_bytes_per_launch": algorithmic.get(workload),
                             "dispatches_averaged": min(nf, nw), "kernel_sources_sha16": kernel_sources.fingerprint(kernel)}
        ratio = total / algorithmic[workload] if algorithmic.get(workload) else float("nan")
        lines.append("| %s | %s | %.0f | %.0f | %.3f | %.3f | %.3f | %.3f |" % (
            workload, kernel, fetch, write, read_bytes / 1e9, write_bytes / 1e9, (algorithmic.get(workload) or 0) / 1e9, ratio))
    by_kernel = {}
    for workload, t in traffic.items():
        by_kernel.setdefault(t["kernel"], t)   # bench.py looks traffic up by kernel name (first workload wins)
    with open(os.path.join(OUT, "hbm_traffic.json"), "w") as f:
        try:
            # (MEASURED_AT: the commit the passes ran on, when the summary is made later)
            commit = os.environ.get("MEASURED_AT") or subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        except (OSError, subprocess.CalledProcessError):
            commit = "unknown"
        json.dump({"round": ROUND, "measured_at_commit": commit, "read_correction": read_factor, "write_correction": write_factor,
                   "by_workload": traffic, **by_kernel}, f, indent=1)
    with open(os.path.join(OUT, ROUND + "_pmc_hbm.md"), "w") as f:
        f.write("# HBM traffic per launch from rocprofv3 PMC passes (%s)\n\n" % ROUND)
        f.write("Separate `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of `tools/bench_one.py <workload> 30 3` "
                "(DZB-V: 2^28 elements), mean over the profiled dispatches of the decoder kernel.\n\n"
                "Calibration (tools/pmc_calibrate.hip, each kernel moves exactly 1 GiB): raw counters are KiB; "
                "FETCH_SIZE read %.4f of the true bytes for 4-, 8- and 16-byte-per-lane streaming loads alike "
                "(%s) -> reads are multiplied by %.4f; WRITE_SIZE read %.4f of the true bytes -> x %.4f.\n\n" % (
                    1 / read_factor, ", ".join("%s=%.0f" % kv for kv in sorted(cal["FETCH_SIZE"].items()) if "read" in kv[0]),
                    read_factor, 1 / write_factor, write_factor))
        f.write("| workload | kernel | FETCH_SIZE raw KiB | WRITE_SIZE raw KiB | HBM read GB | HBM write GB | algorithmic GB | HBM / algorithmic |\n|---|---|---|---|---|---|---|---|\n")
        f.write("\n".join(lines) + "\n")
    print(open(os.path.join(OUT, ROUND + "_pmc_hbm.md")).read())


if __name__ == "__main__":
    main()
