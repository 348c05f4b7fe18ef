#!/usr/bin/env python3
"""Headline benchmark of the lightweight-decompression hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 30] [--workload dictionary]
                    [--schemes all|none|name,...] [--no-cpu-baseline]

One "step" = one pass of the hot path over one batch of synthetic, device-resident input.
The N=1 workload is BASELINE.json configs[1]: Dictionary, 8-bit keys -> 32-bit values, 256-entry
dictionary, 2^30 elements on one MI355X.  `value` = decompressed GB/s of the whole job (all ranks'
output bytes / max-over-ranks time), inputs resident in HBM when the timed region starts.  For N > 1
(launched by torch.distributed.run, one rank per GPU) every rank decodes its own 2^30-element shard —
columns shard by independent anchored segment ranges, there is no data-path collective — so scaling
is "weak".  BASELINE.json configs[3]/[4] (ONE RLE column of 2^30 / ONE DZB-variable column of 2^31 elements cut over the GPUs)
are measured as stated — strong scaling, shards cut on the device by libgiddy_amd.sharding — under `sharded` at every N, and as the
headline line with --scaling strong --workload NAME.

The JSON line also carries
  roofline      algorithmic HBM bytes of one launch / that kernel's mean duration (HIP events on
                the launch stream), against the 8 TB/s HBM3E peak;
  cpu_baseline  the CPU oracle (a port: the reference has no CPU path) on a bounded sample of the same
                workload on this host's cores (OpenMP, thread count stated; single-thread beside it);
  schemes       the same measurement for the other BASELINE.json configs that fit one GPU
                (FOR, Delta + patches, RLE, RPE, DZB-variable) — bit-exactness of every one of them is
                asserted against the oracle before anything is timed.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (≈6.3 TB/s achievable)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--log2n", type=int, default=30)
    p.add_argument("--workload", default="dictionary")
    p.add_argument("--schemes", default="all", help="extra per-scheme measurements at N=1: all (the BASELINE configs) | "
                   "coverage (all + DZB-fixed, Model, incidence bitmaps) | none | comma list")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--extras", choices=("all", "none"), default="all", help="none: without the blocks that are not a BASELINE measurement (first call after idle, small columns, "
                   "the three modes of use) — what tools/profile_round.sh runs under rocprofv3, so that a kernel's dispatches in the trace are its timed steps")
    p.add_argument("--details", default=None, help="where the whole record goes (default: bench_details.json beside bench.py); the final stdout line is the compact one")
    p.add_argument("--gather", action="store_true", help="N > 1: after the timed region also time the one optional collective — the RCCL "
                   "point-to-point gather of every rank's decoded shard into one contiguous buffer on rank 0 (reported separately, never part of `value`)")
    p.add_argument("--gather-log2n", type=int, default=28, help="elements of each rank's decoded shard that are gathered")
    p.add_argument("--cpu-sample-log2", type=int, default=27)
    p.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="strong: the headline line is ONE column of 2^log2n elements "
                   "(same seed on every rank) cut into rank shards on the device by libgiddy_amd.sharding — BASELINE.json configs[3]/[4] as stated; "
                   "use with --workload run_length_encoding | run_position_encoding_relative | run_position_encoding_absolute | discard_zero_bytes_variable")
    p.add_argument("--sharded", default="default", help="extra strong-scaling measurements appended to the line under `sharded` (every N, so that the "
                   "driver's N = 1, 2, 4, 8 runs give the curve): default = run_length_encoding and run_position_encoding_relative at 2^30 and discard_zero_bytes_variable at 2^31 elements "
                   "(with the gather into rank 0) | none | comma list of workload[:log2n]")
    return p.parse_args()


def make_workload(name, n, device, per_rank_seed=True, cap=1 << 28, seed_offset=None):
    """`cap`: elements of the workloads whose single-GPU BASELINE shape is a 2^28-element shard (DZB-variable, incidence bitmaps)"""
    import workloads as w
    w.SEED_OFFSET = seed_offset if seed_offset is not None else (1000 * int(os.environ.get("RANK", "0")) if per_rank_seed else 0)
    table = {
        "dictionary": lambda: w.Dictionary(n, device),
        "dictionary_zipf": lambda: w.Dictionary(n, device, zipf=True),
        "frame_of_reference": lambda: w.FrameOfReference(n, device),
        "delta_patched": lambda: w.DeltaPatched(n, device),
        "delta_fused_patches": lambda: w.DeltaFused(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "delta_compressed_patches": lambda: w.DeltaCompressedPatches(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "delta_fused_compressed_patches": lambda: w.DeltaFusedCompressedPatches(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "frame_of_reference_fused_patches": lambda: w.FrameOfReferenceFused(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "dictionary_fused_patches": lambda: w.DictionaryFused(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "run_length_encoding": lambda: w.RunLength(n, device, mean_run=int(os.environ.get("MEAN_RUN", "32"))),
        "run_position_encoding_relative": lambda: w.RunPosition(n, device, relative=True),
        "run_position_encoding_absolute": lambda: w.RunPosition(n, device, relative=False),
        "discard_zero_bytes_variable": lambda: w.DiscardZeroBytesVariable(min(n, cap), device),
        "discard_zero_bytes_variable_3bit": lambda: w.DiscardZeroBytesVariable(min(n, cap), device, bits=3),
        "discard_zero_bytes_fixed": lambda: w.DiscardZeroBytesFixed(n, device),
        "model": lambda: w.Model(n, device),
        "model_fused_patches": lambda: w.ModelFused(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "discard_zero_bytes_fixed_fused_patches": lambda: w.DiscardZeroBytesFixedFused(n, device, patch_fraction=float(os.environ.get("PATCH_FRACTION", "0.001"))),
        "incidence_bitmaps": lambda: w.IncidenceBitmaps(min(n, cap), device),
    }
    work = table[name]()
    work.workload_name = name
    # how often the step has been launched in this process (tools/summarize_profiles.py finds a workload's dispatches in the rocprofv3 kernel trace by it,
    # where two workloads of one run launch the same kernel: DZB-variable with 2- and 3-bit sizes, Delta with plain and compressed patch positions)
    import functools
    work.step_runs = 0
    uncounted = work.run
    @functools.wraps(uncounted)
    def counted(*args, **kwargs):
        work.step_runs += 1
        return uncounted(*args, **kwargs)
    work.run = counted
    return work


SETTLE_SECONDS = 0.15
RUN_DECODER_WORKLOADS = ("run_length_encoding", "run_position_encoding_relative", "run_position_encoding_absolute")


def timed_steps(work, steps, warmup, barrier, per_step_events=True):
    """EXACTLY `steps` timed passes between barrier+synchronize brackets.  The rank's wall time is read after its own synchronize() and
    BEFORE the closing barrier (the barrier brackets the region; it is not part of any rank's time — a strong-scaled step is tens of
    microseconds, a barrier as long).  The dominant kernel's launch duration comes from HIP events on the launch stream (torch's current
    stream, the one the kernels are enqueued on): a pair around every launch by default — a step with a second small kernel (Delta +
    patches) passes the pair down so that only the decoder is bracketed — or, `per_step_events=False`, ONE pair around the whole region
    (no event between the steps: the mean then includes the gaps between launches, and short steps are not stretched by the events)."""
    import inspect
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    brackets_itself = "around_dominant_kernel" in inspect.signature(work.run).parameters
    # Before the W warm-up steps, `SETTLE_SECONDS` of the same work, untimed: what precedes a timed region here (generating and verifying a
    # column) leaves the GPU idle for seconds, and the first timed region after that has been seen 3-26 % slower than the same region a moment later
    settle_until = time.perf_counter() + SETTLE_SECONDS
    while time.perf_counter() < settle_until:
        for _ in range(8):
            work.run(stream)
        torch.cuda.synchronize()
    for _ in range(warmup):
        work.run(stream)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps if per_step_events else 1)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if per_step_events:
        for begin, end in events:
            if brackets_itself:
                work.run(stream, around_dominant_kernel=(begin.record, end.record))
            else:
                begin.record()
                work.run(stream)
                end.record()
    else:
        events[0][0].record()
        for _ in range(steps):
            work.run(stream)
        events[0][1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    per_launch_ms = [b.elapsed_time(e) for b, e in events]
    timed_steps.last_median_launch_ms = sorted(per_launch_ms)[len(per_launch_ms) // 2] if per_step_events else None   # (SURVEY.md section 8(d) asks for the median; `frac` is of the mean)
    return elapsed, sum(per_launch_ms) / steps


def first_call_after_idle(work, idle_seconds=1.0, repetitions=3):
    """ONE pass of the hot path after the GPU has idled for `idle_seconds` — what a caller's first decode gets, which the steady-state
    figures above (a settle loop, warm-up steps) never show: by HIP events on the launch stream and by the host's clock around launch +
    synchronize.  Median of `repetitions`."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream
    by_events, on_host = [], []
    for _ in range(repetitions):
        torch.cuda.synchronize()
        time.sleep(idle_seconds)
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        begin.record(); work.run(stream); end.record()
        torch.cuda.synchronize()
        on_host.append((time.perf_counter() - t0) * 1e3); by_events.append(begin.elapsed_time(end))
    ms = sorted(by_events)[len(by_events) // 2]
    return {"after_idle_seconds": idle_seconds, "ms_by_events": round(ms, 4), "ms_on_the_hosts_clock": round(sorted(on_host)[len(on_host) // 2], 4),
            "frac_of_hbm_peak": round(work.algorithmic_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
            "note": "one step after %.1f s of idle (median of %d): clocks and power states have to come up; NOT the steady-state rate of `value`" % (idle_seconds, repetitions)}


def small_columns(name, work, device, log2_piece=18, pieces=1024):
    """`pieces` columns of 2^log2_piece elements — consecutive segment-range shards of the workload's column, decoded into their ranges of its output —
    (a) a launch each, back to back, (b) as ONE batch launch (giddy_enqueue_batch, include/giddy/batch.hpp): % of the HBM peak over the
    algorithmic bytes of the pieces, beside the rate of the same bytes as part of the whole column."""
    import torch
    from libgiddy_amd import schemes
    stream = torch.cuda.current_stream().cuda_stream
    total_pieces = work.n >> log2_piece
    pieces = min(pieces, total_pieces)
    if pieces < 2:
        return None
    fields = work.column_fields()
    shards = [work.shard_function(fields, r, total_pieces) for r in range(pieces)]
    outs = [work.output[s.start:s.stop] for s in shards]
    nbytes = sum(o.numel() * o.element_size() + sum(v.numel() * v.element_size() for v in s.fields.values() if isinstance(v, torch.Tensor)) for s, o in zip(shards, outs))
    configs = [work.launch_fields(o, s.fields, stream, None) for o, s in zip(outs, shards)]
    with schemes.into_batch() as batch:
        for o, s in zip(outs, shards):
            work.launch_fields(o, s.fields, stream, None)

    def each():
        for o, s, c in zip(outs, shards, configs):
            work.launch_fields(o, s.fields, stream, c)

    def ms_of(run):
        run(); torch.cuda.synchronize()
        times = []
        for _ in range(5):
            begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            begin.record(); run(); end.record(); torch.cuda.synchronize()
            times.append(begin.elapsed_time(end))
        return sorted(times)[len(times) // 2]

    frac = lambda ms: round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    ms_each, ms_batch = ms_of(each), ms_of(lambda: batch.enqueue(stream))
    return {"columns": pieces, "elements_each": 1 << log2_piece, "algorithmic_bytes": nbytes,
            "a_launch_per_column": {"ms": round(ms_each, 4), "frac_of_hbm_peak": frac(ms_each), "us_per_column": round(ms_each * 1e3 / pieces, 2)},
            "one_batch_launch": {"ms": round(ms_batch, 4), "frac_of_hbm_peak": frac(ms_batch), "workgroups": batch.num_blocks},
            "note": "giddy_enqueue_batch: every column a piece of ONE launch (the columns' bytes are re-read from the Infinity Cache between repetitions when they fit it; "
                    "cold figures per size: profiles/r05_size_sweep.txt)"}


def roofline(work, mean_launch_ms, traffic):
    """`traffic`: None, or (bytes, source) as committed_traffic returns"""
    achieved = work.algorithmic_bytes / (mean_launch_ms * 1e-3) / 1e9
    traffic, traffic_source = traffic if isinstance(traffic, tuple) else (traffic, None)
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "kernel": work.kernel, "algorithmic_bytes_per_launch": work.algorithmic_bytes,
            "mean_launch_ms": round(mean_launch_ms, 4),
            "median_launch_ms": round(timed_steps.last_median_launch_ms, 4) if getattr(timed_steps, "last_median_launch_ms", None) else None}


DETAILS_FILE = "bench_details.json"
FINAL_LINE_LIMIT = 4096   # bytes: the driver recovers the line from the tail of stdout (round 5's 22 kB line came back `parsed: null`)


def _fracs(entries, cold_first=True):
    """{name: roofline.frac} of a `schemes` / `sharded` block of the full record"""
    out = {}
    for name, entry in (entries or {}).items():
        r = entry.get("roofline") or entry.get("roofline_rank0") or {}
        if "frac" in r:
            out[name] = r["frac"]
    return out


def compact_line(line, details_file=DETAILS_FILE):
    """The FINAL stdout line: the contract's keys + roofline + cpu_baseline + one flat map of roofline fractions per scheme.
    Everything else of the record (`schemes`, `modes`, `sharded`, `small_columns`, `first_call`, the predictions) is in `details_file`."""
    pick = lambda d, keys: {k: d[k] for k in keys if k in d}
    out = pick(line, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    out["config"] = pick(line.get("config", {}), ("workload", "instantiation", "elements_per_gpu", "elements_total", "launch"))
    if "roofline" in line:
        out["roofline"] = pick(line["roofline"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "mean_launch_ms",
                                                  "measured_copy_GBps", "frac_of_measured_copy", "measured_widening_stream_GBps", "frac_of_measured_widening_stream"))
    if "cpu_baseline" in line:
        out["cpu_baseline"] = pick(line["cpu_baseline"], ("value", "unit", "cores", "kind", "single_thread_value", "cpu", "sample"))
    if line.get("schemes"):
        out["schemes_frac"] = _fracs(line["schemes"])
    if line.get("sharded"):
        out["sharded_frac"] = _fracs(line["sharded"])
        predicted = {name: entry["predicted_strong_efficiency_at_8"] for name, entry in line["sharded"].items() if "predicted_strong_efficiency_at_8" in entry}
        if predicted:
            out["predicted_strong_efficiency_at_8"] = predicted
    out["details"] = details_file
    return out


def emit(line, details_path=None):
    """rank 0: the whole record -> bench_details.json (and stderr, for the log); the compact line LAST on stdout"""
    details_path = details_path or os.path.join(ROOT, DETAILS_FILE)
    text = json.dumps(line, allow_nan=False)
    try:
        with open(details_path, "w") as f:
            f.write(text + "\n")
    except OSError as problem:   # (a read-only checkout must not cost the line)
        print("bench.py: could not write %s: %r" % (details_path, problem), file=sys.stderr)
    print(text, file=sys.stderr)
    sys.stderr.flush()
    final = json.dumps(compact_line(line, os.path.basename(details_path)), allow_nan=False)
    assert len(final) < FINAL_LINE_LIMIT, "the final line has grown to %d bytes" % len(final)
    print(final)
    sys.stdout.flush()


def measured_copy_ceiling():
    """What this box sustains on the library's own streaming loop with the identity in place of a decoder (tools/stream_ceiling.py,
    tests/cpp/stream_ceiling.hip: 16 bytes a lane, the decoders' non-temporal stores): a plain copy, and the headline's 1 B in : 4 B out shape.
    bytes read + written per second.  None when the measurement library is not built."""
    import stream_ceiling
    try:
        return stream_ceiling.measure()
    except (FileNotFoundError, OSError) as problem:
        return {"skipped": repr(problem)}


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def committed_traffic(kernel, algorithmic_bytes=None, workload=None):
    """(HBM bytes per launch, where the number comes from) from the committed rocprofv3 --pmc passes (profiles/hbm_traffic.json;
    PMC counters cannot be collected from inside the run).  The value is None — and the source says why — when there is no
    measurement for this kernel AT THIS SIZE (the passes ran the default workloads: 2^30 elements, DZB-V 2^28), or when the
    headers the kernel is compiled from have changed since the pass (tools/kernel_sources.py fingerprints them)."""
    import kernel_sources
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    source = {"file": "profiles/hbm_traffic.json", "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, calibrated (tools/profile_round.sh)"}
    try:
        with open(path) as f:
            committed = json.load(f)
    except (OSError, ValueError):
        return None, dict(source, status="no committed measurement")
    source.update(round=committed.get("round"), measured_at_commit=committed.get("measured_at_commit"))
    entry = committed.get("by_workload", {}).get(workload) or committed.get(kernel, {})
    if not entry:
        return None, dict(source, status="no measurement for this kernel")
    measured_for = entry.get("algorithmic_bytes_per_launch")
    if algorithmic_bytes is not None and measured_for and abs(measured_for - algorithmic_bytes) > 0.001 * algorithmic_bytes:
        return None, dict(source, status="measured at another size")
    recorded, current = entry.get("kernel_sources_sha16"), kernel_sources.fingerprint(kernel)
    if recorded != current:
        return None, dict(source, status="stale: the kernel's headers changed since the pass (was %s, is %s)" % (recorded, current))
    return entry.get("hbm_bytes_per_launch"), dict(source, status="current", kernel_sources_sha16=current)


def cpu_baseline(work, sample, unit_bytes_label="decompressed GB/s"):
    from oracle import oracle
    cores = oracle.max_threads()
    results = {}
    for threads in (1, cores):
        fn, out_bytes = work.cpu_run(oracle, sample, threads)
        fn()  # warm
        best = float("inf")
        deadline = time.perf_counter() + 6.0
        reps = 0
        while reps < 3 or (time.perf_counter() < deadline and reps < 20):
            t = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t); reps += 1
        results[threads] = out_bytes / best / 1e9
    return {"value": round(results[cores], 3), "unit": unit_bytes_label, "cores": cores, "kind": "port",
            "sample": "first %d elements of the same workload, CPU oracle (oracle/giddy_oracle.c, gcc -O3 -fopenmp), best of >=3" % sample,
            "single_thread_value": round(results[1], 3), "cpu": cpu_model()}


def time_gather(local, counts, rank, world, device, barrier, repetitions=3, expected=None):
    """The single optional collective: rank r's decoded shard `local` (counts[r] elements) gathered into one
    contiguous buffer on rank 0 by point-to-point RCCL transfers — each shard crosses one xGMI link once.  `expected`: on rank 0,
    what the contiguous buffer must hold (strong scaling: the whole decoded column)."""
    import torch
    import torch.distributed as dist
    from libgiddy_amd.distributed import gather_decoded
    total = sum(int(c) for c in counts)
    out = torch.empty(total, dtype=local.dtype, device=device) if rank == 0 else None
    best = float("inf")
    synchronize = torch.cuda.synchronize if local.is_cuda else (lambda: None)   # (CPU tensors: the gloo test of this function)
    for repetition in range(repetitions + 1):   # the first one sets the connections up
        barrier(); synchronize()
        t0 = time.perf_counter()
        gather_decoded(local, counts, dst=0, out=out)
        synchronize(); barrier()
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        if repetition > 0:
            best = min(best, float(elapsed[0]))
    report = {}
    if rank == 0:
        report["own_shard_in_place"] = bool((out[: int(counts[0])] == local).all())
        if expected is not None:
            report["gathered_equals_the_column"] = bool((out == expected.reshape(-1)).all())
    inbound = (total - int(counts[0])) * local.element_size()
    report.update({"bytes_per_rank": [int(c) * local.element_size() for c in counts] if len(set(int(c) for c in counts)) > 1 else int(counts[0]) * local.element_size(),
                   "ms": round(best * 1e3, 3), "inbound_GBps_at_root": round(inbound / best / 1e9, 1), "links": world - 1,
                   "note": "point-to-point receives from every peer (one xGMI link each, about 153 GB/s per link); not part of `value`"})
    return report


def placement_and_cache(name, work, n, device, steps):
    """Two things a timing loop over one column on a 2 MiB boundary does not show (round-2 verdict):
    the rate with the OUTPUT displaced (what a pool allocator or a shard boundary gives), and the rate when the compressed INPUT does
    not come out of the 256 MiB Infinity Cache — a run column of 2^30 outputs is 200 MB of runs, which repeated launches find there:
    three columns of different seeds are decoded in turn, into the same output."""
    import torch
    import workloads as w
    stream = torch.cuda.current_stream().cuda_stream
    out = {}
    def frac(ms): return round(work.algorithmic_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    def timed(sequence):
        for x in sequence[:2]: x.run(stream)
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); begin.record()
        for x in sequence: x.run(stream)
        end.record(); torch.cuda.synchronize()
        return begin.elapsed_time(end) / len(sequence)
    original, size = work.output, work.output.element_size()
    big = torch.empty(n + (1 << 20), dtype=original.dtype, device=device)
    for offset in (128, 32768):
        work.output = big[offset // size: offset // size + n]
        work.config = work._launch(stream, None) if hasattr(work, "_launch") else work.config
        out["frac_output_at_+%d_bytes" % offset] = frac(timed([work] * steps))
    work.output = original
    del big
    keep = w.SEED_OFFSET
    others = []
    for k in (1, 2):
        other = make_workload(name, n, device, seed_offset=keep + 7777 * k)
        other.output = original
        others.append(other)
    w.SEED_OFFSET = keep
    warm_ms = timed([work] * steps)
    cold_ms = timed([x for _ in range(steps) for x in [work] + others][: 3 * steps])
    out["frac_one_column_repeated"], out["ms_one_column_repeated"] = frac(warm_ms), round(warm_ms, 4)
    out["frac_three_columns_in_turn"], out["ms_three_columns_in_turn"] = frac(cold_ms), round(cold_ms, 4)
    out["note"] = ("one column repeated: its 200 MB of runs come out of the Infinity Cache; three in turn: from HBM.  Since round 3 the wrappers launch, for columns of "
                   "mean runs >= 16, a build of the kernel that is set for the second case (DESIGN.md section 4, 'Cold inputs': round 2's kernel 0.85 / 0.62-0.74, this one 0.80-0.82 / 0.74-0.79)")
    del others
    torch.cuda.empty_cache()
    return out


def time_gather_compressed(whole, shard, counts, rank, world, device, barrier, repetitions=3, expected=None, pieces=4, through_host=False):
    """The same contiguous column on rank 0, the other way round (libgiddy_amd.distributed.gather_compressed): the peers send their
    COMPRESSED shards, in pieces, and rank 0 decodes them into place while the next pieces arrive.  Reported beside `gather`, never
    part of `value`."""
    import torch
    import torch.distributed as dist
    from libgiddy_amd.distributed import gather_compressed
    total = sum(int(c) for c in counts)
    stream = lambda: torch.cuda.current_stream().cuda_stream
    if through_host:   # (the one-GPU rehearsal: gloo moves host tensors; a piece goes up to the device, is decoded there and comes back)
        repetitions = 1
        out = torch.zeros(total, dtype=whole.output.dtype) if rank == 0 else None
        fields = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in shard.fields.items()}
        def decode(fs, view):
            on_device = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in fs.items()}
            decoded = torch.empty(view.numel(), dtype=view.dtype, device=device)
            whole.launch_fields(decoded, on_device, stream(), None)
            view.copy_(decoded.cpu())
        where = torch.device("cpu")
    else:
        out = torch.zeros(total, dtype=whole.output.dtype, device=device) if rank == 0 else None
        fields = shard.fields
        decode = lambda fs, view: whole.launch_fields(view, fs, stream(), None)
        where = device
    best, moved = float("inf"), None
    for repetition in range(repetitions + 1):
        barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        moved = gather_compressed(fields, whole.shard_function, decode, out, counts, dst=0, pieces=pieces, device=where)
        torch.cuda.synchronize()
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=where)
        barrier()
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        if repetition > 0:
            best = min(best, float(elapsed[0]))
    report = {"ms": round(best * 1e3, 3), "pieces_per_shard": pieces,
              "note": "peers send compressed shards point to point (one xGMI link each), rank 0 decodes the whole column; includes the decoding; not part of `value`"}
    if rank == 0:
        report.update({"bytes_received": moved[0], "bytes_a_decoded_gather_receives": moved[1]})
        if expected is not None:
            report["gathered_equals_the_column"] = bool((out.to(expected.device) == expected.reshape(-1)).all())
    return report


def modes_of_use(n, device, steps):
    """A = the factory's type-erased wrapper over the C ABI (what everything above times), B = an instantiated wrapper +
    cuda::kernel::enqueue_launch(kernel, stream, config, typed parameters...), C = the kernel template launched directly — the reference's three
    documented modes (README.md:94-126) — on C4 (run lengths; ordinary segments and segments of 65 536) and C3 (Delta): % of the HBM peak each, and
    whether B's and C's outputs equal A's.  tests/cpp/modes_shim.hip behind ctypes (tools/modes.py); the full table: profiles/r04_modes_bench.txt."""
    import torch
    import modes
    import workloads as w
    modes.shim(build_if_missing=False)   # (never a compiler from inside the benchmark: absent -> "skipped" in the line)
    stream = torch.cuda.current_stream()
    out = {"note": "A and B reach the same typed launch (include/giddy/kernel_wrapper.hpp); C is the bare kernel of the reference's signature"}

    def ms_of(run):
        for _ in range(2):
            run(stream.cuda_stream)
        begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        begin.record(stream)
        for _ in range(steps):
            run(stream.cuda_stream)
        end.record(stream); end.synchronize()
        return begin.elapsed_time(end) / steps

    columns = [("run_length_encoding<4,4,2>, segments of 4096", "rle<4,4,2>", lambda: w.RunLength(n, device)),
               ("run_length_encoding<4,4,2>, segments of 65536", "rle<4,4,2>", lambda: w.RunLength(n, device, segment_length=65536)),
               ("delta<4,8,2>, segments of 4096", "delta<4,8,2>", lambda: w.DeltaPatched(n, device))]
    for label, scheme, make in columns:
        work = make()
        fields = work.column_fields()
        entry, reference = {}, None
        for mode in ("A", "B", "C"):
            run = modes.launcher(scheme, fields, work.output, mode)
            work.output.fill_(-1)
            run(stream.cuda_stream); torch.cuda.synchronize()
            if reference is None:
                reference = work.output.clone()
            else:
                entry["%s_equals_A" % mode] = bool((work.output == reference).all())
            ms = ms_of(run)
            entry["frac_" + mode] = round(work.algorithmic_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
        out[label] = entry
        del work, fields, reference
        torch.cuda.empty_cache()
    return out


def all_reduce_u64_sum(value, world, device, backend):
    """sum of 64-bit checksums over the ranks, wrapping — carried as 16-bit limbs so that no backend's integer addition can overflow"""
    import torch
    import torch.distributed as dist
    if world == 1:
        return value & ((1 << 64) - 1)
    limbs = torch.tensor([(value >> (16 * i)) & 0xFFFF for i in range(4)], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(limbs, op=dist.ReduceOp.SUM)
    return sum(int(limbs[i]) << (16 * i) for i in range(4)) & ((1 << 64) - 1)


def strong_scaling_leg(name, log2n, args, rank, world, device, backend, barrier, oracle, gather=False):
    """BASELINE.json configs[3]/[4] as stated: ONE column of 2^log2n elements — the same seed on every rank — cut into `world`
    segment-range shards ON THE DEVICE (libgiddy_amd.sharding: views of the big arrays, anchors rebased by a device subtraction), rank r
    decodes shard r; no collective on the data path.  Checked: the shards' position-weighted checksums sum to the whole column's (decoded
    by every rank by itself first), and rank 0 verifies the whole column against the oracle / the full-size identities."""
    import torch
    import torch.distributed as dist
    import workloads as w
    w.SEED_OFFSET = 0
    stream = torch.cuda.current_stream().cuda_stream
    whole = make_workload(name, 1 << log2n, device, per_rank_seed=False, cap=1 << log2n)
    whole.run(stream)
    if rank == 0:
        whole.verify(oracle)
    torch.cuda.synchronize()
    column_checksum = w.checksum(whole.output)
    whole.output.zero_()
    shard = w.Sharded(whole, rank, world)
    shard.run(stream)
    torch.cuda.synchronize()
    shards_checksum = all_reduce_u64_sum(w.checksum(shard.output, first_index=shard.start), world, device, backend)
    assert shards_checksum == column_checksum, "%s: the shards' checksums (%x) do not sum to the column's (%x)" % (name, shards_checksum, column_checksum)
    steps, warmup = args.steps, args.warmup
    # TWO timed regions of `steps` steps, the faster one reported (both stated): the checks above leave the GPU idle for seconds, and a first region
    # has been seen 26 % slower than the next on the same box (clocks, or the host thread behind the 16 launches a 2^31 DZB-variable step takes)
    regions = [timed_steps(shard, steps, warmup, barrier, per_step_events=False) for _ in range(2)]
    # ... the SAME region for every rank: each region's time is its slowest rank's, and the region with the smaller such time is the one reported
    # (a maximum over ranks of each rank's own faster region could pair rank 0's first region with rank 1's second: optimistic)
    slowest_per_region = torch.tensor([r[0] for r in regions], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(slowest_per_region, op=dist.ReduceOp.MAX)
    chosen = int(torch.argmin(slowest_per_region))
    elapsed, launch_ms = regions[chosen]
    # every rank's wall time (read before the closing barrier) and every rank's event-timed step, side by side
    times = torch.tensor([[elapsed, launch_ms * 1e-3] if r == rank else [0.0, 0.0] for r in range(world)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.SUM)
    event_ms = [round(float(x) * 1e3, 4) for x in times[:, 1]]
    times = times[:, 0]
    slowest = float(times.max())
    counts = [0] * world
    if world > 1:
        dist.all_gather_object(counts, shard.n)
    else:
        counts = [shard.n]
    entry = {"workload": name, "instantiation": whole.name, "elements_total": whole.n, "scaling": "strong",
             "sharding": "libgiddy_amd.sharding on device tensors: %d segment-range shards of one column, rebased anchors, no collective" % world,
             "decompressed_GBps": round(whole.decompressed_bytes * steps / slowest / 1e9, 1), "ms_per_step": round(slowest / steps * 1e3, 4),
             "per_rank_ms_per_step": [round(float(t) / steps * 1e3, 4) for t in times], "per_rank_event_ms_per_step": event_ms,
             "timing": "wall: each rank's clock from after the opening barrier to after its own synchronize(), MAX over ranks; events: one HIP event pair around the rank's %d steps; "
                       "two such regions are timed; a region's time is its slowest rank's, and the region with the smaller such time is reported for every rank (region %d here)" % (steps, chosen),
             "this_rank_ms_per_step_of_both_regions": [round(r[0] / steps * 1e3, 4) for r in regions],
             "per_rank_elements": [int(c) for c in counts],
             "world_size_seen_by_the_process_group": dist.get_world_size() if world > 1 else 1,
             "checksum": "%016x" % column_checksum, "shards_checksum_equals_column_checksum": True,
             "this_rank_roofline": roofline(shard, launch_ms, None)}
    if world == 1:
        # What ONE rank of 8 would be given, decoded alone on this GPU: shards 0, 3 and 7 of the column cut 8 ways.  shard rate / column rate
        # is the strong-scaling efficiency to expect at N = 8 from the decode itself (launch tails, the fixed cost of the wrappers' pre-passes on a
        # column an eighth the size) — a PREDICTION from one GPU; the curve itself needs the 8-GPU node.
        column_rate = whole.decompressed_bytes * steps / slowest
        predicted = {}
        for r8 in (0, 3, 7):
            eighth = w.Sharded(whole, r8, 8)
            el8, _ = min((timed_steps(eighth, steps, warmup, barrier, per_step_events=False) for _ in range(2)), key=lambda r: r[0])
            predicted["rank_%d_of_8" % r8] = {"elements": eighth.n, "ms_per_step": round(el8 / steps * 1e3, 4),
                                             "decompressed_GBps": round(eighth.decompressed_bytes * steps / el8 / 1e9, 1),
                                             "shard_rate_over_column_rate": round(eighth.decompressed_bytes * steps / el8 / column_rate, 4)}
            del eighth
        # what the one optional collective WOULD move per GPU for this column cut 8 ways: the decoded shards (gather_decoded) against the compressed ones (gather_compressed)
        eighths = [w.Sharded(whole, r8, 8) for r8 in range(8)]
        entry["bytes_a_gather_into_rank_0_would_move_at_8"] = {
            "decoded_per_rank": [int(e.decompressed_bytes) for e in eighths],
            "compressed_per_rank": [int(e.algorithmic_bytes - e.decompressed_bytes) for e in eighths],
            "decoded_over_compressed": round(sum(e.decompressed_bytes for e in eighths) / max(1, sum(e.algorithmic_bytes - e.decompressed_bytes for e in eighths)), 2),
            "note": "7 of the 8 shards cross xGMI into rank 0 (~153 GB/s per link): libgiddy_amd.distributed.gather_compressed moves the second row and decodes at the root; UNMEASURED on hardware"}
        del eighths
        entry["shard_of_8_alone_on_this_gpu"] = predicted
        entry["predicted_strong_efficiency_at_8"] = min(v["shard_rate_over_column_rate"] for v in predicted.values())
        entry["predicted_strong_efficiency_note"] = ("min over ranks 0, 3, 7 of (that rank's shard-of-8 decoded alone, rate) / (the whole column, rate), both on this one GPU: "
                                                    "what the decode itself would scale like; NOT a measured scaling curve — no 8-GPU node has been available")
    if gather and world > 1:
        expected = getattr(whole, "values", None) if rank == 0 else None   # (DZB-V keeps the column it compressed)
        # (the optional collective is reported beside the decode and never part of `value`: a failure in it — it has only ever been rehearsed over gloo, no
        # multi-GPU node having been available — must not cost the line; an exception is reported in its place)
        if backend == "nccl":
            try:
                entry["gather"] = time_gather(shard.output, counts, rank, world, device, barrier, expected=expected)
            except Exception as problem:
                entry["gather"] = {"failed": repr(problem)}
        else:   # (the one-GPU rehearsal: gloo moves host tensors)
            entry["gather"] = time_gather(shard.output.cpu(), counts, rank, world, torch.device("cpu"), barrier, repetitions=1,
                                          expected=expected.cpu() if expected is not None else None)
            entry["gather"]["note"] = "REHEARSAL over gloo through host copies: checks the flow, says nothing about xGMI"
        try:
            entry["gather_compressed"] = time_gather_compressed(whole, shard, counts, rank, world, device, barrier, expected=expected, through_host=backend != "nccl")
        except Exception as problem:
            entry["gather_compressed"] = {"failed": repr(problem)}
        if backend != "nccl":
            entry["gather_compressed"]["note"] = "REHEARSAL over gloo through host copies: checks the flow, says nothing about xGMI"
    del shard, whole
    torch.cuda.empty_cache()
    return entry


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus, args.gpus))
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: no HIP device is visible")
    # (GIDDY_BENCH_SHARE_DEVICE=1 GIDDY_BENCH_BACKEND=gloo: rehearse the N > 1 flow with every rank on GPU 0 of a 1-GPU box)
    if os.environ.get("GIDDY_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("GIDDY_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def barrier():
        if world > 1:
            dist.barrier()

    from oracle import oracle
    n = 1 << args.log2n

    if args.scaling == "strong":
        # ---- one column cut over the ranks: BASELINE.json configs[3] / configs[4] as stated ------------
        entry = strong_scaling_leg(args.workload, args.log2n, args, rank, world, device, backend, barrier, oracle, gather=args.gather)
        line = {
            "metric": "decompressed GB/s (whole job), %s, ONE column of 2^%d elements cut into %d segment-range shards" % (entry["instantiation"], args.log2n, world),
            "value": entry["decompressed_GBps"], "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": entry["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64" if "variable" in args.workload else "u32", "data": "synthetic",
            "config": {"workload": args.workload, "instantiation": entry["instantiation"], "elements_total": entry["elements_total"], "sharding": entry["sharding"]},
            "roofline": entry.pop("this_rank_roofline"), "sharded": {args.workload: entry},
        }
        if rank == 0:
            emit(line, args.details)
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- the headline workload ------------------------------------------------------------------
    work = make_workload(args.workload, n, device)
    work.run(torch.cuda.current_stream().cuda_stream)
    work.verify(oracle)                                   # bit-exact before anything is reported
    first_call = first_call_after_idle(work) if world == 1 and args.extras == "all" else None
    elapsed, mean_launch_ms = timed_steps(work, args.steps, args.warmup, barrier)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    value = world * work.decompressed_bytes * args.steps / elapsed / 1e9

    line = {
        "metric": "decompressed GB/s (whole job; per-GPU = value / n_gpus), %s, 2^%d elements per GPU" % (work.name, args.log2n),
        "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: Dictionary, 8-bit keys -> int32 values, 256-entry dictionary, 2^%d elements per GPU" % args.log2n
                   if args.workload == "dictionary" else args.workload,
                   "instantiation": work.name, "elements_per_gpu": n, "sharding": "independent segment ranges, one shard per GPU, no collective",
                   "launch": repr(work.config),
                   "untimed_before_the_warmup_steps": "%.2f s of the same steps (the GPU has idled while the column was generated and verified)" % SETTLE_SECONDS},
        "per_gpu_value": round(value / world, 1),
        "pct_hbm_peak_decompressed": round(100.0 * value / world / HBM_PEAK_GBPS, 2),
        "roofline": roofline(work, mean_launch_ms, committed_traffic(work.kernel, work.algorithmic_bytes, getattr(work, "workload_name", None))),
    }

    if first_call is not None:
        line["first_call"] = first_call
    if rank == 0 and world == 1 and args.schemes != "none" and args.extras == "all" and hasattr(work, "shard_function"):
        line["small_columns"] = {args.workload: small_columns(args.workload, work, device)}
    if args.gather and world > 1:
        count = min(n, 1 << args.gather_log2n)
        try:
            line["gather"] = time_gather(work.output.view(-1)[:count], [count] * world, rank, world, device, barrier)
        except Exception as problem:
            line["gather"] = {"failed": repr(problem)}

    if rank == 0 and world == 1:
        ceiling = measured_copy_ceiling()
        if "copy_GBps" in ceiling:
            line["roofline"]["measured_copy_GBps"] = ceiling["copy_GBps"]
            line["roofline"]["frac_of_measured_copy"] = round(line["roofline"]["achieved"] / ceiling["copy_GBps"], 4)
            line["roofline"]["measured_widening_stream_GBps"] = ceiling["widen_1_to_4_GBps"]
            line["roofline"]["frac_of_measured_widening_stream"] = round(line["roofline"]["achieved"] / ceiling["widen_1_to_4_GBps"], 4)
            line["roofline"]["measured_ceiling"] = ceiling
            line["roofline"]["measured_ceiling_note"] = ("the library's tile loop (stream_tiles::transform, 16 B per lane) with the identity in place of a decoder, 2^30 elements, median of 10 launches, "
                                                         "best of the decoders' launch shapes (residency throttle x store policy): a 4 B : 4 B copy, and a 1 B : 4 B widening stream (the headline's traffic shape)")
        else:
            line["roofline"]["measured_copy"] = ceiling
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(work, min(n, 1 << args.cpu_sample_log2))

    # ---- the other single-GPU configs, same treatment (rank 0 of a 1-GPU run only) ----------------
    if world == 1 and args.schemes != "none":
        names = ["dictionary_zipf", "frame_of_reference", "delta_patched", "run_length_encoding", "run_position_encoding_relative",
                 "run_position_encoding_absolute", "discard_zero_bytes_variable"]
        if args.schemes == "coverage":
            names += ["discard_zero_bytes_variable_3bit", "discard_zero_bytes_fixed", "model", "incidence_bitmaps",
                      "delta_fused_patches", "dictionary_fused_patches", "frame_of_reference_fused_patches",
                      "delta_compressed_patches", "delta_fused_compressed_patches", "model_fused_patches", "discard_zero_bytes_fixed_fused_patches"]
        elif args.schemes != "all":
            names = [s for s in args.schemes.split(",") if s]
        line["step_runs"] = work.step_runs
        del work
        torch.cuda.empty_cache()
        schemes_out = {}
        for name in names:
            w = make_workload(name, n, device)
            w.run(torch.cuda.current_stream().cuda_stream)
            w.verify(oracle)
            el, launch_ms = timed_steps(w, max(5, args.steps // 2), 2, barrier)
            steps = max(5, args.steps // 2)
            entry = {"instantiation": w.name, "elements": w.n,
                     "decompressed_GBps": round(w.decompressed_bytes * steps / el / 1e9, 1),
                     "ms_per_step": round(el / steps * 1e3, 4),
                     "roofline": roofline(w, launch_ms, committed_traffic(w.kernel, w.algorithmic_bytes, getattr(w, "workload_name", None))), "bit_exact": True}
            if name in RUN_DECODER_WORKLOADS:
                # THE number of a run decoder's entry is the one with its input COLD (three columns of different seeds decoded in turn):
                # one column's 200 MB of runs, decoded again and again, come out of the 256 MiB Infinity Cache — a rate only a benchmark sees
                pc = placement_and_cache(name, w, n, device, steps)
                warm = entry["roofline"]
                entry["placement_and_cache"] = pc
                entry["roofline"] = roofline(w, pc["ms_three_columns_in_turn"], committed_traffic(w.kernel, w.algorithmic_bytes, getattr(w, "workload_name", None)))
                entry["roofline"].update({"median_launch_ms": None, "input": "cold: three columns of different seeds decoded in turn into one output, one HIP event pair around the %d launches" % (3 * steps),
                                          "frac_input_in_infinity_cache": warm["frac"], "mean_launch_ms_input_in_infinity_cache": warm["mean_launch_ms"]})
                entry["decompressed_GBps_input_in_infinity_cache"], entry["ms_per_step_input_in_infinity_cache"] = entry["decompressed_GBps"], entry["ms_per_step"]
                entry["decompressed_GBps"] = round(w.decompressed_bytes / (pc["ms_three_columns_in_turn"] * 1e-3) / 1e9, 1)
                entry["ms_per_step"] = pc["ms_three_columns_in_turn"]
            if args.extras == "all" and name in ("frame_of_reference", "delta_patched", "run_length_encoding", "discard_zero_bytes_variable"):
                line.setdefault("small_columns", {})[name] = small_columns(name, w, device)
            # the same stream with the identity in place of the decoder, where the workload's traffic has one of the measured shapes
            shape = {"dictionary_zipf": "widen_1_to_4", "frame_of_reference": "widen_2_to_4", "delta_patched": "widen_2_to_8", "discard_zero_bytes_fixed": "widen_2_to_8"}.get(name)
            measured = line["roofline"].get("measured_ceiling", {}) if shape else {}
            if shape and (shape + "_GBps") in measured:
                entry["roofline"]["measured_stream_GBps"] = measured[shape + "_GBps"]
                entry["roofline"]["frac_of_measured_stream"] = round(entry["roofline"]["achieved"] / measured[shape + "_GBps"], 4)
            if not args.no_cpu_baseline:
                entry["cpu_baseline"] = cpu_baseline(w, min(w.n, 1 << (args.cpu_sample_log2 - 1)))
            entry["step_runs"] = w.step_runs
            schemes_out[name] = entry
            del w
            torch.cuda.empty_cache()
        line["schemes"] = schemes_out

    # ---- the reference's three modes of use on two of the columns whose wrappers pick among several kernels (N = 1 only) ----
    if world == 1 and args.schemes != "none" and args.extras == "all":
        try:
            line["modes"] = modes_of_use(n, device, max(5, args.steps // 2))
        except Exception as problem:   # (the C++ shim is test infrastructure: its absence must not cost the bench line)
            line["modes"] = {"skipped": repr(problem)}

    # ---- BASELINE.json configs[3] / configs[4] as stated: one column cut over the ranks (strong scaling), at every N ----
    if args.sharded != "none":
        wanted = [("run_length_encoding", 30, False), ("run_position_encoding_relative", 30, False), ("discard_zero_bytes_variable", 31, True)] if args.sharded == "default" else \
                 [(item.split(":")[0], int(item.split(":")[1]) if ":" in item else args.log2n, args.gather) for item in args.sharded.split(",") if item]
        if "work" in dir():
            del work
        torch.cuda.empty_cache()
        line["sharded"] = {}
        for name, log2_total, with_gather in wanted:
            entry = strong_scaling_leg(name, min(log2_total, args.log2n + 1), args, rank, world, device, backend, barrier, oracle, gather=with_gather)
            entry["roofline_rank0"] = entry.pop("this_rank_roofline")
            line["sharded"][name] = entry
        if world == 1:
            line["predicted_strong_efficiency_at_8"] = {name: entry.get("predicted_strong_efficiency_at_8") for name, entry in line["sharded"].items()}
            line["predicted_strong_efficiency_at_8"]["note"] = ("shard-of-8 rate / column rate on ONE GPU (sharded.*.predicted_strong_efficiency_note): a prediction of what the decode "
                                                                "itself would scale like.  The scaling curve at N = 2, 4, 8 is UNMEASURED: no multi-GPU node has been available to any round")

    if rank == 0:
        emit(line, args.details)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
