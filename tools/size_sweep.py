"""What a decoder makes of SMALL columns (round-4 verdict, next 1a).  The seven BASELINE workloads at 2^16 ... 2^30 elements, every figure with the
input COLD: a size's "columns" are consecutive shards of one big column (whole segments, anchors rebased — libgiddy_amd.sharding), as many of
them as move >= 1.5 GiB, so no launch finds its input or its output's lines in the 256 MiB Infinity Cache; each is decoded into its own range of
the big output.

    python tools/size_sweep.py [WORKLOAD,... | all] [min_log2=16] [max_log2=30] [--batch] [--quick]

Per workload and size, % of the 8 TB/s HBM peak over the algorithmic bytes:
  stream   the columns launched back to back from the host (what a loop over chunks gets: launch overhead included)
  graph    the same launches replayed from ONE hipGraph (no host in the way: the device side of a launch alone — ramp and tail)
  first    ONE launch after >= 0.6 s of idle, by events / by the host's clock around launch + synchronize (microseconds)
  batch    (--batch) all the columns as pieces of ONE launch: giddy_enqueue_batch
and at the end the size below which each decoder drops under 70 % and under 50 %.
"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench

ALL = ("dictionary", "frame_of_reference", "delta_patched", "run_length_encoding", "run_position_encoding_relative",
       "run_position_encoding_absolute", "discard_zero_bytes_variable")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
with_batch = "--batch" in sys.argv
quick = "--quick" in sys.argv   # no graph, no launch after idle
names = ALL if not args or args[0] == "all" else tuple(args[0].split(","))
min_log2 = int(args[1]) if len(args) > 1 else 16
max_log2 = int(args[2]) if len(args) > 2 else 30
COLD_BYTES = 3 << 29          # bytes the rotated columns of one size move at least (6 x the Infinity Cache)
MAX_COLUMNS = 4096
IDLE_SECONDS = 0.6

device = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def pct(nbytes, ms): return nbytes / ms / 1e6 / 80


def events_ms(run, repetitions=3):
    times = []
    for _ in range(repetitions):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record(); run(); b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b))
    return sorted(times)[len(times) // 2]


def bytes_of(shard, out):
    return out.numel() * out.element_size() + sum(v.numel() * v.element_size() for v in shard.fields.values() if isinstance(v, torch.Tensor))


table, batch_table = {}, {}
for name in names:
    top = min(max_log2, 28) if name.startswith("discard_zero_bytes_variable") else max_log2
    w = bench.make_workload(name, 1 << top, device, cap=1 << top)
    fields = w.column_fields()
    print("# %s: one column of 2^%d elements, %.3f GB algorithmic; its shards are the smaller columns" % (w.name, top, w.algorithmic_bytes / 1e9), flush=True)
    for log2n in range(min_log2, top + 1):
        if log2n < 22 and (log2n - min_log2) % 2:   # every second size below 2^22
            continue
        pieces = 1 << (top - log2n)
        if pieces == 1:
            # the whole column: three of them in turn would be the honest cold figure (bench.py has it); here the one column, repeated
            config = w.launch_fields(w.output, fields, stream, None)
            ms = events_ms(lambda: w.launch_fields(w.output, fields, stream, config), 5)
            print("%-34s 2^%d  x1 (repeated)  stream %5.1f %%   %.1f us" % (name, log2n, pct(w.algorithmic_bytes, ms), ms * 1e3), flush=True)
            table.setdefault(name, {})[log2n] = pct(w.algorithmic_bytes, ms)
            continue
        per_piece = w.algorithmic_bytes / pieces
        m = int(min(pieces, MAX_COLUMNS, max(3, math.ceil(COLD_BYTES / per_piece))))
        shards = [w.shard_function(fields, r, pieces) for r in range(m)]
        outs = [w.output[s.start:s.stop] for s in shards]
        total = sum(bytes_of(s, o) for s, o in zip(shards, outs))
        configs = [w.launch_fields(o, s.fields, stream, None) for o, s in zip(outs, shards)]
        torch.cuda.synchronize()

        def all_columns():
            for o, s, c in zip(outs, shards, configs): w.launch_fields(o, s.fields, stream, c)

        stream_ms = events_ms(all_columns)
        graph_pct = float("nan")
        try:
            if quick: raise RuntimeError("skipped (--quick)")
            side = torch.cuda.Stream()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                inner = torch.cuda.current_stream().cuda_stream
                for o, s, c in zip(outs, shards, configs): w.launch_fields(o, s.fields, inner, c)
            graph.replay(); torch.cuda.synchronize()
            graph_ms = events_ms(graph.replay)
            graph_pct = pct(total, graph_ms)
            del graph
        except Exception as e:   # (a wrapper that needs scratch refuses a capturing stream)
            torch.cuda.synchronize()
            if not quick: print("#   graph capture failed at 2^%d: %s" % (log2n, str(e).splitlines()[0][:120]), flush=True)
        batch_text = ""
        if with_batch:
            from libgiddy_amd import schemes
            with schemes.into_batch() as batched:   # (the same calls, added to one batch instead of launched)
                for o, s in zip(outs, shards): w.launch_fields(o, s.fields, stream, None)
            batched.enqueue(stream); torch.cuda.synchronize()
            batch_ms = events_ms(lambda: batched.enqueue(stream))
            batch_text = "   batch %5.1f %% (%d workgroups)" % (pct(total, batch_ms), batched.num_blocks)
            if hasattr(w, "values"):
                assert bool((w.output[:shards[-1].stop] == w.values[:shards[-1].stop]).all())
        # one launch after idle: a column in the middle of the rotation (not touched for >= COLD_BYTES)
        k = m // 2
        first_events, first_wall = [], []
        for _ in range(3):
            if not quick: time.sleep(IDLE_SECONDS)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            a.record(); w.launch_fields(outs[k], shards[k].fields, stream, configs[k]); b.record()
            torch.cuda.synchronize()
            first_wall.append((time.perf_counter() - t0) * 1e6); first_events.append(a.elapsed_time(b) * 1e3)
        one = bytes_of(shards[k], outs[k])
        fe, fw = sorted(first_events)[1], sorted(first_wall)[1]
        print("%-34s 2^%d  x%-4d  stream %5.1f %%  (%.1f us each)   graph %5.1f %%%s   first after idle: %.1f us by events = %4.1f %%, %.0f us on the host's clock"
              % (name, log2n, m, pct(total, stream_ms), stream_ms * 1e3 / m, graph_pct, batch_text, fe, pct(one, fe / 1e3), fw), flush=True)
        table.setdefault(name, {})[log2n] = max(pct(total, stream_ms), graph_pct if graph_pct == graph_pct else 0)
        if with_batch: batch_table.setdefault(name, {})[log2n] = pct(total, batch_ms)
        if hasattr(w, "values") and log2n == min_log2:
            torch.cuda.synchronize()
            hi = shards[-1].stop
            assert bool((w.output[:hi] == w.values[:hi]).all())
        del shards, outs, configs
    del w, fields
    torch.cuda.empty_cache()

for title, results in (("a launch per column (best of stream / graph)", table), ("ONE launch for all the columns of a size (batch)", batch_table)):
  if not results: continue
  print("# the smallest size at which a decoder still holds 70 %% / 50 %% of the HBM peak, %s:" % title)
  for name, by_size in results.items():
    def smallest(bar):
        held = [k for k in sorted(by_size) if all(by_size[j] >= bar for j in by_size if j >= k)]
        return ("2^%d" % held[0]) if held else "never"
    print("#   %-34s 70 %%: %-6s 50 %%: %-6s" % (name, smallest(70), smallest(50)))
