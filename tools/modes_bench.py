"""What each of the reference's modes of use gets: the BASELINE columns whose wrappers pick among several kernels, and one
long-segment column per scheme, decoded through
    A   the factory's type-erased wrapper over the C ABI (what bench.py, the sweeps and the profiles time),
    B   an instantiated wrapper + cuda::kernel::enqueue_launch(kernel, stream, config, typed parameters...),
    C   the kernel template launched directly (the kernel of the reference's signature, configuration from its parameter struct),
    C+  <scheme>::launch<...>() — the one-line replacement for C's launch —,
each mode's bits compared with A's, times by HIP events around `--steps` launches on the launch stream.
    python tools/modes_bench.py [--log2n 30] [--steps 10] > profiles/r04_modes_bench.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

import modes
import workloads as w

PEAK = 8.0e12


def time_ms(run, stream, steps, warmup=3):
    for _ in range(warmup):
        run(stream.cuda_stream)
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    begin.record(stream)
    for _ in range(steps):
        run(stream.cuda_stream)
    end.record(stream)
    end.synchronize()
    return begin.elapsed_time(end) / steps


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--log2n", type=int, default=30)
    parser.add_argument("--steps", type=int, default=10)
    parser.add_argument("--only", default="")
    args = parser.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    n = 1 << args.log2n
    stream = torch.cuda.current_stream()

    def delta(n, L):
        work = w.DeltaPatched(n, device, segment_length=L)
        return {"compressed_input": work.deltas, "anchor_values": work.anchors, "length": n, "segment_length": L}, work

    columns = [
        # (label, scheme, constructor)
        ("C3 Delta <4,8,2>, L = 4096", "delta<4,8,2>", lambda: delta(n, 4096)),
        ("Delta <4,8,2>, L = 2^20 (2^28 elements)", "delta<4,8,2>", lambda: delta(min(n, 1 << 28), 1 << 20)),
        ("Delta <4,8,2>, ONE segment (2^28 elements)", "delta<4,8,2>", lambda: delta(min(n, 1 << 28), min(n, 1 << 28))),
        ("C4 RLE <4,4,2>, mean run 32, L = 4096", "rle<4,4,2>", lambda: (lambda k: (k.column_fields(), k))(w.RunLength(n, device))),
        ("RLE <4,4,2>, mean run 32, L = 65536", "rle<4,4,2>", lambda: (lambda k: (k.column_fields(), k))(w.RunLength(n, device, segment_length=65536))),
        ("RLE <4,4,2>, mean run 4, L = 4096", "rle<4,4,2>", lambda: (lambda k: (k.column_fields(), k))(w.RunLength(min(n, 1 << 28), device, mean_run=4))),
        ("C4 RPE <4,4,2,true>, mean run 32, L = 4096", "rpe<4,4,2,true>", lambda: (lambda k: (k.column_fields(), k))(w.RunPosition(n, device, relative=True))),
        ("RPE <4,4,2,true>, mean run 32, L = 65536", "rpe<4,4,2,true>", lambda: (lambda k: (k.column_fields(), k))(w.RunPosition(n, device, segment_length=65536, relative=True))),
        ("C4 RPE <4,4,4,false>, mean run 32, L = 4096", "rpe<4,4,4,false>", lambda: (lambda k: (k.column_fields(), k))(w.RunPosition(n, device, relative=False))),
        ("RPE <4,4,4,false>, mean run 32, L = 2^20", "rpe<4,4,4,false>", lambda: (lambda k: (k.column_fields(), k))(w.RunPosition(n, device, segment_length=1 << 20, relative=False))),
        ("C5 DZB-V <8,8,4> as ONE column of 2n elements, L = 1024", "dzbv<8,8,4>", lambda: (lambda k: (k.column_fields(), k))(w.DiscardZeroBytesVariable(2 * n, device))),
        ("DZB-V <8,8,4>, L = 2^16 (2^28 elements)", "dzbv<8,8,4>", lambda: (lambda k: (k.column_fields(), k))(w.DiscardZeroBytesVariable(min(n, 1 << 28), device, segment_length=1 << 16))),
    ]
    print("modes of use, %d steps per figure, peak %.1f TB/s; ms per decode and %% of peak (algorithmic bytes / time); 'same bits' = the mode's output equals mode A's" % (args.steps, PEAK / 1e12))
    print("%-58s %18s %18s %18s %18s   %s" % ("column", "A (C ABI)", "B (typed)", "C (bare kernel)", "C+ (launch<>)", "B - A, C - A (points)"))
    for label, scheme, make in columns:
        if args.only and args.only not in label:
            continue
        fields, work = make()
        out = work.output
        torch.cuda.synchronize()
        cells, fractions = [], {}
        reference = None
        for mode in ("A", "B", "C", "C+"):
            run = modes.launcher(scheme, fields, out, mode)
            out.fill_(-1)
            run(stream.cuda_stream)
            torch.cuda.synchronize()
            if reference is None:
                reference = out.clone() if out.numel() * out.element_size() <= (8 << 30) else None
                digest = w.checksum(out)
                same = True
            else:
                same = (w.checksum(out) == digest) if reference is None else bool((out == reference).all())
            ms = time_ms(run, stream, args.steps)
            fractions[mode] = work.algorithmic_bytes / (ms * 1e-3) / PEAK
            cells.append("%7.3f ms %5.1f %%%s" % (ms, 100 * fractions[mode], "" if same else " BITS DIFFER"))
        print("%-58s %18s %18s %18s %18s   %+.1f, %+.1f" % (label, *cells, 100 * (fractions["B"] - fractions["A"]), 100 * (fractions["C"] - fractions["A"])), flush=True)
        del fields, work, out, reference
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
