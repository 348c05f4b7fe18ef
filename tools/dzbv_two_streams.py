"""DZB-variable, a long column as consecutive PIECES: on one stream (what the wrapper does: a launch boundary re-aligns the XCDs' write fronts, and costs a
tail and a ramp) against alternately on TWO streams, the first piece half as long, so that one piece's tail and the next one's ramp overlap the other
stream's piece in mid-flight (round 4 experiment).
    python tools/dzbv_two_streams.py [log2n=31]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import workloads as w

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 31
n = 1 << log2n
device = torch.device("cuda", 0)
os.environ.setdefault("GIDDY_DZBV_LAUNCH_BYTES_OFF", "1")
work = w.DiscardZeroBytesVariable(n, device)
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def pct(ms): return work.algorithmic_bytes / ms / 1e6 / 80


def timed(run, steps=8):
    for _ in range(2): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(main)
    for _ in range(steps): run()
    b.record(main); b.synchronize()
    return a.elapsed_time(b) / steps


ms = timed(lambda: work.run(main.cuda_stream))
print("dzbv<8,8,4> 2^%d, the wrapper (pieces of ~1 GiB on one stream): %.3f ms = %.1f %%" % (log2n, ms, pct(ms)), flush=True)
fields = work.column_fields()
for log2piece in (25, 26, 27):
    pieces = n >> log2piece
    if pieces < 4:
        continue
    # cuts: 2 * pieces half-pieces; piece 0 = one half, then whole pieces, the rest
    halves = [w.Sharded(work, r, 2 * pieces) for r in range(2 * pieces)]

    def launch(shards, stream):
        for sh in shards:
            sh.run(stream.cuda_stream)

    groups = [[halves[0]]] + [halves[i:i + 2] for i in range(1, 2 * pieces - 1, 2)] + [[halves[-1]]]

    def one_stream():
        for g in groups: launch(g, main)

    def two_streams():
        side.wait_stream(main)
        for i, g in enumerate(groups): launch(g, main if i % 2 == 0 else side)
        main.wait_stream(side)

    a, b = timed(one_stream), timed(two_streams)
    work.output.zero_(); two_streams(); torch.cuda.synchronize()
    ok = bool((work.output == work.values).all())
    print("   pieces of 2^%d elements (%4d MiB of output): one stream %.3f ms = %.1f %%;  two streams, staggered %.3f ms = %.1f %%  %s"
          % (log2piece, (8 << log2piece) >> 20, a, pct(a), b, pct(b), "ok" if ok else "MISMATCH"), flush=True)
