"""GPU box: every decoder that takes a segment length (or an anchoring period), over lengths that are and are not powers of two or whole rows —
where are the holes?   python tools/segment_length_sweep.py [log2 n] [scheme ...]       (profiles/r0N_segment_lengths.txt)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
only = sys.argv[2:]
LENGTHS = [int(x) for x in os.environ.get("LS", "64 128 256 520 1000 1024 3000 4096 5000 10000 16384 65536").split()]
CASES = [
    ("frame_of_reference<4,4,2,linear>", lambda n, L: sw.run_for(4, 4, 2, "linear", 4, n, L), 4),
    ("frame_of_reference<4,4,1,constant>", lambda n, L: sw.run_for(4, 4, 1, "constant", 4, n, L), 4),
    ("run_length_encoding<4,4,2,4>", lambda n, L: sw.run_rle(4, 4, 2, 4, n, L), 1),
    ("run_length_encoding<4,8,1,4>", lambda n, L: sw.run_rle(4, 8, 1, 4, n, L), 1),
    ("run_position_encoding<4,4,2,relative>", lambda n, L: sw.run_rpe(4, 4, 2, True, 4, n, L), 1),
    ("run_position_encoding<4,4,4,absolute>", lambda n, L: sw.run_rpe(4, 4, 4, False, 4, n, L), 1),
    ("discard_zero_bytes_variable<4,4,4>", lambda n, L: sw.run_dzb_variable(4, 4, 4, n, L), 16),
    ("discard_zero_bytes_variable<4,8,4>", lambda n, L: sw.run_dzb_variable(4, 8, 4, n, L, bits=3), 8),
]
for name, run, multiple in CASES:
    if only and not any(o in name for o in only):
        continue
    for L in LENGTHS:
        if L % multiple:
            continue
        n = (1 << log2n) // L * L
        try:
            ms, nbytes, ok = run(n, L)
            print("%-40s L=%-6d %8.3f ms %5.1f%% %s" % (name, L, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
        except Exception as e:   # a length a launch refuses (the reference's own limits) is a finding too
            print("%-40s L=%-6d refused: %s" % (name, L, str(e)[:100]), flush=True)
