"""development aid (GPU box): run lengths in long segments, the wrapper's choice against the pre-pass as a launch of its own (GIDDY_RUN_FUSED_PREPASS=0) — run once per setting,
the switches are read once per process.     python tools/run_long_segments.py [log2 n ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
label = "fused" if os.environ.get("GIDDY_RUN_FUSED_PREPASS", "1") != "0" else "separate"
for log2n in [int(x) for x in sys.argv[1:]] or [27, 28, 30]:
    for L in (32768, 65536, 131072, 524288):
        for mean_run in (16, 32, 100, 1000):
            n = (1 << log2n) // L * L
            which = os.environ.get("SCHEMES", "rle").split()
            if "rle" in which:
                ms, nbytes, ok = sw.run_rle(4, 4, 2, 4, n, L, mean_run)
                print("%-9s rle<4,4,2,4> 2^%d L=%-7d mean run %-5d %8.3f ms %5.1f%% %s" % (label, log2n, L, mean_run, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
            if "rpe" in which:
                if L <= 65536:
                    ms, nbytes, ok = sw.run_rpe(4, 4, 2, True, 4, n, L, mean_run)
                    print("%-9s rpe<4,4,2,relative> 2^%d L=%-7d mean run %-5d %8.3f ms %5.1f%% %s" % (label, log2n, L, mean_run, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
                ms, nbytes, ok = sw.run_rpe(4, 4, 4, False, 4, n, L, mean_run)
                print("%-9s rpe<4,4,4,absolute> 2^%d L=%-7d mean run %-5d %8.3f ms %5.1f%% %s" % (label, log2n, L, mean_run, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
