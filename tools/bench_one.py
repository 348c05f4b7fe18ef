"""python tools/bench_one.py WORKLOAD [log2n] [steps] — time one workload with the library selected by GIDDY_LIBRARY_PATH"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
name = sys.argv[1]; log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 30; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
w = bench.make_workload(name, 1 << log2n, torch.device("cuda", 0), cap=1 << log2n)
w.run(torch.cuda.current_stream().cuda_stream)
if os.environ.get("VERIFY", "1") == "1":
    from oracle import oracle
    w.verify(oracle)
el, ms = bench.timed_steps(w, steps, 2, lambda: None)
print("%-34s %-28s %8.4f ms (whole step %8.4f ms)  %7.1f GB/s algorithmic (%.1f%% of 8 TB/s)  cfg=%s" % (
    os.path.basename(os.environ.get("GIDDY_LIBRARY_PATH", "default")), name, ms, el * 1e3 / steps, w.algorithmic_bytes / ms / 1e6,
    w.algorithmic_bytes / ms / 1e6 / 80.0, w.config))
# (tools/summarize_profiles.py reads this line out of the --pmc passes' logs: algorithmic bytes of every workload, whether or not a bench line carries it)
print("WORKLOAD_RECORD " + json.dumps({"workload": name, "elements": w.n, "algorithmic_bytes_per_launch": w.algorithmic_bytes, "kernel": w.kernel,
                                       "instantiation": w.name, "mean_launch_ms": round(ms, 4)}))
