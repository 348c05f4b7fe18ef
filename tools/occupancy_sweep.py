"""Development sweep: every representative instantiation timed with its launch's dynamic LDS padded to throttle
occupancy (workgroups per CU).    python tools/occupancy_sweep.py [log2n=30]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
PADS = tuple(int(p) for p in os.environ.get("PADS", "0,22,26,31,39,52,79").split(","))
current_pad = [0]
original_run = schemes._run


def padded_run(key, resolve_arguments, launch_arguments, stream, device, limits, launch_config):
    k = schemes.kernel(key)
    if launch_config is None:
        # the kernels' own occupancy throttles are refused (2 KiB is all a small dictionary needs); the pad is the throttle here
        from libgiddy_amd import registered as r
        launch_config = k.resolve_launch_configuration(device, resolve_arguments, r.launch_configuration_limits_t(dynamic_shared_memory=2048))
        launch_config.dynamic_shared_memory_size = max(launch_config.dynamic_shared_memory_size, current_pad[0] * 1024)
    k.enqueue_launch(stream, launch_config, launch_arguments)
    return launch_config


schemes._run = padded_run
CASES = []
for U, C, model in ((4, 2, "constant"), (4, 2, "linear"), (4, 1, "linear"), (2, 1, "linear")):
    CASES.append(("for<%d,%d,%s>" % (U, C, model), lambda U=U, C=C, model=model: sw.run_for(4, U, C, model, 4, n)))
for U, C in ((8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (2, 1)):
    CASES.append(("dzb_fixed<%d,%d>" % (U, C), lambda U=U, C=C: sw.run_dzb_fixed(4, U, C, n)))
for I, U in ((4, 2), (4, 4), (8, 4)):
    CASES.append(("model<%d,%d,linear>" % (I, U), lambda I=I, U=U: sw.run_model(I, U, "linear", n)))
for U, D in ((1, 1), (2, 1), (4, 1), (8, 1)):
    CASES.append(("dictionary<%d,%d>" % (U, D), lambda U=U, D=D: sw.run_dictionary(4, U, D, n)))
for U, C in ((8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (2, 1)):
    CASES.append(("delta<%d,%d>" % (U, C), lambda U=U, C=C: sw.run_delta(4, U, C, n)))
for U in (2, 4, 8):
    CASES.append(("rle<%d,2>" % U, lambda U=U: sw.run_rle(4, U, 2, 4, n)))
CASES.append(("rpe<4,2,rel>", lambda: sw.run_rpe(4, 4, 2, True, 4, n)))
CASES.append(("rpe<4,4,abs>", lambda: sw.run_rpe(4, 4, 4, False, 4, n)))
CASES.append(("rle<4,1>", lambda: sw.run_rle(4, 4, 1, 4, n)))
CASES.append(("rle<4,4>", lambda: sw.run_rle(4, 4, 4, 4, n)))
CASES.append(("rle<1,1>", lambda: sw.run_rle(4, 1, 1, 4, n)))
for U in (4, 8):
    CASES.append(("dzbv<8,%d,4>" % U, lambda U=U: sw.run_dzb_variable(8, U, 4, min(n, 1 << 28))))
for U, D in ((4, 1), (8, 1), (2, 1)):
    CASES.append(("patched_dictionary<%d,%d>" % (U, D), lambda U=U, D=D: sw.run_dictionary(4, U, D, n, patched=True)))
for U, C, model in ((4, 2, "constant"), (4, 1, "linear")):
    CASES.append(("patched_for<%d,%d,%s>" % (U, C, model), lambda U=U, C=C, model=model: sw.run_for(4, U, C, model, 4, n, patched=True)))
for U, C in ((8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (2, 1)):
    CASES.append(("patched_dzb_fixed<%d,%d>" % (U, C), lambda U=U, C=C: sw.run_dzb_fixed(4, U, C, n, patched=True)))
for U in (2, 4):
    CASES.append(("patched_model<%d,linear>" % U, lambda U=U: sw.run_model(4, U, "linear", n, patched=True)))
for U, C in ((8, 1), (8, 2), (8, 4), (4, 1), (4, 2), (2, 1)):
    CASES.append(("patched_delta<%d,%d>" % (U, C), lambda U=U, C=C: sw.run_delta(4, U, C, n, patched=True)))
CASES.append(("ib<4,2,true>", lambda: sw.run_ib(4, 4, 2, True, min(n, 1 << 28))))

only = os.environ.get("ONLY")
if only:
    CASES = [c for c in CASES if any(c[0].startswith(o) for o in only.split(","))]
print("%-24s" % "instantiation" + "".join("%9s" % ("%dK" % p) for p in PADS) + "   (% of 8 TB/s by KiB of dynamic LDS per workgroup)")
for name, fn in CASES:
    row = []
    for pad in PADS:
        current_pad[0] = pad
        try:
            ms, nbytes, ok = fn()
            row.append("%8.1f%s" % (nbytes / ms / 1e6 / 80, "" if ok else "!"))
        except Exception as e:
            row.append("%9s" % "fail")
        torch.cuda.empty_cache()
    print("%-24s" % name + "".join("%9s" % r for r in row), flush=True)
