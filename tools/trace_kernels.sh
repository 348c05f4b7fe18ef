#!/bin/bash
# Runs ON THE GPU BOX: per-kernel durations of one command under rocprofv3's kernel trace, grouped by kernel name
#   bash tools/trace_kernels.sh <name-filter> <command ...>      e.g.  bash tools/trace_kernels.sh fine python3 tools/long_segments.py rle 30
filter=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_trace
timeout 900 rocprofv3 --kernel-trace -d gpurun_out/prof_trace -o t --output-format csv -- "$@" > gpurun_out/trace_command.log 2>&1
python3 - "$filter" <<'PY'
import csv, collections, glob, sys
f = glob.glob("gpurun_out/prof_trace/**/t_kernel_trace.csv", recursive=True)[0]
d = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if sys.argv[1] in name: d.setdefault(name[:110], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000)
for k, v in d.items():
    print("%-110s n=%d  us: %s" % (k, len(v), " ".join("%.0f" % x for x in v[:: max(1, len(v) // 24)])))
PY
