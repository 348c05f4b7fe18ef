cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r6i
rocprofv3 --kernel-trace --stats -d gpurun_out/r6i/trace -o u --output-format csv -- python3 tools/unaligned_bases.py 27 > gpurun_out/r6i/log.txt 2>&1
f=$(find gpurun_out/r6i/trace -name "*kernel_stats.csv" | head -1)
grep -i "window\|delta" $f | cut -c1-260
