# development aid (GPU box): the C4 workloads (+ RLE with 4-byte lengths) for library variants
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for w in ${WORKLOADS:-run_length_encoding run_position_encoding_relative run_position_encoding_absolute}; do python tools/bench_one.py $w 30 20 | cut -c1-150; done
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
for name, fn in (("rle<4,4,4>", lambda: sw.run_rle(4, 4, 4, 4, 1 << 30)), ("rpe<8,4,8,abs>", lambda: sw.run_rpe(8, 4, 8, False, 8, 1 << 30)), ("rle<8,4,8>", lambda: sw.run_rle(8, 4, 8, 8, 1 << 30))):
    ms, nbytes, ok = fn()
    print("%-10s %-16s %8.3f ms %5.1f%% %s" % (sys.argv[1], name, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done; done
