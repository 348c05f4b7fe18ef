# development aid (GPU box): Delta at segment lengths that are not whole rows, for library variants:  [LS="1000 3000"] [TUPLES="4,8,2 4,4,2"] bash tools/run_ragged_delta.sh default ragged_old
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
n = 1 << int(os.environ.get('LOG2N', '28'))
for (I, U, C) in [tuple(int(c) for c in t.split(',')) for t in os.environ.get('TUPLES', '4,8,2 4,8,1 4,4,1 4,4,2 8,8,4 4,2,1').split()]:
    for L in [int(x) for x in os.environ.get('LS', '1000 3000 3504 4096 5000 10000').split()]:
        if L % (16 // U) and not os.environ.get("ODD"): continue
        m = n // L * L
        ms, nbytes, ok = sw.run_delta(I, U, C, m, L)
        print("%-10s delta<%d,%d,%d> L=%-6d %8.3f ms %5.1f%% %s" % (sys.argv[1], I, U, C, L, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done
