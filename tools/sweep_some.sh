#!/bin/bash
# tools/sweep_some.sh PATTERN [log2n] : sweep only the instantiations whose key matches PATTERN
python - "$@" <<'PY'
import sys, re, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import sweep_all
from libgiddy_amd import registered as r
pat = re.compile(sys.argv[1])
keys = [k for k in r.kernel_t.registered_keys() if pat.search(k)]
r.kernel_t.registered_keys = staticmethod(lambda: keys)
sys.argv = [sys.argv[0]] + sys.argv[2:]
sweep_all.main()
PY
