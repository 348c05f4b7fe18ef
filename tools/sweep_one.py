"""python tools/sweep_one.py PATTERN [log2n] — tools/sweep_all.py restricted to the registered keys matching PATTERN, as a plain python program
(what rocprofv3 wants after its `--`; tools/sweep_some.sh is the shell form)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import sweep_all
from libgiddy_amd import registered as r
pattern = re.compile(sys.argv[1])
keys = [k for k in r.kernel_t.registered_keys() if pattern.search(k)]
r.kernel_t.registered_keys = staticmethod(lambda: keys)
sys.argv = [sys.argv[0]] + sys.argv[2:]
sweep_all.main()
