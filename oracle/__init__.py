"""CPU oracle for the lightweight-decompression hot path.  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never
from libgiddy_amd/ (the product path).  See giddy_oracle.c for the parity-pinning status."""
