"""Developer tool (timing build, `make lib-timing`): per-phase times inside sort_pass_kernel, random 64-bit keys.
    NBODY_AMD_LIB=$PWD/n-body_amd/lib/libnbody_amd_timing.so python3 tools/sort_timing.py [N ...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
for n in [int(a) for a in sys.argv[1:]] or [1_000_000]:
    rng = np.random.default_rng(1)
    k = rng.integers(0, 2 ** 64, size=n, dtype=np.uint64)
    nb.sort_pairs_u64(k, np.arange(n, dtype=np.int32), 3)
