"""Developer tool (test infrastructure: uses oracle/_ref): the reference's OWN physics clock (physicsStart..physicsEnd,
benchmark.cpp:352,513) in its unmodified build and in the drop-in build (physics block -> include/nbody.h)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
O = e.load_oracle()
for bh, n, k in ((1, 10000, 20), (0, 10000, 20), (0, 65536, 10), (1, 1000000, 5)):
    cpu = O.ref_run(n, use_barnes_hut=bh, nsteps=min(k, 5), seed=12345)["step_ms"]
    gpu = O.ref_run(n, use_barnes_hut=bh, nsteps=k, seed=12345, dropin=True)["step_ms"]
    print(f"{'BarnesHut' if bh else 'N^2':9s} N={n:8d}: reference physics clock  CPU build {np.median(cpu):9.3f} ms   "
          f"drop-in build {np.median(gpu[1:]):8.3f} ms   x{np.median(cpu) / np.median(gpu[1:]):.0f}", flush=True)
