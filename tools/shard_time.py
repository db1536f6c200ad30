"""Developer tool: one rank's share of the N = 2^20 naive step for P = 1, 2, 4, 8 on ONE GPU (predicts the
strong-scaling efficiency of the force kernel; the all-gather is not included)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = 1 << 20
b = ic.disc(n, seed=12345)
for P in (1, 2, 4, 8):
    sim = nb.NBodySim(n, shard=nb.shard_range(n, P, P - 1)); sim.upload(*b)
    sim.step(1).sync()
    sim.profile(True).profile_reset()
    reps = 3
    t0 = time.perf_counter(); sim.step(reps).sync(); wall = (time.perf_counter() - t0) / reps * 1e3
    f, c = sim.profile_read(0); g, c1 = sim.profile_read(1)
    inter = n / P * n
    print(f"P={P}: wall {wall:.2f} ms/step, force {f/c:.2f} ms = {20*inter/(f/c*1e-3)/1e12/157.3*100:.1f}% of peak, "
          f"integrate {g/c1*1e3:.1f} us, info {sim.info()}", flush=True)
    sim.close()
