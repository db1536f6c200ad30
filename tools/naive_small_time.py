"""Developer tool: steady-state time per naive step at the small BASELINE sizes (config 1: N = 10 000, config 2: N = 65 536),
steps enqueued back to back (the captured graph unless NBODY_NO_GRAPH is set), one sync at the end."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
for n, steps in ((10000, 2000), (65536, 300), (131072, 100)):
    b = ic.disc(n, seed=12345)
    with nb.NBodySim(n, useBarnesHut=False) as sim:
        sim.upload(*b)
        sim.step(20).sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); sim.step(steps).sync(); dt = (time.perf_counter() - t0) / steps
            best = min(best, dt)
        info = sim.info()
        print(f"N={n:7d}: {best*1e6:8.2f} us/step  {n*n/best/1e9:8.1f} G/s  {n*n/best/1e12*20/157.3*100:5.1f} % of peak  (slices {info['j_slices']}, eager launches)", flush=True)
