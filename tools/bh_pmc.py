"""Developer tool: three Barnes-Hut force evaluations at N = 10^6 (theta 1) for a rocprofv3 --pmc pass."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = 1_000_000
sim = nb.NBodySim(n, useBarnesHut=True, bh_fast_math="--fast" in sys.argv); sim.upload(*ic.disc(n, seed=12345))
for _ in range(3): sim.compute_accel()
sim.sync(); print(sim.bh_counters()); sim.close()
