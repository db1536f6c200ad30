"""Developer tool: phase times of bh_prep_small_kernel (library built with -DNBODY_WIDE_TIMING).
    make lib-timing && NBODY_DEBUG=1 NBODY_AMD_LIB=n-body_amd/lib/libnbody_amd_timing.so python tools/prep_timing.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sim = nb.NBodySim(n, theta=1.0, useBarnesHut=True, useParallel=False); sim.upload(*ic.disc(n, seed=12345))
for rep in range(4):
    sim.compute_accel(); sim.sync()
    sim.bh_counters()          # prints the phases on stderr
sim.close()
