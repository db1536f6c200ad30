"""Developer tool: Barnes-Hut build time with this repository's sort/scan (default) and with rocPRIM's (NBODY_BH_HIPCUB=1),
same process, alternating; eager launches with event timing around the whole build."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
sizes = [int(a) for a in sys.argv[1:]] or [20000, 65536, 262144, 1_000_000, 4_000_000]
for n in sizes:
    b = ic.disc(n, seed=12345)
    os.environ["NBODY_BH_HIPCUB"] = "1"                   # at creation: the context allocates rocPRIM's scratch too
    with nb.NBodySim(n, theta=1.0, useBarnesHut=True) as sim:
        sim.upload(*b)
        out = {}
        for rep in range(3):
            for name, env in (("own", None), ("rocprim", "1")):
                os.environ.pop("NBODY_BH_HIPCUB", None)
                if env: os.environ["NBODY_BH_HIPCUB"] = env
                sim.compute_accel(); sim.sync()
                sim.profile(True).profile_reset()
                for _ in range(10): sim.compute_accel()
                sim.sync()
                build, c2 = sim.profile_read(2)
                out.setdefault(name, []).append(build / c2 * 1e3)
                sim.profile(False)
        os.environ.pop("NBODY_BH_HIPCUB", None)
        print(f"N={n:9d}: build us  own sort/scan {np.round(out['own'], 1)}   rocPRIM {np.round(out['rocprim'], 1)}", flush=True)
