"""Developer tool: a few Barnes-Hut steps at the reference's default size (N = 10 000) for a kernel trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
sim = nb.NBodySim(n, useBarnesHut=True); sim.upload(*ic.disc(n, seed=12345))
os.environ["NBODY_NO_GRAPH"] = "1"
sim.step(5).sync()
sim.close()
