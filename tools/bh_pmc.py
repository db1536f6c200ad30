"""Developer tool: a few Barnes-Hut force evaluations (eager launches, so every kernel is a dispatch of its own) for a
rocprofv3 --kernel-trace or --pmc pass.

    python3 tools/bh_pmc.py [N] [THETA] [--plummer] [--world P] [--fast] [--reps R]

--world P: rank 0's share of P ranks (the tree build is replicated, the walk covers 1/P of the targets)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NBODY_NO_GRAPH", "1")
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 1_000_000
theta = float(args[1]) if len(args) > 1 else 1.0
world = int(sys.argv[sys.argv.index("--world") + 1]) if "--world" in sys.argv else 1
reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 3
if "--plummer" in sys.argv:
    O = e.load_oracle()                       # the build's own Plummer generator lives with the test infrastructure
    bodies = O.init_plummer(n, scale=50.0, seed=12345).arrays()
else:
    bodies = ic.disc(n, seed=12345)
span = nb.shard_range(n, world, 0)
sim = nb.NBodySim(n, theta=theta, useBarnesHut=True, bh_fast_math="--fast" in sys.argv, **(dict(shard=span) if world > 1 else {}))
sim.upload(*bodies)
for _ in range(reps): sim.compute_accel()
sim.sync(); print(n, theta, world, sim.bh_counters()); sim.close()
