"""Developer tool: the one-launch centre-of-mass pass (bh_com_climb_kernel) is evaluated by whichever thread arrives
last at each cell.  Its RESULT must not depend on that: build the same tree many times and compare every node record
with the first build's, bit for bit.  usage: python tools/bh_com_repeat.py [N] [builds]"""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
O = e.load_oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
b = O.init_plummer(n, scale=50.0, seed=7)
with nb.NBodySim(n, theta=1.0, useBarnesHut=True, useParallel=False) as sim:
    sim.upload(*b.arrays())
    first, differ = None, 0
    for k in range(reps):
        sim.compute_accel().sync()
        t = sim.bh_tree()
        h = hashlib.sha256(np.ascontiguousarray(t[["mass", "comX", "comY"]]).tobytes()).hexdigest()
        if first is None: first = h
        differ += h != first
    print(f"N = {n}: {reps} builds of {len(t)} nodes, {differ} differ from the first (sha256 of mass/comX/comY of every node)", flush=True)
