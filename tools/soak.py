"""Developer tool: long runs through the regimes a short test never reaches — the reference's unbound disc leaves the
fixed Barnes-Hut root box within ~15 steps, so the tree shrinks to a handful of bodies and finally to none."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
for n, bh, steps, kw in ((65536, True, 400, {}), (10007, True, 400, dict(ref_compat=2)), (65536, True, 200, dict(adaptive_box=True)),
                         (20000, False, 300, {}), (1000000, True, 60, dict(bh_fast_math=True))):
    sim = nb.NBodySim(n, useBarnesHut=bh, **kw); sim.upload(*ic.disc(n, seed=12345))
    t0 = time.perf_counter()
    inside = []
    for k in range(steps // 20):
        sim.step(20).sync()
        st = sim.download()
        assert all(np.all(np.isfinite(st[f])) for f in st), (n, bh, k)
        inside.append(int(((np.abs(st["posX"]) <= 400) & (np.abs(st["posY"]) <= 400)).sum()))
    nodes = len(sim.bh_tree(max_nodes=1)) and sim.bh_nodes_total if bh else 0
    print(f"N={n} {'BH' if bh else 'N^2'} {kw}: {steps} steps in {time.perf_counter() - t0:.2f}s, bodies inside the box every 20 steps "
          f"{inside[:4]}..{inside[-1]}, last tree {nodes} nodes, diagnostics {sim.diagnostics()['Ekin']:.4g}", flush=True)
    sim.close()
print("soak ok")
