"""Developer tool: time the Barnes-Hut step on the GPU (and the compiled reference on the host)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
O = e.load_oracle() if "--no-ref" not in sys.argv else None      # the compiled reference, timed beside the GPU
fast = "--fast" in sys.argv
srt = "--sorted" in sys.argv          # bh_sorted: compact groups, vote at 0.6 theta (ref_compat 0)
for n, theta in ((10000, 1.0), (65536, 1.0), (1_000_000, 1.0), (1_000_000, 0.5)):
    b = ic.disc(n, seed=12345)
    sim = nb.NBodySim(n, theta=theta, useBarnesHut=True, bh_fast_math=fast, **(dict(bh_sorted=True, ref_compat=0) if srt else {})); sim.upload(*b)
    sim.compute_accel(); sim.sync()
    sim.profile(True).profile_reset()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps): sim.compute_accel()
    sim.sync(); wall = (time.perf_counter() - t0) / reps * 1e3
    walk, c0 = sim.profile_read(0); build, c2 = sim.profile_read(2); integ, c1 = sim.profile_read(1)
    cnt = sim.bh_counters(); nodes = len(sim.bh_tree())
    line = f"{'sorted ' if srt else ''}{'fast ' if fast else 'exact'} N={n} theta={theta}: wall {wall:.3f} ms/step | build {build/c2:.3f} ms | walk {walk/c0:.3f} ms |  nodes {nodes} visits/group {cnt['visits']/(n//8):.0f} accepts/group {cnt['accepts']/(n//8):.0f}"
    if O is not None and O.ref_available() and n <= 1_000_000:
        r = O.ref_run(n, use_barnes_hut=1, nsteps=3, bodies=O.init_disc(n, seed=12345), theta=theta)
        line += f" | reference CPU {np.median(r['step_ms']):.1f} ms/step"
    print(line, flush=True)
    sim.close()
