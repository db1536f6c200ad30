"""Developer tool: Barnes-Hut BUILD time alone (useParallel = false, benchmark.cpp:362) over sizes and inputs.
usage: python tools/bh_build_time.py [disc|plummer] N [N ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
O = e.load_oracle()
kind = sys.argv[1] if len(sys.argv) > 1 else "disc"
sizes = [int(a) for a in sys.argv[2:]] or [10000, 65536, 1_000_000]
for n in sizes:
    b = O.init_plummer(n, scale=50.0, seed=12345) if kind == "plummer" else O.init_disc(n, seed=12345)
    with nb.NBodySim(n, theta=0.5, useBarnesHut=True, useParallel=False) as sim:
        sim.upload(*b.arrays())
        sim.compute_accel(); sim.sync()
        sim.profile(True).profile_reset()
        for _ in range(10): sim.compute_accel()
        sim.sync()
        build, c = sim.profile_read(2)
        root = sim.bh_tree(max_nodes=1)[0]
        print(f"{kind} N = {n:9d}: build {build / c * 1e3:9.1f} us   cells {sim.bh_counters()['internal'] if 'internal' in sim.bh_counters() else '-'}  root mass {root['mass']!r}", flush=True)
