"""Developer tool (test infrastructure): BASELINE config 5's tree at FULL size against the oracle — the oracle's serial
insertion of 2^24 bodies takes about a minute and 26 GB of host memory, too much for the test suite, fine for a
one-off: node count, root record and the accelerations of the first 8192 targets must be the oracle's bits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()
import ctypes as C
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
theta = 0.5
b = O.init_plummer(n, scale=50.0, seed=12345)
nt = 8192
with nb.NBodySim(n, theta=theta, useBarnesHut=True, shard=(0, nt)) as sim:
    sim.upload(*b.arrays()); sim.compute_accel().sync()
    ax, ay = sim.accel(); root = sim.bh_tree(max_nodes=1)[0]; total = sim.bh_nodes_total; cnt = sim.bh_counters()
print(f"GPU: {total} nodes, root mass {root['mass']!r} com ({root['comX']!r}, {root['comY']!r}), counters {cnt}", flush=True)
t0 = time.time()
ot = O.Tree(n).build(b, 400.0)
print(f"oracle tree: {ot.nodes_used} nodes in {time.time() - t0:.1f}s", flush=True)
pool0 = np.frombuffer((C.c_char * 48).from_address(O.lib().oracle_tree_pool(ot.h)), dtype=O.NODE_DTYPE)[0]
print(f"oracle root mass {pool0['mass']!r} com ({pool0['comX']!r}, {pool0['comY']!r})", flush=True)
ox = np.zeros(n); oy = np.zeros(n); oc = np.zeros(2, np.int64)
O.lib().oracle_bh_accel(ot.h, nt, b.posX, b.posY, b.radius, theta, 0, 8, ox, oy, oc)
same = np.array_equal(ax[:nt], ox[:nt].astype(np.float32)) and np.array_equal(ay[:nt], oy[:nt].astype(np.float32))
print(f"nodes equal: {total == ot.nodes_used}; root equal: {root['mass'] == pool0['mass'] and root['comX'] == pool0['comX'] and root['comY'] == pool0['comY']}; "
      f"first {nt} accelerations bit-identical: {same}; counters equal: {cnt == dict(visits=int(oc[0]), accepts=int(oc[1]))}", flush=True)
sys.exit(0 if (same and total == ot.nodes_used) else 1)
