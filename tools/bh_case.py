"""Developer tool (test infrastructure): compare GPU and oracle Barnes-Hut on a saved case (tools/cases/*.npz)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()
from test_gpu_bh import canonical
d = np.load(sys.argv[1]); theta = float(d["theta"])
n = len(d["posX"]); b = O.Bodies(n)
for f in b.FIELDS: setattr(b, f, d[f].copy())
with nb.NBodySim(n, theta=theta, useBarnesHut=True) as sim:
    sim.upload(*b.arrays()); sim.compute_accel().sync()
    ax, ay = sim.accel(); cnt = sim.bh_counters(); tree = sim.bh_tree()
ot = O.Tree(n).build(b, 400.0); ox, oy, ocnt = ot.accel(b, theta)
print("nodes", len(tree), ot.nodes_used, "counters", cnt, ocnt)
cg, co = canonical(tree), canonical(ot.pool())
if cg.shape == co.shape:
    diff = np.argwhere(cg != co)
    print("canonical rows differing:", len(set(diff[:, 0])), "first", diff[:5].tolist())
    for r in sorted(set(diff[:, 0]))[:5]:
        print(" gpu", cg[r]); print(" ora", co[r])
else:
    k = min(len(cg), len(co)); first = next((i for i in range(k) if not np.array_equal(cg[i], co[i])), k)
    print("shapes", cg.shape, co.shape, "first differing row", first)
    for i in range(max(0, first - 1), min(k, first + 4)): print(i, " gpu", cg[i]); print(i, " ora", co[i])
bad = np.flatnonzero((ax != ox.astype(np.float32)) | (ay != oy.astype(np.float32)))
print("bodies with differing accel:", len(bad), bad[:16])
pool = ot.pool()
def inst(p):
    leaf = p[p["children"][:, 0] == -1]; bi = leaf["bodyIndex"]; bi = bi[bi >= 0]
    return np.bincount(bi, minlength=n)
ig, io = inst(tree), inst(pool)
d = np.flatnonzero(ig != io)
print("bodies with different instance counts:", len(d))
for i in d[:8]:
    x, y = b.posX[i], b.posY[i]
    # replay the descent in float32 to find tie levels
    tl = np.float32(-400); br = np.float32(400); tly = np.float32(-400); bry = np.float32(400); ties = []
    for l in range(31):
        mx = np.float32((tl + br) / np.float32(2)); my = np.float32((tly + bry) / np.float32(2))
        if x == mx: ties.append(("x", l))
        if y == my: ties.append(("y", l))
        if x > mx: tl = mx
        else: br = mx
        if y > my: tly = my
        else: bry = my
    print(f" body {i}: pos ({x!r}, {y!r}) gpu instances {ig[i]} oracle {io[i]} ties {ties}")
err = np.hypot(ax - ox, ay - oy) / np.maximum(np.hypot(ox, oy), 1e-30)
print("accel rel diff vs oracle: median %.3g p99 %.3g max %.3g; bodies affected %d of %d; root mass gpu %.6g oracle %.6g" % (
    np.median(err), np.percentile(err, 99), err.max(), int((err > 0).sum()), n, tree[0]["mass"], pool[0]["mass"]))
print("instances gpu %d oracle %d" % (ig.sum(), io.sum()))
