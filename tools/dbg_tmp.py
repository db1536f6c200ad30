import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()
for n in (4096, 5000, 65536):
    b = O.init_disc(n, seed=12345)
    sim = nb.NBodySim(n, useBarnesHut=False); sim.upload(*b.arrays()); sim.compute_accel()
    ax, ay = sim.accel(); info = sim.info(); sim.close()
    idx = np.arange(0, n, max(1, n // 4096), dtype=np.int32)
    rx, ry = O.naive_accel_f64_subset(b, idx)
    err = np.hypot(ax[idx]-rx, ay[idx]-ry)/np.hypot(rx, ry)
    bad = idx[err > 1e-4]
    print(os.environ.get("NBODY_FORCE_K"), os.environ.get("NBODY_FORCE_SLICES"), os.environ.get("NBODY_FORCE_TILE"), n, info, "max err", err.max(), "nbad", len(bad), bad[:10], (bad % 256)[:10], (bad // 256)[:10])
