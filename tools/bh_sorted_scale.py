"""Developer tool: error against the f64 direct sum and visits/group of the bh_sorted walk for several vote-threshold
scales, beside the reference-order walk at the same nominal theta (how nbody_capi.hip's kBhSortedThetaScale was chosen)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()

def run(b, theta, srt):
    with nb.NBodySim(b.n, theta=theta, useBarnesHut=True, ref_compat=0, bh_sorted=srt) as sim:
        sim.upload(*b.arrays()); sim.compute_accel(); sim.sync()
        ax, ay = sim.accel(); c = sim.bh_counters()
    return ax, ay, c["visits"] / (b.n // 8)

for kind, n, theta in (("disc", 200000, 1.0), ("disc", 200000, 0.5), ("plummer", 65536, 1.0), ("plummer", 65536, 0.5), ("plummer", 1000000, 0.5)):
    b = O.init_disc(n, seed=12345) if kind == "disc" else O.init_plummer(n, scale=50.0, seed=4)
    idx = np.arange(0, n, max(1, n // 1024), dtype=np.int32)
    nx, ny = O.naive_accel_f64_subset(b, idx)
    def stats(ax, ay):
        er = np.hypot(ax[idx] - nx, ay[idx] - ny) / np.hypot(nx, ny)
        return f"median {np.median(er):.4f} mean {er.mean():.4f} p90 {np.percentile(er, 90):.4f} max {er.max():.4f}"
    os.environ.pop("NBODY_BH_SORTED_SCALE", None)
    ax, ay, v = run(b, theta, False)
    print(f"{kind} N={n} theta={theta}: reference order      {stats(ax, ay)} visits/group {v:.0f}", flush=True)
    for sc in ("0.7", "0.65", "0.6", "0.55", "0.5"):
        os.environ["NBODY_BH_SORTED_SCALE"] = sc
        ax, ay, v = run(b, theta, True)
        print(f"{kind} N={n} theta={theta}: sorted, scale {sc:5s} {stats(ax, ay)} visits/group {v:.0f}", flush=True)
