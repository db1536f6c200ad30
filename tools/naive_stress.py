"""Developer tool (test infrastructure: uses the oracle): randomized parity stress of the naive pair kernel on the
GPU against the float64 oracle, on inputs chosen to hit every branch of every formula variant: bodies inside each
other's clamp radius, coincident bodies, per-body masses (some zero) and radii (some below 4: exact variant), huge
coordinate ranges, ragged N.  Metric: |a_gpu - a_f64| / sum_j |term_ij| (the fp32 accumulation error relative to the
magnitudes summed; |a| itself can be arbitrarily small where the terms cancel)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()


def make(kind, n, rng):
    b = O.init_disc(n, seed=int(rng.integers(1, 1 << 30)))
    if kind == "crowd":                       # most pairs inside the clamp radius
        b.posX = (b.posX * 0.02).astype(np.float32); b.posY = (b.posY * 0.02).astype(np.float32)
    elif kind == "coincident":
        idx = rng.integers(0, max(1, n // 2), n)
        b.posX = b.posX[idx].copy(); b.posY = b.posY[idx].copy()
    elif kind == "wide":
        s = np.float32(10.0 ** rng.uniform(1, 4))
        b.posX = (b.posX * s).astype(np.float32); b.posY = (b.posY * s).astype(np.float32)
    elif kind == "masses":
        b.m = (rng.random(n) * 5000.0).astype(np.float32); b.m[rng.random(n) < 0.2] = 0.0
    elif kind == "radii":
        b.radius = np.exp(rng.uniform(np.log(4.0), np.log(60.0), n)).astype(np.float32)
        b.m = (rng.random(n) * 2000.0 + 1.0).astype(np.float32)
    elif kind == "small_radii":               # exact variant
        b.radius = np.exp(rng.uniform(np.log(0.01), np.log(10.0), n)).astype(np.float32)
    elif kind == "pow2":
        b.radius[:] = np.float32(2.0 ** int(rng.integers(3, 8)))
    return b


def sum_abs_terms(b):
    x = b.posX.astype(np.float64); y = b.posY.astype(np.float64); m = b.m.astype(np.float64); r = b.radius.astype(np.float64)
    out = np.zeros(b.n)
    for i0 in range(0, b.n, 512):
        dx = x[None, :] - x[i0:i0 + 512, None]; dy = y[None, :] - y[i0:i0 + 512, None]
        d = np.hypot(dx, dy)
        R = np.maximum(r[None, :], r[i0:i0 + 512, None])
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where(d > 0, m[None, :] / (np.maximum(d, R) ** 2 + 1e-6), 0.0)   # |term| = F since |delta/d| = 1
        out[i0:i0 + 512] = t.sum(1)
    return out


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.default_rng(seed)
    kinds = ["disc", "crowd", "coincident", "wide", "masses", "radii", "small_radii", "pow2"]
    bad = 0
    worst = 0.0
    for r in range(rounds):
        kind = kinds[r % len(kinds)]
        n = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 255, 1000, 2049, 4097, 10000]))
        rc = int(rng.integers(0, 2))
        b = make(kind, n, rng)
        with nb.NBodySim(n, useBarnesHut=False, ref_compat=rc) as sim:
            sim.upload(*b.arrays()); sim.compute_accel()
            ax, ay = sim.accel(); info = sim.info()
        rx, ry = O.naive_accel_f64(b, ref_compat=rc)
        lim = n - n % 8 if rc else n
        s = sum_abs_terms(b)
        err = np.hypot(ax - rx, ay - ry)[:lim] / np.maximum(s[:lim], 1e-300)
        m = float(err.max()) if lim else 0.0
        tail_ok = bool(np.all(ax[lim:] == 0) and np.all(ay[lim:] == 0))
        ok = m <= 2e-6 and tail_ok and np.all(np.isfinite(ax)) and np.all(np.isfinite(ay))
        worst = max(worst, m)
        if not ok: bad += 1
        print(f"[{r:3d}] {kind:11s} n={n:6d} ref_compat={rc} variant={info['force_variant']} slices={info['j_slices']:2d}: "
              f"max err/sum|terms| {m:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    print(f"{rounds - bad}/{rounds} ok, worst {worst:.2e}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
