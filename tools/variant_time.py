"""Developer tool: pair-kernel throughput of the four formula variants at N = 2^18 (same launch geometry as N = 2^20)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = 1 << 18
for v, name in ((1, "scaled"), (2, "uniform"), (3, "per-body"), (4, "exact")):
    m, x, y, vx, vy, r = ic.disc(n, seed=3)
    if v == 2: r[:] = 6.0
    if v == 3:
        rng = np.random.default_rng(1); m = (rng.random(n) * 2000 + 1).astype(np.float32); r = rng.uniform(4, 30, n).astype(np.float32)
    if v == 4: r[:] = 0.5
    sim = nb.NBodySim(n, force_variant=v); sim.upload(m, x, y, vx, vy, r)
    sim.compute_accel().sync(); sim.profile(True).profile_reset()
    for _ in range(5): sim.compute_accel()
    sim.sync(); ms, c = sim.profile_read(0)
    g = n * n / (ms / c * 1e-3) / 1e9
    print(f"variant {v} ({name}): {ms / c:.3f} ms  {g:.0f} G interactions/s  {g * 20 / 157.3e3 * 100:.1f} % of FP32 peak  {sim.info()}", flush=True)
    sim.close()
