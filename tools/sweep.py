"""Developer tool: sweep the force-launch geometry knobs (NBODY_FORCE_K / NBODY_FORCE_SLICES) on the GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]); reps = int(sys.argv[2])
b = ic.disc(n, seed=12345)
sim = nb.NBodySim(n, useBarnesHut=False); sim.upload(*b)
sim.compute_accel(); sim.sync()
sim.profile(True).profile_reset()
for _ in range(reps): sim.compute_accel()
ms, cnt = sim.profile_read(0); ms_i, _ = sim.profile_read(1)
info = sim.info()
k = ms / cnt
print("N=%%d K=%%s slices=%%d blocks=%%d  force %%.4f ms  %%.1f Ginter/s (%%.1f%%%%)  reduce %%.1f us" %% (n, sys.argv[3], info["j_slices"], info["blocks"], k, n * n / k / 1e6, n * n / k / 1e6 * 20 / 157.3e3 * 100, ms_i / cnt * 1e3))
''' % ROOT

for n, reps, ks, sls, tile in ((10000, 300, (2,), (10, 20, 40), "small"), (10000, 300, (2,), (5,), "big"),
                               (65536, 200, (2, 4), (16, 32), "big"), (65536, 200, (2, 4), (16, 32, 64), "small"),
                               (262144, 20, (2, 4), (4, 8, 16), "big"), (1 << 20, 2, (4,), (8, 16), "big")):
    for k in ks:
        for sl in sls:
            env = dict(os.environ, NBODY_FORCE_K=str(k), NBODY_FORCE_SLICES=str(sl), NBODY_FORCE_TILE=tile)
            print(tile, end=" ")
            r = subprocess.run([sys.executable, "-c", CODE, str(n), str(reps), str(k)], env=env, capture_output=True, text=True)
            print(r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
