"""Developer tool: N = 10 000 naive step time against the number of j-slices (NBODY_FORCE_SLICES), whole steps back to back."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time
sys.path.insert(0, %r)
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1])
with nb.NBodySim(n, useBarnesHut=False) as sim:
    sim.upload(*ic.disc(n, seed=12345)); sim.step(50).sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); sim.step(500).sync(); best = min(best, (time.perf_counter() - t0) / 500)
    i = sim.info()
    print("N=%%d slices=%%d blocks=%%d: %%.2f us/step  %%.1f %%%% of peak" %% (n, i["j_slices"], i["blocks"], best * 1e6, n * n / best / 1e12 * 20 / 157.3 * 100), flush=True)
''' % ROOT
sizes = [int(a) for a in sys.argv[1:]] or [10000, 5000, 20000]
for n in sizes:
    for sl in (None, 12, 16, 19, 25, 32, 38, 47, 64):
        env = dict(os.environ)
        env.pop("NBODY_FORCE_SLICES", None)
        if sl: env["NBODY_FORCE_SLICES"] = str(sl)
        r = subprocess.run([sys.executable, "-c", CODE, str(n)], env=env, capture_output=True, text=True)
        print(("plan  " if sl is None else "forced"), r.stdout.strip() or r.stderr.strip()[-300:], flush=True)
