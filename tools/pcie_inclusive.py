"""Developer tool: the naive step at N = 2^20 when the caller hands over HOST arrays every step (nbody_upload + nbody_step
+ nbody_download per step) instead of keeping the state resident — the PCIe-inclusive rate DESIGN.md §6 quotes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = 1 << 20
b = ic.disc(n, seed=12345)
sim = nb.NBodySim(n, useBarnesHut=False); sim.upload(*b); sim.step(1).sync()
for name, fn in (("resident", lambda: sim.step(1).sync()),
                 ("upload_state + step + download (pos, vel) every step", lambda: (sim.upload_state(*b[1:5]), sim.step(1), sim.download())),
                 ("full upload + step + download every step", lambda: (sim.upload(*b), sim.step(1), sim.download()))):
    fn()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:.2f} ms/step = {n*n/dt/1e9:.0f} G interactions/s", flush=True)
sim.close()
