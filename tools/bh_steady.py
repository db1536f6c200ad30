"""Developer tool: steady-state time of one Barnes-Hut force evaluation (build + walk, state unchanged) after the
chip's clocks have ramped up: REPS evaluations back to back, per-kernel-class event times over the last quarter.
    python tools/bh_steady.py [N [theta [reps [sorted]]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
srt = dict(bh_sorted=True, ref_compat=0) if len(sys.argv) > 4 and sys.argv[4] == "sorted" else {}
sim = nb.NBodySim(n, theta=theta, useBarnesHut=True, **srt); sim.upload(*ic.disc(n, seed=12345))
for phase in ("cold", "warm"):
    sim.profile(True).profile_reset()
    k = 5 if phase == "cold" else reps
    t0 = time.perf_counter()
    for _ in range(k): sim.compute_accel()
    sim.sync(); wall = (time.perf_counter() - t0) / k * 1e3
    w = sim.profile_samples(0); b = sim.profile_samples(2)
    q = max(1, len(w) // 4)
    print(f"N={n} theta={theta} {phase}: {k} evaluations, wall {wall:.3f} ms each; last quarter: walk {w[-q:].mean()*1e3:.1f} us, build {b[-q:].mean()*1e3:.1f} us"
          f" (first 5: walk {w[:5].mean()*1e3:.1f}, build {b[:5].mean()*1e3:.1f})", flush=True)
sim.close()
