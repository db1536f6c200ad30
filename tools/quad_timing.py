"""Developer tool: phase times of bh_walk_quad_kernel (library built with -DNBODY_WIDE_TIMING, which makes the kernel
return 100 MHz ticks per phase instead of accelerations).
    make lib-timing && NBODY_AMD_LIB=n-body_amd/lib/libnbody_amd_timing.so python tools/quad_timing.py [N theta]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
sim = nb.NBodySim(n, theta=theta, useBarnesHut=True); sim.upload(*ic.disc(n, seed=12345))
for rep in range(2):
    sim.compute_accel(); sim.sync()
ax, ay = sim.accel()
v, a = sim.bh_group_counters()
g = n // 8
us = 0.01
A = ax[:8 * g].reshape(g, 4, 2); B = ay[:8 * g].reshape(g, 4, 2)
p1, srt, pb, tot = A[:, :, 0], B[:, :, 0], A[:, :, 1], B[:, :, 1]       # [group, wave]
f = lambda x: f"{np.median(x) * us:.1f} / {x.max() * us:.1f}"
print(f"N={n} theta={theta}, per group in us, median / max over groups")
print(f"  pass 1: slowest wave {f(p1.max(1))}   mean wave {f(p1.mean(1))}")
print(f"  sort + held terms: slowest wave {f(srt.max(1))}")
print(f"  phase A (slowest wave, pass 1 + sort): {f((p1 + srt).max(1))}")
print(f"  phase B: sum of the four waves {f(pb.sum(1))}")
print(f"  whole group: {f(tot.max(1))}")
print(f"  visits median {np.median(v[:g]):.0f} max {v[:g].max()}; accepts median {np.median(a[:g]):.0f} max {a[:g].max()}")
sim.close()
