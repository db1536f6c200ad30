"""Developer tool: phase times of bh_walk_wide_kernel (library built with -DNBODY_WIDE_TIMING, which makes the kernel
return 100 MHz ticks per phase instead of accelerations).
    make lib-timing && NBODY_AMD_LIB=n-body_amd/lib/libnbody_amd_timing.so python tools/wide_timing.py [N theta]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as e
nb = e.load_package()
from nbody_amd import initial_conditions as ic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
os.environ["NBODY_BH_WIDE"] = "1"
sim = nb.NBodySim(n, theta=theta, useBarnesHut=True); sim.upload(*ic.disc(n, seed=12345))
for rep in range(2):
    sim.compute_accel(); sim.sync()
ax, ay = sim.accel()
v, a = sim.bh_group_counters()
sim.bh_counters()     # NBODY_DEBUG=1: phase times of the one-workgroup build kernels on stderr
g = n // 8
p1, srt = ax[0:8 * g:8], ay[0:8 * g:8]
p2, it = ax[1:8 * g:8], ay[1:8 * g:8]
us = 0.01
print(f"N={n} theta={theta}: per group (us) median / max:  pass1 {np.median(p1)*us:.1f} / {p1.max()*us:.1f}   sort {np.median(srt)*us:.1f} / {srt.max()*us:.1f}   "
      f"pass2 {np.median(p2)*us:.1f} / {p2.max()*us:.1f}   total {np.median(p1+srt+p2)*us:.1f} / {(p1+srt+p2).max()*us:.1f}")
print(f"iterations median {np.median(it):.0f} max {it.max():.0f}; visits median {np.median(v[:g]):.0f} max {v[:g].max()}; accepts median {np.median(a[:g]):.0f} max {a[:g].max()}")
print(f"us per pass-1 iteration: {np.median(p1 / np.maximum(it, 1)) * us:.2f}; us per pass-2 round: {np.median(p2 / np.maximum(np.ceil(a[:g] / 8), 1)) * us:.2f}")
sim.close()
