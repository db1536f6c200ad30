"""The drop-in claim, end to end: the REFERENCE'S OWN benchmark.cpp — its generator, loop, timing and footer, compiled
from /root/reference by oracle/build_ref.sh — with exactly the edit INTEGRATION.md describes (QuadTree init ->
nbody_create + nbody_upload, physics block benchmark.cpp:354-511 -> nbody_step + nbody_sync).  Both builds of that
program run on the same seed; the GPU-physics build must track the CPU build within the trajectory gates of
DESIGN.md §5 and equal a plain NBodySim run bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bh,n", [(0, 1024), (0, 1003), (1, 1024), (1, 10000)])
def test_reference_program_with_gpu_physics(nb, O, bh, n):
    if not (O.ref_available() and O.ref_dropin_available()):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time)")
    k = 3
    cpu = O.ref_run(n, use_barnes_hut=bh, nsteps=k, seed=4242)
    gpu = O.ref_run(n, use_barnes_hut=bh, nsteps=k, seed=4242, dropin=True)
    for f in ("m", "posX", "posY", "velX", "velY", "radius"):          # the reference's generator ran in both
        assert np.array_equal(getattr(cpu["init"], f), getattr(gpu["init"], f)), f
    # same library, same inputs, driven through the ctypes binding instead: identical bits
    with nb.NBodySim(n, dt=0.05, theta=1.0, useBarnesHut=bool(bh)) as sim:
        sim.upload(*gpu["init"].arrays())
        sim.step(k)
        st = sim.download()
        ax, ay = sim.accel()
    for f in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(getattr(gpu["final"], f), st[f]), f
    assert np.array_equal(gpu["acc"][0], ax) and np.array_equal(gpu["acc"][1], ay)
    # against the reference's own CPU physics (12-bit rcp, and for Barnes-Hut the discontinuous vote)
    for f in ("posX", "posY"):
        d = np.abs(getattr(gpu["final"], f) - getattr(cpu["final"], f))
        assert np.median(d) <= 2e-5 * 400.0 and d.max() <= 1e-3 * 400.0, (f, np.median(d), d.max())
    for f in ("velX", "velY"):
        d = np.abs(getattr(gpu["final"], f) - getattr(cpu["final"], f))
        vmax = np.abs(getattr(cpu["final"], f)).max()
        assert np.median(d) <= 2e-4 * vmax and d.max() <= 2e-2 * vmax, (f, np.median(d) / vmax, d.max() / vmax)
    assert len(gpu["step_ms"]) == k and np.all(gpu["step_ms"] > 0)    # the reference's own physicsStart/physicsEnd clock


@pytest.mark.parametrize("bh,n", [(1, 1003), (1, 10007), (0, 1003)])
def test_ref_compat_2_tracks_main_cpp_physics(nb, O, bh, n):
    """main.cpp cannot be built as a program here (SFML), but its physics block can (oracle/build_ref.sh streams
    main.cpp:2-315 and 485-642 into a callable loop).  ref_compat = 2 is that program's flavour: the N%8 tail bodies of
    its Barnes-Hut branch drift force-free — bit-identical here, one fma per coordinate — and everything else stays
    within the trajectory gates of its 12-bit-rcp CPU arithmetic."""
    if not O.ref_mainphys_available():
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time)")
    k = 3
    b = O.init_disc(n, seed=808, vel_factor=0.05)                       # main.cpp:431-432
    cpu = O.ref_run(n, use_barnes_hut=bh, nsteps=k, bodies=b, main_cpp=True)["final"]
    with nb.NBodySim(n, dt=0.05, theta=1.0, useBarnesHut=bool(bh), ref_compat=2) as sim:
        sim.upload(*b.arrays())
        sim.step(k)
        st = sim.download()
    lim = n - n % 8
    for f in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(st[f][lim:], getattr(cpu, f)[lim:]), f        # the tail: same bits
    for f in ("posX", "posY"):
        d = np.abs(st[f][:lim] - getattr(cpu, f)[:lim])
        assert np.median(d) <= 2e-5 * 400.0 and d.max() <= 1e-3 * 400.0, (f, np.median(d), d.max())
    for f in ("velX", "velY"):
        d = np.abs(st[f][:lim] - getattr(cpu, f)[:lim])
        vmax = np.abs(getattr(cpu, f)).max()
        assert np.median(d) <= 2e-4 * vmax and d.max() <= 2e-2 * vmax, (f, np.median(d) / vmax, d.max() / vmax)
