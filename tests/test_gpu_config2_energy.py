"""BASELINE config 2: naive O(N^2), N = 65 536, single MI355X, fp32, 100 steps, energy-drift check vs the CPU.

The CPU side is the compiled reference itself (oracle/_ref, benchmark.cpp's own naive loop) when it is present on the
box, advanced in 4 legs of 25 steps from the same seeded disc; energies of both trajectories are evaluated by the
f64-accumulating GPU diagnostics kernel (itself checked against the oracle in test_gpu_naive.py).  The system is
chaotic and forward Euler at dt = 0.05 with mass 1000 per body is violent: trajectories are NOT
compared, the conserved-quantity drift curves are.  Measured (round 2, profiles/r2_config2_energy_drift.txt): the
energy itself changes by +115 % of |E0| within 25 steps (the disc is unbound and forward Euler heats it), and the GPU
and reference curves differ by 3.6e-5, 1.8e-5, 9e-6, 5e-6 of |E0| at steps 25/50/75/100.  Stated bound: ten times the
largest observed difference, |drift_gpu - drift_cpu| <= 4e-4 of |E0| at every checkpoint; momentum change <= 1e-4 of
sum m|v| on both."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config2_energy_and_momentum_drift(nb, O):
    if not O.ref_available():
        pytest.skip("oracle/_ref/libnbody_ref.so not on this box")
    n, dt, legs, per = 65536, 0.05, 4, 25
    b = O.init_disc(n, seed=12345)

    def energy(state):
        with nb.NBodySim(n, dt=dt, useBarnesHut=False) as s:
            s.upload(*state.arrays())
            d = s.diagnostics()
        return d["Ekin"] + d["Epot"], np.array([d["Px"], d["Py"]])

    e_gpu, p_gpu = [], []
    with nb.NBodySim(n, dt=dt, useBarnesHut=False) as sim:
        sim.upload(*b.arrays())
        d = sim.diagnostics(); e_gpu.append(d["Ekin"] + d["Epot"]); p_gpu.append(np.array([d["Px"], d["Py"]]))
        for _ in range(legs):
            sim.step(per)
            d = sim.diagnostics(); e_gpu.append(d["Ekin"] + d["Epot"]); p_gpu.append(np.array([d["Px"], d["Py"]]))
    cur = b.copy()
    e0, p0 = energy(cur)
    e_cpu, p_cpu = [e0], [p0]
    for _ in range(legs):
        cur = O.ref_run(n, use_barnes_hut=0, nsteps=per, bodies=cur, dt=dt)["final"]
        e, p = energy(cur)
        e_cpu.append(e); p_cpu.append(p)
    e_gpu, e_cpu = np.array(e_gpu), np.array(e_cpu)
    assert e_gpu[0] == e_cpu[0]
    drift_gpu = (e_gpu - e_gpu[0]) / abs(e_gpu[0])
    drift_cpu = (e_cpu - e_cpu[0]) / abs(e_cpu[0])
    scale = np.sum(b.m.astype(np.float64) * np.hypot(b.velX, b.velY))
    mom = [[float(np.linalg.norm(q - p[0]) / scale) for q in p] for p in (p_gpu, p_cpu)]
    print("energy drift gpu", drift_gpu, "cpu(reference)", drift_cpu, "difference", drift_gpu - drift_cpu)
    print("momentum change / sum m|v|: gpu", mom[0], "cpu(reference)", mom[1])
    assert np.all(np.abs(drift_gpu - drift_cpu) <= 4e-4), (drift_gpu, drift_cpu)
    assert max(mom[0]) <= 1e-4 and max(mom[1]) <= 1e-4
