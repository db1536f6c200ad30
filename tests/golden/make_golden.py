"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE ITSELF.

The reference ships no golden vectors.  These fixtures are outputs of
/root/reference/benchmark.cpp compiled by oracle/build_ref.sh (reference flags, x86-64-v3) and run in
the build container (Intel Xeon; the low bits of the naive accelerations depend on the CPU vendor's
`vrcpps`).  Fixtures are data only: inputs + the reference's outputs.

    python tests/golden/make_golden.py        # needs oracle/_ref/libnbody_ref.so
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle as O  # noqa: E402

SEED = 12345


def state(b):
    return dict(posX=b.posX, posY=b.posY, velX=b.velX, velY=b.velY)


def naive_case(n, nsteps_final, dt=0.05):
    r1 = O.ref_run(n, use_barnes_hut=0, nsteps=1, seed=SEED, dt=dt)
    rk = O.ref_run(n, use_barnes_hut=0, nsteps=nsteps_final, seed=SEED, dt=dt)
    init = r1["init"]
    out = dict(n=n, seed=SEED, spread=400.0, mass=1000.0, dt=dt, nsteps_final=nsteps_final,
               m=init.m[:1].copy(), radius=init.radius[:1].copy(),
               accX=r1["acc"][0], accY=r1["acc"][1])
    out.update({"init_" + k: v for k, v in state(init).items()})
    out.update({"step1_" + k: v for k, v in state(r1["final"]).items()})
    out.update({"final_" + k: v for k, v in state(rk["final"]).items()})
    return out


def bh_case(n, theta, nsteps_final, dt=0.05, want_pool=False):
    r1 = O.ref_run(n, use_barnes_hut=1, nsteps=1, seed=SEED, dt=dt, theta=theta, want_pool=want_pool)
    rk = O.ref_run(n, use_barnes_hut=1, nsteps=nsteps_final, seed=SEED, dt=dt, theta=theta)
    init = r1["init"]
    out = dict(n=n, seed=SEED, spread=400.0, mass=1000.0, dt=dt, theta=theta, nsteps_final=nsteps_final,
               m=init.m[:1].copy(), radius=init.radius[:1].copy(),
               accX=r1["acc"][0], accY=r1["acc"][1], nodes_used=r1["nodes_used"], root=r1["root"],
               final_nodes_used=rk["nodes_used"],
               # the reference's own vote (hooks at benchmark.cpp:245 and :267): per group of 8 for step 1, totals
               # of every step of the nsteps_final run
               grp_visits=r1["grp_visits"], grp_accepts=r1["grp_accepts"], grp_hash=r1["grp_hash"],
               step_visits=rk["step_visits"], step_accepts=rk["step_accepts"])
    out.update({"init_" + k: v for k, v in state(init).items()})
    out.update({"step1_" + k: v for k, v in state(r1["final"]).items()})
    out.update({"final_" + k: v for k, v in state(rk["final"]).items()})
    if want_pool:
        out["pool"] = r1["pool"]
    return out


def main():
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libnbody_ref.so missing: run oracle/build_ref.sh first")
    cases = {
        "naive_n1024": naive_case(1024, 10),
        "naive_n1003": naive_case(1003, 5),          # N % 8 != 0: frozen tail (benchmark.cpp:423,486)
        "naive_n10000": naive_case(10000, 10),       # BASELINE config 1
        "bh_n1024_theta1": bh_case(1024, 1.0, 5, want_pool=True),
        "bh_n1003_theta1": bh_case(1003, 1.0, 3),
        "bh_n10000_theta1": bh_case(10000, 1.0, 5),
        "bh_n10000_theta05": bh_case(10000, 0.5, 3),
    }
    for name, d in cases.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
