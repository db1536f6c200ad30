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
    r1 = O.ref_run(n, use_barnes_hut=1, nsteps=1, seed=SEED, dt=dt, theta=theta, want_pool=want_pool, count_votes=True)
    rk = O.ref_run(n, use_barnes_hut=1, nsteps=nsteps_final, seed=SEED, dt=dt, theta=theta, count_votes=True)
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


def close_pair_case():
    """Two DISTINCT bodies 1.2e-7 apart (closer than the finest cell of the GPU's 31-level path key, 800/2^31 = 3.7e-7):
    the reference keeps splitting (to depth 33 here) and still terminates; the GPU build keeps them in one leaf
    (DESIGN.md §3.4).  Fixture: the reference's node count / root / accelerations / per-group vote counters."""
    n = 512
    b = O.init_disc(n, seed=9)
    b.posX[7], b.posY[7] = np.float32(0.5), np.float32(0.25)
    b.posX[100], b.posY[100] = np.nextafter(np.float32(0.5), np.float32(1), dtype=np.float32), np.float32(0.25)
    b.posX[100] = np.nextafter(b.posX[100], np.float32(1), dtype=np.float32)          # 2 ulp = 1.2e-7
    r1 = O.ref_run(n, use_barnes_hut=1, nsteps=1, bodies=b, theta=1.0, want_pool=True, count_votes=True)
    depth = 0
    pool = r1["pool"]
    stack = [(0, 0)]
    while stack:
        ni, d = stack.pop()
        depth = max(depth, d)
        if pool[ni]["children"][0] != -1:
            stack.extend((int(c), d + 1) for c in pool[ni]["children"])
    out = dict(n=n, seed=9, spread=400.0, mass=1000.0, dt=0.05, theta=1.0, nsteps_final=1,
               m=b.m[:1].copy(), radius=b.radius[:1].copy(), accX=r1["acc"][0], accY=r1["acc"][1],
               nodes_used=r1["nodes_used"], root=r1["root"], final_nodes_used=r1["nodes_used"], tree_depth=depth,
               pair=np.array([7, 100], np.int32),
               grp_visits=r1["grp_visits"], grp_accepts=r1["grp_accepts"], grp_hash=r1["grp_hash"],
               step_visits=r1["step_visits"], step_accepts=r1["step_accepts"])
    out.update({"init_" + k: v for k, v in state(b).items()})
    out.update({"step1_" + k: v for k, v in state(r1["final"]).items()})
    out.update({"final_" + k: v for k, v in state(r1["final"]).items()})
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
        "bh_close_pair_n512": close_pair_case(),
    }
    for name, d in cases.items():
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
