"""Developer tool: every Barnes-Hut size class has two implementations of several build and walk phases (rank sort /
device radix sort, one-workgroup preparation / separate kernels, one-launch centre of mass / one launch per depth,
four-wave walk / 8 groups per wave).  For random sizes and adversarial inputs (tools/bh_stress.py's generators plus
near-coincident bodies) this builds and walks the SAME input with the size's default kernels, three times, and with
every alternative forced, and compares trees, counters and accelerations bit for bit.  GPU against GPU: no oracle,
so sizes up to 300 000 and many rounds.      python tools/bh_paths_soak.py [seed [rounds]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as e
nb = e.load_package(); O = e.load_oracle()
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bh_stress import make

ALT = {"NBODY_BH_NO_FUSE": "1", "NBODY_BH_RANKSORT_MAX": "0", "NBODY_BH_CLIMB_MAX": "0", "NBODY_BH_QUAD_MAX": "0",
       "NBODY_NO_GRAPH": "1",                 # eager launches instead of the captured step graph
       "NBODY_BH_NO_CORE": "1",               # hipcc's IEEE sqrt and divisions instead of exact_core.hip.h's sequences
       "NBODY_BH_HIPCUB": "1"}                # rocPRIM's radix sort and scan instead of csrc/bh_sort.hip.h


def run(b, theta, env, steps, **kw):
    for k in ALT: os.environ.pop(k, None)
    os.environ.update(env)
    with nb.NBodySim(b.n, theta=theta, useBarnesHut=True, **kw) as sim:
        sim.upload(*b.arrays())
        if steps: sim.step(steps)
        sim.compute_accel().sync()
        ax, ay = sim.accel()
        t = sim.bh_tree()
        return ax, ay, t, sim.bh_counters(), sim.bh_group_counters()


def soak(seed, rounds):
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for r in range(rounds):
        kind = ["lattice", "cluster", "wide", "axis", "disc", "close"][r % 6]
        n = int(rng.choice([9, 100, 1000, 4097, 9000, 12288, 12289, 20000, 32768, 40000, 70000, 300000]))
        if kind in ("lattice", "axis", "cluster"): n = min(n, 4097)      # stay inside the tie buffer
        theta = float(rng.choice([1.0, 0.5, 0.3]))
        if n > 50000: theta = 1.0
        b = make("disc" if kind == "close" else kind, n, rng)
        if kind == "close":                     # neighbouring floats near the origin: equal path keys, distinct bodies
            k = max(2, b.n // 200)
            p = rng.choice(b.n, size=(k, 2), replace=False)
            x = (0.5 + 0.4 * rng.random(k)).astype(np.float32)
            b.posX[p[:, 0]] = x; b.posX[p[:, 1]] = np.nextafter(x, np.float32(2.0))
            b.posY[p[:, 0]] = b.posY[p[:, 1]] = (0.6 + 0.3 * rng.random(k)).astype(np.float32)
        steps = int(rng.choice([0, 0, 3]))
        ref = run(b, theta, {}, steps)
        status = "ok"
        variants = [("default again", {}), ("default again", {})] + [(f"{k}={v}", {k: v}) for k, v in ALT.items()] + [("all alternatives", ALT)]
        for name, env in variants:
            ax, ay, t, c, g = run(b, theta, env, steps)
            same_tree = len(t) == len(ref[2]) and all(np.array_equal(t[f], ref[2][f]) for f in ("mass", "comX", "comY", "children"))
            if not same_tree: status = f"TREE differs with {name}"; break
            if c != ref[3] or not np.array_equal(g[0], ref[4][0]) or not np.array_equal(g[1], ref[4][1]): status = f"COUNTERS differ with {name}"; break
            if not (np.array_equal(ax, ref[0]) and np.array_equal(ay, ref[1])): status = f"ACCELERATIONS differ with {name}"; break
        if status == "ok" and steps == 0:
            # shards: the targets of every rank of a random world size, walked on their own (i_begin != 0, ragged last
            # shard), put together = the unsharded accelerations; with the default walk and with the alternative one
            world = int(rng.integers(2, 9))
            for env in ({}, {"NBODY_BH_QUAD_MAX": "0"}, {"NBODY_BH_QUAD_MAX": "100000000"}):
                sx, sy = np.zeros_like(ref[0]), np.zeros_like(ref[1])
                for rk in range(world):
                    span = nb.shard_range(b.n, world, rk)
                    ax, ay, t, c, g = run(b, theta, env, 0, shard=span)
                    sx[span[0]:span[0] + span[1]] = ax[span[0]:span[0] + span[1]]
                    sy[span[0]:span[0] + span[1]] = ay[span[0]:span[0] + span[1]]
                if not (np.array_equal(sx, ref[0]) and np.array_equal(sy, ref[1])):
                    status = f"SHARDS of {world} differ with {env}"
                    ax, ay = sx, sy
                    break
        if status == "ok" and steps == 0 and kind != "close":
            # bh_sorted (groups in tree order, ref_compat 0): its own result, every alternative against it
            sref = run(b, theta, {}, 0, bh_sorted=True, ref_compat=0)
            for name, env in [(f"{k}={v}", {k: v}) for k, v in ALT.items()] + [("sorted again", {})]:
                ax, ay, t, c, g = run(b, theta, env, 0, bh_sorted=True, ref_compat=0)
                if c != sref[3] or not (np.array_equal(ax, sref[0]) and np.array_equal(ay, sref[1])):
                    status = f"BH_SORTED differs with {name}"
                    break
        if status != "ok":                       # details, the input for a closer look, and is the DEFAULT path reproducible?
            out = os.path.join(ROOT, "gpurun_out", f"soak_fail_seed{seed}_round{r}.npz")
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savez(out, theta=theta, steps=steps, **{f: getattr(b, f) for f in b.FIELDS})
            da = np.nonzero((ax != ref[0]) | (ay != ref[1]))[0]
            diff = {f: int((t[f] != ref[2][f]).sum()) for f in ("mass", "comX", "comY", "bodyIndex")} if len(t) == len(ref[2]) else {}
            again = [run(b, theta, {}, steps) for _ in range(4)]
            repro = all(np.array_equal(a[0], ref[0]) and len(a[2]) == len(ref[2]) for a in again)
            status += (f" | nodes {len(t)} vs {len(ref[2])}, differing node fields {diff}, {len(da)} accelerations differ (first bodies {da[:4]}),"
                       f" default path reproducible over 4 more runs: {repro}; input saved to {os.path.relpath(out, ROOT)}")
        bad += status.split(" |")[0] != "ok"
        print(f"[{r:3d}] {kind:8s} n={b.n:7d} theta={theta} steps={steps}: {status}", flush=True)
    for k in ALT: os.environ.pop(k, None)
    print(f"{rounds - bad}/{rounds} ok in {time.time() - t0:.1f}s", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(soak(int(sys.argv[1]) if len(sys.argv) > 1 else 1, int(sys.argv[2]) if len(sys.argv) > 2 else 40))
