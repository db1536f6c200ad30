"""nbody_config.bh_sorted (opt-in, ref_compat = 0, no reference counterpart): the walk's groups of 8
are consecutive bodies of the build's own quadrant-path order instead of consecutive body indices, and the vote
threshold is 0.6 * theta.  Same tree, same vote rule, same node kernel — so against the oracle walking the SAME groups
the comparison is exact — and against the direct sum the error must be no worse than the reference-order walk's at the
same nominal theta (mean and 90th percentile at or below it, median within 10 %; measured table:
profiles/r2_bh_sorted_scale.txt), at >= 1.5x fewer node visits (BASELINE configs 4 and 5 flavours: seeded disc
theta = 1, clustered Plummer theta = 0.5)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
SCALE = np.float32(0.6)


def run(nb, b, *, theta, sorted_, nsteps=0, shard=None, **kw):
    with nb.NBodySim(b.n, dt=0.05, theta=theta, useBarnesHut=True, ref_compat=0, bh_sorted=sorted_, shard=shard, **kw) as sim:
        sim.upload(*b.arrays())
        if nsteps:
            sim.step(nsteps)
        else:
            sim.compute_accel()
        sim.sync()
        ax, ay = sim.accel()
        return ax, ay, sim.download(), sim.bh_counters(), sim.bh_group_counters()


@pytest.mark.parametrize("n,theta,fast", [(10000, 1.0, False), (4099, 0.5, False), (100003, 1.0, False), (100003, 0.5, True)])
def test_sorted_walk_equals_the_oracle_walking_the_same_groups(nb, O, n, theta, fast):
    b = O.init_disc(n, seed=7)
    b.posX[5] *= 3.0                                               # one body outside the root box: last in the order
    ax, ay, _, cnt, (gv, ga) = run(nb, b, theta=theta, sorted_=True, bh_fast_math=fast)
    t = O.Tree(n).build(b, 400.0)
    ox, oy, oc = t.accel_sorted(b, float(np.float32(theta) * SCALE), 400.0)
    assert (cnt["visits"], cnt["accepts"]) == (oc["visits"], oc["accepts"])
    assert np.array_equal(gv, oc["groups"][0]) and np.array_equal(ga, oc["groups"][1])
    if fast:
        e = rel_err(ax, ay, ox, oy)
        # hardware sqrt/rcp: same votes (counters above), terms differ by a few ulp; at an effective theta of 0.3 a body
        # sums ~5000 terms that partly cancel (measured max 8e-5 of |a| over 100 003 bodies; north-star tolerance 1e-4)
        assert np.median(e) <= 2e-7 and e.max() <= 1e-4
    else:
        assert np.array_equal(ax, ox.astype(np.float32)) and np.array_equal(ay, oy.astype(np.float32))
    assert np.hypot(ax, ay).min() > 0                              # every body is a target, the outside one too


@pytest.mark.parametrize("kind,n,theta", [("disc", 200000, 1.0), ("plummer", 65536, 0.5)])
def test_sorted_walk_error_no_worse_than_reference_order_at_fewer_visits(nb, O, kind, n, theta):
    b = O.init_disc(n, seed=12345) if kind == "disc" else O.init_plummer(n, scale=50.0, seed=4)
    idx = np.arange(0, n, max(1, n // 2048), dtype=np.int32)
    nx, ny = O.naive_accel_f64_subset(b, idx)
    rx, ry, _, rc, _ = run(nb, b, theta=theta, sorted_=False)
    sx, sy, _, sc, _ = run(nb, b, theta=theta, sorted_=True)
    e_ref = rel_err(rx[idx], ry[idx], nx, ny)
    e_srt = rel_err(sx[idx], sy[idx], nx, ny)
    print(f"[bh_sorted {kind} N={n} theta={theta}] reference order: median {np.median(e_ref):.4f} mean {e_ref.mean():.4f} "
          f"p90 {np.percentile(e_ref, 90):.4f} visits/group {rc['visits'] / (n // 8):.0f} | sorted: median {np.median(e_srt):.4f} "
          f"mean {e_srt.mean():.4f} p90 {np.percentile(e_srt, 90):.4f} visits/group {sc['visits'] / (n // 8):.0f}")
    assert e_srt.mean() <= e_ref.mean() and np.percentile(e_srt, 90) <= np.percentile(e_ref, 90)
    assert np.median(e_srt) <= 1.1 * np.median(e_ref)
    assert sc["visits"] * 1.5 <= rc["visits"]


def test_sorted_walk_on_a_shard_and_while_stepping(nb, O):
    """One rank of 4: its groups are ITS bodies in key order (a shard-sized second sort); results equal the oracle
    walking those groups.  And three steps of the whole array keep every body finite and moving."""
    n, theta = 40000, 1.0
    b = O.init_disc(n, seed=11)
    beg, cnt = nb.shard_range(n, 4, 2)
    ax, ay, _, c, _ = run(nb, b, theta=theta, sorted_=True, shard=(beg, cnt))
    order = O.sorted_order(b, 400.0)
    own = order[(order >= beg) & (order < beg + cnt)].astype(np.int32)
    assert len(own) == cnt
    t = O.Tree(n).build(b, 400.0)
    full = np.concatenate([own, np.zeros(0, np.int32)])
    ox = np.zeros(n); oy = np.zeros(n); oc = np.zeros(2, np.int64)
    gv = np.zeros((cnt + 7) // 8, np.int32); ga = np.zeros_like(gv)
    O.lib().oracle_bh_accel_ordered(t.h, cnt, b.posX, b.posY, b.radius, float(np.float32(theta) * SCALE), full, ox, oy, oc, gv, ga)
    assert (c["visits"], c["accepts"]) == (int(oc[0]), int(oc[1]))
    assert np.array_equal(ax[beg:beg + cnt], ox[beg:beg + cnt].astype(np.float32))
    assert not ax[:beg].any() and not ax[beg + cnt:].any()
    _, _, st, _, _ = run(nb, b, theta=theta, sorted_=True, nsteps=3)
    assert all(np.all(np.isfinite(st[k])) for k in st) and not np.array_equal(st["posX"], b.posX)


def test_sorted_needs_ref_compat_off(nb):
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodySim(1024, useBarnesHut=True, ref_compat=1, bh_sorted=True)
    assert e.value.code == nb.ERR_INVALID
    with pytest.raises(nb.NBodyError):
        nb.NBodySim(1024, useBarnesHut=False, ref_compat=0, bh_sorted=True)
