"""CPU tests: pin the oracle (oracle/nbody_oracle.c) against the golden vectors produced by running the
reference itself (tests/golden/make_golden.py).  No GPU needed."""
import numpy as np
import pytest

from conftest import load_golden, rel_err


def ulp_diff(a, b):
    ai = a.view(np.int32).astype(np.int64)
    bi = b.view(np.int32).astype(np.int64)
    return np.abs(ai - bi)


def test_mt19937_known_answer(O):
    # C++11 [rand.predef]: 10000th output of default-seeded mt19937
    out = np.zeros(10000, np.uint32)
    O.lib().oracle_mt19937_raw(5489, 10000, out)
    assert int(out[-1]) == 4123659995


@pytest.mark.parametrize("name", ["naive_n1024", "naive_n1003", "naive_n10000"])
def test_generator_matches_reference(O, name):
    """A8: positions bit-exact; velocities within 1e-6 relative (the reference's -ffast-math reassociates them)."""
    g, _ = load_golden(O, name)
    b = O.init_disc(int(g["n"]), seed=int(g["seed"]))
    assert np.array_equal(b.posX, g["init_posX"])
    assert np.array_equal(b.posY, g["init_posY"])
    assert np.allclose(b.velX, g["init_velX"], rtol=1e-6, atol=1e-7)
    assert np.allclose(b.velY, g["init_velY"], rtol=1e-6, atol=1e-7)
    assert np.all(b.m == g["m"][0]) and np.all(b.radius == g["radius"][0])


def test_survey_anchor(O):
    """seed 12345 => pos[0] and the compiled reference's acc[0] at N=10000."""
    g, b = load_golden(O, "naive_n10000")
    assert abs(b.posX[0] - 321.80777) < 1e-4 and abs(b.posY[0] + 152.380524) < 1e-4
    assert abs(g["accX"][0] + 99.0515518) < 2e-2 and abs(g["accY"][0] - 49.2517471) < 2e-2


@pytest.mark.parametrize("name", ["naive_n1024", "naive_n1003", "naive_n10000"])
def test_naive_accel_vs_reference(O, name):
    """A1: the reference is the inaccurate party (12-bit vrcpps), so distribution gates, no max gate:
    median <= 5e-5, p99 <= 3e-4 at N >= 10000; the per-term vrcpps error (<= 3.7e-4)
    averages down more slowly at N ~ 1000, so 1e-4 / 1e-3 there."""
    g, b = load_golden(O, name)
    n = b.n
    lim = n - n % 8
    ax64, ay64 = O.naive_accel_f64(b)
    ax32, ay32 = O.naive_accel_f32(b)
    for ax, ay in ((ax64, ay64), (ax32, ay32)):
        e = rel_err(ax[:lim], ay[:lim], g["accX"][:lim], g["accY"][:lim])
        med_gate, p99_gate = (5e-5, 3e-4) if n >= 10000 else (1e-4, 1e-3)
        assert np.median(e) <= med_gate and np.percentile(e, 99) <= p99_gate
    # f32 restatement against f64 truth: rounding only
    e = rel_err(ax32[:lim], ay32[:lim], ax64[:lim], ay64[:lim])
    assert e.max() <= 2e-5
    # tail quirk: never a target
    assert np.all(g["accX"][lim:] == 0) and np.all(ax64[lim:] == 0) and np.all(ax32[lim:] == 0)


@pytest.mark.parametrize("name", ["naive_n1024", "naive_n1003", "naive_n10000"])
def test_euler_step_vs_reference(O, name):
    """A2: feed the reference's own accelerations through the oracle integrator: bit-exact with the reference's
    state after one step (the oracle uses fmaf, as the reference's -ffast-math build contracts mul+add)."""
    g, b = load_golden(O, name)
    O.lib().oracle_euler_f32(b.n, b.posX, b.posY, b.velX, b.velY, g["accX"], g["accY"], float(g["dt"]), 1)
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(getattr(b, k), g["step1_" + k]), k


def test_tail_bodies_frozen(O):
    g, b = load_golden(O, "naive_n1003")
    lim = 1000
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(g["final_" + k][lim:], g["init_" + k][lim:])
    c = b.copy()
    O.naive_steps(c, float(g["dt"]), int(g["nsteps_final"]))
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(getattr(c, k)[lim:], g["init_" + k][lim:])


@pytest.mark.parametrize("name", ["naive_n1024", "naive_n1003", "naive_n10000"])
def test_short_trajectory_vs_reference(O, name):
    """Short horizon (<= 10 steps): trajectories still agree to ~1e-5 of the box size."""
    g, b = load_golden(O, name)
    O.naive_steps(b, float(g["dt"]), int(g["nsteps_final"]))
    for k in ("posX", "posY"):       # 2e-5 of the box half-width
        assert np.abs(getattr(b, k) - g["final_" + k]).max() <= 2e-5 * 400.0
    for k in ("velX", "velY"):       # 2e-4 of the largest speed
        assert np.abs(getattr(b, k) - g["final_" + k]).max() <= 2e-4 * np.abs(g["final_" + k]).max()


def test_diagnostics_momentum_definition(O):
    b = O.init_disc(512, seed=7)
    d = O.diagnostics(b)
    assert np.isclose(d[2], np.sum(b.m.astype(np.float64) * b.velX), rtol=1e-12)
    assert np.isclose(d[0], 0.5 * np.sum(b.m.astype(np.float64) * (b.velX.astype(np.float64) ** 2 + b.velY.astype(np.float64) ** 2)), rtol=1e-12)
    assert d[1] < 0


def test_energy_force_consistency(O):
    """The potential used by the diagnostics is the antiderivative of the clamped force: finite-difference
    of Epot w.r.t. one body's position equals -m a from the f64 force oracle."""
    b = O.init_disc(256, seed=3)
    b.posX *= 0.05; b.posY *= 0.05          # crowd the bodies so the clamp branch (d < R) is exercised
    ax, ay = O.naive_accel_f64(b)
    i, h = 17, 1e-2
    bp, bm = b.copy(), b.copy()
    bp.posX[i] += h; bm.posX[i] -= h
    hp = float(bp.posX[i]) - float(bm.posX[i])
    dE = (O.diagnostics(bp)[1] - O.diagnostics(bm)[1]) / hp
    assert np.isclose(-dE, b.m[i] * ax[i], rtol=2e-3)


# ---------------------------------------------------------------------------------------------------
# Barnes-Hut (A4-A7)
# ---------------------------------------------------------------------------------------------------
def test_tree_identical_to_reference(O):
    g, b = load_golden(O, "bh_n1024_theta1")
    t = O.Tree(b.n).build(b, float(g["spread"]))
    assert t.nodes_used == int(g["nodes_used"])
    pool, ref = t.pool(), g["pool"]
    for f in ("tlX", "tlY", "brX", "brY", "bodyIndex", "children"):
        assert np.array_equal(pool[f], ref[f]), f
    assert np.array_equal(pool["mass"], ref["mass"])
    assert np.array_equal(pool["comX"], ref["comX"])      # bit-exact incl. the FMA contraction of the reference build
    assert np.array_equal(pool["comY"], ref["comY"])
    st = t.stats()
    assert st["bodies"] >= b.n        # >= : a body exactly on a split line enters 2+ children


@pytest.mark.parametrize("name", ["bh_n1003_theta1", "bh_n10000_theta1", "bh_n10000_theta05", "bh_close_pair_n512"])
def test_tree_node_count(O, name):
    g, b = load_golden(O, name)
    t = O.Tree(b.n).build(b, float(g["spread"]))
    assert t.nodes_used == int(g["nodes_used"])
    root = t.pool()[0]
    assert np.isclose(root["mass"], g["root"][0], rtol=1e-6)
    assert abs(root["comX"] - g["root"][1]) < 1e-2 and abs(root["comY"] - g["root"][2]) < 1e-2


@pytest.mark.parametrize("name", ["bh_n1024_theta1", "bh_n1003_theta1", "bh_n10000_theta1", "bh_n10000_theta05"])
def test_bh_accel_vs_reference(O, name):
    """Same tree, same group-of-8 vote => same accepted node set; what differs is the reference's 12-bit
    rcp (benchmark.cpp:213): median <= 2e-4, p99 <= 2e-3."""
    g, b = load_golden(O, name)
    t = O.Tree(b.n).build(b, float(g["spread"]))
    lim = b.n - b.n % 8
    for f64 in (False, True):
        ax, ay, cnt = t.accel(b, float(g["theta"]), f64=f64)
        e = rel_err(ax[:lim], ay[:lim], g["accX"][:lim], g["accY"][:lim])
        assert np.median(e) <= 2e-4 and np.percentile(e, 99) <= 2e-3
        assert np.all(ax[lim:] == 0)
    assert cnt["accepts"] > 0 and cnt["visits"] >= cnt["accepts"]


BH_GOLDEN = ["bh_n1024_theta1", "bh_n1003_theta1", "bh_n10000_theta1", "bh_n10000_theta05", "bh_close_pair_n512"]


@pytest.mark.parametrize("name", BH_GOLDEN)
def test_reference_vote_equals_oracle_vote(O, name):
    """The accept/open vote of the COMPILED REFERENCE (benchmark.cpp:259-266: width * (vrcpps + one Newton step) <
    theta in its -ffast-math build) against the oracle's IEEE vote, group by group.  The goldens carry the reference's
    own counters (hooks at benchmark.cpp:245 / :267, oracle/ref_hooks.h): nodes popped, nodes accepted and a hash of
    the accepted node indices in walk order, per group of 8 bodies.  Measured: not one group differs (also 0 of
    125 000 groups = 1.03e8 / 3.45e8 votes at N = 10^6, theta = 1 / 0.5, tools/vote_census.py) => gate: exactly 0."""
    g, b = load_golden(O, name)
    t = O.Tree(b.n).build(b, float(g["spread"]))
    v, a, h = t.group_counters(b, float(g["theta"]))
    assert len(v) == b.n // 8 == len(g["grp_visits"])
    differ = int(((v != g["grp_visits"]) | (a != g["grp_accepts"]) | (h != g["grp_hash"])).sum())
    assert differ == 0, f"{differ} of {len(v)} groups vote differently from the compiled reference"
    _, _, cnt = t.accel(b, float(g["theta"]))
    assert cnt["visits"] == int(g["step_visits"][0]) == int(v.sum())
    assert cnt["accepts"] == int(g["step_accepts"][0]) == int(a.sum())


@pytest.mark.parametrize("name", BH_GOLDEN)
def test_reference_vote_totals_per_step(O, name):
    """Over several steps the states drift apart by the reference's 12-bit vrcpps in the node kernel
    (benchmark.cpp:213), so later votes can flip.  Measured on the goldens: steps 1-2 identical everywhere; from
    step 3 on at N = 10 000, theta = 1 the totals differ by <= 20 of 6.2e5 visits (3.2e-5).  Gate: step 1 exact,
    every step within 1e-4 relative."""
    g, b = load_golden(O, name)
    sv, sa = O.bh_steps_counted(b, float(g["spread"]), float(g["theta"]), float(g["dt"]), int(g["nsteps_final"]))
    assert sv[0] == g["step_visits"][0] and sa[0] == g["step_accepts"][0]
    assert np.all(np.abs(sv - g["step_visits"]) <= 1e-4 * g["step_visits"])
    assert np.all(np.abs(sa - g["step_accepts"]) <= 1e-4 * g["step_accepts"])


def test_reference_vote_live_if_present(O):
    """Same census against the reference RUN ON THIS HOST (its vrcpps bits are CPU-vendor specific):
    N = 65 536, theta = 1 — 5.1e6 group-visits."""
    if not O.ref_counted_available():
        pytest.skip("oracle/_ref/libnbody_ref_counted.so not built")
    n = 65536
    b = O.init_disc(n, seed=777)
    r = O.ref_run(n, use_barnes_hut=1, nsteps=1, bodies=b, theta=1.0, count_votes=True)
    v, a, h = O.Tree(n).build(b, 400.0).group_counters(b, 1.0)
    differ = int(((v != r["grp_visits"]) | (a != r["grp_accepts"]) | (h != r["grp_hash"])).sum())
    assert differ == 0, f"{differ} of {n // 8} groups differ"


def test_bh_error_vs_naive_matches_survey(O):
    """theta=1 error vs f64 naive median 3-5 %, theta=0.5 below 1 %."""
    _, b = load_golden(O, "bh_n10000_theta1")
    t = O.Tree(b.n).build(b, 400.0)
    nx, ny = O.naive_accel_f64(b)
    ax, ay, _ = t.accel(b, 1.0)
    e1 = np.median(rel_err(ax, ay, nx, ny))
    ax, ay, _ = t.accel(b, 0.5)
    e05 = np.median(rel_err(ax, ay, nx, ny))
    assert 0.01 < e1 < 0.08 and e05 < 0.015 and e05 < e1


@pytest.mark.parametrize("name", ["bh_n1024_theta1", "bh_n1003_theta1"])
def test_bh_short_trajectory_vs_reference(O, name):
    g, b = load_golden(O, name)
    O.bh_steps(b, float(g["spread"]), float(g["theta"]), float(g["dt"]), int(g["nsteps_final"]))
    for k in ("posX", "posY"):       # 2e-5 of the box half-width
        assert np.abs(getattr(b, k) - g["final_" + k]).max() <= 2e-5 * 400.0
    for k in ("velX", "velY"):       # 2e-4 of the largest speed
        assert np.abs(getattr(b, k) - g["final_" + k]).max() <= 2e-4 * np.abs(g["final_" + k]).max()


def test_bh_use_parallel_false_does_nothing(O):
    """benchmark.cpp:359: useParallel=false in the BH branch builds the tree and nothing else."""
    _, b = load_golden(O, "bh_n1024_theta1")
    c = b.copy()
    O.bh_steps(c, 400.0, 1.0, 0.05, 2, use_parallel=0)
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(getattr(c, k), getattr(b, k))


def test_reference_live_if_present(O):
    """When oracle/_ref is present, re-run the reference and check the committed fixture is reproduced
    (bit-exact on the build container's CPU; distribution-level elsewhere because vrcpps differs by vendor)."""
    if not O.ref_available():
        pytest.skip("oracle/_ref/libnbody_ref.so not built")
    g, b = load_golden(O, "naive_n1024")
    r = O.ref_run(b.n, use_barnes_hut=0, nsteps=1, bodies=b, dt=float(g["dt"]))
    e = rel_err(r["acc"][0], r["acc"][1], g["accX"], g["accY"])
    assert np.percentile(e, 99) <= 3e-4


@pytest.mark.parametrize("bh,n", [(0, 1000), (0, 1003), (1, 1000), (1, 1003), (1, 7)])
def test_main_cpp_physics_vs_benchmark_cpp_physics(O, bh, n):
    """main.cpp's physics block (main.cpp:485-642) against benchmark.cpp's, both compiled from
    /root/reference (oracle/build_ref.sh) and run on the same bodies.  The naive branches are the same code: bit-identical.
    The Barnes-Hut branches differ in two things, both mirrored by ref_compat = 2: main.cpp lets the N%8 tail bodies
    drift force-free (main.cpp:535-544), and its tree normalises a centre of mass with one reciprocal and two products
    where benchmark.cpp divides twice (main.cpp:173-175 vs benchmark.cpp:139-140) — a last-bit difference."""
    if not (O.ref_available() and O.ref_mainphys_available()):
        pytest.skip("oracle/_ref was not built (needs /root/reference at build time)")
    b = O.init_disc(n, seed=33)
    bench = O.ref_run(n, use_barnes_hut=bh, nsteps=1, bodies=b, want_pool=bool(bh))
    main = O.ref_run(n, use_barnes_hut=bh, nsteps=1, bodies=b, main_cpp=True, want_pool=bool(bh))
    lim = n - n % 8
    for f in ("posX", "posY", "velX", "velY"):
        x, y = getattr(bench["final"], f)[:lim], getattr(main["final"], f)[:lim]
        if bh:
            assert np.allclose(x, y, rtol=2e-5, atol=1e-5), f
        else:
            assert np.array_equal(x, y), f
    for f in ("velX", "velY"):
        assert np.array_equal(getattr(main["final"], f)[lim:], getattr(b, f)[lim:])      # never accelerated in either
    for f, v in (("posX", "velX"), ("posY", "velY")):
        assert np.array_equal(getattr(bench["final"], f)[lim:], getattr(b, f)[lim:])     # benchmark.cpp: frozen
        if bh and n % 8:
            drift = (getattr(b, f)[lim:].astype(np.float64) + getattr(b, v)[lim:].astype(np.float64) * np.float64(np.float32(0.05))).astype(np.float32)
            assert np.allclose(getattr(main["final"], f)[lim:], drift, rtol=2e-7, atol=0)    # main.cpp: pos += vel*dt
            assert not np.array_equal(getattr(main["final"], f)[lim:], getattr(b, f)[lim:])
        else:
            assert np.array_equal(getattr(main["final"], f)[lim:], getattr(b, f)[lim:])
    if bh and n >= 1000:
        # the oracle's tree in main.cpp flavour is main.cpp's own node pool, bit for bit (and not benchmark.cpp's)
        try:
            O.set_main_cpp_flavour(True)
            t = O.Tree(n).build(b, 400.0)
            pool = t.pool()
        finally:
            O.set_main_cpp_flavour(False)
        assert t.nodes_used == main["nodes_used"] == bench["nodes_used"]
        for f in ("tlX", "tlY", "brX", "brY", "bodyIndex", "children", "mass", "comX", "comY"):
            assert np.array_equal(pool[f], main["pool"][f]), f
        assert not (np.array_equal(main["pool"]["comX"], bench["pool"]["comX"]) and np.array_equal(main["pool"]["comY"], bench["pool"]["comY"]))


def test_a_timing_call_runs_the_unprobed_reference(O, monkeypatch):
    """the Barnes-Hut CPU baseline must be the reference, not the reference plus a probe.  A default
    ref_run (what bench.py's cpu_baseline*, tools/dropin_time.py and tools/bh_time.py call) goes to libnbody_ref.so,
    which is built WITHOUT the counter hooks at benchmark.cpp:245/267, and hands over NULL counter arrays; only
    count_votes=True reaches the counted library.  Both builds give the same physics."""
    if not (O.ref_available() and O.ref_counted_available()):
        pytest.skip("oracle/_ref not built")
    n = 4096
    b = O.init_disc(n, seed=99)
    plain = O.ref_run(n, use_barnes_hut=1, nsteps=2, bodies=b, theta=1.0)
    assert plain["counter_pointers_null"] and not plain["counted"] and "grp_visits" not in plain
    assert O._ref.ref_votes_counted() == 0
    counted = O.ref_run(n, use_barnes_hut=1, nsteps=2, bodies=b, theta=1.0, count_votes=True)
    assert not counted["counter_pointers_null"] and counted["counted"] and int(counted["step_visits"][0]) > 0
    assert O._ref_counted.ref_votes_counted() == 1
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(getattr(plain["final"], k), getattr(counted["final"], k))
    with pytest.raises(ValueError):
        O.ref_run(n, use_barnes_hut=0, nsteps=1, bodies=b, count_votes=True)
    # bench.py's Barnes-Hut baseline leg: spy on the call it makes
    import bench
    seen = {}
    real = O.ref_run

    def spy(*a, **kw):
        seen.update(kw)
        return real(*a, **kw)
    monkeypatch.setattr(O, "ref_run", spy)
    out = bench.cpu_baseline_bh(2048, 1.0)
    assert out["kind"] == "reference" and not seen.get("count_votes", False)
