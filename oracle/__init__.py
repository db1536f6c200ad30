"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle and the compiled reference.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package, and only as the checker.  The product package (``n-body_amd``) never does.

* ``libnbody_oracle.so``      — oracle/nbody_oracle.c, the plain-C restatement.
* ``_ref/libnbody_ref.so``    — /root/reference/benchmark.cpp itself, compiled by build_ref.sh; nothing added inside its
  hot loops: every TIMING of the reference goes through this one.
* ``_ref/libnbody_ref_counted.so`` — the same with per-group vote counters inside the walk (benchmark.cpp:245, 267);
  only ``ref_run(..., count_votes=True)`` loads it (vote-parity tests), never a timing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "libnbody_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libnbody_ref.so")
REF_COUNTED_SO = os.path.join(_HERE, "_ref", "libnbody_ref_counted.so")   # the same + vote counters inside the walk
REF_MAINPHYS_SO = os.path.join(_HERE, "_ref", "libnbody_ref_mainphys.so")       # main.cpp's physics block (485-642)
REF_DROPIN_SO = os.path.join(_HERE, "_ref", "libnbody_ref_dropin.so")   # same file, physics block -> include/nbody.h

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")

NODE_DTYPE = np.dtype([("mass", "f4"), ("comX", "f4"), ("comY", "f4"),
                       ("tlX", "f4"), ("tlY", "f4"), ("brX", "f4"), ("brY", "f4"),
                       ("bodyIndex", "i4"), ("children", "i4", (4,))])
assert NODE_DTYPE.itemsize == 48

_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_SO)
        L.oracle_mt19937_uniform01.argtypes = [C.c_uint32, C.c_int, _f32p]
        L.oracle_mt19937_raw.argtypes = [C.c_uint32, C.c_int, np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")]
        L.oracle_init_disc.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_float] + [_f32p] * 6
        L.oracle_naive_accel_f64.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, _f64p, _f64p]
        L.oracle_naive_accel_f64_range.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, _f64p, _f64p]
        L.oracle_naive_accel_f64_subset.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _i32p, C.c_int, _f64p, _f64p]
        L.oracle_naive_accel_f32.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, _f32p, _f32p]
        L.oracle_euler_f32.argtypes = [C.c_int, _f32p, _f32p, _f32p, _f32p, _f32p, _f32p, C.c_float, C.c_int]
        L.oracle_naive_steps.argtypes = [C.c_int] + [_f32p] * 6 + [C.c_float, C.c_int, C.c_int]
        L.oracle_diagnostics.argtypes = [C.c_int] + [_f32p] * 6 + [_f64p]
        L.oracle_tree_create.restype = C.c_void_p
        L.oracle_tree_create.argtypes = [C.c_int]
        L.oracle_tree_destroy.argtypes = [C.c_void_p]
        L.oracle_tree_build.argtypes = [C.c_void_p, C.c_int, C.c_float, _f32p, _f32p, _f32p]
        L.oracle_tree_nodes_used.argtypes = [C.c_void_p]
        L.oracle_tree_pool.restype = C.c_void_p
        L.oracle_tree_pool.argtypes = [C.c_void_p]
        L.oracle_tree_stats.argtypes = [C.c_void_p, _i32p]
        L.oracle_bh_accel.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_float, C.c_int, C.c_int,
                                      _f64p, _f64p, _i64p]
        L.oracle_bh_group_counters.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_float, _i32p, _i32p,
                                               np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
        L.oracle_bh_steps.argtypes = [C.c_int] + [_f32p] * 6 + [C.c_float, C.c_float, C.c_float, C.c_int, C.c_int]
        L.oracle_path_keys.argtypes = [C.c_int, C.c_float, _f32p, _f32p, np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")]
        L.oracle_bh_accel_ordered.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, _f32p, C.c_float, _i32p, _f64p, _f64p, _i64p,
                                              _i32p, _i32p]
        L.oracle_bh_steps_counted.argtypes = L.oracle_bh_steps.argtypes + [_i64p, _i64p]
        _lib = L
    return _lib


class Bodies:
    """SoA state with the reference's field names (benchmark.cpp:24-49)."""
    FIELDS = ("m", "posX", "posY", "velX", "velY", "radius")

    def __init__(self, n):
        self.n = n
        for f in self.FIELDS:
            setattr(self, f, np.zeros(n, np.float32))

    def copy(self):
        b = Bodies(self.n)
        for f in self.FIELDS:
            setattr(b, f, getattr(self, f).copy())
        return b

    def arrays(self):
        return [getattr(self, f) for f in self.FIELDS]


def init_disc(n, spread=400.0, mass=1000.0, seed=12345, vel_factor=0.3):
    b = Bodies(n)
    lib().oracle_init_disc(n, spread, mass, seed, vel_factor, *b.arrays())
    return b


def init_plummer(n, scale=50.0, mass=1000.0, seed=12345, rmax_factor=20.0):
    """2-D Plummer-like clustered disc (BASELINE config 5).  The reference has NO such generator
    (SURVEY.md §0); this one is defined by the build: surface density ∝ (1+r²/a²)^-2 =>
    r = a·sqrt(u/(1-u)) truncated at rmax_factor·a, circular-ish tangential velocities."""
    rng = np.random.default_rng(seed)
    u = rng.random(n)
    umax = rmax_factor ** 2 / (1.0 + rmax_factor ** 2)
    u = u * umax
    r = scale * np.sqrt(u / (1.0 - u))
    phi = rng.random(n) * 2 * np.pi
    b = Bodies(n)
    b.m[:] = mass
    b.radius[:] = 8.0
    b.posX[:] = (r * np.cos(phi)).astype(np.float32)
    b.posY[:] = (r * np.sin(phi)).astype(np.float32)
    menc = mass * n * (r ** 2 / (r ** 2 + scale ** 2))
    v = np.sqrt(menc / np.maximum(r, 1.0)) * 0.3
    b.velX[:] = (-np.sin(phi) * v).astype(np.float32)
    b.velY[:] = (np.cos(phi) * v).astype(np.float32)
    return b


def naive_accel_f64(b, ref_compat=1):
    ax = np.zeros(b.n); ay = np.zeros(b.n)
    lib().oracle_naive_accel_f64(b.n, b.m, b.posX, b.posY, b.radius, ref_compat, ax, ay)
    return ax, ay


def naive_accel_f64_range(b, i0, i1):
    """Accelerations of targets [i0, i1) against all sources (one rank's shard)."""
    ax = np.zeros(max(i1 - i0, 0)); ay = np.zeros(max(i1 - i0, 0))
    if i1 > i0:
        lib().oracle_naive_accel_f64_range(b.n, b.m, b.posX, b.posY, b.radius, i0, i1, ax, ay)
    return ax, ay


def naive_accel_f64_subset(b, idx):
    idx = np.ascontiguousarray(idx, np.int32)
    ax = np.zeros(len(idx)); ay = np.zeros(len(idx))
    lib().oracle_naive_accel_f64_subset(b.n, b.m, b.posX, b.posY, b.radius, idx, len(idx), ax, ay)
    return ax, ay


def naive_accel_f32(b, ref_compat=1):
    ax = np.zeros(b.n, np.float32); ay = np.zeros(b.n, np.float32)
    lib().oracle_naive_accel_f32(b.n, b.m, b.posX, b.posY, b.radius, ref_compat, ax, ay)
    return ax, ay


def naive_steps(b, dt, nsteps, ref_compat=1):
    """In-place nsteps of f64-force + f32 Euler."""
    lib().oracle_naive_steps(b.n, *b.arrays(), dt, nsteps, ref_compat)
    return b


def diagnostics(b):
    out = np.zeros(5)
    lib().oracle_diagnostics(b.n, *b.arrays(), out)
    return out


class Tree:
    def __init__(self, n):
        self.n = n
        self.h = lib().oracle_tree_create(n)

    def build(self, b, spread):
        lib().oracle_tree_build(self.h, b.n, spread, b.m, b.posX, b.posY)
        return self

    @property
    def nodes_used(self):
        return lib().oracle_tree_nodes_used(self.h)

    def pool(self):
        n = self.nodes_used
        buf = (C.c_char * (48 * n)).from_address(lib().oracle_tree_pool(self.h))
        return np.frombuffer(buf, dtype=NODE_DTYPE, count=n).copy()

    def stats(self):
        out = np.zeros(4, np.int32)
        lib().oracle_tree_stats(self.h, out)
        return dict(nodes=int(out[0]), depth=int(out[1]), leaves=int(out[2]), bodies=int(out[3]))

    def accel(self, b, theta, f64=False, group_size=8):
        ax = np.zeros(b.n); ay = np.zeros(b.n); cnt = np.zeros(2, np.int64)
        lib().oracle_bh_accel(self.h, b.n, b.posX, b.posY, b.radius, theta, int(f64), group_size, ax, ay, cnt)
        return ax, ay, dict(visits=int(cnt[0]), accepts=int(cnt[1]))

    def accel_sorted(self, b, theta, spread, order=None):
        """bh_sorted walk (no reference counterpart): groups of 8 consecutive bodies in quadrant-path order."""
        if order is None:
            order = sorted_order(b, spread)
        ax = np.zeros(b.n); ay = np.zeros(b.n); cnt = np.zeros(2, np.int64)
        g = (b.n + 7) // 8
        gv = np.zeros(g, np.int32); ga = np.zeros(g, np.int32)
        lib().oracle_bh_accel_ordered(self.h, b.n, b.posX, b.posY, b.radius, theta, order, ax, ay, cnt, gv, ga)
        return ax, ay, dict(visits=int(cnt[0]), accepts=int(cnt[1]), groups=(gv, ga))

    def group_counters(self, b, theta):
        """Per group of 8 bodies: (visits, accepts, hash of the accepted node indices in walk order)."""
        g = b.n // 8
        v = np.zeros(g, np.int32); a = np.zeros(g, np.int32); h = np.zeros(g, np.uint64)
        lib().oracle_bh_group_counters(self.h, b.n, b.posX, b.posY, b.radius, theta, v, a, h)
        return v, a, h

    def __del__(self):
        try:
            lib().oracle_tree_destroy(self.h)
        except Exception:
            pass


def sorted_order(b, spread):
    """Bodies in the order the GPU build lists them: by (quadrant-path key, index); out-of-box bodies last."""
    keys = np.zeros(b.n, np.uint64)
    lib().oracle_path_keys(b.n, spread, b.posX, b.posY, keys)
    return np.argsort(keys, kind="stable").astype(np.int32)


def set_main_cpp_flavour(on):
    """Tree arithmetic of main.cpp (centre of mass = sum * (1/mass), main.cpp:173-175) instead of benchmark.cpp's
    (sum / mass) for every tree built from now on."""
    lib().oracle_set_main_cpp_flavour(int(bool(on)))


def bh_steps(b, spread, theta, dt, nsteps, use_parallel=1):
    lib().oracle_bh_steps(b.n, *b.arrays(), spread, theta, dt, nsteps, use_parallel)
    return b


def bh_steps_counted(b, spread, theta, dt, nsteps):
    """bh_steps + the walk's {visits, accepts} totals of every step."""
    sv = np.zeros(max(nsteps, 1), np.int64); sa = np.zeros(max(nsteps, 1), np.int64)
    lib().oracle_bh_steps_counted(b.n, *b.arrays(), spread, theta, dt, nsteps, 1, sv, sa)
    return sv[:nsteps], sa[:nsteps]


# ---------------------------------------------------------------------------------------------
# The reference itself (oracle/_ref/libnbody_ref.so)
# ---------------------------------------------------------------------------------------------
class _RefIO(C.Structure):
    _fp = C.POINTER(C.c_float)
    _fields_ = ([("use_barnes_hut", C.c_int), ("use_parallel", C.c_int), ("n", C.c_int),
                 ("spread", C.c_float), ("mass", C.c_float), ("theta", C.c_float), ("dt", C.c_float),
                 ("seed", C.c_uint), ("nsteps", C.c_int), ("log_enabled", C.c_int)]
                + [(k, C.POINTER(C.c_float)) for k in ("in_m", "in_px", "in_py", "in_vx", "in_vy", "in_radius")]
                + [(k, C.POINTER(C.c_float)) for k in ("init_m", "init_px", "init_py", "init_vx", "init_vy", "init_radius")]
                + [(k, C.POINTER(C.c_float)) for k in ("out_px", "out_py", "out_vx", "out_vy", "acc_x", "acc_y")]
                + [("tree_nodes_used", C.POINTER(C.c_int)), ("tree_root", C.POINTER(C.c_float)),
                   ("tree_pool_out", C.c_void_p), ("tree_pool_cap", C.c_int),
                   ("step_ms", C.POINTER(C.c_double)),
                   ("grp_visits", C.POINTER(C.c_int)), ("grp_accepts", C.POINTER(C.c_int)),
                   ("grp_hash", C.POINTER(C.c_ulonglong)),
                   ("step_visits", C.POINTER(C.c_longlong)), ("step_accepts", C.POINTER(C.c_longlong))])


_ref = None
_ref_counted = None
_ref_dropin = None


def ref_available():
    return os.path.exists(REF_SO)


def ref_counted_available():
    return os.path.exists(REF_COUNTED_SO)


def ref_dropin_available():
    return os.path.exists(REF_DROPIN_SO)


def ref_mainphys_available():
    return os.path.exists(REF_MAINPHYS_SO)


_ref_mainphys = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def ref_run(n, *, use_barnes_hut, nsteps, bodies=None, seed=12345, spread=400.0, mass=1000.0, theta=1.0,
            dt=0.05, use_parallel=1, want_pool=False, dropin=False, main_cpp=False, count_votes=False):
    """Run the compiled reference loop (benchmark.cpp:349-557) for nsteps.

    bodies=None -> the reference's own generator (benchmark.cpp:317-339) with `seed`.
    main_cpp=True -> main.cpp's physics block (main.cpp:485-642; needs `bodies`, no tree/acceleration read-back) instead
    of benchmark.cpp's.
    dropin=True -> the DROP-IN build: the same reference program with its physics block replaced by the C-ABI
    (needs a GPU; tree read-back fields stay empty).
    count_votes=True -> the COUNTED build (libnbody_ref_counted.so): per-group visit/accept counters and the hash of the
    accepted nodes from inside the reference's walk.  The default runs libnbody_ref.so, whose walk carries no probe and
    whose RefIO counter pointers stay NULL: that is the one to time (the counters cost 3-7x at N >= 2e5).
    Returns dict(init=Bodies, final=Bodies, acc=(ax, ay) of the last step, nodes_used, root, pool, step_ms)."""
    global _ref, _ref_counted, _ref_dropin, _ref_mainphys
    if count_votes and (main_cpp or dropin or not use_barnes_hut):
        raise ValueError("count_votes=True is for benchmark.cpp's own Barnes-Hut walk only")
    if main_cpp:
        if bodies is None:
            raise ValueError("main_cpp=True needs bodies (main.cpp's generator sits behind its SFML code)")
        if _ref_mainphys is None:
            _ref_mainphys = C.CDLL(REF_MAINPHYS_SO)
            _ref_mainphys.ref_main_physics_run.argtypes = [C.POINTER(_RefIO)]
            _ref_mainphys.ref_main_physics_run.restype = C.c_int
        entry = _ref_mainphys.ref_main_physics_run
    elif dropin:
        if _ref_dropin is None:
            _ref_dropin = C.CDLL(REF_DROPIN_SO)
            _ref_dropin.ref_benchmark_run_dropin.argtypes = [C.POINTER(_RefIO)]
            _ref_dropin.ref_benchmark_run_dropin.restype = C.c_int
        entry = _ref_dropin.ref_benchmark_run_dropin
    elif count_votes:
        if _ref_counted is None:
            _ref_counted = C.CDLL(REF_COUNTED_SO)
            _ref_counted.ref_benchmark_run.argtypes = [C.POINTER(_RefIO)]
            _ref_counted.ref_benchmark_run.restype = C.c_int
            if _ref_counted.ref_votes_counted() != 1:
                raise RuntimeError(f"{REF_COUNTED_SO} was built without REF_COUNT_VOTES")
        entry = _ref_counted.ref_benchmark_run
    else:
        if _ref is None:
            _ref = C.CDLL(REF_SO)
            if _ref.ref_votes_counted() != 0:
                raise RuntimeError(f"{REF_SO} carries the vote counters: timings of it would not be the reference's")
            _ref.ref_benchmark_run.argtypes = [C.POINTER(_RefIO)]
            _ref.ref_benchmark_run.restype = C.c_int
        entry = _ref.ref_benchmark_run
    io = _RefIO()
    io.use_barnes_hut, io.use_parallel, io.n = int(use_barnes_hut), int(use_parallel), n
    io.spread, io.mass, io.theta, io.dt = spread, mass, theta, dt
    io.seed, io.nsteps, io.log_enabled = seed, nsteps, 0
    keep = []
    if bodies is not None:
        src = bodies.copy()
        keep.append(src)
        io.in_m, io.in_px, io.in_py = _fp(src.m), _fp(src.posX), _fp(src.posY)
        io.in_vx, io.in_vy, io.in_radius = _fp(src.velX), _fp(src.velY), _fp(src.radius)
    init = Bodies(n)
    io.init_m, io.init_px, io.init_py = _fp(init.m), _fp(init.posX), _fp(init.posY)
    io.init_vx, io.init_vy, io.init_radius = _fp(init.velX), _fp(init.velY), _fp(init.radius)
    final = Bodies(n)
    io.out_px, io.out_py, io.out_vx, io.out_vy = _fp(final.posX), _fp(final.posY), _fp(final.velX), _fp(final.velY)
    ax = np.zeros(n, np.float32); ay = np.zeros(n, np.float32)
    io.acc_x, io.acc_y = _fp(ax), _fp(ay)
    used = C.c_int(0)
    root = np.zeros(3, np.float32)
    io.tree_nodes_used = C.pointer(used)
    io.tree_root = _fp(root)
    pool = None
    if want_pool:
        cap = max(8, 32 * n)
        pool = np.zeros(cap, NODE_DTYPE)
        io.tree_pool_out = pool.ctypes.data
        io.tree_pool_cap = cap
    ms = np.zeros(max(nsteps, 1))
    io.step_ms = ms.ctypes.data_as(C.POINTER(C.c_double))
    groups = max(n // 8, 1)
    gv = np.zeros(groups, np.int32); ga = np.zeros(groups, np.int32); gh = np.zeros(groups, np.uint64)
    sv = np.zeros(max(nsteps, 1), np.int64); sa = np.zeros(max(nsteps, 1), np.int64)
    if count_votes:
        io.grp_visits = gv.ctypes.data_as(C.POINTER(C.c_int))
        io.grp_accepts = ga.ctypes.data_as(C.POINTER(C.c_int))
        io.grp_hash = gh.ctypes.data_as(C.POINTER(C.c_ulonglong))
        io.step_visits = sv.ctypes.data_as(C.POINTER(C.c_longlong))
        io.step_accepts = sa.ctypes.data_as(C.POINTER(C.c_longlong))
    rc = entry(C.byref(io))
    if rc != 0:
        raise RuntimeError(f"ref_benchmark_run returned {rc}")
    srcb = bodies if bodies is not None else init
    final.m[:] = srcb.m
    final.radius[:] = srcb.radius
    if pool is not None:
        pool = pool[:used.value].copy()
    out = dict(init=init, final=final, acc=(ax, ay), nodes_used=used.value, root=root, pool=pool, step_ms=ms[:nsteps],
               counted=bool(count_votes),
               # what the call handed to the library (tests assert that a timing call hands over NULL)
               counter_pointers_null=not any(bool(getattr(io, k)) for k in
                                             ("grp_visits", "grp_accepts", "grp_hash", "step_visits", "step_accepts")))
    if count_votes:
        # the reference's own vote (benchmark.cpp:242-276): per group of 8, accumulated over the nsteps of this
        # run, and the totals of every step
        out.update(grp_visits=gv[:n // 8], grp_accepts=ga[:n // 8], grp_hash=gh[:n // 8],
                   step_visits=sv[:nsteps], step_accepts=sa[:nsteps])
    return out


# ---------------------------------------------------------------------------------------------
# main.cpp's own config.json reader (oracle/_ref/libnbody_ref_config.so)
# ---------------------------------------------------------------------------------------------
REF_CONFIG_SO = os.path.join(_HERE, "_ref", "libnbody_ref_config.so")


class _RefConfig(C.Structure):
    _fields_ = [("useBarnesHut", C.c_int), ("useParallel", C.c_int), ("numBodies", C.c_int), ("spread", C.c_float),
                ("mass", C.c_float), ("theta", C.c_float), ("dt", C.c_float), ("windowSize", C.c_int),
                ("windowTitle", C.c_char * 256), ("threw", C.c_int)]


_ref_config = None


def ref_config_available():
    return os.path.exists(REF_CONFIG_SO)


def ref_parse_config(filename="config.json"):
    """The reference's findConfigFile + parameter block (main.cpp:26-56, 337-402) on `filename`, searched from the
    CURRENT working directory as the reference does.  Returns a dict, or None when its stoi/stod threw."""
    global _ref_config
    if _ref_config is None:
        _ref_config = C.CDLL(REF_CONFIG_SO)
        _ref_config.ref_main_parse_config.argtypes = [C.c_char_p, C.POINTER(_RefConfig)]
    out = _RefConfig()
    _ref_config.ref_main_parse_config(filename.encode(), C.byref(out))
    if out.threw:
        return None
    return dict(useBarnesHut=out.useBarnesHut, useParallel=out.useParallel, numBodies=out.numBodies, spread=out.spread,
                mass=out.mass, theta=out.theta, dt=out.dt, windowSize=out.windowSize,
                windowTitle=out.windowTitle.decode(errors="replace"))
