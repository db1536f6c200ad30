"""n-body_amd — MI355X-native physics step for fejzo9/N-Body (Python host binding).

Thin ctypes layer over the C-ABI in ``include/nbody.h`` (``lib/libnbody_amd.so``), which is what a
C++ host (the reference's benchmark.cpp / main.cpp loops) binds directly.
Argument names follow the reference's config.json keys (main.cpp:374-390).

There is NO CPU fallback: if the HIP library is missing or no gfx950 device is visible, creating a
simulation raises.  The directory name contains a hyphen, so import it with
``importlib`` (see ``load_package()`` in ``__graft_entry__.py``) under the module name ``nbody_amd``.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBODY_AMD_LIB") or os.path.join(_HERE, "lib", "libnbody_amd.so")   # env: developer builds

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED = 0, -1, -2, -3, -4

# every symbol include/nbody.h declares
ABI_SYMBOLS = (
    "nbody_default_config", "nbody_shard_range", "nbody_create", "nbody_destroy", "nbody_set_stream",
    "nbody_upload", "nbody_upload_state", "nbody_step", "nbody_compute_accel", "nbody_sync", "nbody_download", "nbody_frame_publish",
    "nbody_frame_acquire", "nbody_comm_unique_id", "nbody_comm_init", "nbody_exchange_positions", "nbody_comm_plan", "nbody_get_accel",
    "nbody_diagnostics", "nbody_device_ptr", "nbody_profile_enable", "nbody_profile_read",
    "nbody_profile_reset", "nbody_profile_samples", "nbody_get_info", "nbody_bh_tree_nodes", "nbody_bh_counters",
    "nbody_bh_group_counters", "nbody_accel_noalloc", "nbody_sort_pairs_u64",
    "nbody_last_error", "nbody_abi_version",
)


class Config(C.Structure):
    """struct nbody_config (include/nbody.h)."""
    _fields_ = [("num_bodies", C.c_int32), ("spread", C.c_float), ("mass", C.c_float), ("dt", C.c_float),
                ("theta", C.c_float), ("use_barnes_hut", C.c_int32), ("use_parallel", C.c_int32),
                ("ref_compat", C.c_int32), ("device", C.c_int32), ("shard_begin", C.c_int32),
                ("shard_count", C.c_int32), ("force_variant", C.c_int32), ("adaptive_box", C.c_int32), ("bh_fast_math", C.c_int32),
                ("bh_sorted", C.c_int32), ("reserved", C.c_int32 * 1)]


class NBodyError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"nbody error {code}: {msg}")
        self.code = code


_lib = None
_torch_first = None          # was PyTorch already imported when the library was loaded?  (see lib())
_f32p = C.POINTER(C.c_float)


def lib():
    """Load libnbody_amd.so (raises if it has not been built — no fallback)."""
    global _lib, _torch_first
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not found: build it with `make lib` (hipcc, gfx950)")
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Whichever runtime a process loads FIRST serves every
        # later user (same SONAME); but if this library comes first it binds /opt/rocm's runtime, torch then brings up
        # its own beside it, and the second one finds no GPU ("No HIP GPUs are available") or aborts on a stream of the
        # other.  A process that uses both must import torch first (bench.py and tests/conftest.py do);
        # torch_interop_ok() lets the torch-facing code (sharded.py) refuse loudly otherwise.
        import sys
        _torch_first = "torch" in sys.modules
        L = C.CDLL(LIB_PATH)
        L.nbody_default_config.argtypes = [C.POINTER(Config)]
        L.nbody_default_config.restype = None
        L.nbody_shard_range.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.nbody_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
        L.nbody_destroy.argtypes = [C.c_void_p]
        L.nbody_destroy.restype = None
        L.nbody_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.nbody_upload.argtypes = [C.c_void_p] + [_f32p] * 6
        L.nbody_upload_state.argtypes = [C.c_void_p] + [_f32p] * 4
        L.nbody_step.argtypes = [C.c_void_p, C.c_int32]
        L.nbody_compute_accel.argtypes = [C.c_void_p]
        L.nbody_sync.argtypes = [C.c_void_p]
        L.nbody_download.argtypes = [C.c_void_p] + [_f32p] * 4
        L.nbody_frame_publish.argtypes = [C.c_void_p]
        L.nbody_frame_acquire.argtypes = [C.c_void_p, C.POINTER(_f32p), C.POINTER(_f32p), C.POINTER(C.c_int64)]
        L.nbody_comm_unique_id.argtypes = [C.c_char_p]
        L.nbody_comm_init.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p]
        L.nbody_exchange_positions.argtypes = [C.c_void_p]
        L.nbody_comm_plan.argtypes = [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.nbody_get_accel.argtypes = [C.c_void_p, _f32p, _f32p]
        L.nbody_diagnostics.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.nbody_device_ptr.argtypes = [C.c_void_p, C.c_int32]
        L.nbody_device_ptr.restype = C.c_void_p
        L.nbody_profile_enable.argtypes = [C.c_void_p, C.c_int32]
        L.nbody_profile_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.nbody_profile_reset.argtypes = [C.c_void_p]
        L.nbody_profile_samples.argtypes = [C.c_void_p, C.c_int32, _f32p, C.c_int64]
        L.nbody_profile_samples.restype = C.c_int64
        L.nbody_get_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 3
        L.nbody_bh_tree_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.nbody_bh_tree_nodes.restype = C.c_int64
        L.nbody_bh_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.nbody_bh_group_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.nbody_accel_noalloc.argtypes = [C.c_int32] + [C.c_void_p] * 7
        L.nbody_sort_pairs_u64.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.nbody_last_error.restype = C.c_char_p
        L.nbody_abi_version.restype = C.c_int
        _lib = L
    return _lib


def _check(rc):
    if rc != OK:
        raise NBodyError(rc, lib().nbody_last_error().decode())


def default_config():
    cfg = Config()
    lib().nbody_default_config(C.byref(cfg))
    return cfg


def torch_interop_ok():
    """False when the library was loaded BEFORE torch in this process (two HIP runtimes: see lib())."""
    return _lib is None or bool(_torch_first)


def shard_range(num_bodies, nranks, rank):
    """Contiguous i-shard [begin, begin+count) of `rank` (nbody_shard_range)."""
    b, c = C.c_int32(), C.c_int32()
    _check(lib().nbody_shard_range(num_bodies, nranks, rank, C.byref(b), C.byref(c)))
    return b.value, c.value


def comm_plan(num_bodies, nranks):
    """(spans, in_place_allgather) of the native exchange (nbody_comm_plan; host arithmetic only)."""
    b, c, eq = (C.c_int32 * nranks)(), (C.c_int32 * nranks)(), C.c_int32()
    _check(lib().nbody_comm_plan(num_bodies, nranks, b, c, C.byref(eq)))
    return [(b[r], c[r]) for r in range(nranks)], bool(eq.value)


def comm_unique_id():
    """128-byte RCCL id, created by rank 0 and handed to every rank's NBodySim.comm_init (nbody_comm_unique_id)."""
    buf = C.create_string_buffer(128)
    _check(lib().nbody_comm_unique_id(buf))
    return buf.raw


def sort_pairs_u64(keys, vals, repeats=1, truncated=False):
    """nbody_sort_pairs_u64 (parity hook): the Barnes-Hut build's own device sort of (uint64 key, int32 value) pairs,
    stable; `repeats` sorts in a row on one state.  Returns (sorted keys, values)."""
    k = np.ascontiguousarray(keys, np.uint64)
    v = np.ascontiguousarray(vals, np.int32)
    ok = np.empty_like(k); ov = np.empty_like(v)
    _check(lib().nbody_sort_pairs_u64(k.ctypes.data, v.ctypes.data, len(k), ok.ctypes.data, ov.ctypes.data, int(repeats), int(bool(truncated))))
    return ok, ov


def accel_noalloc(num_bodies, dev_m, dev_posX, dev_posY, dev_radius, dev_accX, dev_accY, hip_stream=None):
    """nbody_accel_noalloc: all arrays are raw DEVICE addresses (ints) owned by the caller; enqueues on hip_stream of
    the current device and returns.  Allocates nothing (include/nbody.h)."""
    _check(lib().nbody_accel_noalloc(int(num_bodies), *[C.c_void_p(int(p)) for p in
                                                        (dev_m, dev_posX, dev_posY, dev_radius, dev_accX, dev_accY)],
                                     C.c_void_p(hip_stream or 0)))


def _as_f32(a, n, name):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.shape != (n,):
        raise ValueError(f"{name}: expected shape ({n},), got {a.shape}")
    return a


def _ptr(a):
    return a.ctypes.data_as(_f32p)


FIELD_INDEX = dict(m=0, posX=1, posY=2, velX=3, velY=4, radius=5, accX=6, accY=7)


class DeviceArray:
    """Zero-copy view of one SoA device array for frameworks (``__cuda_array_interface__``)."""

    def __init__(self, ptr, n, owner):
        self._owner = owner
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f4", data=(int(ptr), False), version=2)


class NBodySim:
    """One simulation context on one GPU (one i-shard).  Mirrors the reference's parameter block
    (benchmark.cpp:300-306) and loop body (benchmark.cpp:354-511)."""

    def __init__(self, numBodies, *, spread=400.0, mass=1000.0, dt=0.05, theta=1.0, useBarnesHut=False,
                 useParallel=True, ref_compat=True, device=-1, shard=None, force_variant=0, adaptive_box=False,
                 bh_fast_math=False, bh_sorted=False):
        cfg = default_config()
        cfg.num_bodies, cfg.spread, cfg.mass, cfg.dt, cfg.theta = int(numBodies), spread, mass, dt, theta
        cfg.use_barnes_hut, cfg.use_parallel = int(bool(useBarnesHut)), int(bool(useParallel))
        cfg.ref_compat, cfg.device, cfg.force_variant = int(ref_compat), int(device), int(force_variant)
        cfg.adaptive_box, cfg.bh_fast_math = int(bool(adaptive_box)), int(bool(bh_fast_math))
        cfg.bh_sorted = int(bool(bh_sorted))
        if shard is not None:
            cfg.shard_begin, cfg.shard_count = int(shard[0]), int(shard[1])
        self.cfg = cfg
        self.n = int(numBodies)
        self._h = C.c_void_p()
        _check(lib().nbody_create(C.byref(cfg), C.byref(self._h)))

    # -- state ------------------------------------------------------------------------------------
    def upload(self, m, posX, posY, velX, velY, radius):
        arrs = [_as_f32(a, self.n, k) for a, k in zip((m, posX, posY, velX, velY, radius),
                                                      ("m", "posX", "posY", "velX", "velY", "radius"))]
        _check(lib().nbody_upload(self._h, *[_ptr(a) for a in arrs]))
        return self

    def upload_state(self, posX, posY, velX, velY):
        """New positions/velocities for the bodies already uploaded (nbody_upload_state)."""
        arrs = [_as_f32(a, self.n, k) for a, k in zip((posX, posY, velX, velY), ("posX", "posY", "velX", "velY"))]
        _check(lib().nbody_upload_state(self._h, *[_ptr(a) for a in arrs]))
        return self

    def download(self):
        out = [np.empty(self.n, np.float32) for _ in range(4)]
        _check(lib().nbody_download(self._h, *[_ptr(a) for a in out]))
        return dict(posX=out[0], posY=out[1], velX=out[2], velY=out[3])

    # -- position streaming for a viewer (main.cpp:646-655 reads posX/posY every frame) -----------
    def frame_publish(self):
        """Enqueue an asynchronous snapshot of posX/posY (after the steps enqueued so far) into pinned host memory."""
        _check(lib().nbody_frame_publish(self._h))
        return self

    def frame_acquire(self):
        """Wait for the oldest outstanding frame: (posX, posY, step) — zero-copy views of library-owned pinned
        memory, valid until two further frames have been published."""
        px, py, step = _f32p(), _f32p(), C.c_int64()
        _check(lib().nbody_frame_acquire(self._h, C.byref(px), C.byref(py), C.byref(step)))
        return (np.ctypeslib.as_array(px, shape=(self.n,)), np.ctypeslib.as_array(py, shape=(self.n,)), step.value)

    # -- native RCCL exchange (hosts without a framework; sharded.py uses torch.distributed instead) -----------
    def comm_init(self, rank, nranks, unique_id):
        _check(lib().nbody_comm_init(self._h, int(rank), int(nranks), bytes(unique_id)))
        return self

    def exchange_positions(self):
        _check(lib().nbody_exchange_positions(self._h))
        return self

    def accel(self):
        ax, ay = np.empty(self.n, np.float32), np.empty(self.n, np.float32)
        _check(lib().nbody_get_accel(self._h, _ptr(ax), _ptr(ay)))
        return ax, ay

    # -- stepping ---------------------------------------------------------------------------------
    def step(self, nsteps=1):
        _check(lib().nbody_step(self._h, nsteps))
        return self

    def compute_accel(self):
        _check(lib().nbody_compute_accel(self._h))
        return self

    def sync(self):
        _check(lib().nbody_sync(self._h))
        return self

    def set_stream(self, hip_stream):
        _check(lib().nbody_set_stream(self._h, C.c_void_p(hip_stream)))
        return self

    def diagnostics(self):
        out = (C.c_double * 5)()
        _check(lib().nbody_diagnostics(self._h, out))
        return dict(zip(("Ekin", "Epot", "Px", "Py", "Lz"), out))

    # -- interop / introspection ------------------------------------------------------------------
    def device_array(self, field):
        ptr = lib().nbody_device_ptr(self._h, FIELD_INDEX[field])
        if not ptr:
            raise NBodyError(ERR_INVALID, lib().nbody_last_error().decode())
        return DeviceArray(ptr, self.n, self)

    def profile(self, on=True):
        _check(lib().nbody_profile_enable(self._h, int(on)))
        return self

    def profile_reset(self):
        _check(lib().nbody_profile_reset(self._h))
        return self

    def profile_read(self, which=0):
        ms, cnt = C.c_double(), C.c_int64()
        _check(lib().nbody_profile_read(self._h, which, C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def profile_samples(self, which=0):
        """Per-launch durations (ms) of that kernel class since the last profile_reset."""
        n = lib().nbody_profile_samples(self._h, which, None, 0)
        if n < 0:
            raise NBodyError(int(n), lib().nbody_last_error().decode())
        out = np.zeros(max(int(n), 1), np.float32)
        lib().nbody_profile_samples(self._h, which, _ptr(out), int(n))
        return out[:int(n)]

    def info(self):
        v, s, b = C.c_int32(), C.c_int32(), C.c_int32()
        _check(lib().nbody_get_info(self._h, C.byref(v), C.byref(s), C.byref(b)))
        return dict(force_variant=v.value, j_slices=s.value, blocks=b.value)

    def bh_tree(self, max_nodes=None):
        """Node pool of the last build as 48-byte QuadNode records (benchmark.cpp:52-60); node 0 is the root.
        max_nodes limits the copy (a 2^24-body tree has ~4.6e7 nodes = 2.2 GB); len() of the result is then the
        number copied, and self.bh_nodes_total the number of nodes in the tree."""
        n = lib().nbody_bh_tree_nodes(self._h, None, 0)
        if n < 0:
            raise NBodyError(int(n), lib().nbody_last_error().decode())
        self.bh_nodes_total = int(n)
        if max_nodes is not None:
            n = min(n, int(max_nodes))
        from numpy import dtype
        node = dtype([("mass", "f4"), ("comX", "f4"), ("comY", "f4"), ("tlX", "f4"), ("tlY", "f4"),
                      ("brX", "f4"), ("brY", "f4"), ("bodyIndex", "i4"), ("children", "i4", (4,))])
        out = np.zeros(n, node)
        got = lib().nbody_bh_tree_nodes(self._h, out.ctypes.data, n)
        if got < 0:
            raise NBodyError(int(got), lib().nbody_last_error().decode())
        return out[:got]

    def bh_counters(self):
        out = (C.c_int64 * 2)()
        _check(lib().nbody_bh_counters(self._h, out))
        return dict(visits=out[0], accepts=out[1])

    def bh_group_counters(self):
        """(visits, accepts) of the last walk per group of 8 consecutive bodies (nbody_bh_group_counters)."""
        g = (self.n + 7) // 8
        v, a = np.zeros(g, np.int32), np.zeros(g, np.int32)
        _check(lib().nbody_bh_group_counters(self._h, v.ctypes.data_as(C.POINTER(C.c_int32)),
                                             a.ctypes.data_as(C.POINTER(C.c_int32))))
        return v, a

    def close(self):
        if self._h:
            lib().nbody_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
