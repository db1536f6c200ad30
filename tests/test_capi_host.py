"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/nbody.h declares,
its pure-host entry points behave, and (without a GPU) creation fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(nb):
    hdr = open(os.path.join(ROOT, "include", "nbody.h")).read()
    declared = set(re.findall(r"\b(nbody_[a-z_0-9]+)\s*\(", hdr))
    declared -= {"nbody_config", "nbody_ctx"}
    assert declared == set(nb.ABI_SYMBOLS)
    raw = C.CDLL(nb.LIB_PATH)
    for s in declared:
        assert hasattr(raw, s), f"{s} declared in include/nbody.h but not exported"
    assert nb.lib().nbody_abi_version() == int(re.search(r"#define NBODY_ABI_VERSION (\d+)", hdr).group(1)) >= 2


def test_default_config_matches_reference_literals(nb):
    cfg = nb.default_config()       # benchmark.cpp:300-306
    assert (cfg.use_barnes_hut, cfg.use_parallel, cfg.num_bodies) == (1, 1, 10000)
    assert (cfg.spread, cfg.mass, cfg.theta) == (400.0, 1000.0, 1.0)
    assert abs(cfg.dt - 0.05) < 1e-9
    assert C.sizeof(nb.Config) == 64


@pytest.mark.parametrize("n,p", [(1 << 20, 8), (10000, 8), (1003, 4), (65536, 3), (17, 8), (8, 2), (5, 2)])
def test_shard_range_partitions(nb, n, p):
    spans = [nb.shard_range(n, p, r) for r in range(p)]
    pos = 0
    for b, c in spans:
        assert b == pos and c >= 0
        pos += c
    assert pos == n
    for b, c in spans[:-1]:
        assert b % 8 == 0 and c % 8 == 0          # groups of 8 (benchmark.cpp:421) never straddle GPUs
    counts = [c for _, c in spans[:-1]] or [0]
    assert max(counts) - min(counts) <= 8


def test_shard_range_rejects_bad_arguments(nb):
    with pytest.raises(nb.NBodyError):
        nb.shard_range(100, 0, 0)
    with pytest.raises(nb.NBodyError):
        nb.shard_range(100, 4, 4)


@pytest.mark.parametrize("n,p", [(1 << 20, 8), (10000, 8), (1003, 4), (65536, 3), (17, 8), (4096, 2), (5, 2)])
def test_native_exchange_plan_equals_the_torch_drivers(nb, n, p):
    """The native exchange (nbody_exchange_positions / `nbody_benchmark --dist`: one in-place ncclAllGather per array,
    or one ncclBroadcast per owner when the shards are ragged, comm_rccl.hip.h) follows nbody_comm_plan; the torch
    driver (sharded.exchange_slices) decides the same way from the same spans.  Both plans, played on numpy arrays,
    rebuild the full array from the ranks' owned slices."""
    spans, in_place = nb.comm_plan(n, p)
    assert spans == [nb.shard_range(n, p, r) for r in range(p)]
    counts = {c for _, c in spans}
    assert in_place == (len(counts) == 1 and spans[0][1] * p == n)        # exchange_slices' own condition
    truth = np.arange(n, dtype=np.float32)
    ranks = []
    for r, (b, c) in enumerate(spans):                                     # every rank holds only its slice
        a = np.full(n, -1.0, np.float32)
        a[b:b + c] = truth[b:b + c]
        ranks.append(a)
    if in_place:
        cnt = spans[0][1]
        gathered = np.concatenate([ranks[r][r * cnt:(r + 1) * cnt] for r in range(p)])   # ncclAllGather, rank order
        for a in ranks:
            a[:] = gathered
    else:
        for owner, (b, c) in enumerate(spans):                             # ncclBroadcast(root = owner) of its span
            for a in ranks:
                a[b:b + c] = ranks[owner][b:b + c]
    for a in ranks:
        assert np.array_equal(a, truth)


def test_barnes_hut_rejects_sizes_beyond_32_bit_node_indices(nb):
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodySim((1 << 28), useBarnesHut=True)
    assert e.value.code == nb.ERR_INVALID and "32-bit" in str(e.value)


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_create_validates_and_never_falls_back(nb):
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodySim(0)
    assert e.value.code == nb.ERR_INVALID
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodySim(100, shard=(90, 20))
    assert e.value.code == nb.ERR_INVALID
    if not _have_gpu():
        with pytest.raises(nb.NBodyError) as e:
            nb.NBodySim(1024)
        assert e.value.code == nb.ERR_NO_DEVICE
        assert "no CPU fallback" in str(e.value)


def test_host_benchmark_binary_fails_loudly_without_gpu():
    exe = os.path.join(ROOT, "n-body_amd", "bin", "nbody_benchmark")
    if not os.path.exists(exe):
        pytest.skip("host drivers not built")
    if _have_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([exe, "--n", "256", "--naive", "--steps", "1", "--no-log"], capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP device" in r.stderr


# ---------------------------------------------------------------------------------------------------
# config.json reader (host/config_reader.hpp) — semantics of main.cpp:348-398, exercised through a tiny
# C++ probe compiled on the fly with g++ (host-only code, no GPU).
# ---------------------------------------------------------------------------------------------------
PROBE = r'''
#include "config_reader.hpp"
#include <cstdio>
int main(int argc, char** argv) {
    nbody_host::SimConfig c = nbody_host::load_config(argc > 1 ? argv[1] : "config.json", false);
    std::printf("%d %d %d %d %g %g %g %g %d|%s\n", (int)c.loaded, (int)c.useBarnesHut, (int)c.useParallel, c.numBodies,
                c.spread, c.mass, c.theta, c.dt, c.windowSize, c.windowTitle.c_str());
    return 0;
}
'''


@pytest.fixture(scope="module")
def probe(tmp_path_factory):
    d = tmp_path_factory.mktemp("cfgprobe")
    src = d / "probe.cpp"
    src.write_text(PROBE)
    exe = d / "probe"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "n-body_amd", "host"), str(src), "-o", str(exe)],
                   check=True)
    return exe


def run_probe(probe, cwd, text, name="config.json"):
    if text is not None:
        (cwd / name).write_text(text)
    out = subprocess.run([str(probe), name], cwd=str(cwd), capture_output=True, text=True, check=True).stdout.strip()
    head, title = out.split("|", 1)
    v = head.split()
    return dict(loaded=int(v[0]), bh=int(v[1]), par=int(v[2]), n=int(v[3]), spread=float(v[4]), mass=float(v[5]),
                theta=float(v[6]), dt=float(v[7]), window=int(v[8]), title=title)


REFERENCE_CONFIG = '''{
  "simulation": {
    "useBarnesHut": true,
    "useParallel": true,
    "numBodies": 10000,
    "spread": 400.0,
    "mass": 1000.0,
    "theta": 1.0,
    "dt": 0.03
  },
  "display": {
    "windowSize": 1000,
    "windowTitle": "Chaotic N-Body Simulation"
  }
}
'''


def test_config_reader_reference_file(probe, tmp_path):
    c = run_probe(probe, tmp_path, REFERENCE_CONFIG)      # same key/values as the reference's config.json:1-15
    assert c == dict(loaded=1, bh=1, par=1, n=10000, spread=400.0, mass=1000.0, theta=1.0, dt=0.03, window=1000,
                     title="Chaotic N-Body Simulation")


def test_config_reader_defaults_when_missing(probe, tmp_path):
    c = run_probe(probe, tmp_path, None, name="absent.json")   # main.cpp:337-345
    assert (c["loaded"], c["bh"], c["par"], c["n"], c["dt"], c["window"]) == (0, 1, 1, 50000, 0.03, 1000)


def test_config_reader_flat_and_quirks(probe, tmp_path):
    # flat layout (README.md:113-125 style), naive mode, and the untrimmed-boolean quirk (main.cpp:384-385):
    # `true }` with a blank before the brace is NOT "true".
    c = run_probe(probe, tmp_path, '{ "useBarnesHut": false, "numBodies": 65536 , "dt": 0.05, "useParallel": true }')
    assert (c["bh"], c["n"], c["dt"], c["par"]) == (0, 65536, 0.05, 0)
    c = run_probe(probe, tmp_path, '{"useParallel": true}')
    assert c["par"] == 1


def test_config_reader_searches_parent_dirs(probe, tmp_path):
    (tmp_path / "config.json").write_text('{"numBodies": 4242}')
    deep = tmp_path / "build" / "bin"
    deep.mkdir(parents=True)
    assert run_probe(probe, deep, None)["n"] == 4242       # cwd/../.. (main.cpp:41)
    deeper = deep / "x"
    deeper.mkdir()
    assert run_probe(probe, deeper, None)["loaded"] == 0    # three levels up is not searched


def test_host_generator_matches_oracle(O, tmp_path):
    """host/initial_conditions.hpp (used by the C++ drivers) == oracle generator == reference generator."""
    src = tmp_path / "gen.cpp"
    src.write_text(r'''
#include "initial_conditions.hpp"
#include <cstdio>
int main() { nbody_host::BodyStore b; nbody_host::fill_disc(b, 1024, 400.0f, 1000.0f, 12345u, 0.3f);
  std::fwrite(b.posX.data(), 4, 1024, stdout); std::fwrite(b.posY.data(), 4, 1024, stdout);
  std::fwrite(b.velX.data(), 4, 1024, stdout); std::fwrite(b.velY.data(), 4, 1024, stdout); return 0; }
''')
    exe = tmp_path / "gen"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "n-body_amd", "host"), str(src), "-o", str(exe)], check=True)
    raw = subprocess.run([str(exe)], capture_output=True, check=True).stdout
    a = np.frombuffer(raw, np.float32).reshape(4, 1024)
    b = O.init_disc(1024, seed=12345)
    assert np.array_equal(a[0], b.posX) and np.array_equal(a[1], b.posY)
    assert np.allclose(a[2], b.velX, rtol=1e-6, atol=1e-7) and np.allclose(a[3], b.velY, rtol=1e-6, atol=1e-7)


def test_python_generator_equals_oracle_generator(O):
    """n-body_amd/initial_conditions.py (bench.py's synthetic inputs) against the oracle's restatement of
    benchmark.cpp:312-339, itself pinned to the compiled reference by tests/test_oracle.py: bit for bit."""
    import __graft_entry__ as entry
    entry.load_package()
    from nbody_amd.initial_conditions import disc
    for n, seed in ((1000, 12345), (4099, 7)):
        got = disc(n, seed=seed)
        want = O.init_disc(n, seed=seed)
        for a, f in zip(got, ("m", "posX", "posY", "velX", "velY", "radius")):
            assert a.dtype == np.float32 and np.array_equal(a, getattr(want, f)), f
    slow = disc(256, seed=3, vel_factor=0.05)             # main.cpp:431-432
    fast = disc(256, seed=3, vel_factor=0.3)
    assert np.allclose(slow[3] * 6.0, fast[3], rtol=1e-6)


def test_cpp_driver_bootstrap_hands_the_comm_id_to_every_rank(tmp_path):
    """nbody_benchmark --dist shares rank 0's 128-byte RCCL id over a TCP socket (MASTER_PORT + 1).  The hand-off
    itself needs no GPU: three ranks, started in the wrong order, must all print the checksum of rank 0's bytes."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    exe = os.path.join(ROOT, "n-body_amd", "bin", "nbody_benchmark")
    procs = []
    for rank in (2, 1, 0):                                   # rank 0 last: the others have to retry
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port - 1))
        procs.append(subprocess.Popen([exe, "--selftest-bootstrap"], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=60) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    sums = {re.search(r"id checksum (\d+)", o[0]).group(1) for o in outs}
    assert len(sums) == 1
    want = 0
    for k in range(128):
        want = (want * 31 + ((k * 7 + 3) & 0xFF)) & 0xFFFFFFFF
    assert sums == {str(want)}


def test_config_reader_equals_the_references_own_code_on_random_files(O, tmp_path):
    """host/config_reader.hpp against main.cpp's own findConfigFile + parameter block (compiled from /root/reference by
    oracle/build_ref.sh into oracle/_ref/libnbody_ref_config.so) on generated files: nested or flat objects, keys in
    any order or missing, blanks before commas/braces (which turn `true }` into false), quoted numbers, duplicate
    keys, keys hidden inside the title string, garbage numbers (the reference's stoi/stod then throws: both fail)."""
    if not O.ref_config_available():
        pytest.skip("oracle/_ref/libnbody_ref_config.so was not built (needs /root/reference at build time)")
    rng = np.random.default_rng(99)
    exe = os.path.join(ROOT, "n-body_amd", "bin", "nbody_sim")

    def token(kind):
        r = rng.random()
        if kind == "bool":
            return rng.choice(["true", "false", "true ", " false", "True", "1", '"true"'])
        if kind == "int":
            v = str(int(rng.integers(1, 200000)))
            return rng.choice([v, v + " ", '"' + v + '"', v + ".7", "  " + v, "x" + v]) if r < 0.9 else "none"
        v = "%g" % rng.uniform(0.001, 5000)
        return rng.choice([v, v + "  ", '"' + v + '"', v + "e1", "-" + v, v + "f"]) if r < 0.92 else "abc"

    keys = [("useBarnesHut", "bool"), ("useParallel", "bool"), ("numBodies", "int"), ("spread", "float"), ("mass", "float"),
            ("theta", "float"), ("dt", "float"), ("windowSize", "int")]
    cwd = os.getcwd()
    agreed = failed_both = 0
    try:
        for trial in range(120):
            d = tmp_path / f"t{trial}"
            d.mkdir()
            items = [(k, token(t)) for k, t in keys if rng.random() < 0.85]
            if rng.random() < 0.6:
                items.append(("windowTitle", rng.choice(['"plain"', '"has \\"numBodies\\": 5 inside"', '"a, b } c"', '""'])))
            if rng.random() < 0.3 and items:
                items.append(items[int(rng.integers(0, len(items)))])            # duplicate key: the first one wins
            order = rng.permutation(len(items))
            sep = rng.choice([", ", ",\n  ", " ,", ","])
            body = sep.join('"%s"%s:%s%s' % (items[i][0], rng.choice(["", " "]), rng.choice(["", " ", "\n"]), items[i][1]) for i in order)
            text = "{ %s%s }" % (rng.choice(["", '"simulation": { ']), body) + rng.choice(["", " }", "\n"])
            (d / "config.json").write_text(text)
            os.chdir(d)
            want = O.ref_parse_config("config.json")
            os.chdir(cwd)
            r = subprocess.run([exe, "--print-config"], cwd=str(d), capture_output=True, text=True)
            if want is None:
                assert r.returncode != 0, (text, r.stdout)                          # both choke on the same file
                failed_both += 1
                continue
            assert r.returncode == 0, (text, r.stderr)
            got = dict(ln.split("=", 1) for ln in r.stdout.splitlines() if "=" in ln and not ln.startswith(("Config", "Search", "Found")))
            for k in ("useBarnesHut", "useParallel", "numBodies", "windowSize"):
                assert int(got[k]) == want[k], (k, text, got, want)
            for k in ("spread", "mass", "theta", "dt"):
                assert np.float32(float(got[k])) == np.float32(want[k]), (k, text, got, want)
            assert got["windowTitle"] == want["windowTitle"], (text, got, want)
            agreed += 1
    finally:
        os.chdir(cwd)
    assert agreed >= 60 and failed_both >= 3


def test_bench_fails_loudly_without_a_gpu():
    """bench.py and smoke() are product-path callers: on a machine without an MI355X they must stop with a clear message,
    not fall back to anything that runs on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--bodies", "1024"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())           # no JSON line without a measurement
    import __graft_entry__ as entry
    with pytest.raises(Exception, match="no HIP device|no CPU fallback"):
        entry.smoke()


def test_bench_gpus_n_launches_its_own_ranks_and_fails_loudly_without_a_gpu(tmp_path):
    """`python bench.py --gpus 2` from a plain interpreter (no torch.distributed.run around it): bench.py starts the ranks
    itself.  Here there is no GPU, so every rank stops with the no-fallback message, and the
    launcher must hand that failure on as a non-zero exit code without printing a result line."""
    env = dict(os.environ, OMP_NUM_THREADS="1", NBODY_LAUNCH_TIMEOUT="150")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is visible: tests/test_gpu_multirank.py covers the launch that succeeds")
    except ImportError:
        pytest.skip("torch not importable")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--bodies", "4096"],
                       env=env, capture_output=True, text=True, timeout=170, cwd=str(tmp_path))
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
