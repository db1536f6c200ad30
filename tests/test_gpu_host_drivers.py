"""GPU tests of the C++ host drivers (counterparts of the reference's benchmark.cpp / main.cpp loops): they bind
include/nbody.h directly, so this is the drop-in boundary exercised from the reference's own language."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "n-body_amd", "bin")


def run(cmd, cwd):
    return subprocess.run(cmd, cwd=str(cwd), capture_output=True, text=True, timeout=240)


def test_benchmark_driver_naive(tmp_path):
    r = run([os.path.join(BIN, "nbody_benchmark"), "--n", "65536", "--naive", "--steps", "20", "--warmup", "2"], tmp_path)
    assert r.returncode == 0, r.stderr
    m = re.search(r"SUMMARY \| N=65536 \| Mode: N\^2 \| steps=20 \| ([0-9.]+) ms/step \| ([0-9.]+) G interactions/s", r.stdout)
    assert m, r.stdout
    assert float(m.group(2)) > 1000.0                      # three orders of magnitude above one CPU core
    log = (tmp_path / "log.txt").read_text()               # benchmark.cpp:314,548-552: appended to ./log.txt
    assert "SUMMARY" in log and "=== Benchmark ended:" in log


def test_benchmark_driver_log_line_format(tmp_path):
    """>= 1 s of frames produces the reference's line `FPS: x | Physics: y ms | Total frame: z ms` (benchmark.cpp:536-540)."""
    r = run([os.path.join(BIN, "nbody_benchmark"), "--n", "262144", "--naive", "--steps", "120", "--warmup", "1", "--no-log"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert re.search(r"^FPS: [0-9.]+ \| Physics: [0-9.]+ ms \| Total frame: [0-9.]+ ms$", r.stdout, re.M), r.stdout
    assert not (tmp_path / "log.txt").exists()


def test_benchmark_driver_barnes_hut_default(tmp_path):
    """No flags = the reference's hard-coded block (benchmark.cpp:300-306): BH, n=10000, theta=1, dt=0.05."""
    r = run([os.path.join(BIN, "nbody_benchmark"), "--steps", "10", "--no-log"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert re.search(r"SUMMARY \| N=10000 \| Mode: BarnesHut theta=1.00 \| steps=10", r.stdout), r.stdout


def test_sim_driver_reads_config_and_dumps_frames(tmp_path):
    (tmp_path / "config.json").write_text(
        '{ "simulation": { "useBarnesHut": false, "useParallel": true, "numBodies": 4096, "spread": 400.0, "mass": 1000.0,'
        ' "theta": 1.0, "dt": 0.03 }, "display": { "windowSize": 256, "windowTitle": "t" } }')
    frames = tmp_path / "frames"
    frames.mkdir()
    r = run([os.path.join(BIN, "nbody_sim"), "--frames", "3", "--dump-pgm", str(frames), "--no-log"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Config loaded: useBarnesHut=0, useParallel=1, numBodies=4096" in r.stdout
    assert re.search(r"FPS: [0-9.]+ \| Physics: [0-9.]+ ms \| Total frame: [0-9.]+ ms \| N=4096 \| Mode: N\^2", r.stdout)  # main.cpp:679
    pgm = sorted(frames.glob("frame_*.pgm"))
    assert len(pgm) == 3
    raw = pgm[-1].read_bytes()
    assert raw.startswith(b"P5\n256 256\n255\n")
    img = np.frombuffer(raw[len(b"P5\n256 256\n255\n"):], np.uint8)
    assert img.size == 256 * 256 and (img > 0).sum() > 500   # bodies were rasterised from the downloaded posX/posY


def test_sim_driver_async_frames_equal_synchronous_frames(tmp_path):
    """--async-frames (nbody_frame_publish/acquire: frame f is drawn while step f+1 runs) must draw exactly the
    frames the synchronous download path draws."""
    (tmp_path / "config.json").write_text(
        '{ "simulation": { "useBarnesHut": true, "useParallel": true, "numBodies": 8192, "spread": 400.0, "mass": 1000.0,'
        ' "theta": 1.0, "dt": 0.05 }, "display": { "windowSize": 200, "windowTitle": "t" } }')
    out = {}
    for mode in ("sync", "async"):
        d = tmp_path / mode
        d.mkdir()
        cmd = [os.path.join(BIN, "nbody_sim"), "--frames", "5", "--dump-pgm", str(d), "--no-log"]
        r = run(cmd + (["--async-frames"] if mode == "async" else []), tmp_path)
        assert r.returncode == 0, r.stderr
        assert "Mode: BarnesHut" in r.stdout
        out[mode] = [p.read_bytes() for p in sorted(d.glob("frame_*.pgm"))]
    assert len(out["sync"]) == len(out["async"]) == 5
    assert out["sync"] == out["async"]
    assert out["sync"][0] != out["sync"][-1]
