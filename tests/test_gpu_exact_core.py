"""n-body_amd/csrc/exact_core.hip.h: the Barnes-Hut node kernel's sqrt and its two divisions as the FMA cores of hipcc's
own IEEE expansions, without the operand-scaling and special-case instructions around them (valid for operands in a
stated range, which the host and the kernels check).  The prebuilt checker compares them with `sqrtf`, `1.0f / b` and
`a / b` as hipcc compiles them, on the GPU: every float of the sqrt range [2^-96, 2^80] and of the reciprocal range
[2^-21, 2^41], the walk's seeded reciprocal 1 / (sqrt(d^2) + 1e-6) for every d^2 of the sqrt range, 2^34 random operand pairs and 2^30 mantissa-corner pairs of the division's ranges.  Bit-exact or fail.
(The walk kernels themselves are compared with the oracle, which uses the host's IEEE operations, by test_gpu_bh*.py,
and with their own hipcc-expansion variant by tools/bh_paths_soak.py.)"""
import os
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exact_cores_equal_hipccs_ieee_expansions():
    exe = os.path.join(ROOT, "n-body_amd", "bin", "exact_core_check")
    assert os.path.exists(exe), "run `make host` (builds n-body_amd/bin/exact_core_check)"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=150)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if " differ " in l]
    assert len(lines) == 6 and all(", 0 differ" in l or ": 0 differ" in l for l in lines), r.stdout
