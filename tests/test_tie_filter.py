"""CPU check of the argument behind bh_min_tie_level (n-body_amd/csrc/bh_tree.hip.h): the lowest level at which a body
can sit on a split line, read off its path key.  The descent of bh_keys_kernel (benchmark.cpp:98-99, 105-108 in float32)
is replayed in numpy with the exact first tie level per axis; the key-based bound must never lie ABOVE an exact tie at a
level <= 22 — it is a filter in front of the exact record, so a bound that is too low only costs time, one that is too
high loses a second instance (tools/bh_stress.py found exactly that in round 3: a coordinate on the box's upper edge
has 23 one-bits and no candidate of its own, and the two axes must be combined by min, not by the longer run)."""
import numpy as np
import pytest

f32 = np.float32
LEVELS = 31


def descend(x, y, spread):
    n = len(x)
    tlx = np.full(n, -spread, f32); tly = tlx.copy()
    brx = np.full(n, spread, f32); bry = brx.copy()
    key = np.zeros(n, np.uint64)
    tx = np.full(n, LEVELS); ty = np.full(n, LEVELS)
    for l in range(LEVELS):
        mx = ((tlx + brx) / f32(2)).astype(f32); my = ((tly + bry) / f32(2)).astype(f32)
        tx = np.where((x == mx) & (tx == LEVELS), l, tx); ty = np.where((y == my) & (ty == LEVELS), l, ty)
        qx = x > mx; qy = y > my                                   # on the line -> lower child
        key = (key << np.uint64(2)) | (qy.astype(np.uint64) * np.uint64(2) + qx.astype(np.uint64))
        tlx = np.where(qx, mx, tlx); brx = np.where(qx, brx, mx)
        tly = np.where(qy, my, tly); bry = np.where(qy, bry, my)
    return key, tx, ty


def min_tie_level(k):
    """bh_min_tie_level restated: per axis the 0 above the trailing 1s of levels 0..22, 23 when there is none."""
    odd = 0xAAAAAAAAAAAAAAAA
    def run(z):
        v = (~(z | odd)) & 0xFFFFFFFFFFFFFFFF
        return ((v & -v).bit_length() - 1) >> 1
    rx, ry = run(k >> 16), run(k >> 17)
    return min(23 if rx >= 23 else 22 - rx, 23 if ry >= 23 else 22 - ry)


def inputs(kind, rng, n, spread):
    s = f32(spread)
    if kind == "lattice":                                           # every point on split lines, half of them moved off in x
        k = int(rng.integers(3, 14))
        side = (1 << (k + 1)) + 1
        cells = rng.choice(side * side, size=min(n, side * side), replace=False)
        step = spread / (1 << k)
        x = ((cells % side - (1 << k)) * step).astype(f32); y = ((cells // side - (1 << k)) * step).astype(f32)
        off = rng.random(len(x)) < 0.5
        x = np.where(off, x + rng.uniform(0.01, 0.4, len(x)).astype(f32) * f32(step), x).astype(f32)
        return x, y
    if kind == "edges":                                             # the box's own edges and centre lines, one ulp either side
        base = np.array([-s, s, 0.0, s / 2, -s / 2, s / 4, -3 * s / 4], f32)
        v = np.concatenate([base, np.nextafter(base, f32(0)), np.nextafter(base, f32(2) * s * np.sign(base + f32(1e-30)))]).astype(f32)
        v = v[np.abs(v) <= s]
        x = rng.choice(v, n).astype(f32); y = rng.choice(v, n).astype(f32)
        return x, y
    if kind == "dyadic":                                            # few mantissa bits: many exact midpoints at every depth
        m = rng.integers(-(1 << 20), (1 << 20) + 1, (2, n))
        sh = rng.integers(0, 21, (2, n))
        v = (m >> sh << sh).astype(np.float64) / (1 << 20) * spread
        return v[0].astype(f32), v[1].astype(f32)
    if kind == "tiny":                                              # close to the centre: ulps far below the cell width
        return (rng.normal(0, 1e-6 * spread, n)).astype(f32), (rng.normal(0, 1e-3 * spread, n)).astype(f32)
    return rng.uniform(-spread, spread, n).astype(f32), rng.uniform(-spread, spread, n).astype(f32)


@pytest.mark.parametrize("spread", [400.0, 123.456, 1.0, 3.0e6, 7.3e-9])
@pytest.mark.parametrize("kind", ["lattice", "edges", "dyadic", "tiny", "uniform"])
def test_key_bound_never_exceeds_an_exact_tie_level(kind, spread):
    import zlib
    rng = np.random.default_rng(zlib.crc32(repr((kind, spread)).encode()))
    for _ in range(3):
        x, y = inputs(kind, rng, 6000, spread)
        inside = (np.abs(x) <= f32(spread)) & (np.abs(y) <= f32(spread))
        x, y = x[inside], y[inside]
        key, tx, ty = descend(x, y, f32(spread))
        first = np.minimum(tx, ty)
        bound = np.array([min_tie_level(int(k)) for k in key])
        shallow = first <= 22
        assert not np.any(bound[shallow] > first[shallow]), (kind, spread, np.flatnonzero(shallow & (bound > first))[:5])
        assert np.all(bound <= 23)                                  # levels 23..30 stay possible for everybody
    if kind in ("lattice", "edges", "dyadic"):
        assert np.any(first <= 22)                                  # the case is not vacuous
