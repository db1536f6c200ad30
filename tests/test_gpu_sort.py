"""The Barnes-Hut build's own device sort (csrc/bh_sort.hip.h) through its parity hook nbody_sort_pairs_u64: a stable LSD
radix sort of (64-bit key, int) pairs whose digit positions with a one-bin histogram are skipped on the device.
Checked against numpy's stable sort, bit for bit, on key sets chosen for the skip logic (every number of live digit
positions, odd and even, none at all), for the tile edges (sizes around multiples of the 2048- and 4096-pair tiles,
one pair) and for the look-back (hundreds of tiles; several sorts in a row on one state)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def check(nb, keys, repeats=1, truncated=(False, True)):
    keys = np.ascontiguousarray(keys, np.uint64)
    vals = np.arange(len(keys), dtype=np.int32)
    order = np.argsort(keys, kind="stable")
    for tr in truncated:                       # the full 8-sweep sort and the build's form: 5 sweeps + run kernels
        ok, ov = nb.sort_pairs_u64(keys, vals, repeats, truncated=tr)
        assert np.array_equal(ov, order.astype(np.int32)), f"n={len(keys)} truncated={tr}: first difference at {np.nonzero(ov != order)[0][:4]}"
        assert np.array_equal(ok, keys[order])


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2047, 2048, 2049, 4096, 12289, 100_003, 1 << 20, (1 << 21) + 1, 5_000_017])
def test_random_64_bit_keys(nb, n):
    rng = np.random.default_rng(n)
    check(nb, rng.integers(0, 2 ** 64, size=n, dtype=np.uint64))


@pytest.mark.parametrize("live", [0, 1, 2, 3, 5, 7])
def test_only_some_digit_positions_vary(nb, live):
    """`live` of the 8 byte positions vary, the others hold one value: 0 = all keys equal (output = input)."""
    rng = np.random.default_rng(100 + live)
    n = 300_001
    pos = rng.choice(8, size=live, replace=False)
    keys = np.full(n, 0x1122334455667788, np.uint64)
    for p in pos:
        keys = (keys & ~np.uint64(0xFF << (8 * int(p)))) | (rng.integers(0, 256, size=n, dtype=np.uint64) << np.uint64(8 * int(p)))
    check(nb, keys, repeats=2)


def test_few_distinct_keys_and_presorted_input(nb):
    rng = np.random.default_rng(7)
    n = 777_777
    check(nb, rng.integers(0, 3, size=n, dtype=np.uint64) << np.uint64(40))        # three values: long runs, stability
    s = np.sort(rng.integers(0, 2 ** 62, size=n, dtype=np.uint64))
    check(nb, s)
    check(nb, s[::-1].copy())
    check(nb, np.full(n, 2 ** 64 - 1, np.uint64))                                   # the padding value of the tie buffer


def test_path_keys_of_the_disc_many_sorts_on_one_state(nb, O):
    """What the build sorts: quadrant-path keys of the seeded disc with the out-of-box keys (bit 63 + index) behind."""
    n = 1_000_000
    b = O.init_disc(n, seed=12345)
    b.posX[::1000] = 500.0                                                           # some bodies outside the root box
    keys = np.zeros(n, np.uint64)
    O.lib().oracle_path_keys(n, 400.0, b.posX, b.posY, keys)
    check(nb, keys, repeats=5)


@pytest.mark.parametrize("n,prefixes", [(200_000, 150_000), (200_000, 3000), (300_000, 7), (70_000, 1), (1_500_000, 40)])
def test_runs_of_equal_prefixes(nb, n, prefixes):
    """The truncated form sweeps bits 24-63 only; pairs that agree there are finished by the run kernels: short runs
    (many prefixes: runs of 1-5), runs around the 32-pair limit of the insertion sort, and long runs that one wave sorts
    (a few prefixes; one prefix = the whole input is one run)."""
    rng = np.random.default_rng(n + prefixes)
    pre = rng.integers(0, 2 ** 38, size=prefixes, dtype=np.uint64)
    keys = (pre[rng.integers(0, prefixes, size=n)] << np.uint64(24)) | rng.integers(0, 2 ** 24, size=n, dtype=np.uint64)
    keys[::5] = keys[1::5][:len(keys[::5])]                       # and equal FULL keys: stability inside the runs
    check(nb, keys, repeats=2)


def test_run_lengths_around_the_insertion_limit(nb):
    rng = np.random.default_rng(33)
    parts = []
    for length in list(range(1, 40)) * 20:
        parts.append((np.uint64(len(parts) + 1) << np.uint64(24)) | rng.integers(0, 2 ** 24, size=length, dtype=np.uint64))
    keys = np.concatenate(parts)
    check(nb, keys[rng.permutation(len(keys))])
