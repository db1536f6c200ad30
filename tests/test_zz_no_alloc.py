"""GPU test of nbody_accel_noalloc (include/nbody.h): the force phase for callers that cannot allocate.

Through the C-ABI: the caller's inputs and outputs are torch tensors on cuda:0 (the test may allocate; the entry point
may not), results are compared with the existing entry point — nbody_compute_accel + nbody_get_accel of a context with
ref_compat = 1, force_variant = 4 — bit for bit, and with the f64 oracle (<= 1e-4), on two calls in a row, the second
with a smaller input: the static scratch array keeps the first call's contents, none of which may show.
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def _noalloc(nb, torch, b, stream=None, first_call=False):
    n = b.n
    m, px, py, r = (_dev(torch, x) for x in (b.m, b.posX, b.posY, b.radius))
    ax = torch.full((n,), float("nan"), device="cuda"); ay = torch.full((n,), float("nan"), device="cuda")
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    nb.accel_noalloc(n, m.data_ptr(), px.data_ptr(), py.data_ptr(), r.data_ptr(), ax.data_ptr(), ay.data_ptr(),
                     stream.cuda_stream if stream is not None else None)
    if stream is not None:
        stream.synchronize()
    torch.cuda.synchronize()
    # device memory in use must not move (the very first launch from the library loads its code object, which holds
    # the static array, so the first call of a process is exempt)
    assert first_call or torch.cuda.mem_get_info()[0] == free_before
    return ax.cpu().numpy(), ay.cpu().numpy()


def _context(nb, b):
    with nb.NBodySim(b.n, useBarnesHut=False, ref_compat=1, force_variant=4) as sim:
        sim.upload(*b.arrays())
        sim.compute_accel()
        return sim.accel()


def test_two_calls_in_a_row_second_smaller(nb, O):
    import torch
    torch.cuda.set_device(0)
    rng = np.random.default_rng(3)
    first = O.init_disc(65536, seed=11)                       # the stated maximum
    first.m = (rng.random(first.n) * 2000.0 + 1.0).astype(np.float32)
    first.radius = rng.uniform(0.5, 30.0, first.n).astype(np.float32)
    second = O.init_disc(10003, seed=12)                      # smaller, ragged (N % 8 = 3), other launch plan
    for b in (first, second):
        ax, ay = _noalloc(nb, torch, b, first_call=b is first)
        cx, cy = _context(nb, b)
        assert np.array_equal(ax, cx) and np.array_equal(ay, cy)
        lim = b.n - b.n % 8
        assert not ax[lim:].any() and not ay[lim:].any()
        idx = np.linspace(0, lim - 1, 512).astype(np.int32)
        rx, ry = O.naive_accel_f64_subset(b, idx)
        assert rel_err(ax[idx], ay[idx], rx, ry).max() <= 1e-4
    # and once more on a caller's stream, tiny input
    third = O.init_disc(9, seed=13)
    ax, ay = _noalloc(nb, torch, third, stream=torch.cuda.Stream())
    cx, cy = _context(nb, third)
    assert np.array_equal(ax, cx) and np.array_equal(ay, cy)
    # fewer than 8 bodies: nobody is a target (benchmark.cpp:423), the outputs are zeros
    ax, ay = _noalloc(nb, torch, O.init_disc(5, seed=14))
    assert not ax.any() and not ay.any()


def test_rejects_sizes_beyond_the_static_array(nb):
    with pytest.raises(nb.NBodyError) as e:
        nb.accel_noalloc(65537, 8, 8, 8, 8, 8, 8)            # fails on the size check before any pointer is touched
    assert e.value.code == nb.ERR_INVALID and "65536" in str(e.value)
    with pytest.raises(nb.NBodyError):
        nb.accel_noalloc(1024, 0, 8, 8, 8, 8, 8)


def test_captured_into_a_graph_and_replayed(nb, O):
    """The caller the entry point is for: a stream capture.  After one warm-up call outside the capture (it resolves the
    static array's address on this GPU), the call is captured into a graph — nothing in it allocates, synchronises or
    is otherwise illegal under capture — and the graph is replayed on new inputs written into the same buffers."""
    import torch
    torch.cuda.set_device(0)
    a, b2 = O.init_disc(4099, seed=21), O.init_disc(4099, seed=22)
    n = a.n
    m, px, py, r = (_dev(torch, x) for x in (a.m, a.posX, a.posY, a.radius))
    ax = torch.zeros(n, device="cuda"); ay = torch.zeros(n, device="cuda")
    s = torch.cuda.Stream()
    nb.accel_noalloc(n, m.data_ptr(), px.data_ptr(), py.data_ptr(), r.data_ptr(), ax.data_ptr(), ay.data_ptr(), s.cuda_stream)
    s.synchronize()                                          # warm-up, outside the capture
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        nb.accel_noalloc(n, m.data_ptr(), px.data_ptr(), py.data_ptr(), r.data_ptr(), ax.data_ptr(), ay.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    for bodies in (a, b2, a):
        px.copy_(_dev(torch, bodies.posX)); py.copy_(_dev(torch, bodies.posY))
        ax.fill_(float("nan")); ay.fill_(float("nan"))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        cx, cy = _context(nb, bodies)
        assert np.array_equal(ax.cpu().numpy(), cx) and np.array_equal(ay.cpu().numpy(), cy)
