"""Position streaming for a viewer: nbody_frame_publish / nbody_frame_acquire through the C-ABI.
The consumer in the reference is the draw loop main.cpp:646-655, which reads posX/posY once per frame."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bh", [False, True])
def test_frames_show_the_state_after_each_step(nb, O, bh):
    n, nframes = 20000, 6
    b = O.init_disc(n, seed=21)
    # expected: synchronous downloads after every step
    want = []
    with nb.NBodySim(n, dt=0.05, useBarnesHut=bh) as sim:
        sim.upload(*b.arrays())
        for _ in range(nframes):
            sim.step(1)
            st = sim.download()
            want.append((st["posX"].copy(), st["posY"].copy()))
    # pipelined: frame k is fetched while step k+1 is already enqueued
    got = []
    with nb.NBodySim(n, dt=0.05, useBarnesHut=bh) as sim:
        sim.upload(*b.arrays())
        sim.step(1).frame_publish()
        for k in range(1, nframes):
            sim.step(1).frame_publish()              # frame k in flight together with frame k-1
            x, y, step = sim.frame_acquire()         # oldest first: frame k-1
            assert step == k
            got.append((x.copy(), y.copy()))
        x, y, step = sim.frame_acquire()
        assert step == nframes
        got.append((x.copy(), y.copy()))
        with pytest.raises(nb.NBodyError):
            sim.frame_acquire()                      # nothing outstanding
    for k in range(nframes):
        assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), k


def test_at_most_two_frames_in_flight_and_upload_drops_them(nb, O):
    n = 4096
    b = O.init_disc(n, seed=3)
    with nb.NBodySim(n) as sim:
        with pytest.raises(nb.NBodyError):
            sim.frame_publish()                      # nothing uploaded
        sim.upload(*b.arrays())
        sim.frame_publish().frame_publish()
        with pytest.raises(nb.NBodyError, match="in flight"):
            sim.frame_publish()
        x0, y0, s0 = sim.frame_acquire()
        assert s0 == 0 and np.array_equal(x0, b.posX) and np.array_equal(y0, b.posY)
        sim.frame_publish()                          # a slot is free again
        sim.upload(*b.arrays())                      # new state: outstanding frames are dropped
        with pytest.raises(nb.NBodyError):
            sim.frame_acquire()
        sim.step(2).frame_publish()
        x, y, s = sim.frame_acquire()
        assert s == 2 and np.array_equal(x, sim.download()["posX"])
