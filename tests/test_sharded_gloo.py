"""CPU test of the N>1 path: world_size-2 (and 3, ragged) `gloo` process groups run the SAME
shard plan (nbody_shard_range from the C-ABI library) and the SAME exchange code (nbody_amd.sharded.exchange_slices)
that bench.py uses on GPUs; the per-rank force/integrate is played by the CPU oracle.  Result must equal the
single-process oracle bit for bit (every rank evaluates its targets against identical gathered positions)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, n, nsteps, port, out_dir):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    pkg = entry.load_package()
    O = entry.load_oracle()
    from nbody_amd.sharded import exchange_slices, owned_targets
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b = O.init_disc(n, seed=4242)
        dt = np.float32(0.05)
        spans = [pkg.shard_range(n, world, r) for r in range(world)]
        beg, cnt = spans[rank]
        adv = owned_targets(n, spans, rank)             # ref_compat: the N%8 tail is frozen
        pos = [torch.from_numpy(b.posX), torch.from_numpy(b.posY)]   # share memory with the Bodies arrays
        for _ in range(nsteps):
            ax, ay = O.naive_accel_f64_range(b, beg, beg + adv)
            sl = slice(beg, beg + adv)
            newx = b.posX[sl] + b.velX[sl] * dt
            newy = b.posY[sl] + b.velY[sl] * dt
            b.velX[sl] = b.velX[sl] + ax.astype(np.float32) * dt
            b.velY[sl] = b.velY[sl] + ay.astype(np.float32) * dt
            b.posX[sl] = newx
            b.posY[sl] = newy
            exchange_slices(pos, spans, rank, dist)
        # velocities are rank-local: gather them only for the check
        vel = [torch.from_numpy(b.velX), torch.from_numpy(b.velY)]
        exchange_slices(vel, spans, rank, dist)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), posX=b.posX, posY=b.posY, velX=b.velX, velY=b.velY)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 2048), (2, 1003), (3, 1000)])
def test_sharded_steps_equal_single_process(tmp_path, world, n):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as entry
    O = entry.load_oracle()
    nsteps = 3
    port = 29500 + (os.getpid() + world * 7 + n) % 2000
    mp.spawn(_worker, args=(world, n, nsteps, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "out.npz")
    ref = O.init_disc(n, seed=4242)
    # single process, same arithmetic (f64 force rounded to f32, mul-then-add Euler in numpy f32)
    dt = np.float32(0.05)
    lim = n - n % 8
    for _ in range(nsteps):
        ax, ay = O.naive_accel_f64_range(ref, 0, lim)
        sl = slice(0, lim)
        newx = ref.posX[sl] + ref.velX[sl] * dt
        newy = ref.posY[sl] + ref.velY[sl] * dt
        ref.velX[sl] = ref.velX[sl] + ax.astype(np.float32) * dt
        ref.velY[sl] = ref.velY[sl] + ay.astype(np.float32) * dt
        ref.posX[sl] = newx
        ref.posY[sl] = newy
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(got[k], getattr(ref, k)), k
