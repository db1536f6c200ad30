"""Worker for tests/test_gpu_multirank.py: one rank of a sharded run (launched by torch.distributed.run).
Backend comes from NBODY_DIST_BACKEND (gloo lets several ranks share the one GPU of the test box)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import faulthandler
    # a stalled rank reports where it stands and ends (NBODY_DEBUG_TIMEOUT: seconds, as in bench.py; a busy box may need more)
    faulthandler.dump_traceback_later(int(os.environ.get("NBODY_DEBUG_TIMEOUT", "60")), exit=True)
    n, nsteps, bh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out = sys.argv[4]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("NBODY_DIST_BACKEND", "nccl")
    dev = int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist.init_process_group(backend)
    pkg = entry.load_package()
    O = entry.load_oracle()
    from nbody_amd.sharded import ShardedSim
    b = O.init_disc(n, seed=77)
    sim = ShardedSim(pkg, n, rank, world, device=dev, dt=0.05, dist=dist, useBarnesHut=bool(bh))
    sim.upload(*b.arrays())
    sim.step(nsteps)
    sim.sync()
    st = sim.sim.download()
    beg, cnt = sim.spans[rank]
    # positions are complete on every rank after the exchange; velocities are rank-local
    np.savez(os.path.join(out, f"rank{rank}.npz"), posX=st["posX"], posY=st["posY"], velX=st["velX"][beg:beg + cnt],
             velY=st["velY"][beg:beg + cnt], beg=beg, cnt=cnt)
    sim.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
