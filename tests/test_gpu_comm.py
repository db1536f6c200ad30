"""Native RCCL exchange of the C-ABI (nbody_comm_* / nbody_exchange_positions), as far as ONE GPU
allows: a 1-rank communicator exercises the run-time binding of RCCL, communicator set-up on the context's device and
the in-place all-gather on the context's stream.  Two ranks cannot share a device under RCCL, so the N > 1 exchange
itself is only rehearsed through torch.distributed/gloo (tests/test_gpu_multirank.py) and runs for real on the
driver's 8-GPU node."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("bh", [False, True])
def test_one_rank_communicator_steps_like_a_plain_context(nb, O, bh):
    n = 8192
    b = O.init_disc(n, seed=11)
    with nb.NBodySim(n, useBarnesHut=bh) as ref:
        ref.upload(*b.arrays())
        ref.step(3)
        want = ref.download()
    uid = nb.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    with nb.NBodySim(n, useBarnesHut=bh, shard=nb.shard_range(n, 1, 0)) as sim:
        with pytest.raises(nb.NBodyError, match="nbody_comm_init"):
            sim.exchange_positions()
        sim.comm_init(0, 1, uid)
        with pytest.raises(nb.NBodyError, match="already"):
            sim.comm_init(0, 1, uid)
        sim.upload(*b.arrays())
        for _ in range(3):
            sim.step(1).exchange_positions()
        got = sim.download()
    for k in ("posX", "posY", "velX", "velY"):
        assert np.array_equal(got[k], want[k]), k


def test_sharded_driver_native_exchange_one_rank(nb, O):
    """ShardedSim(exchange="native") — what `NBODY_EXCHANGE=native bench.py` runs: the RCCL id travels through
    torch.distributed, the all-gather is the library's own ncclAllGather on the shared stream."""
    import socket
    import torch
    import torch.distributed as dist
    from nbody_amd.sharded import ShardedSim
    n = 8192
    b = O.init_disc(n, seed=6)
    with nb.NBodySim(n) as ref:
        ref.upload(*b.arrays()); ref.step(2)
        want = ref.download()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sim = ShardedSim(nb, n, 0, 1, device=0, dist=dist, exchange="native")
        assert sim.exchange_mode == "native"
        sim.upload(*b.arrays())
        sim.sim.comm_init(0, 1, nb.comm_unique_id())               # world = 1: ShardedSim skips it, do it here
        sim.step(2)
        sim.exchange(force=True)
        sim.sync()
        got = sim.sim.download()
        sim.close()
    finally:
        dist.destroy_process_group()
    for k in want:
        assert np.array_equal(got[k], want[k]), k


def test_comm_init_checks_the_shard_plan(nb, O):
    n = 4096
    uid = nb.comm_unique_id()
    with nb.NBodySim(n, shard=(0, 1024)) as sim:                 # not rank 1 of 2's share
        with pytest.raises(nb.NBodyError, match="nbody_shard_range"):
            sim.comm_init(1, 2, uid)
        with pytest.raises(nb.NBodyError):
            sim.comm_init(2, 2, uid)


def test_cpp_driver_dist_mode_single_rank(tmp_path):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    exe = os.path.join(ROOT, "n-body_amd", "bin", "nbody_benchmark")
    r = subprocess.run([exe, "--dist", "--n", "65536", "--naive", "--steps", "5", "--warmup", "1", "--no-log"], env=env,
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    assert re.search(r"SUMMARY \| N=65536 \| Mode: N\^2 \| steps=5 \| [0-9.]+ ms/step .* \| ranks=1", r.stdout), r.stdout
