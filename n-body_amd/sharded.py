"""Multi-GPU step driver: one process per GPU, i-bodies sharded, positions all-gathered each step.

SURVEY.md §8e (no reference counterpart — the reference is a single OpenMP process).  Each rank owns the
contiguous target range ``nbody_shard_range(N, world, rank)``; every rank holds ALL positions (they are the
sources of the pair loop) and after each local step the owned slices of posX/posY are exchanged with ONE
collective per array over ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU
tests).  Velocities and accelerations stay rank-local; m and radius are constant and uploaded everywhere.
Message size per GPU per step: 8 B x N/P sent, 8 B x N x (P-1)/P received (7 MiB at N=2^20, P=8).

torch is plumbing here (process group, stream ordering); the physics runs in the C-ABI library.
"""
import numpy as np


def exchange_slices(tensors, spans, rank, dist, group=None, staging=None):
    """All-gather: every rank contributes tensors[k][begin:begin+count] of ITS span; afterwards every rank holds
    every span.

    Equal spans -> one all_gather_into_tensor per array into a staging buffer of the same shape, copied back
    (the send buffer is a slice of the array itself, and an aliased in-place gather is not something every
    backend promises); ragged spans (N not divisible) -> one broadcast per rank and array."""
    counts = {c for _, c in spans}
    if len(counts) == 1 and spans[0][1] * len(spans) == tensors[0].numel():
        b, c = spans[rank]
        for k, t in enumerate(tensors):
            stage = staging[k] if staging is not None else t.new_empty(t.shape)
            dist.all_gather_into_tensor(stage, t[b:b + c].contiguous(), group=group)
            t.copy_(stage)
    else:
        for src, (b, c) in enumerate(spans):
            if c == 0:
                continue
            for t in tensors:
                dist.broadcast(t[b:b + c], src=src, group=group)


class ShardedSim:
    """One rank's share of a sharded simulation (naive or Barnes-Hut) on its GPU.  In Barnes-Hut mode every rank
    rebuilds the (cheap) tree from the gathered positions and walks it for its own targets only.

    exchange = "torch" (default): torch.distributed collectives — RCCL ("nccl") on the shared stream, or, for a
    host-staged backend ("gloo": the CPU tests and the one-GPU rehearsal), this class's own staging through pinned host
    memory with the collective on CPU tensors.  exchange = "native" (or NBODY_EXCHANGE=native): the library's own
    in-place ncclAllGather below the C-ABI (nbody_comm_init / nbody_exchange_positions); torch.distributed then only
    carries the 128-byte RCCL id to the ranks."""

    def __init__(self, pkg, num_bodies, rank, world, *, device, dt=0.05, ref_compat=True, dist=None,
                 useBarnesHut=False, exchange=None, **kw):
        import os
        import torch
        if not pkg.torch_interop_ok():
            raise RuntimeError("n-body_amd's library was loaded before `import torch` in this process: the two would run on "
                               "two different HIP runtimes (torch wheels bundle their own).  Import torch first.")
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.n = rank, world, num_bodies
        self.spans = [pkg.shard_range(num_bodies, world, r) for r in range(world)]
        self.sim = pkg.NBodySim(num_bodies, dt=dt, useBarnesHut=useBarnesHut, ref_compat=ref_compat, device=device,
                                shard=self.spans[rank], **kw)
        self.device = torch.device("cuda", device)
        # One dedicated (non-default) stream carries BOTH the library's kernels and the collectives, so the
        # all-gather is ordered after the integrate kernel and before the next force kernel without host syncs.
        # (The default stream's handle is 0, which nbody_set_stream reads as "use the context's own stream".)
        self.stream = torch.cuda.Stream(device=self.device)
        self.sim.set_stream(self.stream.cuda_stream)
        self._pos = None
        self._stage = None
        self._host = None
        self.exchange_mode = exchange or os.environ.get("NBODY_EXCHANGE", "torch")
        if self.exchange_mode not in ("torch", "native"):
            raise ValueError(f"exchange={self.exchange_mode!r}: expected 'torch' or 'native'")
        # RCCL collectives are stream-ordered device work.  A host-staged backend (gloo) is NOT handed CUDA tensors:
        # its CUDA path copies through pinned memory from helper threads on private streams that wait for an event
        # recorded on the compute stream, and on torch 2.10 / ROCm 7 both ranks were seen to stall inside that wait
        # (profiles/r2_dist_probe_*.txt).  Host staging needs a host round trip anyway, so this class
        # does it itself: D2H of the owned slice on the compute stream, stream sync, CPU collective, H2D.
        self._host_staged = dist is not None and dist.get_backend() != "nccl"
        if self.exchange_mode == "native" and world > 1:
            if dist is None:
                raise ValueError("exchange='native' needs a process group to hand the RCCL id to the ranks")
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(pkg.comm_unique_id()), dtype=torch.uint8).clone()
            if dist.get_backend() == "nccl":
                uid = uid.to(self.device)
            dist.broadcast(uid, src=0)
            self.sim.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))

    def upload(self, m, posX, posY, velX, velY, radius):
        self.sim.upload(m, posX, posY, velX, velY, radius)
        t = self.torch
        self._pos = [t.as_tensor(self.sim.device_array(k), device=self.device) for k in ("posX", "posY")]
        self.prepare_exchange()
        return self

    def upload_state(self, posX, posY, velX, velY):
        """Back to another state of the same bodies (bench.py re-uploads the initial state after the warm-up)."""
        self.sim.upload_state(posX, posY, velX, velY)
        return self

    def prepare_exchange(self, force=False):
        """Staging buffers, once per upload (force: also with one rank — tools/dist_probe.py)."""
        t = self.torch
        if (self.world > 1 or force) and self.exchange_mode == "torch":
            if self._host_staged:
                self._host = [(t.empty(self.n, dtype=t.float32, pin_memory=True),
                               t.empty(self.n, dtype=t.float32, pin_memory=True)) for _ in self._pos]
            else:
                self._stage = [t.empty_like(p) for p in self._pos]
        return self

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self.sim.step(1)                               # force over all sources + Euler on the owned shard
            if self.world > 1:
                self.exchange()
        return self

    def exchange(self, force=False):
        if self.world == 1 and not force:
            return
        if self.exchange_mode == "native":
            self.sim.exchange_positions()                  # in-place ncclAllGather on the context's (= this) stream
            return
        t = self.torch
        if self._host_staged:
            b, c = self.spans[self.rank]
            with t.cuda.stream(self.stream):
                for p, (h, _) in zip(self._pos, self._host):
                    h[b:b + c].copy_(p[b:b + c], non_blocking=True)
            self.stream.synchronize()                      # the owned slices are in host memory
            exchange_slices([h for h, _ in self._host], self.spans, self.rank, self.dist,
                            staging=[g for _, g in self._host])
            with t.cuda.stream(self.stream):               # ordered before the next step's kernels
                for p, (h, _) in zip(self._pos, self._host):
                    p.copy_(h, non_blocking=True)
            return
        with t.cuda.stream(self.stream):
            exchange_slices(self._pos, self.spans, self.rank, self.dist, staging=self._stage)

    def reduce_max(self, value):
        """max over ranks of a host float (bench.py's timing), on the backend's own kind of tensor."""
        if self.dist is None or self.world == 1:
            return float(value)
        t = self.torch
        x = t.tensor([value], dtype=t.float64, device="cpu" if self._host_staged else self.device)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.MAX)
        return float(x.item())

    def sync(self):
        self.stream.synchronize()

    def targets(self):
        return self.spans[self.rank]

    def close(self):
        self._pos = None
        self._host = None
        self.sim.close()


ShardedNaiveSim = ShardedSim      # earlier name


def owned_targets(num_bodies, spans, rank, ref_compat=True):
    """Number of bodies rank actually advances (the N%8 tail is frozen in ref_compat, benchmark.cpp:423)."""
    b, c = spans[rank]
    lim = num_bodies - num_bodies % 8 if ref_compat else num_bodies
    return max(0, min(b + c, lim) - b)


def to_numpy(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False)
