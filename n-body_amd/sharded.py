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
    rebuilds the (cheap) tree from the gathered positions and walks it for its own targets only."""

    def __init__(self, pkg, num_bodies, rank, world, *, device, dt=0.05, ref_compat=True, dist=None,
                 useBarnesHut=False, **kw):
        import torch
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.n = rank, world, num_bodies
        self.spans = [pkg.shard_range(num_bodies, world, r) for r in range(world)]
        self.sim = pkg.NBodySim(num_bodies, dt=dt, useBarnesHut=useBarnesHut, ref_compat=ref_compat, device=device,
                                shard=self.spans[rank], **kw)
        self.device = torch.device("cuda", device)
        # One dedicated (non-default) stream carries BOTH the library's kernels and torch's collectives, so the
        # all-gather is ordered after the integrate kernel and before the next force kernel without host syncs.
        # (The default stream's handle is 0, which nbody_set_stream reads as "use the context's own stream".)
        self.stream = torch.cuda.Stream(device=self.device)
        self.sim.set_stream(self.stream.cuda_stream)
        self._pos = None
        self._stage = None
        # RCCL collectives are stream-ordered device work.  Host-staged backends (gloo: the CPU tests and the
        # one-GPU rehearsal) copy through pinned memory from a helper thread; torch 2.10/ROCm 7 was seen to stall
        # there when the helper's stream had to wait on a still-busy compute stream (tools/dist_probe.py), so for
        # those backends the stream is drained on the host first.
        self._host_staged = dist is not None and world > 1 and dist.get_backend() != "nccl"

    def upload(self, m, posX, posY, velX, velY, radius):
        self.sim.upload(m, posX, posY, velX, velY, radius)
        t = self.torch
        self._pos = [t.as_tensor(self.sim.device_array(k), device=self.device) for k in ("posX", "posY")]
        self._stage = [t.empty_like(p) for p in self._pos] if self.world > 1 else None
        return self

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self.sim.step(1)                               # force over all sources + Euler on the owned shard
            if self.world > 1:
                self.exchange()
        return self

    def exchange(self):
        if self._host_staged:
            self.stream.synchronize()
        with self.torch.cuda.stream(self.stream):
            exchange_slices(self._pos, self.spans, self.rank, self.dist, staging=self._stage)

    def sync(self):
        self.stream.synchronize()

    def targets(self):
        return self.spans[self.rank]

    def close(self):
        self._pos = None
        self.sim.close()


ShardedNaiveSim = ShardedSim      # earlier name


def owned_targets(num_bodies, spans, rank, ref_compat=True):
    """Number of bodies rank actually advances (the N%8 tail is frozen in ref_compat, benchmark.cpp:423)."""
    b, c = spans[rank]
    lim = num_bodies - num_bodies % 8 if ref_compat else num_bodies
    return max(0, min(b + c, lim) - b)


def to_numpy(t):
    return t.detach().cpu().numpy().astype(np.float32, copy=False)
