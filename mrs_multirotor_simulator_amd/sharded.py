"""Multi-GPU swarm: one process per GPU, static contiguous index shards.

makeStep needs no communication (UAVs are independent).  Only the collision pass exchanges data: every rank packs
{x, y, z, mass, arm_length, prop_radius} (48 B) per UAV, ONE all-gather (RCCL over xGMI, `nccl` backend; `gloo` on CPU
for the tests) assembles the records of the whole swarm on every rank, and each rank runs the spatial-hash pass for its
own UAVs against all records.  Shards are padded to equal length with NaN records (NaN never collides), so the plain
equal-size all-gather applies.
"""
import math

import torch
import torch.distributed as dist

REC = 6  # doubles per record


def shard_range(n_total, world, rank):
    """Contiguous index shard [lo, hi) of `rank`: sizes differ by at most one."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_shard(n_total, world):
    return math.ceil(n_total / world)


class GpuEngine:
    """Adapter of mrs_multirotor_simulator_amd.Swarm to the three calls ShardedSwarm needs."""

    def __init__(self, swarm, device):
        self.swarm, self.device = swarm, device
        # the swarm's own HIP stream as a torch stream: the hand-over to and from the collective (which runs on torch's current
        # stream) is then stream-to-stream ordering through events — no host synchronisation inside a tick
        self.ext = torch.cuda.ExternalStream(swarm.stream(), device=device)

    def step(self, dt):
        self.swarm.step(dt)

    def write_records(self, out):  # out: (n_local, 6) float64 slice of the send buffer on self.device
        self.swarm.pack_positions_to(out.data_ptr())
        torch.cuda.current_stream(self.device).wait_stream(self.ext)  # the collective starts after the pack kernel

    def collide(self, records, n_records, my_offset, enabled, crash, rebounce):
        self.ext.wait_stream(torch.cuda.current_stream(self.device))  # the collision pass starts after the collective
        self.swarm.handle_collisions_gathered(records.data_ptr(), n_records, my_offset, enabled, crash, rebounce)
        # the next tick's pack overwrites the send buffer and the next collective the receive buffer: both are ordered behind
        # this pass on the swarm's stream / by the wait above
        torch.cuda.current_stream(self.device).wait_stream(self.ext)


class ShardedSwarm:
    def __init__(self, n_total, engine, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        self.n_local = self.hi - self.lo
        self.n_max = max_shard(n_total, self.world)
        self.engine = engine
        self.send = torch.full((self.n_max, REC), float("nan"), dtype=torch.float64, device=device)
        self.recv = torch.empty((self.world * self.n_max, REC), dtype=torch.float64, device=device)

    def exchange(self):
        """all-gather of the position records; returns (records tensor, offset of this shard inside it)."""
        if self.n_local:
            self.engine.write_records(self.send[: self.n_local])
        if self.world > 1:
            dist.all_gather_into_tensor(self.recv, self.send, group=self.group)
        else:
            self.recv.copy_(self.send)
        return self.recv, self.rank * self.n_max

    def global_index(self, rec_index):
        """index in the gathered buffer -> global UAV index (or -1 for padding)."""
        r, k = divmod(rec_index, self.n_max)
        lo, hi = shard_range(self.n_total, self.world, r)
        return lo + k if k < hi - lo else -1

    def handle_collisions(self, enabled, crash, rebounce):
        if not (enabled or crash):
            return
        rec, off = self.exchange()
        self.engine.collide(rec, self.world * self.n_max, off, enabled, crash, rebounce)

    def tick_n(self, dt, n_ticks, enabled, crash, rebounce):
        """timerMain order (src/multirotor_simulator.cpp:211-217) on every shard."""
        for _ in range(n_ticks):
            self.engine.step(dt)
            self.handle_collisions(enabled, crash, rebounce)
