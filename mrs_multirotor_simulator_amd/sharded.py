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


def bind_native_exchange(swarm, n_total, transport="rccl", group=None):
    """Binds the library's OWN sharded tick (mrs_swarm_tick_sharded_n) to the ranks of a torch.distributed job, one swarm per rank:
    "rccl" — an RCCL communicator of the library's own (the 128-byte id travels over `group`);
    "peer" — the peer-window exchange (direct writes into the peers' device memory; the 64-byte IPC handles travel over `group`).
    Afterwards `swarm.tick_sharded_n(...)` on every rank; `dist.barrier()` before `swarm.comm_destroy()` (peer windows)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if transport == "peer":
        _, handle = swarm.peer_window_create(world, rank, n_total)
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)
        swarm.comm_init_peer(handles=handles)
        return
    if transport != "rccl":
        raise ValueError(f"unknown transport {transport!r}")
    from .swarm import rccl_unique_id
    box = [bytes(rccl_unique_id()) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    swarm.comm_init(world, rank, box[0], n_total)


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


# ------------------------------------------------------------------------------------------------------------------------------
# Export-set exchange driven from Python: the synchronous form of the protocol the library runs natively in
# mrs_swarm_tick_sharded_n (tick_sharded.hip: export_ticks / export_search).  Between two neighbour searches a rank only needs the
# positions of the foreign UAVs within sqrt(3) + SKIN of one of its own at the last search, and — the relation being symmetric —
# only has to publish the own UAVs that have such a foreign neighbour.  Per tick:
#   step -> all-gather of [stale flag | padded export positions]
#   nobody stale : collide against own UAVs + received exports            (24 B per EXPORTED UAV on the wire)
#   somebody stale (moved > SKIN/2 since the search): all-gather of every record, search, new export sets, all-gather of the slot maps
# The engine is anything with step(dt), records() -> (n_local, 6) array [x y z mass arm prop] and
# collide(foreign_global_index, foreign_records, enabled, crash, rebounce); the world_size-2/-4 gloo tests run it on the CPU.
SKIN = 0.5
LIST_RADIUS = 3.0 ** 0.5 + SKIN


def slab_order(pos, world):
    """public indices sorted by x (stable): rank r holds order[shard_range(n, world, r)] — the numpy twin of mrs_slab_partition"""
    import numpy as np
    x = np.asarray(pos, dtype=np.float64)[:, 0]
    return np.argsort(np.where(np.isnan(x), np.inf, x), kind="stable")


class ExportSetSwarm:
    def __init__(self, n_total, engine, device, group=None):
        import numpy as np
        self.np = np
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_total = n_total
        self.lo, self.hi = shard_range(n_total, self.world, self.rank)
        self.n_local, self.n_max = self.hi - self.lo, max_shard(n_total, self.world)
        self.engine, self.device = engine, device
        self.live = False          # export sets valid
        self.cap = 0               # export slots per rank in the padded collective
        self.stats = dict(ticks=0, searches=0, bytes_list_ticks=0, bytes_search_ticks=0)

    def _allgather(self, t):
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=self.device)
        if self.world > 1:
            dist.all_gather_into_tensor(out.view(-1, *t.shape[1:]) if t.dim() > 1 else out.view(-1), t.contiguous(), group=self.group)
        else:
            out[0] = t
        return out

    def _search(self, rec):
        """full gather + neighbour search + export sets.  Returns the gathered records (world, n_max, 6) as numpy."""
        np = self.np
        from scipy.spatial import cKDTree
        send = torch.full((self.n_max, REC), float("nan"), dtype=torch.float64, device=self.device)
        send[: self.n_local] = torch.from_numpy(rec)
        allrec = self._allgather(send).cpu().numpy()
        self.stats["searches"] += 1
        self.stats["bytes_search_ticks"] += self.n_max * REC * 8
        flat = allrec.reshape(-1, REC)
        usable = np.isfinite(flat[:, :3]).all(axis=1)
        mine = np.arange(self.rank * self.n_max, self.rank * self.n_max + self.n_local)
        ui = np.flatnonzero(usable)  # NaN padding and non-finite positions take no part
        tree = cKDTree(flat[ui, :3])
        foreign_of = [[] for _ in range(self.n_local)]
        for k, g in enumerate(mine):
            if usable[g]:
                for jj in tree.query_ball_point(flat[g, :3], LIST_RADIUS * (1 + 1e-12)):
                    j = int(ui[jj])
                    if j // self.n_max != self.rank:
                        foreign_of[k].append(j)
        exported = np.array([len(f) > 0 for f in foreign_of], dtype=bool)
        self.exp_local = np.flatnonzero(exported)                       # own UAVs some other rank lists (symmetry)
        slot = np.full(self.n_max + 1, -1, dtype=np.int64)
        slot[0] = len(self.exp_local)
        slot[1 + self.exp_local] = np.arange(len(self.exp_local))
        maps = self._allgather(torch.from_numpy(slot).to(self.device)).cpu().numpy()
        self.stats["bytes_search_ticks"] += (self.n_max + 1) * 8
        self.cap = int(maps[:, 0].max())
        # foreign UAVs this rank needs: (gathered slot g, position in the padded export collective)
        need = sorted({j for f in foreign_of for j in f})
        self.need_g = np.array(need, dtype=np.int64)
        q, j = self.need_g // self.n_max, self.need_g % self.n_max
        e = maps[q, 1 + j]
        assert (e >= 0).all(), "a listed foreign UAV is not in its owner's export set: the relation must be symmetric"
        self.need_slot = q * self.cap + e
        self.need_const = flat[self.need_g, 3:]
        self.ref_pos = rec[:, :3].copy()
        self.ref_const = rec[:, 3:].copy()
        self.live = True
        return flat

    def _global_index(self, g):
        q, k = self.np.divmod(g, self.n_max)
        base = self.np.array([shard_range(self.n_total, self.world, r)[0] for r in range(self.world)])
        return base[q] + k

    def tick(self, dt, enabled, crash, rebounce):
        np = self.np
        self.engine.step(dt)
        self.stats["ticks"] += 1
        if not (enabled or crash):
            return
        rec = np.asarray(self.engine.records(), dtype=np.float64).reshape(self.n_local, REC)
        stale = 1.0
        if self.live:
            d2 = ((rec[:, :3] - self.ref_pos) ** 2).sum(axis=1)
            ok_now, ok_ref = np.isfinite(rec[:, :3]).all(axis=1), np.isfinite(self.ref_pos).all(axis=1)
            moved = (ok_now != ok_ref) | (ok_now & ok_ref & ~(d2 <= (0.5 * SKIN) ** 2 * (1 - 1e-9)))
            stale = float(moved.any() or not np.array_equal(rec[:, 3:], self.ref_const))
        send = torch.zeros((1 + self.cap, 3), dtype=torch.float64, device=self.device)
        send[0, 0] = stale
        if self.live and len(self.exp_local):
            send[1: 1 + len(self.exp_local)] = torch.from_numpy(rec[self.exp_local, :3])
        got = self._allgather(send).cpu().numpy()
        self.stats["bytes_list_ticks"] += (1 + self.cap) * 24
        if got[:, 0, 0].any():  # some rank's lists are stale: every rank searches (same decision everywhere)
            flat = self._search(rec)
            fg = np.array([g for g in range(len(flat)) if g // self.n_max != self.rank and np.isfinite(flat[g, 0])], dtype=np.int64)
            # the search tick collides against every gathered record, like the full exchange
            self.engine.collide(self._global_index(fg), flat[fg], enabled, crash, rebounce)
            return
        pos = got[:, 1:, :].reshape(-1, 3)[self.need_slot] if len(self.need_g) else np.zeros((0, 3))
        self.engine.collide(self._global_index(self.need_g), np.concatenate([pos, self.need_const], axis=1), enabled, crash, rebounce)

    def tick_n(self, dt, n_ticks, enabled, crash, rebounce):
        for _ in range(n_ticks):
            self.tick(dt, enabled, crash, rebounce)
