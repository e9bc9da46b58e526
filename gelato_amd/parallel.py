"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).  Two modes (DESIGN.md "Multi-GPU"):

replicas      the defect path has no coupling between decision vectors, so a batch is split by
              vectors: rank r owns vectors [lo, hi).  No data-path collective; results stay on the
              rank that produced them.  This is the weak-scaling mode bench.py measures.

phase shards  ONE batch evaluated by all ranks together (BASELINE.json config 4): the path is
              block-diagonal per phase (lib/con_dynamics.py:46,132,237,320,512,554) and its forward-difference
              columns are independent, so the UNITS (work item, part) of every vector are dealt to ranks in
              contiguous, cost-balanced ranges.  A rank evaluates its units into the ordinary res / jvar buffers
              (entries of other ranks are simply not touched: no zero fill), packs the entries it owns, and ONE
              all-gather (all_gather_into_tensor) hands every rank every other rank's entries: (N-1)/N of
              8*(11N + V) bytes per vector (up to padding to the largest share) -- latency-bound over xGMI, which
              is why replicas are preferred whenever there is more than one vector.  `UnitShards` is that
              exchange; bench.py --mode phase-shard and the gloo world_size-2 test drive the same object.
"""
import numpy as np


def replica_range(total, rank, world):
    """Contiguous, balanced split of `total` decision vectors: -> (lo, hi) of this rank."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def unit_costs(engine_or_desc):
    """Relative cost of every unit = (work item, part) in unit-id order 4 * item + part: part 0 is the centre
    evaluation with the light sweeps, parts 1..3 one position sweep (plus a centre evaluation) each; phases
    without aerodynamics have empty parts 1..3 (gel_eval_shard_units_device)."""
    if isinstance(engine_or_desc, dict):
        ph, area, hold = engine_or_desc["chunk_phase"], engine_or_desc["reference_area"], engine_or_desc["attitude_hold"]
    else:
        e = engine_or_desc
        ph, area, hold = e.chunk_phase(), e.prob["reference_area"], e.prob["attitude_hold"]
    air = np.asarray(area)[ph] != 0.0
    free = np.asarray(hold)[ph] == 0
    cost = np.zeros((len(ph), 4))
    cost[:, 0] = np.where(air, 4.5, 1.5) + np.where(free, 0.5, 0.0)
    cost[:, 1:] = np.where(air, 4.0, 0.0)[:, None]
    return cost.ravel()


def shard_chunks(costs, world):
    """Contiguous partition of the work items into `world` ranges with balanced total cost.
    -> list of (begin, count); ranges are disjoint, ordered, and cover every item."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    world = int(world)
    if world <= 1:
        return [(0, n)]
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, cuts[-1]), n)
        if k > cuts[-1] and k <= n and abs(cum[k - 1] - target) < abs(cum[k] - target):
            k -= 1
        cuts.append(max(k, cuts[-1]))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


class UnitShards:
    """The unit partition of one problem over `world` ranks and the exchange of owned entries.

    engine: an Engine (a host-only handle is enough: only the partition is read from it).  The evaluator is any
    callable ``evaluate(unit_begin, unit_count, res, jvar)`` that fills the entries owned by those units in the
    [B, 11N] / [B, V] tensors -- Engine.eval_shard_units_device on a GPU."""

    def __init__(self, engine, world, rank):
        import torch
        self.world, self.rank = int(world), int(rank)
        self.ranges = shard_chunks(unit_costs(engine), world)
        ro, jo = engine.unit_owner()
        self.res_idx, self.jv_idx = [], []
        for (u0, cnt) in self.ranges:
            self.res_idx.append(torch.from_numpy(np.nonzero((ro >= u0) & (ro < u0 + cnt))[0]))
            self.jv_idx.append(torch.from_numpy(np.nonzero((jo >= u0) & (jo < u0 + cnt))[0]))
        self.counts = [(len(a), len(b)) for a, b in zip(self.res_idx, self.jv_idx)]
        assert sum(c[0] for c in self.counts) == engine.nres and sum(c[1] for c in self.counts) == engine.V
        self.width = max(a + b for a, b in self.counts)     # doubles per vector and rank in the exchange (padded)
        self._dev = None
        self._send = self._recv = None

    def bytes_received_per_vector(self):
        """what the one collective delivers to a rank per decision vector, padding included"""
        return 8 * self.width * (self.world - 1)

    def _buffers(self, B, like):
        if self._send is None or self._send.shape[0] != B or self._send.device != like.device:
            import torch
            self._send = torch.empty((B, self.width), dtype=like.dtype, device=like.device)
            self._recv = torch.empty((self.world, B, self.width), dtype=like.dtype, device=like.device)
            self.res_idx = [i.to(like.device) for i in self.res_idx]
            self.jv_idx = [i.to(like.device) for i in self.jv_idx]
        return self._send, self._recv

    def pack(self, res, jvar, out, rank=None):
        """the entries rank `rank` (default: this one) owns, of every vector: res rows then compact values -> out [B, width]"""
        import torch
        r = self.rank if rank is None else rank
        nr, nj = self.counts[r]
        torch.index_select(res, 1, self.res_idx[r], out=out[:, :nr])
        torch.index_select(jvar, 1, self.jv_idx[r], out=out[:, nr:nr + nj])
        return out

    def unpack(self, recv, res, jvar, skip=None):
        """recv [world, B, width]: every rank's packed entries -> their places in res / jvar (rank `skip` left alone)"""
        for r in range(self.world):
            if r != skip:
                nr, nj = self.counts[r]
                res.index_copy_(1, self.res_idx[r], recv[r, :, :nr])
                jvar.index_copy_(1, self.jv_idx[r], recv[r, :, nr:nr + nj])
        return res, jvar

    def step(self, evaluate, res, jvar, group=None):
        """res [B, 11N], jvar [B, V] (torch).  Evaluates this rank's units, exchanges owned entries with ONE
        all-gather, and leaves the complete result in res / jvar on every rank.  No buffer is ever zero-filled."""
        import torch.distributed as dist
        u0, cnt = self.ranges[self.rank]
        if cnt > 0:
            evaluate(u0, cnt, res, jvar)
        if self.world == 1:
            return res, jvar
        send, recv = self._buffers(res.shape[0], res)
        self.pack(res, jvar, send)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
        return self.unpack(recv, res, jvar, skip=self.rank)


def max_over_ranks(value, device=None, group=None):
    """bench.py timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
