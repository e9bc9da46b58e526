"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).  Two modes (DESIGN.md "Multi-GPU"):

replicas      the defect path has no coupling between decision vectors, so a batch is split by
              vectors: rank r owns vectors [lo, hi).  No data-path collective; results stay on the
              rank that produced them.  This is the weak-scaling mode bench.py measures.

phase shards  ONE batch evaluated by all ranks together (BASELINE.json config 4): the path is
              block-diagonal per phase (lib/con_dynamics.py:46,132,237,320,512,554) and its forward-difference
              columns are independent, so the UNITS (work item, part) of every vector are dealt to ranks in
              contiguous, cost-balanced ranges.  A rank's kernel writes the entries its units own STRAIGHT into its
              slice of one exchange buffer out [world][B][width] (every unit's entries one contiguous run), and ONE
              in-place all-gather (all_gather_into_tensor, send = out[rank]) completes the buffer on every rank:
              no pack, no unpack, no zero fill; (N-1)/N of 8*(11N + V) bytes per vector (up to padding to the
              largest share) -- latency-bound over xGMI, which is why replicas are preferred whenever there is more
              than one vector.  `UnitShards` is that exchange; bench.py --mode phase-shard and the gloo tests
              (world sizes 2 and 4, one rank without units) drive the same object.
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
    """The unit partition of one problem over `world` ranks and the exchange of owned entries with ZERO pack / unpack launches.

    engine: an Engine (a host-only handle is enough to plan).  Every rank agrees on one exchange buffer
    ``out [world][B][width]`` (gel_shard_plan): rank r's kernel writes the entries its units own, of all B vectors, straight
    into its slice out[r] (Engine.eval_shard_packed_device), and ONE in-place all-gather over the slices completes the
    buffer on every rank.  ``res_pos`` / ``jv_pos`` map the ordinary res / compact-value layouts into it
    (pos = rank * width + offset); `gather` reads a finished buffer through them, Engine.shard_unpack_device does it in one
    launch on the device."""

    def __init__(self, engine, world, rank, unit_begin=None):
        """unit_begin [world + 1] (optional): the caller's own contiguous unit ranges instead of the cost-balanced cut"""
        self.world, self.rank = int(world), int(rank)
        if unit_begin is None:
            self.ranges = shard_chunks(unit_costs(engine), world)
            self.unit_begin = np.array([r[0] for r in self.ranges] + [self.ranges[-1][0] + self.ranges[-1][1]], dtype=np.int32)
        else:
            self.unit_begin = np.ascontiguousarray(unit_begin, dtype=np.int32)
            assert len(self.unit_begin) == self.world + 1
            self.ranges = [(int(self.unit_begin[r]), int(self.unit_begin[r + 1] - self.unit_begin[r])) for r in range(self.world)]
        self.engine = engine
        import os
        # [r6] default: the NON-aliased send buffer (a copy of this rank's slice: one device copy of width * B doubles per step).  The
        # in-place form (send = a view of the receive buffer at the rank's offset: documented for NCCL / RCCL all-gather, a memcpy
        # with src == dst on gloo) has never run on RCCL with more than one rank from this repository -- gpurun boxes have one GPU,
        # the driver's multi-GPU records were skipped in every round -- so it is opt-in until it has: GELATO_AMD_ALLGATHER_INPLACE=1
        # (GELATO_AMD_ALLGATHER_COPY=1, the older switch, still forces the copy)
        self.inplace = (os.environ.get("GELATO_AMD_ALLGATHER_INPLACE", "0") not in ("", "0")
                        and os.environ.get("GELATO_AMD_ALLGATHER_COPY", "0") in ("", "0"))
        self.width, self.res_pos, self.jv_pos = engine.shard_plan(self.unit_begin)
        self.plan_key = engine.shard_plan_key
        self.nres, self.V = engine.nres, engine.V
        # entries per rank (what a rank really contributes; the slices are padded to the largest share)
        self.counts = [(int(np.count_nonzero(self.res_pos // self.width == r)), int(np.count_nonzero(self.jv_pos // self.width == r)))
                       for r in range(self.world)]
        assert sum(c[0] for c in self.counts) == self.nres and sum(c[1] for c in self.counts) == self.V
        assert max(a + b for a, b in self.counts) <= self.width

    def bytes_received_per_vector(self):
        """what the one collective delivers to a rank per decision vector, padding included"""
        return 8 * self.width * (self.world - 1)

    def check_current(self):
        """The plan lives in the engine's handle and a later UnitShards(engine, ...) replaces it: this object's width and maps
        then describe a layout the device no longer writes.  Raises instead of letting a stale object size a buffer."""
        if getattr(self.engine, "shard_plan_key", None) != self.plan_key:
            raise RuntimeError("UnitShards: the engine's shard plan was replaced by a later gel_shard_plan call "
                               "(one Engine holds one plan: build the other UnitShards on its own Engine, or re-create this one)")

    @property
    def plan(self):
        """(nranks, width): what Engine.eval_shard_packed_device / shard_unpack_device check the handle's plan against"""
        return (self.world, self.width)

    def buffer(self, B, device=None, dtype=None):
        """the exchange buffer out [world][B][width] (never zero-filled: every entry that is read has an owner)"""
        import torch
        self.check_current()
        return torch.empty((self.world, int(B), self.width), dtype=dtype or torch.float64, device=device)

    def step(self, evaluate_packed, out, group=None):
        """out [world][B][width] (torch).  ``evaluate_packed(out, rank)`` writes this rank's entries into out[rank]
        (Engine.eval_shard_packed_device on a GPU); then ONE all-gather, in place: send = out[rank], receive = out.
        Afterwards every rank holds every entry."""
        import torch.distributed as dist
        self.check_current()
        if tuple(out.shape[::2]) != (self.world, self.width):
            raise ValueError("UnitShards.step: out is %s, the plan needs [%d][B][%d]" % (tuple(out.shape), self.world, self.width))
        if self.ranges[self.rank][1] > 0:
            evaluate_packed(out, self.rank)
        if self.world > 1:
            send = out[self.rank].view(-1)
            if not self.inplace:
                # the non-aliased form: the send buffer is a copy of the slice (one device copy of width * B doubles).  The in-place
                # form (send = a view of the receive buffer at this rank's offset) is what NCCL / RCCL document for all-gather and
                # what gloo executes as a memcpy with src == dst; it has never run on RCCL with more than one rank from this
                # repository (gpurun boxes have one GPU), so GELATO_AMD_ALLGATHER_COPY=1 keeps a way out
                send = send.clone()
            dist.all_gather_into_tensor(out.view(-1), send, group=group)
        return out

    def flat_index(self, B):
        """(res_idx [11N], jv_idx [V], stride): entry i of vector b sits at out.view(-1)[idx[i] + b * stride]"""
        w = self.width
        f = lambda pos: (pos // w) * (int(B) * w) + pos % w          # noqa: E731
        return f(self.res_pos), f(self.jv_pos), w

    def gather(self, out):
        """a finished exchange buffer -> (res [B][11N], jvar [B][V]) in the ordinary layouts (host-side reader of the map)"""
        import torch
        B = out.shape[1]
        ri, ji, w = self.flat_index(B)
        flat = out.reshape(-1)
        boff = (torch.arange(B, device=out.device) * w)[:, None]
        ri_t, ji_t = torch.from_numpy(ri).to(out.device)[None, :], torch.from_numpy(ji).to(out.device)[None, :]
        return flat[ri_t + boff], flat[ji_t + boff]

    def scatter_owned(self, res, jvar, out, rank=None):
        """the inverse, for one rank's entries: what a packed evaluation of `rank` writes, given the ordinary-layout results
        (the CPU tests' stand-in for the kernel)"""
        import torch
        r = self.rank if rank is None else rank
        w = self.width
        rsel, jsel = np.nonzero(self.res_pos // w == r)[0], np.nonzero(self.jv_pos // w == r)[0]
        out[r][:, torch.from_numpy(self.res_pos[rsel] % w)] = res[:, torch.from_numpy(rsel)]
        out[r][:, torch.from_numpy(self.jv_pos[jsel] % w)] = jvar[:, torch.from_numpy(jsel)]
        return out


def max_over_ranks(value, device=None, group=None):
    """bench.py timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
