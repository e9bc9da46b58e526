"""Multi-GPU plumbing: one process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).  Two modes (DESIGN.md "Multi-GPU"):

replicas      the defect path has no coupling between decision vectors, so a batch is split by
              vectors: rank r owns vectors [lo, hi).  No data-path collective; results stay on the
              rank that produced them.  This is the weak-scaling mode bench.py measures.

phase shards  ONE batch evaluated by all ranks together (BASELINE.json config 4): the path is
              block-diagonal per phase (lib/con_dynamics.py:46,132,237,320,512,554), so the 64-node
              work items of every vector are dealt to ranks in contiguous, cost-balanced ranges; every
              rank fills its own entries of zero-initialised res / jvar buffers and ONE sum all-reduce
              (each entry has exactly one owner) leaves the complete result on every rank.  The
              exchange is <= 8*(11N + V) bytes per vector (243 KB at 6x64) -- latency-bound over xGMI,
              which is why replicas are preferred whenever there is more than one vector.
"""
import numpy as np


def replica_range(total, rank, world):
    """Contiguous, balanced split of `total` decision vectors: -> (lo, hi) of this rank."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def chunk_costs(engine_or_desc):
    """Relative cost of every work item (64-node chunk), from its phase type: an aerodynamic phase
    runs the heavy chain 4x + 2 time sweeps, a NoAir phase only gravity.  Accepts an Engine (host-only
    is enough) or a dict with chunk_phase / reference_area / attitude_hold arrays."""
    if isinstance(engine_or_desc, dict):
        ph, area, hold = engine_or_desc["chunk_phase"], engine_or_desc["reference_area"], engine_or_desc["attitude_hold"]
    else:
        e = engine_or_desc
        ph, area, hold = e.chunk_phase(), e.prob["reference_area"], e.prob["attitude_hold"]
    cost = np.where(np.asarray(area)[ph] != 0.0, 10.0, 1.5) + np.where(np.asarray(hold)[ph] != 0, 0.0, 0.5)
    return cost


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


def all_reduce_owned(buffers, group=None):
    """Sum all-reduce of buffers whose every entry is non-zero on at most one rank (its owner)."""
    import torch.distributed as dist
    for b in buffers:
        if b is not None:
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
    return buffers


def phase_sharded_eval(evaluate_range, res, jvar, ranges, rank, group=None):
    """Generic driver of the phase-shard mode.  `evaluate_range(begin, count, res, jvar)` must fill
    ONLY the entries owned by work items [begin, begin+count) (gel_eval_shard_device does exactly
    that); res / jvar are torch tensors that this function zeroes first and all-reduces after."""
    res.zero_()
    if jvar is not None:
        jvar.zero_()
    begin, count = ranges[rank]
    if count > 0:
        evaluate_range(begin, count, res, jvar)
    all_reduce_owned([res, jvar], group)
    return res, jvar


def max_over_ranks(value, device=None, group=None):
    """bench.py timing rule: the slowest rank defines the step time."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
