"""CPU-only: (1) the engine's host-side sparsity pattern / constant values / compact index map against
the reference goldens and the oracle (bit-exact for int32 indices and for constant entries), through a
host-only handle (device = GEL_DEVICE_NONE, which can describe but never evaluate); (2) the multi-GPU
plumbing (gelato_amd.parallel) in a real world_size-2 `gloo` run, with the oracle standing in for the
GPU evaluator of each rank."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import D_tau_from_golden, ROOT, load_golden, problem_from_golden

import oracle
from gelato_amd import Engine, _lib, parallel
from gelato_amd.engine import BLOCKS


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def host_engine(prob, D=None, tau=None):
    return Engine(prob, D=D, tau=tau, device=-1)


@pytest.mark.parametrize("name", ["example", "3x32", "mixed6x64", "dense6x64", "negarea"])
def test_pattern_bit_exact_vs_reference_golden(name):
    g = load_golden("g6_%s.npz" % name)
    prob = problem_from_golden(g)
    D, tau = D_tau_from_golden(g, prob)
    E = host_engine(prob, D, tau)
    pat = E.pattern()
    cv = E.const_values()
    vidx = E.var_index()
    is_var = E.var_mask()
    assert len(np.unique(vidx)) == E.V                                   # one COO entry per compact slot
    for b, (grp, var) in enumerate(BLOCKS):
        key = "jac_%s_%s" % (grp, var)
        r, c = pat[b]
        assert r.dtype == np.int32 and c.dtype == np.int32
        assert int(g[key + "_nnz"]) == len(r) == E.block_nnz[b]
        assert tuple(g[key + "_shape"]) == E.block_shape[b]
        assert str(g[key + "_rows_sha"]) == sha(r) and str(g[key + "_cols_sha"]) == sha(c), key
        if key + "_rows" in g:
            assert np.array_equal(r, g[key + "_rows"]) and np.array_equal(c, g[key + "_cols"])
        if key + "_vals" in g:                                          # constants: D / 0 / +-1 / massflow, bit for bit
            ref = g[key + "_vals"]
            m = ~is_var[E.block_off[b]:E.block_off[b + 1]]
            assert np.array_equal(cv[E.block_off[b]:E.block_off[b + 1]][m], ref[m]), key
    assert np.all(cv[is_var] == 0.0)


def test_pattern_matches_oracle_on_ragged_problem():
    g = load_golden("g6_example.npz")
    prob = dict(problem_from_golden(g))
    prob["num_nodes"] = np.array([2, 3, 100, 2, 17, 64, 5], dtype=np.int32)
    prob["thrust"] = np.array([420000.0, 0.0, 420000.0, 30700.0, 0.0, 30700.0, 1000.0])
    prob["massflow"] = np.array([140.0, 0.0, 140.0, 9.8, 0.0, 9.8, 0.3])
    prob["reference_area"] = np.array([2.21, 2.21, 2.21, 0.0, 0.0, 2.21, 0.0])
    prob["nozzle_area"] = np.array([0.68, 0.0, 0.68, 0.0, 0.0, 0.1, 0.0])
    prob["engine_on"] = np.array([1, 0, 1, 1, 0, 1, 1], dtype=np.int32)
    prob["attitude_hold"] = np.array([1, 0, 0, 1, 1, 0, 0], dtype=np.int32)
    P = oracle.Problem(prob)
    E = host_engine(prob, [P.D(i) for i in range(P.S)], [P.tau(i) for i in range(P.S)])
    x = np.random.default_rng(3).random(P.nvars) + 0.5
    pat = E.pattern()
    b = 0
    for grp in oracle.GROUPS:
        Jo = P.jacobian(grp, x)
        for var in oracle.BLOCK_VARS[grp]:
            r, c = pat[b]
            assert np.array_equal(r, Jo[var]["coo"][0]) and np.array_equal(c, Jo[var]["coo"][1]), (grp, var)
            assert E.block_shape[b] == Jo[var]["shape"]
            b += 1
    assert E.num_chunks() == 1 + 1 + 2 + 1 + 1 + 1 + 1
    assert list(E.chunk_phase()) == [0, 1, 2, 2, 3, 4, 5, 6]


def test_host_only_handle_cannot_evaluate():
    g = load_golden("g6_3x32.npz")
    E = host_engine(problem_from_golden(g))
    assert E.nvars == 1285 and E.nres == 1056 and E.stored_bytes == 8 * (1056 + E.V)
    with pytest.raises(_lib.GelatoAmdError, match="host-only"):
        E.eval_residual(g["x"])
    with pytest.raises(_lib.GelatoAmdError, match="host-only"):
        E.eval_batch_device(1, 1, 1, 1)
    with pytest.raises(_lib.GelatoAmdError, match="host-only"):
        E.jac_fd("mass", g["x"])


def test_algorithmic_bytes_match_survey():
    from gelato_amd import con_dynamics, problem
    for name, a_min in [("dense-6x64", 320072), ("3x32", None)]:
        pdict, unitdict, _, _ = problem.make_problem(name)
        E = host_engine(con_dynamics.problem_arrays(pdict, unitdict))
        if a_min:
            assert E.algorithmic_bytes == a_min and E.total_nnz == 745728
        # every x-dependent value the reference computes; the compact vector holds the distinct
        # ones among them (negated tf columns, the node-uniform diagonal and the structurally constant half of the
        # quaternion diagonal blocks are restored by the gather map)
        n_ref = int(np.count_nonzero(E.var_mask()))
        hold = np.asarray(E.prob["attitude_hold"]) != 0
        n_const_quat = int(8 * np.asarray(E.prob["num_nodes"])[~hold].sum())
        assert E.algorithmic_bytes == 8 * (E.nvars + E.nres + n_ref + n_const_quat)
        assert E.stored_bytes == 8 * (E.nres + E.V) and E.V < n_ref


# --------------------------------------------------------------------------
def test_replica_and_chunk_partitions():
    for total, world in [(4096, 8), (10, 3), (3, 8), (1, 1)]:
        rs = [parallel.replica_range(total, r, world) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == total
        assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in rs]
        assert max(sizes) - min(sizes) <= 1
    costs = np.array([10.5, 10.5, 10.5, 10.5, 10.0, 2.0])
    for world in [1, 2, 3, 4, 6, 8]:
        sh = parallel.shard_chunks(costs, world)
        assert len(sh) == world and sh[0][0] == 0 and sum(c for _, c in sh) == len(costs)
        assert all(sh[i][0] + sh[i][1] == sh[i + 1][0] for i in range(world - 1))
    assert parallel.shard_chunks(costs, 2) == [(0, 3), (3, 3)]
    # units = (work item, part): a 6-phase mesh gives 8 ranks something to do each
    from gelato_amd import con_dynamics, problem
    pdict, unitdict, _, _ = problem.make_problem("mixed-6x64")
    E = host_engine(con_dynamics.problem_arrays(pdict, unitdict))
    uc = parallel.unit_costs(E)
    assert len(uc) == 24 and uc.reshape(6, 4)[5, 1:].sum() == 0.0     # the NoAir phase has no position-sweep units
    sh = parallel.shard_chunks(uc, 8)
    assert all(c > 0 for _, c in sh) and sum(c for _, c in sh) == 24
    loads = [uc[b:b + c].sum() for b, c in sh]
    assert max(loads) <= 1.6 * (uc.sum() / 8)


WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
from conftest import load_golden, problem_from_golden
import oracle
from gelato_amd import Engine, parallel

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = load_golden("g6_3x32.npz")
prob = problem_from_golden(g)
P = oracle.Problem(prob)
E = Engine(prob, D=[P.D(i) for i in range(P.S)], tau=[P.tau(i) for i in range(P.S)], device=-1)
x = g["x"]

# ---- phase-shard mode: the SAME object bench.py --mode phase-shard drives (parallel.UnitShards), on the unit partition
#      of a host-only handle.  The oracle is only the source of the values a rank "computes": the stand-in for the kernel
#      writes exactly the entries its units own (the plan's map, cross-checked against gel_unit_owner) into ITS slice of an
#      exchange buffer that is otherwise NaN, so nothing can lean on a zero fill, and ONE in-place all-gather completes it. ----
B = 3
X = np.tile(x, (B, 1)) * (1 + 1e-7 * np.arange(B))[:, None]
ores, ovals = P.eval_batch(X)
full_jv = ovals[:, E.var_index()]
ro, jo = E.unit_owner()
assert ro.min() == 0 and jo.max() < 4 * E.num_chunks()
sh = parallel.UnitShards(E, world, rank)
assert len(sh.ranges) == world and sum(c for _, c in sh.ranges) == 4 * E.num_chunks()
if {empty_rank} >= 0:
    # a rank without units (more ranks than a cost-balanced cut can feed): its slice is never written, never read
    ub = sh.unit_begin.copy()
    ub[{empty_rank} + 1:-1] = np.maximum(ub[{empty_rank} + 1:-1], ub[{empty_rank}])
    ub[{empty_rank} + 1] = ub[{empty_rank}]
    sh = parallel.UnitShards(E, world, rank, unit_begin=ub)      # a new plan on the handle: a new object (the old one is stale now)
    assert sh.ranges[{empty_rank}][1] == 0
# the plan agrees with the ownership table: an entry lies in the slice of the rank that holds its unit, and the entries of
# one unit form one contiguous run
owner_rank = np.searchsorted(sh.unit_begin, ro, side="right") - 1
assert np.array_equal(sh.res_pos // sh.width, owner_rank)
owner_rank_j = np.searchsorted(sh.unit_begin, jo, side="right") - 1
assert np.array_equal(sh.jv_pos // sh.width, owner_rank_j)
allpos = np.concatenate([sh.res_pos, sh.jv_pos]); allown = np.concatenate([ro, jo])
assert len(np.unique(allpos)) == len(allpos)
for u in np.unique(allown):
    pu = np.sort(allpos[allown == u])
    assert pu[-1] - pu[0] + 1 == len(pu), "unit %d is not one contiguous run" % u
calls = []

def evaluate_packed(out, r):
    calls.append(r)
    sh.scatter_owned(torch.from_numpy(ores), torch.from_numpy(full_jv), out, r)

out = sh.buffer(B)
for it in range(2):                       # twice: the exchange buffer is reused
    out.fill_(float("nan"))
    sh.step(evaluate_packed, out)
    res, jv = sh.gather(out)
    assert np.array_equal(res.numpy(), ores), "residual after the all-gather"
    assert np.array_equal(jv.numpy(), full_jv), "jacobian after the all-gather"
assert calls == ([rank] * 2 if sh.ranges[rank][1] > 0 else [])
assert sh.inplace == (os.environ.get("GELATO_AMD_ALLGATHER_INPLACE", "0") == "1")
# one collective per step, delivering (world-1)/world of the outputs up to padding to the largest share
if {empty_rank} < 0:
    assert sh.bytes_received_per_vector() <= 8 * (E.nres + E.V) * (world - 1) / world * 1.6

# ---- replica mode: vectors split across ranks, slowest rank defines the time ----
B = 5
lo, hi = parallel.replica_range(B, rank, world)
X = np.tile(x, (B, 1)) * (1 + 1e-7 * np.arange(B))[:, None]
mine, _ = P.eval_batch(X[lo:hi])
gathered = [None] * world
dist.all_gather_object(gathered, (lo, hi, mine))
allres = np.concatenate([m for _, _, m in sorted(gathered, key=lambda t: t[0])])
ref, _ = P.eval_batch(X)
assert np.array_equal(allres, ref)
tmax = parallel.max_over_ranks(1.0 + rank)
assert tmax == float(world)
dist.destroy_process_group()
print("WORKER_OK", rank)
"""


def run_gloo_world(tmp_path, world, empty_rank=-1, inplace=False):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.replace("{empty_rank}", str(empty_rank)).replace("{root!r}", repr(ROOT)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1", GELATO_AMD_ALLGATHER_INPLACE="1" if inplace else "0")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "WORKER_OK %d" % r in out, out[-3000:]


@pytest.mark.parametrize("inplace", [False, True])
def test_world_size_2_gloo(tmp_path, inplace):
    """inplace = False: the default since round 6 (the send buffer is a copy of the rank's slice); True: the aliased form
    (GELATO_AMD_ALLGATHER_INPLACE=1), opt-in until it has run on RCCL with more than one rank"""
    run_gloo_world(tmp_path, 2, inplace=inplace)


def test_world_size_4_gloo_with_a_rank_without_units(tmp_path):
    run_gloo_world(tmp_path, 4, empty_rank=2)


@pytest.mark.parametrize("seed", range(12))
def test_random_problem_structures_pattern_and_constants(seed):
    """Random phase structures (1..8 phases, 2..70 nodes, every combination of aerodynamic / NoAir, engine on / off,
    free / held attitude): sparsity pattern, block shapes and every constant entry of the host-only engine equal the
    oracle's, and the compact map covers exactly the entries that move with x."""
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(1, 9))
    g = load_golden("g6_example.npz")
    prob = dict(problem_from_golden(g))
    nn = rng.integers(2, 24, S)
    if seed % 4 == 0:
        nn[rng.integers(0, S)] = int(rng.integers(64, 71))       # a multi-chunk phase with a ragged tail
    prob["num_nodes"] = nn.astype(np.int32)
    on = rng.integers(0, 2, S)
    prob["engine_on"] = on.astype(np.int32)
    prob["thrust"] = np.where(on, rng.uniform(1e4, 5e5, S), 0.0)
    prob["massflow"] = np.where(on, rng.uniform(1.0, 150.0, S), 0.0)
    prob["reference_area"] = np.where(rng.integers(0, 2, S), rng.uniform(0.5, 3.0, S), 0.0)
    prob["nozzle_area"] = np.where(on, rng.uniform(0.0, 1.0, S), 0.0)
    prob["attitude_hold"] = rng.integers(0, 2, S).astype(np.int32)
    P = oracle.Problem(prob)
    E = host_engine(prob, [P.D(i) for i in range(S)], [P.tau(i) for i in range(S)])
    N = int(nn.sum())
    M = N + S
    assert (E.N, E.M, E.nvars) == (N, M, 11 * M + 2 * N + S + 1)
    x1 = rng.random(E.nvars) + 0.5
    x2 = rng.random(E.nvars) + 0.5
    for x in (x1, x2):                                          # keep positions above ground, times increasing
        x[M:4 * M] = 0.6 + 0.05 * x[M:4 * M]
        x[-(S + 1):] = np.sort(x[-(S + 1):])
    pat, cv, vidx = E.pattern(), E.const_values(), E.var_index()
    is_var = E.var_mask()
    b = 0
    for grp in oracle.GROUPS:
        J1, J2 = P.jacobian(grp, x1), P.jacobian(grp, x2)
        for var in oracle.BLOCK_VARS[grp]:
            r, c = pat[b]
            ro, co, v1 = J1[var]["coo"]
            v2 = J2[var]["coo"][2]
            assert np.array_equal(r, ro) and np.array_equal(c, co), (grp, var)
            assert E.block_shape[b] == J1[var]["shape"]
            lo, hi = E.block_off[b], E.block_off[b + 1]
            m = is_var[lo:hi]
            assert np.array_equal(cv[lo:hi][~m], v1[~m]) and np.array_equal(v1[~m], v2[~m]), (grp, var)  # constants
            # an entry outside the compact map never moves with x (the converse need not hold: an x-dependent
            # entry may be 0 at both points, e.g. d/dquaternion of a zero-thrust phase)
            b += 1
    assert b == 13 and len(np.unique(vidx)) == E.V


# --------------------------------------------------------------------------
# bench.py --gpus N: the launcher (N ranks as a child torch.distributed.run, before anything touches HIP)
# --------------------------------------------------------------------------
def _bench(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout)


def test_bench_launcher_command_is_formed():
    import json
    pr = _bench(["--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launcher"])
    assert pr.returncode == 0, pr.stderr
    cmd = json.loads(pr.stdout)["launcher"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5"]     # the ranks get the caller's own flags


def test_bench_gpus_2_without_gpus_fails_loudly_instead_of_running_one_rank():
    pr = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert pr.returncode != 0
    assert '"metric"' not in pr.stdout                      # no line, in particular no n_gpus: 1 line
    assert "2-rank child run failed" in pr.stderr


def test_bench_rank_count_must_agree_with_gpus_flag():
    pr = _bench(["--gpus", "8"], env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert pr.returncode != 0 and "must agree" in pr.stderr and '"metric"' not in pr.stdout
