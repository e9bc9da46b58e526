#!/usr/bin/env python3
"""bench.py -- defect-residual + FD-Jacobian evaluations per second on the 6-phase x 64-node LGR mesh.

A "step" is one pass of the hot path over one batch of B synthetic decision vectors per GPU:
one fused launch that writes the four defect residuals and every x-dependent Jacobian value of
each vector (1 eval = the hot-path share of one objfunc + one sens call).  Inputs are
resident in HBM when the timed region starts; outputs stay in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the path shards by independent decision vectors (replicas of the static problem, B
vectors per rank, no data-path collective) -> "scaling": "weak".  Rank 0 prints ONE JSON line, with a per-rank `roofline`
and the `cpu_baseline` leg (rank 0's host cores) at every N.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_FLOP_PER_CYCLE = 256 * 4 * 16 * 2   # CUs x SIMDs x fp64 FMA lanes x 2 (half the fp32 vector rate of that guide: 157.3 TF at 2.4 GHz)
FP64_SPEC_CLOCK_GHZ = 2.4


def launcher_command(n_gpus, argv, port=None):
    """The command that runs this script as `n_gpus` ranks, one per GPU, over RCCL: what the driver itself uses for N > 1
    (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`)."""
    if port is None:
        import socket
        with socket.socket() as sk:   # a free port of this host; nothing here touches the GPU
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n_gpus, argv, dry=False):
    """`bench.py --gpus N` typed without torch.distributed.run: start the N ranks as a CHILD process (never an exec:
    this process has not touched HIP yet and must not, the child ranks own the GPUs), relay rank 0's JSON line and
    the exit code.  Fails loudly -- it never falls back to one rank."""
    cmd = launcher_command(n_gpus, argv)
    if dry:
        print(json.dumps({"launcher": cmd}))
        return 0
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    pr = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if pr.returncode != 0 or len(lines) != 1:
        sys.stdout.write(pr.stdout)
        raise SystemExit("bench.py --gpus %d: the %d-rank child run failed (exit code %d, %d result lines)"
                         % (n_gpus, n_gpus, pr.returncode, len(lines)))
    line = json.loads(lines[0])
    if line.get("n_gpus") != n_gpus:
        raise SystemExit("bench.py --gpus %d: the child run reported n_gpus = %r" % (n_gpus, line.get("n_gpus")))
    print(lines[0])
    return 0


def cpu_model():
    """model name of the host's CPU (/proc/cpuinfo), for the cpu_baseline leg"""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(prob, D, tau, X, gpu_first=None, budget_s=12.0, residual_only=False):
    """The oracle (scalar C port of the reference algorithm) timed on this box's host cores, 1 thread,
    on a bounded sample of the same workload.  This leg is the only place bench.py touches oracle/: besides
    the timing it checks element 0 of what the GPU just produced (`gpu_first` = (res, full values)) against
    it, after the timed region.  residual_only: the four residual functions alone (BASELINE configs[1])."""
    import oracle
    P = oracle.Problem(prob, D=D, tau=tau)

    def run(Xs):
        if residual_only:
            for x in Xs:
                for g in oracle.GROUPS:
                    P.residual(g, x)
        else:
            P.eval_batch(Xs, nthreads=1, keep_vals=False)

    t0 = time.perf_counter()
    run(X[:2])
    per = (time.perf_counter() - t0) / 2
    # per-eval distribution first (>= 200 evals, median + p10/p90), one eval per call
    nd = int(max(8, min(len(X), 256, 0.25 * budget_s / max(per, 1e-6))))
    ts = np.empty(nd)
    for k in range(nd):
        t0 = time.perf_counter()
        run(X[k:k + 1])
        ts[k] = time.perf_counter() - t0
    n = int(max(4, min(len(X), 0.75 * budget_s / max(per, 1e-6))))
    t0 = time.perf_counter()
    run(X[:n])
    dt = time.perf_counter() - t0
    what = "4 residuals" if residual_only else "4 residuals + 4 COO Jacobians each, every COO value computed"
    out = {"value": n / dt, "unit": "evals/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "host_cores_visible": len(os.sched_getaffinity(0)),
           "sample": "%d evals (%s) of the same workload, oracle/libgelato_oracle.so, 1 thread, %.1f s; "
                     "per-eval distribution over %d single-eval calls" % (n, what, dt, nd),
           "ms_per_eval": 1e3 * dt / n,
           "ms_per_eval_median": 1e3 * float(np.median(ts)), "ms_per_eval_p10": 1e3 * float(np.percentile(ts, 10)),
           "ms_per_eval_p90": 1e3 * float(np.percentile(ts, 90)), "distribution_evals": nd}
    if residual_only:
        if gpu_first is not None:
            ores = np.concatenate([P.residual(g, X[0]) for g in oracle.GROUPS])
            out["parity_spot_check"] = {"residual_max_abs_diff": float(np.max(np.abs(gpu_first[0] - ores)))}
        return out
    try:  # informational: many host cores = independent single-thread worker PROCESSES over slices of the
        # sample (threads of one process do not run concurrently in this pool's sandbox: measured).  The
        # workers import only oracle/ and numpy -- nothing that touches the GPU -- and start from a fresh
        # interpreter (no fork of this GPU-initialised process).
        import subprocess
        import tempfile
        nc = len(os.sched_getaffinity(0))
        W = min(nc, 32)
        if W > 1:
            per_w = max(8, int(3.0 / max(dt / n, 1e-6)))        # ~3 s of work per worker
            with tempfile.TemporaryDirectory() as td:
                np.savez(os.path.join(td, "job.npz"), X=X[:min(len(X), 64)], D=np.concatenate([d.ravel() for d in D]),
                         tau=np.concatenate(tau), **{"p_" + k: np.asarray(v) for k, v in prob.items()})
                code = ("import sys, time, numpy as np; sys.path.insert(0, %r); import oracle\n"
                        "j = np.load(sys.argv[1]); prob = {k[2:]: j[k] for k in j.files if k.startswith('p_')}\n"
                        "nn = prob['num_nodes']; D = []; tau = []; o = 0; q = 0\n"
                        "for n in nn:\n"
                        "    D.append(j['D'][o:o + n * (n + 1)].reshape(n, n + 1)); o += n * (n + 1)\n"
                        "    tau.append(j['tau'][q:q + n]); q += n\n"
                        "P = oracle.Problem(prob, D=D, tau=tau); X = np.tile(j['X'], (int(sys.argv[2]) // len(j['X']) + 1, 1))[:int(sys.argv[2])]\n"
                        "P.eval_batch(X[:2], nthreads=1, keep_vals=False); print('READY', flush=True); sys.stdin.readline()\n"
                        "t0 = time.perf_counter(); P.eval_batch(X, nthreads=1, keep_vals=False); print(time.perf_counter() - t0, flush=True)\n"
                        % ROOT)
                env = dict(os.environ, OMP_NUM_THREADS="1")
                procs = [subprocess.Popen([sys.executable, "-c", code, os.path.join(td, "job.npz"), str(per_w)], env=env,
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True) for _ in range(W)]
                for pr in procs:                                  # all workers built their problem: start together
                    assert pr.stdout.readline().strip() == "READY"
                t0 = time.perf_counter()
                for pr in procs:
                    pr.stdin.write("go\n"); pr.stdin.flush()
                for pr in procs:
                    float(pr.stdout.readline())
                wall = time.perf_counter() - t0
                for pr in procs:
                    pr.wait(timeout=30)
            out["all_cores"] = {"value": W * per_w / wall, "cores": W, "kind": "port, %d single-thread worker processes" % W,
                                # SURVEY 8(d) names OpenMP over phases x sweeps: threads of one process do not run concurrently in this
                                # pool's sandbox (measured, round 3), so the many-core leg is W independent processes over slices
                                "form": "%d processes x 1 thread (not OpenMP threads of one process)" % W,
                                "sample": "%d worker processes x %d evals, 1 thread each, %.1f s" % (W, per_w, wall)}
    except Exception as ex:  # noqa: BLE001
        out["all_cores"] = {"error": str(ex)[:200]}
    if gpu_first is not None:
        ores, ovals = P.eval_batch(X[:1])
        out["parity_spot_check"] = {"residual_max_abs_diff": float(np.max(np.abs(gpu_first[0] - ores[0]))),
                                    "jacobian_max_abs_diff": float(np.max(np.abs(gpu_first[1] - ovals[0])))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="decision vectors per GPU per step.  65536 vectors = 96 rounds of the 4096 "
                    "wavefronts the chip holds (at 4096 vectors the ramp and tail of a launch cost a quarter) and, at 6 x 64, "
                    "13.8 GB of inputs and outputs resident in HBM: sized for the 288 GB of the part, and a step (3.4 ms) long "
                    "enough that W = 5 warm-up steps reach the chip's steady power state (at 16384 vectors the K timed steps of "
                    "a short run sit inside the start-up transient: 17.1 M evals/s against 18.8 M settled)")
    ap.add_argument("--settle-ms", type=float, default=250.0, dest="settle_ms",
                    help="untimed launches before the warm-up steps until the power state has settled (0 = none)")
    ap.add_argument("--workload", default="mixed-6x64", help="mixed-6x64 | dense-6x64 | 3x32 | stress-12x128 | example")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "phase-shard"],
                    help="replicas: B vectors per GPU, no collective (weak scaling, the headline). phase-shard: ONE "
                         "batch of B vectors evaluated by all GPUs together, units (work item, FD column part) dealt to "
                         "ranks, one RCCL all-gather of the owned entries per step (strong scaling; BASELINE.json config 4)")
    ap.add_argument("--residual-only", action="store_true", dest="residual_only",
                    help="RHS + defect residuals only, no Jacobian (BASELINE.json configs[1]: 3x32 residual only vs CPU)")
    ap.add_argument("--flags", type=int, default=0,
                    help="engine creation flags (include/gelato_amd.h): 8 = GEL_FLAG_FD_RECOMPUTE, the reference-literal form that "
                         "re-runs the RHS chain on every perturbed column (lib/con_dynamics.py:353-480,580-604); 1 / 2 force D.X onto "
                         "the matrix pipe / the vector unit; 4 = never two vectors per wavefront")
    ap.add_argument("--placement-tries", type=int, default=8, dest="placement_tries",
                    help="candidate placements of the resident batch buffers measured before anything is timed (1: take what the allocator gives)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs of the other BASELINE.json configurations (key `other_configs`; --no-extras skips them too)")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational full-COO and B=1 legs")
    ap.add_argument("--dry-launcher", action="store_true", help="print the N-rank launcher command and exit (no GPU touched)")
    a = ap.parse_args()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    # ---- N ranks: before torch is imported and before anything touches HIP ----
    world_env = os.environ.get("WORLD_SIZE")
    if "RANK" not in os.environ:
        if a.gpus > 1 or a.dry_launcher:
            argv = [v for v in sys.argv[1:] if v != "--dry-launcher"]
            raise SystemExit(launch_ranks(a.gpus, argv, dry=a.dry_launcher))
    elif int(world_env or "1") != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE = %s: the rank count and --gpus must agree "
                         "(the line reports n_gpus = WORLD_SIZE and nothing else)" % (a.gpus, world_env))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # under torch.distributed.run (RANK set) the process group is always created, also for one rank, so
    # that the barrier / max-over-ranks path is the same code at every N
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
        # the communicator is set up lazily by the first collective: here, not by the barrier in front of the timed region (which would
        # leave the GPU idle for the hundreds of milliseconds of that set-up right before the K timed steps)
        dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    from gelato_amd import Engine, con_dynamics, pack_x, problem

    pdict, unitdict, condition, xdict = problem.make_problem(a.workload)
    prob = con_dynamics.problem_arrays(pdict, unitdict)
    S = pdict["num_sections"]
    ps = pdict["ps_params"]
    D = [ps.D(i) for i in range(S)]
    tau = [ps.tau(i) for i in range(S)]
    E = Engine(prob, D=D, tau=tau, device=local, flags=a.flags)
    B, K, W = a.batch, a.steps, a.warmup

    x0 = pack_x(xdict)
    shard = a.mode == "phase-shard" and world > 1
    X = problem.synthetic_batch(x0, E.M, B, seed=20260313 + (0 if shard else rank * B))
    stream = torch.cuda.current_stream().cuda_stream  # the engine launches on torch's current stream
    dX = torch.from_numpy(X).to(dev)
    # [r6] Three timed regions: from an idle GPU (`value_cold`), steady state on the buffers as the allocator places them
    # (`value_default_placement`), then -- gelato_amd/placement.py: candidate allocations measured, the fastest kept -- steady state on
    # placed buffers (`value`).  Rounds 4-5 placed first and reported one figure; now every launch in front of `value` is counted
    # (`warmup_steps_run`) and the allocator's figure stands beside it.
    placement = None
    dres = torch.empty((B, E.nres), dtype=torch.float64, device=dev)
    djv = None if a.residual_only else torch.empty((B, E.V), dtype=torch.float64, device=dev)
    djv_ptr = 0 if djv is None else djv.data_ptr()

    if shard:
        from gelato_amd import parallel
        # units = (work item, part): the position-sweep columns of the FD Jacobian are dealt to ranks too, so
        # that 8 GPUs have something to do on a 6-phase mesh (BASELINE.json configs[3]).  Every output entry has one
        # owning unit; a rank's kernel writes the entries of its units straight into its slice of ONE exchange buffer
        # out [world][B][width], and ONE in-place all-gather completes it on every rank: no pack / unpack launches, no
        # fills, (N-1)/N of the outputs received per rank.
        shards = parallel.UnitShards(E, world, rank)
        dout = shards.buffer(B, dev)

        def evaluate(out_t, r):
            E.eval_shard_packed_device(B, dX.data_ptr(), out_t.data_ptr(), r, stream, plan=shards.plan)

        def step():
            shards.step(evaluate, dout)

        def kernel_only():
            if shards.ranges[rank][1] > 0:
                evaluate(dout, rank)
    else:
        def step():
            E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), 0 if djv is None else djv.data_ptr(), stream)
        kernel_only = step

    def barrier():
        if use_dist:
            dist.barrier(device_ids=[local])

    def timed(K):
        """barrier + synchronize, K steps, synchronize + barrier -> (wall seconds, HIP-event ms per step)"""
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        for k in range(K):
            step()
        ev1.record()
        torch.cuda.synchronize()
        barrier()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) / K

    # (1) `value_cold`: W untimed warm-up steps, then K timed steps.  From an idle GPU these sit inside the
    # chip's start-up power transient (per-launch durations from an idle GPU: first launch fast, a dip ~3 ms later, steady after ~40 ms).
    # Warm-up: the W untimed steps of the contract -- and, when a step is short, more of them until WARM_MS of launches have
    # gone by (same count on every rank): 5 steps of 0.3 ms end inside the start-up dip of the clock, and the K timed steps
    # would measure the transient, not the kernel (3 x 32 residual-only: 167 M evals/s against 204 M).
    WARM_MS = 40.0
    # every launch of this run is under HIP events, group by group: their mean is what a kernel trace of the whole run averages
    # (`roofline.kernel_ms_mean_of_all_launches`, beside the timed region's `kernel_ms`)
    all_ms, all_n = 0.0, 0

    def untimed(n):
        nonlocal all_ms, all_n
        if n <= 0:
            return
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for _ in range(n):
            step()
        g1.record()
        torch.cuda.synchronize()
        all_ms += g0.elapsed_time(g1)
        all_n += n
    untimed(W)
    # The step's length is measured on ONE more untimed step, after the W steps have absorbed the one-time costs (module load,
    # first launch, the lazy RCCL communicator of the first all-gather): timed over the W steps themselves those costs made
    # a 0.3-ms step look like tens of milliseconds and the extra warm-up came out as zero.
    w_probe = 0
    per_step_ms = WARM_MS
    if W > 0:
        t0 = time.perf_counter()
        untimed(1)
        per_step_ms = 1e3 * (time.perf_counter() - t0)
        w_probe = 1
    w_extra = 0 if W == 0 else max(0, min(4000, int(WARM_MS / max(per_step_ms, 1e-3)) - W - w_probe))
    if use_dist:
        we = torch.tensor([w_extra], dtype=torch.int64, device=dev)
        dist.all_reduce(we, op=dist.ReduceOp.MAX)
        w_extra = int(we.item())
    untimed(w_extra)
    elapsed, kern_ms = timed(K)
    all_ms += kern_ms * K
    all_n += K
    # (2) `value`: the same K steps after ~0.25 s of untimed launches (same count on every rank) and W warm-up steps again -- the
    # steady state a batch workload lives in.
    n_settle = 0 if a.settle_ms <= 0 else min(5000, max(4, int(a.settle_ms / 1.0 * 16384 / B)))
    untimed(n_settle)
    untimed(W)
    elapsed_settled, kern_ms_settled = timed(K)
    all_ms += kern_ms_settled * K
    all_n += K
    # [r6] `value` is the STEADY-STATE figure (the second timed region), `value_cold` the first one.  The K timed steps right behind W
    # warm-up steps from an idle GPU sit inside the chip's start-up power / clock transient (19.1 against 20.8 M evals/s on one box);
    # rounds 4-5 reported that region as `value` -- and in round 5 some 300 placement launches had run in front of it, which made it
    # the steady-state figure in all but name.  Now the line says what it is: `warmup` = W as asked for,
    # `warmup_steps_run` = every launch in front of the timed region of `value`, `value_cold` = the contract read literally.
    warm_total = W + w_probe + w_extra + K + n_settle + W
    elapsed_cold, kern_ms_cold = elapsed, kern_ms
    elapsed, kern_ms = elapsed_settled, kern_ms_settled
    status = E.sync(stream)
    # [r6] buffer placement, after the contract's timed regions: the same K steps on the fastest of `--placement-tries` candidate
    # placements of jvar and of (x, res), W warm-up steps in front (informational; every rank does the same)
    elapsed_placed, kern_ms_placed = float("nan"), float("nan")
    do_place = a.placement_tries > 1 and not shard
    if do_place:
        # free memory first: a rank that cannot hold a second set of buffers beside the first makes EVERY rank skip
        # the placement, so that the barriers of the timed region behind it stay matched
        free_b, _tot = torch.cuda.mem_get_info(dev)
        need = 2 * 8 * B * (E.nvars + E.nres + (0 if a.residual_only else E.V)) + (5 << 30)   # two more sets (kept + candidate) and the pads
        ok_t = torch.tensor([1 if free_b >= need else 0], dtype=torch.int64, device=dev)
        if use_dist:
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        if int(ok_t.item()) == 0:
            do_place = False
            placement = {"error": "placement skipped: %.1f GB free on rank %d, %.1f GB wanted for the candidate buffers" % (free_b / 1e9, rank, need / 1e9)}
    if do_place:
        from gelato_amd.placement import place_batch_buffers
        placed_ok = 1
        try:
            dX, dres, djv, placement = place_batch_buffers(E, dX, want_jac=not a.residual_only, tries=a.placement_tries, stream=stream,
                                                          seed=rank)
            all_ms += placement["all_launches_ms"]
            all_n += placement["all_launches"]
        except Exception as ex:  # noqa: BLE001: what the allocator gave stays
            torch.cuda.empty_cache()
            placement = {"error": str(ex)[:200]}
            placed_ok = 0
        if use_dist:   # every rank times the placed steps, or none does (the timed region holds barriers)
            ok_t = torch.tensor([placed_ok], dtype=torch.int64, device=dev)
            dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
            placed_ok = int(ok_t.item())
        if placed_ok:
            untimed(W)
            elapsed_placed, kern_ms_placed = timed(K)
            all_ms += kern_ms_placed * K
            all_n += K
        elif placement is not None and "error" not in placement:
            placement["error"] = "another rank could not place its buffers: the placed steps were not timed"
    djv_ptr = 0 if djv is None else djv.data_ptr()

    tmax = torch.tensor([elapsed, elapsed_cold, elapsed_placed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    T, T_cold, T_placed = float(tmax[0].item()), float(tmax[1].item()), float(tmax[2].item())
    # [r6] WHICH region is `value`.  On the allocator's placement the steady-state figure is one of two levels by the physical pages
    # the process happened to get (20.4 / 20.5 / 21.0 / 21.9 M evals/s in four calls, profiles/r06/boxes.json) -- and over N ranks the
    # slowest rank's level, i.e. almost always the slow one: a coin in the headline and a false scaling loss.  On placed buffers it
    # is 21.7-21.8 M every time.  So `value` is the PLACED steady-state figure when the placement ran (the K timed steps behind
    # `warmup_steps_run` launches, every one of them counted), `value_default_placement` the allocator's, `value_cold` the literal
    # W + K region from an idle GPU: all three in the line.  --placement-tries 1: `value` = `value_default_placement`.
    T_default, kern_ms_default = T, kern_ms
    placed_used = T_placed == T_placed      # not NaN: every rank placed its buffers and timed the K steps on them
    if placed_used:
        T, kern_ms = T_placed, kern_ms_placed
        warm_total += K + (placement.get("all_launches", 0) if isinstance(placement, dict) else 0) + W

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # element 0 of what was just timed, kept for the oracle check inside the cpu_baseline leg
    gpu_first = None
    if not a.no_cpu_baseline:
        if shard:
            r_t, j_t = shards.gather(dout[:, :1])
            gpu_first = (r_t[0].cpu().numpy(), E.expand(j_t[0].cpu().numpy()))
        else:
            gpu_first = (dres[0].cpu().numpy(), None if djv is None else E.expand(djv[0].cpu().numpy()))

    evals = (1 if shard else world) * B * K
    # per launch: SURVEY.md 8(d) A_min x evals per launch (residual only: read x once, write the residual once)
    a_min = 8 * (E.nvars + E.nres) if a.residual_only else E.algorithmic_bytes
    abytes = a_min * B
    achieved = abytes / (kern_ms * 1e-3) / 1e9
    wl_tag = a.workload + ("_resonly" if a.residual_only else "") + ("_flags%d" % a.flags if a.flags else "")
    from gelato_amd import _lib
    build = _lib.build_info()

    def static_counters(kind, tag=None, batch=None, engine=None, jac=None):
        """profiles/<kind>_<workload>_B<batch>.json (separate rocprofv3 --pmc passes of this command, tools/gpu_record.sh) --
        only if it was recorded with THE library that is loaded now (build_so_sha256); -> (dict or None, why not)"""
        path = os.path.join(ROOT, "profiles", "%s_%s_B%d.json" % (kind, tag or wl_tag, batch or B))
        if not os.path.exists(path):
            return None, "no profiles/%s on record for this workload and batch" % os.path.basename(path)
        try:
            d = json.load(open(path))
        except Exception as ex:  # noqa: BLE001
            return None, "unreadable: %s" % ex
        same_device_code = d.get("build_device_code_sha256") is not None and d.get("build_device_code_sha256") == build.get("device_code_sha256")
        if d.get("build_so_sha256") != build["so_sha256"] and not same_device_code:      # counters describe the DEVICE code
            return None, ("profiles/%s describes another build (so_sha256 %s..., git %s; loaded: %s...): not reported"
                          % (os.path.basename(path), str(d.get("build_so_sha256"))[:12], str(d.get("build_git_head"))[:10], build["so_sha256"][:12]))
        # the launch policy the counters were taken under (instantiation, wavefront count, work items) must be the one this library
        # picks now: the same device code launched differently moves different bytes
        pol = d.get("launch_policy")
        if isinstance(pol, dict) and "launch_info" in pol:
            eng = engine or E
            now = {"launch_info": eng.launch_info(batch or B, True, (not a.residual_only) if jac is None else jac), "num_chunks": eng.num_chunks()}
            if pol != now:
                return None, "profiles/%s was recorded under another launch policy (%r, now %r): not reported" % (os.path.basename(path), pol, now)
        d["_file"] = os.path.basename(path)
        return d, None

    tdata, traffic_why = (None, "phase-shard mode: no counters on record") if shard else static_counters("traffic")
    traffic = None if tdata is None else tdata.get("hbm_bytes_per_launch")
    info = E.launch_info(B, True, not a.residual_only)   # which instantiation the launcher picked
    kname = "gel::eval_kernel<%s, %s, %s, %s>" % tuple("true" if v else "false" for v in (info[0], info[1], info[2], info[4]))
    if shard:   # a unit range always runs the split (latency) form, one decision vector per wavefront
        kname = "gel::eval_kernel<true, %s, true, false>" % ("true" if info[1] else "false")
    out = {
        "metric": "residual+Jacobian evals/sec (and ms/eval), 6-phase x 64-node LGR mesh",
        "value": evals / T, "unit": "evals/s", "n_gpus": world, "steps": K, "warmup": W,
        "world_size": world, "collective_backend": ("nccl (RCCL over xGMI)" if use_dist else None),
        "warmup_steps_run": warm_total,
        "warmup_steps_run_detail": {"cold_region_warmup": W + w_probe + w_extra, "cold_region_timed": K, "settling_launches": n_settle,
                                    "default_placement_region": (W + K) if placed_used else W,
                                    "placement_candidate_launches": (placement.get("all_launches", 0) if (placed_used and isinstance(placement, dict)) else 0),
                                    "warmup_before_value": W},
        "value_is": ("steady state on placed buffers (gelato_amd/placement.py)" if placed_used else "steady state on the allocator's buffer placement"),
        "value_default_placement": evals / T_default, "ms_per_step_default_placement": 1e3 * T_default / K,
        "ms_per_step": 1e3 * T / K, "ms_per_eval": 1e3 * T / (B * K), "higher_is_better": True,
        "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        # `value`: K timed steps in the settled power state (warmup_steps_run launches in front of them, W of them directly), on the
        # placed buffers when the placement ran; `value_default_placement`: the same on the buffers as the allocator placed them;
        # `value_cold`: the K timed steps right behind W (+ time-based, >= 40 ms) warm-up steps from an idle GPU -- inside the
        # start-up power transient.  value_settled = value (the key of rounds 3-5, kept for comparisons)
        "value_cold": evals / T_cold, "ms_per_step_cold": 1e3 * T_cold / K,
        "value_settled": evals / T, "ms_per_step_settled": 1e3 * T / K,
        "config": {"workload": a.workload + (" (residual only)" if a.residual_only else ""), "engine_flags": int(a.flags), "phases": int(S),
                   "nodes_per_phase": [int(n) for n in prob["num_nodes"]],
                   "batch_per_gpu": B, "settle_launches_before_second_timing": n_settle, "decision_vars": E.nvars, "residual_rows": E.nres,
                   "jacobian_values_per_eval": 0 if a.residual_only else E.V, "coo_nnz": E.total_nnz,
                   "parallelism": ("phase+column shards x%d + all-gather" if shard else "replicas x%d") % world,
                   "output": ("4 defect residuals, in HBM" if a.residual_only else
                              "4 defect residuals + all x-dependent COO Jacobian values (compact), in HBM")},
        "status": int(status),
        # AFTER the timed regions of `value_cold` / `value_default_placement` [r6]: candidate placements of the resident x / res /
        # jvar buffers (a few launches each), the fastest kept (gelato_amd/placement.py), then W warm-up steps and the same K steps on
        # them: `value` (= `value_placed` here).  null: --placement-tries 1 / phase-shard mode
        "buffer_placement": placement,
    }
    if placement is not None and "error" not in placement:
        placement["value_placed"] = evals / T_placed
        placement["ms_per_step_placed"] = 1e3 * T_placed / K
        placement["kernel_ms_placed"] = kern_ms_placed
        placement["launches_before_value_placed"] = placement.get("all_launches", 0) + W
        placement["value_placed_over_default"] = T_default / T_placed
    out["build"] = build
    if not shard:
        # one launch = B evals on this rank; HIP events on the launch stream over the K timed launches
        stored = 8 * E.nres if a.residual_only else E.stored_bytes
        fp64 = None
        f, fp64_why = static_counters("fp64")
        if f is not None:   # fp64 instruction counters + datapath occupancy of this command (tools/gpu_record.sh)
            peak_tf = FP64_FLOP_PER_CYCLE * FP64_SPEC_CLOCK_GHZ / 1e3
            fp64 = {"flops_per_launch": f["fp64_flops_per_launch"], "achieved": f["fp64_flops_per_launch"] / (kern_ms * 1e-3) / 1e12,
                    "peak": peak_tf, "unit": "TFLOP/s", "frac": f["fp64_flops_per_launch"] / (kern_ms * 1e-3) / 1e12 / peak_tf,
                    "peak_note": "256 CUs x 4 SIMDs x 16 fp64 FMA lanes x 2 at the 2.4 GHz spec clock; vector fp64 and "
                                 "v_mfma_f64 share that datapath (DESIGN.md 3.1)",
                    # the datapath's occupancy in TIME (profiled pass): fp64 adds / multiplies, conversions, integer and move
                    # instructions occupy issue slots without counting two flops per lane, so this, not `frac`, says how close
                    # the kernel is to the pipe
                    "pipe_busy": f.get("fp64_pipe_busy"), "valu_busy": f.get("valu_busy"), "mfma_busy": f.get("mfma_busy"),
                    "wait_inst_share": f.get("wait_inst_share"),
                    "valu_instructions_per_wave": f.get("valu_instructions_per_wave"), "clock_ghz_profiled": f.get("clock_ghz"),
                    "source": "static: profiles/%s (rocprofv3 --pmc passes of this command with this build; not re-measured in this run)" % f["_file"]}
        hbm_real = None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        # which limit binds, from the data: the share of the HBM peak the bytes that REALLY moved reach, against the share of
        # time the fp64 datapath was occupied.  Without counters of THIS build: SURVEY 8(d)'s a-priori bound, and said so.
        bound, bound_source = "hbm", "SURVEY.md 8(d) (a priori): no counters of the loaded build on record"
        if fp64 is not None and fp64.get("pipe_busy") is not None and hbm_real is not None:
            bound = "mfma" if fp64["pipe_busy"] > hbm_real else "hbm"
            bound_source = "counters of this build: fp64.pipe_busy %.2f against hbm_frac_of_bytes_moved %.2f" % (fp64["pipe_busy"], hbm_real)
        out["roofline"] = {"bound": bound, "bound_source": bound_source, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                           "traffic_source": traffic_why if traffic is None else
                           "static: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command with this build, "
                           "(2*FETCH_SIZE + WRITE_SIZE)*1024; not re-measured in this run)" % tdata["_file"],
                           "kernel": kname, "kernel_ms": kern_ms,
                           # mean over EVERY launch of this kernel in this run up to here (placement candidates, warm-ups, both timed
                           # regions and the settling launches between them): what `rocprofv3 --kernel-trace --stats` of the run averages
                           "kernel_ms_mean_of_all_launches": all_ms / max(all_n, 1), "launches_so_far": all_n,
                           "frac_cold": abytes / (kern_ms_cold * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms_cold": kern_ms_cold,
                           # frac / kernel_ms belong to `value` (placed buffers when the placement ran); the allocator's placement beside them
                           "frac_default_placement": abytes / (kern_ms_default * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms_default_placement": kern_ms_default,
                           "frac_placed": (abytes / (kern_ms_placed * 1e-3) / 1e9 / HBM_PEAK_GBS) if kern_ms_placed == kern_ms_placed else None,
                           "kernel_ms_placed": kern_ms_placed if kern_ms_placed == kern_ms_placed else None,
                           "algorithmic_bytes_per_eval": a_min, "algorithmic_bytes_per_launch": abytes,
                           # what one eval actually writes (residual + the DISTINCT x-dependent values; the gather map
                           # restores negated / shared / structurally constant entries) -- `achieved` uses SURVEY 8(d)'s
                           # A_min as the contract prescribes, `traffic` shows the bytes that really moved
                           "stored_bytes_per_eval": stored, "hbm_frac_of_bytes_moved": hbm_real,
                           "fp64": fp64, "fp64_source": fp64_why,
                           "bound_note": "`achieved` / `peak` / `frac` are the HBM figures of SURVEY 8(d) (algorithmic bytes); `bound` names "
                                         "the limit the counters show nearer: 'mfma' = the fp64 datapath that v_mfma_f64 and vector fp64 "
                                         "share (fp64.pipe_busy) is busier than HBM is with the bytes that really move "
                                         "(hbm_frac_of_bytes_moved); see DESIGN.md 3.1"}
    else:
        # this rank's kernel alone (its unit range of all B vectors, split form, written into its slice of the exchange buffer):
        # HIP events over K launches without the collective
        for _ in range(3):
            kernel_only()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            kernel_only()
        e1.record()
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / K
        own = sum(shards.counts[rank])
        r_bytes = 8 * (E.nvars + own) * B     # this rank reads every vector once and writes the entries its units own once
        out["roofline"] = {"bound": "hbm", "bound_source": "SURVEY.md 8(d) (a priori); the split (latency) form at this batch is launch- and "
                           "latency-bound, see `shard`", "achieved": r_bytes / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": r_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "traffic_source": traffic_why,
                           "kernel": kname, "kernel_ms": k_ms, "rank": rank,
                           "algorithmic_bytes_per_launch": r_bytes, "owned_entries_per_vector": own}
        out["shard"] = {"step_ms": kern_ms, "step_ms_cold": kern_ms_cold, "kernel_ms": k_ms,
                        "exchange_ms": max(kern_ms - k_ms, 0.0), "pack_launches": 0, "unpack_launches": 0,
                        "units_per_rank": [c for _, c in shards.ranges], "slice_doubles_per_vector": shards.width,
                        "all_gather_bytes_received_per_rank_per_step": shards.bytes_received_per_vector() * B,
                        "note": "one step = this rank's unit range (split-form kernel writing straight into its slice of the exchange "
                                "buffer) + ONE in-place all-gather; a consumer reads the buffer through gel_shard_plan's map"}

    if not a.no_extras and not a.residual_only and not shard:
        # informational: materialise every COO value like the reference does (compact -> full expansion)
        try:
            Bf = min(B, 1024)
            dfull = torch.empty((Bf, E.total_nnz), dtype=torch.float64, device=dev)
            for _ in range(2):
                E.expand_full_device(Bf, djv.data_ptr(), dfull.data_ptr(), stream)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5):
                E.expand_full_device(Bf, djv.data_ptr(), dfull.data_ptr(), stream)
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / 5
            # the fused launch at THIS batch (not the big batch's per-vector share: a 1024-vector launch does not fill the chip as well)
            for _ in range(3):
                E.eval_batch_device(Bf, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), stream)
            s.record()
            for _ in range(10):
                E.eval_batch_device(Bf, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), stream)
            e.record()
            torch.cuda.synchronize()
            ms_fused = s.elapsed_time(e) / 10
            # update-in-place mode: the constants laid down once, then only the x-dependent entries per evaluation
            nvar_entries = int(np.count_nonzero(E.var_mask()))
            E.fill_full_device(Bf, dfull.data_ptr(), stream)
            for _ in range(3):
                E.update_full_device(Bf, djv.data_ptr(), dfull.data_ptr(), stream)
            s.record()
            for _ in range(10):
                E.update_full_device(Bf, djv.data_ptr(), dfull.data_ptr(), stream)
            e.record()
            torch.cuda.synchronize()
            ms_upd = s.elapsed_time(e) / 10
            s.record()
            for _ in range(10):
                E.eval_full_device(Bf, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), dfull.data_ptr(), stream)
            e.record()
            torch.cuda.synchronize()
            ms_both = s.elapsed_time(e) / 10
            out["full_coo_expand"] = {"batch": Bf, "kernel_ms": ms,
                                      "write_GBps": Bf * E.total_nnz * 8 / (ms * 1e-3) / 1e9,
                                      # expand_kernel: reads the compact values once, writes every COO value once
                                      "algorithmic_bytes_per_launch": 8 * (E.V + E.total_nnz) * Bf,
                                      "hbm_frac": 8 * (E.V + E.total_nnz) * Bf / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "fused_kernel_ms_at_this_batch": ms_fused,
                                      "evals_per_s_fused_plus_full_rewrite": Bf / ((ms_fused + ms) * 1e-3),
                                      # every COO value of every vector valid in HBM after each step: the buffer holds the constants,
                                      # gel_update_full_device writes the x-dependent entries (fused launch + update launch, back to back)
                                      "update_in_place": {"x_dependent_entries": nvar_entries, "update_kernel_ms": ms_upd,
                                                          "fused_plus_update_ms": ms_both,
                                                          "algorithmic_bytes_per_launch": 8 * (E.V + nvar_entries) * Bf},
                                      # key names as in rounds 1-4: ..._fused_plus_expand = fused launch + FULL rewrite of every COO
                                      # value (expand_kernel); the update-in-place mode of round 5 has its own key
                                      "evals_per_s_fused_plus_expand": Bf / ((ms_fused + ms) * 1e-3),
                                      "evals_per_s_fused_plus_update_in_place": Bf / (ms_both * 1e-3),
                                      "update_in_place_mode": "gel_fill_full_device once, gel_eval_full_device per step: fused launch + update of "
                                                              "the x-dependent entries (every COO value valid in HBM after each step)"}
            del dfull
        except Exception as ex:  # noqa: BLE001
            out["full_coo_expand"] = {"error": str(ex)}
        # informational: the B = 1 host-buffer callback path (launch-latency bound), PCIe inclusive
        # (a) into the engine's own pinned buffers -- what the Python mirrors of the reference's functions hand out (gel_pinned_buffers:
        # COO-direct kernel output, no host copy of the residual rows or of the all-x-dependent blocks); (b) into caller arrays
        pres, pvals = E.pinned_buffers()

        def b1_stats(call, n):
            """per-call wall times of n one-vector evaluations -> (median, mean, p90) in ms: a latency figure, so the median is the
            headline (one call in a few hundred meets a host hiccup of tens of milliseconds) with mean and p90 beside it"""
            for _ in range(10):
                call()
            ts = np.empty(n)
            for k in range(n):
                t0 = time.perf_counter()
                call()
                ts[k] = time.perf_counter() - t0
            return 1e3 * float(np.median(ts)), 1e3 * float(ts.mean()), 1e3 * float(np.percentile(ts, 90))

        med, mean, p90 = b1_stats(lambda: E.eval(x0, out=pvals, res_out=pres), 300)
        out["b1_pinned_buffers_ms_median"], out["b1_pinned_buffers_ms_mean"], out["b1_pinned_buffers_ms_p90"] = med, mean, p90
        vals = E.eval(x0)[1]
        med_c, mean_c, p90_c = b1_stats(lambda: E.eval(x0, out=vals), 50)
        # key as in rounds 1-4: the MEAN of 50 calls into the caller's numpy arrays
        out["b1_host_callback_ms"] = mean_c
        out["b1_host_callback_ms_caller_arrays_median"] = med_c
        out["b1_note"] = ("one residual + full-COO Jacobian evaluation of one decision vector through Engine.eval (ctypes included): "
                          "b1_host_callback_ms = mean of 50 calls into numpy arrays of the caller (the key's meaning since round 1); "
                          "b1_pinned_buffers_ms_* = 300 calls into the handle's pinned buffers (zero-copy, COO-direct), median / mean / p90")
        # informational: the batched host-buffer entry point (pageable caller buffers -> pinned staging -> H2D,
        # launch, D2H of residuals + compact Jacobian values): PCIe inclusive, never `value`
        Bh = min(B, 512)
        r_h, j_h, _ = E.eval_batch(X[:Bh])
        t0 = time.perf_counter()
        E.eval_batch(X[:Bh], out=(r_h, j_h))
        dt = time.perf_counter() - t0
        out["host_batch_pcie_inclusive"] = {"batch": Bh, "evals_per_s": Bh / dt, "ms": 1e3 * dt,
                                            "bytes_moved": Bh * E.algorithmic_bytes}
        try:   # the same with page-locked caller buffers (torch pin_memory): the copy engines use them directly, no staging copies
            Bp = min(B, 2048)
            xp = torch.from_numpy(X[:Bp]).pin_memory()
            rp = torch.empty((Bp, E.nres), dtype=torch.float64).pin_memory()
            jp = torch.empty((Bp, E.V), dtype=torch.float64).pin_memory()
            E.eval_batch(xp.numpy(), out=(rp.numpy(), jp.numpy()))
            dts = []
            for _ in range(3):
                t0 = time.perf_counter()
                E.eval_batch(xp.numpy(), out=(rp.numpy(), jp.numpy()))
                dts.append(time.perf_counter() - t0)
            dtp = min(dts)
            out["host_batch_pcie_inclusive"].update({"pinned_caller_buffers": {"batch": Bp, "evals_per_s": Bp / dtp, "ms": 1e3 * dtp,
                                                                                "ms_of_three_calls": [1e3 * v for v in dts],
                                                                                "GBps_over_pcie": Bp * 8 * (E.nvars + E.nres + E.V) / dtp / 1e9}})
            del xp, rp, jp
        except Exception as ex:  # noqa: BLE001
            out["host_batch_pcie_inclusive"]["pinned_caller_buffers"] = {"error": str(ex)[:200]}

        # informational: generic column-batched forward difference (lib/jac_fd.py): dense
        # d(eqcon_dyn_vel)/dx over all columns = num_vars + 1 residual evaluations in one launch, host arrays out
        try:
            E.jac_fd("vel", x0)
            t0 = time.perf_counter()
            Jd, _ = E.jac_fd("vel", x0)
            dtj = time.perf_counter() - t0
            out["jac_fd_generic"] = {"group": "vel", "rows": int(Jd.shape[0]), "columns": int(Jd.shape[1]),
                                     "ms": 1e3 * dtj, "residual_evals_per_s": (Jd.shape[1] + 1) / dtj}
            del Jd
        except Exception as ex:  # noqa: BLE001
            out["jac_fd_generic"] = {"error": str(ex)}
        # informational: the aero path constraints on every phase but the last, "all" nodes:
        # value + forward-difference gradient of the three kinds, host buffers in and out
        try:
            S_ = len(prob["num_nodes"])
            rows = 0
            for kind, lim in (("alpha", 0.2), ("q", 4.0e4), ("qalpha", 5.0e3)):
                E.aero_configure(kind, [(i, 1, lim) for i in range(S_ - 1)])
                rows += E.aero_dims(kind)[0]
            # all three kinds, values + gradients, ONE launch; output arrays reused like the callback path does
            aero_all = lambda Xa: E.eval_aero_all(Xa, reuse=True)   # noqa: E731
            aero_all(x0)
            t0 = time.perf_counter()
            for _ in range(50):
                aero_all(x0)
            b1 = (time.perf_counter() - t0) / 50
            Ba = min(B, 1024)
            aero_all(X[:Ba])
            t0 = time.perf_counter()
            aero_all(X[:Ba])
            dtb = time.perf_counter() - t0
            # inputs resident in HBM, outputs stay in HBM: the device-pointer entry point, HIP events on the launch stream
            dims = [E.aero_dims(k) for k in E.AERO_KINDS]
            dcon = [torch.empty((Ba, d[0]), dtype=torch.float64, device=dev) for d in dims]
            djac = [torch.empty((Ba, sum(d[1])), dtype=torch.float64, device=dev) for d in dims]
            cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]
            for _ in range(3):
                E.eval_aero_all_device(Ba, dX.data_ptr(), cp, jp, stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                E.eval_aero_all_device(Ba, dX.data_ptr(), cp, jp, stream)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            grads = sum(sum(d[1]) for d in dims)
            out["aero_constraints"] = {"rows": rows, "b1_ms_3_kinds": 1e3 * b1, "batch": Ba,
                                       # aero_kernel: reads x once, writes the rows and their gradient values once (compute-bound:
                                       # six runs of the atmosphere chain per constrained node; profiles/r03/others)
                                       "algorithmic_bytes_per_launch": 8 * (E.nvars + rows + grads) * Ba,
                                       "hbm_frac": 8 * (E.nvars + rows + grads) * Ba / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "host_buffers_vectors_per_s": Ba / dtb,
                                       "device_resident_vectors_per_s": Ba / (ms * 1e-3), "device_kernel_ms": ms}
            del dcon, djac
            # the same launch at the batch where the kernel's throughput is quoted (profiles/r05/others: B = 16384)
            Bl_ = min(B, 16384)
            if Bl_ > Ba:
                cand = []
                for t_ in range(max(1, min(a.placement_tries, 4))):      # output buffers placed like the headline's (gelato_amd/placement.py)
                    pad_ = torch.empty((1 + 37 * t_) << 22, dtype=torch.float64, device=dev) if t_ else None
                    dcon = [torch.empty((Bl_, d[0]), dtype=torch.float64, device=dev) for d in dims]
                    djac = [torch.empty((Bl_, sum(d[1])), dtype=torch.float64, device=dev) for d in dims]
                    del pad_
                    cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]
                    for _ in range(20):
                        E.eval_aero_all_device(Bl_, dX.data_ptr(), cp, jp, stream)
                    e0.record()
                    for _ in range(10):
                        E.eval_aero_all_device(Bl_, dX.data_ptr(), cp, jp, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    cand.append(e0.elapsed_time(e1) / 10)
                    del dcon, djac
                    torch.cuda.empty_cache()
                msl = cand[0]      # the allocator's own placement; the fastest of the candidates beside it (informational)
                out["aero_constraints"]["large_batch"] = {"batch": Bl_, "device_kernel_ms": msl, "device_resident_vectors_per_s": Bl_ / (msl * 1e-3),
                                                          "hbm_frac": 8 * (E.nvars + rows + grads) * Bl_ / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                          "hbm_frac_best_placement": 8 * (E.nvars + rows + grads) * Bl_ / (min(cand) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                          "buffer_placement_ms": [round(c, 4) for c in cand]}
        except Exception as ex:  # noqa: BLE001
            out["aero_constraints"] = {"error": str(ex)}
        # [r6] defect groups + aero rows of the SAME resident batch: one call whose fused launch writes the aero rows from the
        # lanes of the aerodynamic phases (gel_eval_batch_aero_device), against the two kernels one after the other
        try:
            width, _oc, _oj = E.aero_record_layout()
            n_ref = sum(len(v) for v in _oc.values()) + sum(len(v) for v in _oj.values())                     # the reference's values per vector
            n_stored = sum(int((v >= 0).sum()) for v in _oc.values()) + sum(int((v >= 0).sum()) for v in _oj.values())   # without the exact zeros
            dims = [E.aero_dims(k) for k in E.AERO_KINDS]
            daero = torch.empty((B, width), dtype=torch.float64, device=dev)
            dcon = [torch.empty((B, d[0]), dtype=torch.float64, device=dev) for d in dims]
            djac = [torch.empty((B, sum(d[1])), dtype=torch.float64, device=dev) for d in dims]
            cp, jp = [t.data_ptr() for t in dcon], [t.data_ptr() for t in djac]

            def fused_call():
                E.eval_batch_aero_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), daero.data_ptr(), stream)

            def two_calls():
                E.eval_batch_device(B, dX.data_ptr(), dres.data_ptr(), djv.data_ptr(), stream)
                E.eval_aero_all_device(B, dX.data_ptr(), cp, jp, stream)

            def time_of(fn, n=12, warm=12):
                for _ in range(warm):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / n
            ms_two = time_of(two_calls)
            ms_one = time_of(fused_call)
            ms_two2 = time_of(two_calls)
            ms_one2 = time_of(fused_call)
            ms_def = time_of(step)
            a_bytes = E.algorithmic_bytes + 8 * n_ref      # SURVEY 8(d)'s A_min of the defect path + every aero value of the reference once
            out["defect_plus_aero"] = {
                "batch": B, "aero_rows": rows, "aero_values_per_vector": n_ref, "aero_values_stored_per_vector": n_stored, "aero_record_doubles": width,
                "one_call_ms": min(ms_one, ms_one2), "two_kernels_ms": min(ms_two, ms_two2), "defect_alone_ms": ms_def,
                "ns_per_vector_one_call": 1e6 * min(ms_one, ms_one2) / B, "ns_per_vector_two_kernels": 1e6 * min(ms_two, ms_two2) / B,
                "ns_per_vector_defect_alone": 1e6 * ms_def / B,
                "ms_all_runs": {"one_call": [ms_one, ms_one2], "two_kernels": [ms_two, ms_two2]},
                # SURVEY 8(d)'s A_min of the defect path + the aero rows and gradient values written once (x is read once for both)
                "algorithmic_bytes_per_vector": a_bytes,
                "frac": a_bytes * B / (min(ms_one, ms_one2) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "gel_eval_batch_aero_device: ONE fused launch (defect groups + the aero rows of the aerodynamic phases' nodes 1..n from "
                        "the same chain, written as whole 64-byte lines into part A of a per-vector record) + a small launch for every other "
                        "row (state node 0 of every phase), against gel_eval_batch_device followed by gel_eval_aero_all_device (dense per-kind "
                        "arrays); 12 warm-up + 12 timed calls each, twice, alternating; HIP events; same values bit for bit "
                        "(tests/test_aero_engine.py); GEL_AERO_FUSED=0 in the environment: aero_kernel writes part A in a launch of its own"}
            del daero, dcon, djac
        except Exception as ex:  # noqa: BLE001
            out["defect_plus_aero"] = {"error": str(ex)[:300]}

    if not a.no_other_configs and not a.no_extras and not shard and not a.residual_only and a.workload == "mixed-6x64" and a.flags == 0:
        # The other BASELINE.json configurations in front of the driver, after the headline's timed region:
        # configs[1] 3 x 32 residual-only, configs[4] stress-12x128, dense-6x64 (maximum work), and the headline mesh in the
        # reference-literal form that re-runs the RHS chain on every perturbed column (GEL_FLAG_FD_RECOMPUTE).  Each leg: its own
        # engine, LEG_DISTINCT distinct synthetic vectors tiled to the batch on the device, >= 40 ms of untimed
        # launches, then LEG_K launches under HIP events.  `frac` = SURVEY 8(d)'s A_min x B / kernel time / 8 TB/s;
        # `frac_of_bytes_moved` from the PMC passes on record for the loaded build (null otherwise).
        LEG_DISTINCT, LEG_K = 256, 8
        legs = [("3x32_resonly", "3x32", True, 65536, 0), ("stress-12x128", "stress-12x128", False, 16384, 0),
                ("dense-6x64", "dense-6x64", False, 65536, 0), ("mixed-6x64_fd_recompute", "mixed-6x64", False, 65536, 8)]
        oc = {}
        t_legs = time.perf_counter()
        for key, wl, resonly, Bl, fl in legs:
            try:
                if wl == a.workload:
                    pd_l, ud_l, xd_l, prob_l, D_l, tau_l = pdict, unitdict, xdict, prob, D, tau
                else:
                    pd_l, ud_l, _c, xd_l = problem.make_problem(wl)
                    prob_l = con_dynamics.problem_arrays(pd_l, ud_l)
                    ps_l = pd_l["ps_params"]
                    D_l = [ps_l.D(i) for i in range(pd_l["num_sections"])]
                    tau_l = [ps_l.tau(i) for i in range(pd_l["num_sections"])]
                El = Engine(prob_l, D=D_l, tau=tau_l, device=local, flags=fl)
                Xl = problem.synthetic_batch(pack_x(xd_l), El.M, LEG_DISTINCT)
                dXl = torch.from_numpy(Xl).to(dev).repeat(Bl // LEG_DISTINCT, 1).contiguous()
                # [r6] as for the headline: `value` / `frac` on the allocator's placement; then the leg's buffers placed (the fastest of
                # four candidates) -> `value_placed` / `frac_placed`, informational
                dresl = torch.empty((Bl, El.nres), dtype=torch.float64, device=dev)
                djvl = None if resonly else torch.empty((Bl, El.V), dtype=torch.float64, device=dev)

                def measure_leg():
                    jp_ = 0 if djvl is None else djvl.data_ptr()

                    def leg_step():
                        El.eval_batch_device(Bl, dXl.data_ptr(), dresl.data_ptr(), jp_, stream)
                    leg_step()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    leg_step()
                    torch.cuda.synchronize()
                    one_ms = 1e3 * (time.perf_counter() - t0)
                    for _ in range(max(2, min(2000, int(WARM_MS / max(one_ms, 1e-3))))):
                        leg_step()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(LEG_K):
                        leg_step()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / LEG_K
                ms = measure_leg()
                place_l, ms_placed = None, None
                if a.placement_tries > 1:
                    from gelato_amd.placement import place_batch_buffers
                    dXl, dresl, djvl, place_l = place_batch_buffers(El, dXl, want_jac=not resonly, tries=min(a.placement_tries, 4),
                                                                    launches=8, warm=4, stream=stream, seed=1)
                    ms_placed = measure_leg()
                st_l = El.sync(stream)
                amin_l = 8 * (El.nvars + El.nres) if resonly else El.algorithmic_bytes
                tag_l = wl + ("_resonly" if resonly else "") + ("_flags%d" % fl if fl else "")
                td_l, _why = static_counters("traffic", tag_l, Bl, engine=El, jac=not resonly)
                moved = None if td_l is None else td_l.get("hbm_bytes_per_launch")
                inf_l = El.launch_info(Bl, True, not resonly)
                ms_def = ms
                if ms_placed is not None:      # as the headline: the leg's figure is the placed one, the allocator's beside it
                    ms = ms_placed
                oc[key] = {"workload": wl + (" (residual only)" if resonly else ""), "engine_flags": fl, "batch": Bl,
                           "value": Bl / (ms * 1e-3), "unit": "evals/s", "kernel_ms": ms,
                           "algorithmic_bytes_per_eval": amin_l, "frac": amin_l * Bl / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "value_default_placement": Bl / (ms_def * 1e-3), "frac_default_placement": amin_l * Bl / (ms_def * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "frac_of_bytes_moved": None if moved is None else moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "kernel": "gel::eval_kernel<%s, %s, %s, %s>" % tuple("true" if v else "false" for v in (inf_l[0], inf_l[1], inf_l[2], inf_l[4])),
                           "status": int(st_l),
                           "value_placed": None if ms_placed is None else Bl / (ms_placed * 1e-3),
                           "frac_placed": None if ms_placed is None else amin_l * Bl / (ms_placed * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "buffer_placement_ms": None if place_l is None else [round(c["ms_per_launch"], 4) for c in place_l["candidates"]]}
                El.close()
                del dXl, dresl, djvl, El
            except Exception as ex:  # noqa: BLE001
                oc[key] = {"error": str(ex)[:300]}
        oc["note"] = ("%d distinct synthetic vectors tiled to the batch on the device; >= 40 ms of untimed launches, then %d launches under "
                      "HIP events -> value / frac on the fastest of four candidate placements (as the headline), value_default_placement / "
                      "frac_default_placement on the allocator's; after the headline's timed regions; %.1f s in all" % (LEG_DISTINCT, LEG_K, time.perf_counter() - t_legs))
        out["other_configs"] = oc

    if not a.no_cpu_baseline:   # rank 0, at every N
        out["cpu_baseline"] = cpu_baseline(prob, D, tau, X, gpu_first, residual_only=a.residual_only)
        # informational only (the roofline fraction is the kernel-quality figure): the GPU writes compact
        # Jacobian values, the scalar port materialises every COO value
        out["cpu_baseline"]["gpu_over_cpu_informational"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
