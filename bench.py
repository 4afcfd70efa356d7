"""bench.py -- D-optimal design iterations/second on MI355X (BASELINE.json metric).

A "step" is one outer iteration of the solver over one resident instance.  Default workload
(N=1) is BASELINE config 2: D_opt_design(2048, 32768), ABPG_gain(gamma=2), fp64.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config 2|3|4|5]
                    [--workload abpg_gain|abpg|bpg|fw|fw_away|poisson_abpg|poisson_bpg]
                    [--m M --n N] [--mode instances|shard] [--instances-per-gpu P]

--gpus N > 1 started WITHOUT a torch.distributed environment makes this process a launcher: before
anything touches the GPU it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child, relays its output and exits with its status.  Started by torch.distributed.run (the
driver's form) it is one rank and WORLD_SIZE must equal --gpus.

BASELINE configs:  2 = (2048,32768) ABPG_gain, one independent instance per GPU (weak scaling);
3 = the same shape, D_opt_FW_away;  4 = 8 x D_opt_design(512,8192) ABPG per GPU (64 over 8 GPUs),
dealt to the ranks, solved concurrently, results gathered;  5 = ONE (8192,262144) ABPG instance whose
design points are sharded over the GPUs, one RCCL all-reduce of the Gram matrix per evaluation.

Prints ONE JSON line on rank 0 (contract in the task description), including
  value        -- the driver-timed region: exactly K steps after W warm-up steps from x0
  steady_state -- the same solver timed over >= 200 further iterations starting at k >= 100, where the
                  line search of ABPG_gain retries about once per iteration (2 gradient + 3 value
                  evaluations per step instead of 1 + 2 in the first iterations)
  roofline     -- dominant kernel (Gram stream-K, fp64 MFMA bound; FW workloads: the V pass, HBM bound)
  cpu_baseline -- the NumPy oracle timed on this box's host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X vendor figure (fp64 matrix); confirmed on the box by accbpg_mfma_f64_peak
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_traffic.json")

CONFIGS = {      # BASELINE.json configs -> defaults of the switches below
    2: dict(workload="abpg_gain", m=2048, n=32768, mode="instances", instances_per_gpu=1),
    3: dict(workload="fw_away", m=2048, n=32768, mode="instances", instances_per_gpu=1),
    4: dict(workload="abpg", m=512, n=8192, mode="instances", instances_per_gpu=8),
    5: dict(workload="abpg", m=8192, n=262144, mode="shard", instances_per_gpu=1),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS),
                    help="BASELINE.json config whose workload / shape / sharding the switches below default to")
    ap.add_argument("--workload", default=None,
                    choices=["abpg_gain", "abpg", "bpg", "fw", "fw_away", "poisson_abpg", "poisson_bpg"])
    ap.add_argument("--m", type=int, default=None)
    ap.add_argument("--n", type=int, default=None)
    ap.add_argument("--mode", default=None, choices=["instances", "shard"],
                    help="'instances' = independent instances per GPU (weak scaling); 'shard' = ONE instance, "
                         "design points partitioned over the GPUs with one RCCL all-reduce of the Gram matrix "
                         "per evaluation (strong scaling, BASELINE config 5)")
    ap.add_argument("--instances-per-gpu", type=int, default=None,
                    help="independent instances per GPU advanced concurrently (BASELINE config 4: 8 per GPU)")
    ap.add_argument("--host-threads", action="store_true",
                    help="with --instances-per-gpu: one host thread + stream per instance instead of the lock-step batch")
    ap.add_argument("--steady-start", type=int, default=100, help="first iteration of the steady-state window")
    ap.add_argument("--steady-iters", type=int, default=200, help="iterations timed in the steady-state window")
    ap.add_argument("--no-steady", action="store_true")
    ap.add_argument("--library-collectives", action="store_true",
                    help="shard mode: the all-reduce / all-gather inside libaccbpg_hip.so (accbpg_dopt_shard_*, its own "
                         "RCCL communicator) instead of torch.distributed")
    ap.add_argument("--no-overlap", action="store_true",
                    help="evaluate F[k] = f(x) on the solver's own stream instead of beside the gradient evaluation")
    ap.add_argument("--speculation", action="store_true",
                    help="ABPG_gain: start the next trial's gradient evaluation beside the current value test "
                         "(DOptimalObj.speculate; off by default, see its docstring)")
    ap.add_argument("--linear-gram", action="store_true",
                    help="measure with Gram-matrix reuse through linearity switched on (extension); the "
                         "default run reports it separately as linear_gram_variant")
    ap.add_argument("--memo-values", action="store_true",
                    help="measure with DOptimalObj.memoize_values switched on (extension: F[k+1] = f(x) at the accepted "
                         "line-search point is not evaluated a second time; identical results)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the variant segments (profiling runs: the kernel trace then holds the headline "
                         "and steady-state regions only)")
    ap.add_argument("--logdet-refresh", type=int, default=None,
                    help="fw_away: refactor H for F[k] = log det(H_k) every this many iterations (1 = every iteration, "
                         "the reference's computation; default: the package's, D_opt_alg.LOGDET_REFRESH_DEFAULT)")
    ap.add_argument("--logdet-ring", type=int, default=None,
                    help="fw_away with --logdet-refresh 1: factorisations in flight beside the steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--launch-check", action="store_true",
                    help="rank plumbing only (no GPU work): rendezvous, deal 8 instances per rank, gather, print")
    args = ap.parse_args(argv)
    for key, val in CONFIGS[args.config].items():
        if getattr(args, key) is None:
            setattr(args, key, val)
    return args


# ---------------------------------------------------------------------------- rank launcher
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(args):
    """--gpus N > 1 outside a torch.distributed environment: start the N ranks as a child job.  Runs
    before torch is imported, so this process never initialises the GPU."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s; start it with --nproc-per-node %d"
                             % (args.gpus, world_env, args.gpus))
        return
    if args.gpus <= 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


# ---------------------------------------------------------------------------- instances
def make_instance(m, n, seed, device):
    """np.random.seed(s); randn(m,n) as accbpg/applications.py:47-49, resident in HBM."""
    import torch
    np.random.seed(seed)
    V = np.random.randn(m, n)
    return torch.from_numpy(V).to(device)


def fw_generator(f, x0, away, maxitrs, logdet_refresh, logdet_ring):
    """The package's own Frank-Wolfe loops as step generators (accbpg_and_fw_amd/D_opt_alg.py); eps < 0: never stops."""
    from accbpg_and_fw_amd.D_opt_alg import D_opt_FW_away_steps, D_opt_FW_steps
    if away:
        return D_opt_FW_away_steps(f, x0, -1.0, maxitrs, verbose=False, logdet_refresh=logdet_refresh,
                                   logdet_ring=logdet_ring)
    return D_opt_FW_steps(f, x0, -1.0, maxitrs, verbose=False)


# ---- Poisson linear inverse problem (HBM-bound objective), single GPU
def poisson_instance(acc, torch, m, n, seed, lamda):
    gen = torch.Generator(device="cuda").manual_seed(seed)
    A = torch.rand(m, n, dtype=torch.float64, device="cuda", generator=gen)
    A /= A.sum(0, keepdim=True)
    x = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) / n
    x = torch.clamp(x - x.mean(), min=0) * 10
    b = A @ x + 0.001 * (torch.rand(m, dtype=torch.float64, device="cuda", generator=gen) - 0.5) / m
    b = torch.clamp(b, min=1e-12)
    f = acc.PoissonRegression(A, b)
    return f, acc.BurgEntropyL2(lamda), float(b.sum()), torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")


def poisson_main(args):
    import torch
    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd.algorithms import ABPG_steps, BPG_steps
    m, n = (args.m, args.n) if (args.m, args.n) != (2048, 32768) else (8192, 65536)
    solver = args.workload.split('_')[1]
    f, h, L, x0 = poisson_instance(acc, torch, m, n, 1, 0.001)
    total = args.steps + args.warmup
    if solver == "abpg":
        gen = ABPG_steps(f, h, L, x0, gamma=2.0, maxitrs=total + 1, theta_eq=False, verbose=False)
    else:
        gen = BPG_steps(f, h, L, x0, maxitrs=total + 1, linesearch=True, ls_ratio=1.5, verbose=False)
    for _ in range(args.warmup):
        next(gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next(gen)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    # the two passes, timed with events on the launch stream
    x = x0.clone()
    reps = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f.func_grad(x, 0); f.func_grad(x, 2)
    ev[0].record()
    for _ in range(reps):
        f.func_grad(x, 0)
    ev[1].record()
    ev[2].record()
    for _ in range(reps):
        f.func_grad(x, 2)
    ev[3].record()
    torch.cuda.synchronize()
    ms_val = ev[0].elapsed_time(ev[1]) / reps
    ms_fg = ev[2].elapsed_time(ev[3]) / reps
    byt = 8.0 * m * n
    out = {
        "metric": "Poisson iters/sec (m=%d,n=%d)" % (m, n), "value": args.steps / dt, "unit": "iterations/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "Poisson_regrL2-shaped instance (%d,%d) lamda=1e-3, %s, fp64" % (m, n, solver)},
        "roofline": {"bound": "hbm", "kernel": "poisson_ax_kernel + fw_vgemv_partial_kernel (A x, then A^T r)",
                     "achieved": 2 * byt / (ms_fg * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                     "frac": 2 * byt / (ms_fg * 1e-3) / (PEAK_HBM_GBS * 1e9), "traffic": None,
                     "avg_func_grad_ms": ms_fg, "value_only_ms": ms_val,
                     "value_only_GBps": byt / (ms_val * 1e-3) / 1e9,
                     "note": "includes the value readback (one stream sync) per call"},
    }
    if not args.no_cpu_baseline:
        from oracle import np_oracle as O
        Ah, bh = f._A.cpu().numpy(), f._b.cpu().numpy()
        fo, ho = O.PoissonOracle(Ah, bh), O.BurgL2Oracle(0.001)
        it = 3
        t0 = time.perf_counter()
        if solver == "abpg":
            O.ABPG(fo, ho, L, x0.cpu().numpy(), gamma=2.0, maxitrs=it, theta_eq=False)
        else:
            O.BPG(fo, ho, L, x0.cpu().numpy(), maxitrs=it, linesearch=True, ls_ratio=1.5)
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": it / cdt, "unit": "iterations/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "%d oracle iterations of the same instance (NumPy, threaded BLAS)" % it}
    print(json.dumps(out))


# ---------------------------------------------------------------------------- rank plumbing check (no GPU work)
def launch_check(args):
    """What a rank does around the solve, without the solve: rendezvous over gloo, deal 8 instances per rank with
    split_instances, 'solve' them with a stand-in, gather by instance index, max-reduce a timing."""
    import torch
    import torch.distributed as dist
    from accbpg_and_fw_amd.batched import solve_instances
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    total = 8 * world
    res = solve_instances(lambda i: (i, None, 1.0, np.full(4, float(i))), total,
                          lambda f, h, L, x0: (x0 * 2.0, np.array([float(f)])), world=world, rank=rank,
                          concurrent=False)
    assert len(res) == total and all(res[i][1][0] == float(i) for i in range(total))
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "backend": "gloo" if world > 1 else None,
                          "instances": total, "max_rank_token": float(t.item())}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------- main measurement
def main():
    args = parse()
    launch_ranks_if_needed(args)
    if args.launch_check:
        return launch_check(args)
    if args.workload.startswith("poisson"):      # SURVEY 8(f) row 4, single GPU, reported apart from the headline
        return poisson_main(args)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)
    device = torch.device("cuda", torch.cuda.current_device())
    # RCCL needs one GPU per rank.  When fewer GPUs than ranks are visible (rehearsals on a one-GPU
    # box) the control-plane collectives (barrier, max of the timings, result gather) go over gloo; the
    # instances mode has no data-path collective, shard mode requires RCCL.
    use_nccl = (world > 1 and ndev >= world)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if use_nccl else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
        if args.mode == "shard" and not use_nccl:
            raise SystemExit("--mode shard needs one GPU per rank (RCCL)")

    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd import algorithms as alg
    from accbpg_and_fw_amd.sharded import split_instances

    m, n = args.m, args.n
    shard = (args.mode == "shard" and world > 1)
    ipg = max(1, args.instances_per_gpu)
    # instance indices of this rank: world * ipg instances dealt round-robin (seed = 1 + index)
    mine = split_instances(world * ipg, world, rank)
    if shard:
        # synthetic standard-normal design points generated on the device, per shard (the whole
        # matrix of config 5 is 16 GiB and never exists on one host)
        from accbpg_and_fw_amd.sharded import make_sharded, shard_bounds
        lo, hi = shard_bounds(n, world, rank)
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + rank)
        V = torch.randn(m, hi - lo, dtype=torch.float64, device=device, generator=gen)
        if args.library_collectives:
            from accbpg_and_fw_amd.sharded import make_sharded_native
            f = make_sharded_native(V, m, n, rank, world)
        else:
            f = make_sharded(V, m, n, rank, world)
        prof_objs = [f.local]
    elif m * n * 8 > (4 << 30):
        # one GPU's worth of a config-5-sized instance: generated on the device like the shards
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + rank)
        f = acc.DOptimalObj(torch.randn(m, n, dtype=torch.float64, device=device, generator=gen))
        prof_objs = [f]
    else:
        f = acc.DOptimalObj(make_instance(m, n, 1 + mine[0], device))
        prof_objs = [f]
    overlap = (not args.no_overlap) and (not shard) and hasattr(f, "overlap_values")
    if hasattr(f, "overlap_values"):
        f.overlap_values(overlap)
    if hasattr(f, "speculate"):
        f.speculate(args.speculation)
    if args.linear_gram and not shard:
        f.linear_gram(True)
    if args.memo_values and hasattr(f, "memoize_values"):
        f.memoize_values(True)
    h = acc.BurgEntropySimplex()
    x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device=device)
    # (the steady-state window is 300 further iterations: only where an iteration takes tens of milliseconds)
    steady = (not args.no_steady) and args.workload in ("abpg_gain", "abpg", "bpg") and args.steady_iters > 0 \
        and float(m) * n <= 2.0 * 2048 * 32768
    total = args.warmup + args.steps
    # (the lock-step batch times its kernels over `steps` further steps behind the driver-timed region)
    horizon = total + args.steps + (max(0, args.steady_start - total) + args.steady_iters if steady else 0) + 2

    def make_gen(ff, hh, length):
        if args.workload == "abpg_gain":
            return alg.ABPG_gain_steps(ff, hh, 1.0, x0.clone(), 2, length, verbose=False)
        if args.workload == "abpg":
            return alg.ABPG_steps(ff, hh, 1.0, x0.clone(), 2, length, verbose=False)
        return alg.BPG_steps(ff, hh, 1.0, x0.clone(), length, verbose=False)

    batch = None
    lockstep = None
    objs = [f]
    if ipg > 1 and args.workload in ("bpg", "abpg", "abpg_gain") and not shard and not args.host_threads:
        # BASELINE config 4: the instances of this GPU advance in lock-step, one launch per kernel family for all of
        # them (accbpg_dopt_batch_*); results are bit-identical to solving them one after the other
        from accbpg_and_fw_amd.batched import ABPG_batch_steps, ABPG_gain_batch_steps, BPG_batch_steps, DOptimalBatch
        lockstep = DOptimalBatch([f.V_dev] + [make_instance(m, n, 1 + idx, device) for idx in mine[1:]])
        objs = [lockstep]
        prof_objs = [lockstep]
        if args.workload == "abpg":
            gen = ABPG_batch_steps(lockstep, acc.BurgEntropySimplex(), 1.0, x0, 2, horizon, overlap=overlap)
        elif args.workload == "bpg":
            # epsilon=0: BPG's |F[k]-F[k-1]| stopping test never fires, so the steady-state region below still has
            # running instances at this small shape (the timed region itself ends long before the test would fire)
            gen = BPG_batch_steps(lockstep, acc.BurgEntropySimplex(), 1.0, x0, horizon, epsilon=0.0)
        else:
            gen = ABPG_gain_batch_steps(lockstep, acc.BurgEntropySimplex(), 1.0, x0, 2, horizon, overlap=overlap)

        def advance(count=1):
            for _ in range(count):
                next(gen)
    elif ipg > 1:
        if shard or args.workload.startswith("fw"):
            raise SystemExit("--instances-per-gpu applies to the BPG family in instances mode")
        from accbpg_and_fw_amd.batched import BatchStepper
        objs = [f] + [acc.DOptimalObj(make_instance(m, n, 1 + idx, device)) for idx in mine[1:]]
        for ff in objs[1:]:
            ff.overlap_values(overlap)
        batch = BatchStepper([(lambda ff=ff: make_gen(ff, acc.BurgEntropySimplex(), horizon)) for ff in objs],
                             device, threads=ipg)
        prof_objs = objs
        advance = batch.step
    elif args.workload in ("abpg_gain", "abpg", "bpg"):
        gen = make_gen(f, h, horizon)

        def advance(count=1):
            for _ in range(count):
                next(gen)
    else:
        gen = fw_generator(f, x0, args.workload == "fw_away", total + 1, args.logdet_refresh, args.logdet_ring)

        def advance(count=1):
            for _ in range(count):
                next(gen)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=device if use_nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def calls_now():
        return {k: sum(o.calls[k] for o in objs) for k in ("value", "grad")}
    ninst_local = ipg if lockstep is not None else len(objs)

    def timed(count):
        """(seconds, max over ranks; oracle calls per step and instance) of `count` further steps."""
        c0 = calls_now()
        barrier()
        t0 = time.perf_counter()
        advance(count)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        c1 = calls_now()
        return dt, {k: (c1[k] - c0[k]) / (count * ninst_local) for k in c0}

    if shard and args.workload.startswith("fw"):
        raise SystemExit("shard mode covers the BPG family (the FW solvers are not a multi-GPU config)")

    # ---- the driver-timed region: W warm-up steps, then exactly K steps
    advance(args.warmup)
    is_fw = args.workload.startswith("fw")
    # Kernel timing inside the driver-timed region where a step is long (one instance of the BPG family: two event
    # records per launch against milliseconds of kernels).  Where a step is short -- Frank-Wolfe (~0.1-0.2 ms), the
    # lock-step batch (its value evaluations then also leave the side stream) -- the kernels are timed over the SAME
    # NUMBER of further steps of the same run right behind the region instead, and the region stays as users run it.
    prof_after = is_fw or lockstep is not None
    if not prof_after:
        for o in prof_objs:
            o.profile(True)
    elapsed, calls = timed(args.steps)
    if prof_after:
        if is_fw:
            fw_gen_extra = fw_generator(f, x0, args.workload == "fw_away", args.steps + 1, args.logdet_refresh,
                                        args.logdet_ring)
            more = lambda: next(fw_gen_extra)
        else:
            more = lambda: advance(1)
        for o in prof_objs:
            o.profile(True)
        for _ in range(args.steps):
            more()
        torch.cuda.synchronize()
    prof = {}
    for o in prof_objs:
        for k, v in o.profile_read().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + v[0], a[1] + v[1])
        o.profile(False)

    # ---- the regime the solver lives in: continue the same run to k >= steady_start, then time
    steady_out = None
    if steady:
        done = total
        if done < args.steady_start:
            advance(args.steady_start - done)
            done = args.steady_start
        spec0 = getattr(f, "spec_unused", 0)
        sdt, scalls = timed(args.steady_iters)
        steady_out = {"value": (1 if shard else world * ipg) * args.steady_iters / sdt, "unit": "iterations/s",
                      "ms_per_step": 1e3 * sdt / args.steady_iters, "first_iteration": done,
                      "iterations": args.steady_iters, "oracle_calls_per_step": scalls,
                      "gradients_started_ahead_unused_per_step":
                          (getattr(f, "spec_unused", 0) - spec0) / args.steady_iters if hasattr(f, "spec_unused") else None,
                      "note": "same run continued; for ABPG_gain the gain is cut by ls_dec before every search, so "
                              "after the first ~20-30 iterations every iteration retries about once "
                              "(accbpg/algorithms.py:358-390)"}

    variants_ok = (not args.no_variants) and (not args.linear_gram) and (not args.memo_values) and (not shard) and ipg == 1 \
        and args.workload in ("abpg_gain", "abpg")

    def variant_run():
        g2 = make_gen(f, h, total + 2)
        for _ in range(args.warmup):
            next(g2)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            next(g2)
        barrier()
        return max_over_ranks(time.perf_counter() - t1)

    # same K steps once more as the package runs them by default: F[k] = f(x) on a second stream beside the
    # gradient evaluation (the timed region above keeps it on the solver's stream for uncontended kernel timings)
    serial_variant = None
    if variants_ok and overlap:
        dt = variant_run()
        serial_variant = {"value": world * args.steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / args.steps,
                          "note": "the package default: F[k] = f(x) on a second stream beside func_grad(y), which does "
                                  "not depend on it -- the latency-bound Cholesky of one evaluation runs under the "
                                  "MFMA-bound products of the other; identical kernels and results "
                                  "(test_overlapped_value_*); same transient window as `value`, whose timed region "
                                  "keeps both evaluations on one stream so that the HIP-event kernel timings are "
                                  "uncontended"}

    # same K steps once more with Gram-matrix reuse through linearity (extension, reported apart)
    lin_variant = None
    if variants_ok:
        f.linear_gram(True)
        gl0, vh0 = f.gram_launches, f.value_hits
        dt = variant_run()
        lin_variant = {"value": world * args.steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / args.steps,
                       "gram_launches_per_step": (f.gram_launches - gl0) / total,
                       "repeated_value_lookups_per_step": (f.value_hits - vh0) / total,
                       "note": "V diag(x) V^T is linear in x: Gram matrices at x, z stay resident and the ones at "
                               "y and x+ are O(m^2) combinations; f at a vector object already evaluated (the "
                               "line-search point, re-read as F[k+1]) is looked up; results equal to rounding "
                               "(test_linear_gram_*); same transient window as `value`"}
        f.linear_gram(False)

    # config 4: gather per-instance bookkeeping by instance index (the full results travel the same way through
    # batched.solve_instances, which is what a caller of the package uses; bench.py steps generators instead)
    gathered = None
    if lockstep is not None:
        counts = {idx: lockstep.calls["grad"] // ipg for idx in mine}
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, counts)
            gathered = {k: v for p in parts for k, v in p.items()}
        else:
            gathered = counts
    if batch is not None:
        counts = {idx: o.calls["grad"] for idx, o in zip(mine, objs)}
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, counts)
            gathered = {k: v for p in parts for k, v in p.items()}
        else:
            gathered = counts
        batch.close()

    if rank == 0:
        ninst = 1 if shard else world * ipg
        value = ninst * args.steps / elapsed
        if shard:
            layout = "ONE instance, design points sharded over the GPUs, one RCCL all-reduce of the Gram matrix per evaluation" \
                + (" (collectives inside the library)" if args.library_collectives else "")
        elif lockstep is not None:
            layout = ("%d independent instances per GPU advancing in lock-step (one launch per kernel family for all of "
                      "them%s), dealt round-robin over the ranks" % (ipg, "" if lockstep.fused else "; NOT fused at this shape"))
        elif ipg > 1:
            layout = "%d independent instances per GPU advanced concurrently from host threads, dealt round-robin over the ranks" % ipg
        else:
            layout = "one independent instance per GPU"
        out = {
            "metric": "D-opt iters/sec (m=%d,n=%d)" % (m, n),
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "D_opt_design(%d,%d) %s%s fp64, %s"
                                   % (m, n, args.workload,
                                      " gamma=2" if "bpg" in args.workload and args.workload != "bpg" else "", layout),
                       "baseline_config": args.config, "instances": ninst, "instances_per_gpu": ipg,
                       "seeds": "1..%d" % (world * ipg), "collectives": backend,
                       "window": "transient: iterations %d..%d from x0 = 1/n" % (args.warmup, total - 1),
                       "oracle_calls_per_step": calls,
                       "value_overlap": ("on" if lockstep is not None else
                                         "off in the timed region (kernel timing on), on in steady_state and overlap_variant")
                                        if overlap else "off",
                       "linear_gram": bool(args.linear_gram), "memo_values": bool(args.memo_values)},
        }
        if steady_out is not None:
            out["steady_state"] = steady_out
        if gathered is not None:
            out["config"]["instances_gathered"] = len(gathered)
        gram_ms, gram_cnt = prof.get("gram", (0.0, 0))
        grad_ms, grad_cnt = prof.get("grad", (0.0, 0))
        out["kernels"] = {k: {"ms_total": v[0], "launches": v[1], "ms_avg": (v[0] / v[1] if v[1] else None)}
                          for k, v in prof.items()}
        if serial_variant is not None:
            out["overlap_variant"] = serial_variant
        if lin_variant is not None:
            out["linear_gram_variant"] = lin_variant
        if args.workload in ("abpg_gain", "abpg", "bpg"):
            # dominant kernel: Gram stream-K.  Algorithmic flops per launch = m^2 * n_local (SURVEY 8(d):
            # the SYRK share of 2 m^2 n + m^3/3 + 2 m n).
            n_local = (f.hi - f.lo) if shard else n
            # a lock-step launch covers every instance of the GPU, or `chunk` of them where their one-launch
            # factorisations do not fit the chip together (64 instances: chunks of 15, the last one smaller -- the
            # average launch then covers ipg / ceil(ipg / chunk) instances)
            per_launch = 1
            if lockstep is not None:
                per_launch = ipg / float(-(-ipg // max(1, lockstep.chunk)))
            flops = float(m) * m * n_local * per_launch
            # rows of 65536 columns or more: the Gram matrix is formed by one launch per column block of 32768 (the
            # library's choice); the algorithmic flops of ONE launch are those of its block
            evals = (calls["value"] + calls["grad"]) * args.steps * (1 if lockstep is not None else ninst_local)
            gram_blocks = 1
            if lockstep is None and gram_cnt and evals:
                gram_blocks = max(1, int(round(gram_cnt / float(evals))))
            gram_flops = flops / gram_blocks
            achieved = gram_flops / (gram_ms / gram_cnt * 1e-3) * 1e-12 if gram_cnt else None
            traffic = None
            try:        # HBM bytes per launch from this round's PMC passes (same workload only)
                tj = json.load(open(TRAFFIC_FILE))
                if (m, n) == (2048, 32768) and not shard:
                    traffic = tj["gram_streamk_glds_kernel"]["hbm_bytes"]
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "mfma", "kernel": "gram_streamk_glds_kernel (weighted Gram matrix, stream-K)",
                               "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_FP64_MFMA_TFLOPS if achieved else None, "traffic": traffic,
                               "traffic_source": ("replayed from %s: HBM bytes per launch counted by separate rocprofv3 --pmc "
                                                  "passes of this command (a counter pass cannot run inside the timed "
                                                  "process), not in this run" % os.path.relpath(TRAFFIC_FILE, ROOT))
                                                 if traffic else None,
                               "avg_launch_ms": gram_ms / gram_cnt if gram_cnt else None, "launches": gram_cnt,
                               "instances_per_launch": per_launch, "column_blocks_per_evaluation": gram_blocks,
                               "timing": ("HIP events around every launch on the launching stream, "
                                          + ("over %d further steps of the same run right behind the timed region"
                                             % args.steps if lockstep is not None else "inside the timed region"))
                                         + (" (a launch covers %.1f instances of the lock-step batch on average; all of them "
                                            "active in this window)" % per_launch if lockstep is not None else
                                            (" (concurrent instances share the chip)" if ipg > 1 else ""))}
            if grad_cnt:
                ga = flops / (grad_ms / grad_cnt * 1e-3) * 1e-12
                out["roofline_grad_kernel"] = {"bound": "mfma", "kernel": "colnorm_glds_kernel (triangular product + column norms)",
                                               "achieved": ga, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                               "frac": ga / PEAK_FP64_MFMA_TFLOPS,
                                               "avg_launch_ms": grad_ms / grad_cnt, "launches": grad_cnt}
            # whole-step algorithmic rate from the measured call mix
            fl_grad = 2.0 * m * m * n + m ** 3 / 3.0 + 2.0 * m * n
            fl_val = 1.0 * m * m * n + m ** 3 / 3.0
            step_fl = calls["grad"] * fl_grad + calls["value"] * fl_val
            out["whole_step"] = {"algorithmic_tflops": step_fl * ninst / (elapsed / args.steps) * 1e-12 / world,
                                 "frac_of_mfma_peak_per_gpu": step_fl * ninst / (elapsed / args.steps) * 1e-12 / world
                                 / PEAK_FP64_MFMA_TFLOPS}
        else:
            # Frank-Wolfe step: HBM bound, algorithmic bytes 8 m n + 24 m^2 + 48 n per step
            bytes_step = 8.0 * m * n + 24.0 * m * m + 48.0 * n
            achieved = bytes_step / (elapsed / args.steps) * 1e-9
            vp_ms, vp_cnt = prof.get("fw_vpass", (0.0, 0))
            vach = 8.0 * m * n / (vp_ms / vp_cnt * 1e-3) * 1e-9 if vp_cnt else None
            # dominant kernel: the pass over V (u = Hv^T V), algorithmic 8 m n bytes per launch
            out["roofline"] = {"bound": "hbm", "kernel": "fw_vgemv_partial_kernel (the pass over V of one step)",
                               "achieved": vach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": vach / PEAK_HBM_GBS if vach else None, "traffic": None,
                               "avg_launch_ms": vp_ms / vp_cnt if vp_cnt else None, "launches": vp_cnt,
                               "timing": "HIP events around the launch on its stream, over %d further steps of the same "
                                         "solver right behind the driver-timed region (the event pair would be a "
                                         "measurable share of a 0.1-0.2 ms step)" % args.steps}
            out["whole_step"] = {"algorithmic_bytes": bytes_step, "achieved_GBps": achieved,
                                 "frac_of_hbm_peak": achieved / PEAK_HBM_GBS,
                                 "note": "8 m n + 24 m^2 + 48 n bytes (SURVEY 8(d)) over the host-timed step"}
            if args.workload == "fw_away":
                from accbpg_and_fw_amd import D_opt_alg as DA
                out["config"]["logdet_refresh"] = DA.LOGDET_REFRESH_DEFAULT if args.logdet_refresh is None else args.logdet_refresh
                out["config"]["logdet_ring"] = DA.LOGDET_RING_DEFAULT if args.logdet_ring is None else args.logdet_ring
        # measured fp64 MFMA peak on this device
        import ctypes as C
        from accbpg_and_fw_amd import _lib
        tf = C.c_double(0.0)
        _lib.load().accbpg_mfma_f64_peak(20000, C.byref(tf), None)
        out["mfma_f64_peak_measured_tflops"] = tf.value

        if world == 1 and not args.no_cpu_baseline and m * n <= 2048 * 32768:
            out["cpu_baseline"] = cpu_baseline(args, m, n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, m, n):
    """The NumPy oracle (kind "port") on this box's host cores, bounded sample: a few outer iterations from
    x0 (the transient window `value` is timed in), plus one gradient and one value evaluation timed alone,
    from which the steady-state mix (2 gradient + 3 value evaluations per ABPG_gain iteration) is priced."""
    from oracle import np_oracle as O
    np.random.seed(1)
    V = np.random.randn(m, n)
    fo, ho = O.DOptOracle(V), O.BurgSimplexOracle()
    x0 = np.ones(n) / n
    iters = max(1, args.cpu_iters)
    t0 = time.perf_counter()
    if args.workload == "abpg_gain":
        O.ABPG_gain(fo, ho, 1.0, x0, 2, iters)
    elif args.workload == "abpg":
        O.ABPG(fo, ho, 1.0, x0, 2, iters)
    elif args.workload == "bpg":
        O.BPG(fo, ho, 1.0, x0, iters)
    elif args.workload == "fw":
        iters = max(iters, 3)
        O.D_opt_FW(V, x0, 1e-8, iters)
    else:
        iters = max(iters, 3)
        O.D_opt_FW_away(V, x0, 1e-8, iters)
    dt = time.perf_counter() - t0
    try:
        import threadpoolctl
        threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    out = {"value": iters / dt, "unit": "iterations/s", "cores": threads, "kind": "port",
           "sample": "%d outer iteration(s) of the NumPy oracle %s at (%d,%d) from x0 -- the transient window of "
                     "`value` -- %.1f s wall%s" %
                     (iters, args.workload, m, n, dt,
                      " (includes the one-off O(m^2 n) setup)" if args.workload.startswith("fw") else ""),
           "numpy": np.__version__, "os_cpu_count": os.cpu_count()}
    if args.workload == "abpg_gain":
        t0 = time.perf_counter()
        fo.func_grad(x0, 2)
        tg = time.perf_counter() - t0
        t0 = time.perf_counter()
        fo.func_grad(x0, 0)
        tv = time.perf_counter() - t0
        out["steady_state_estimate"] = {"value": 1.0 / (2 * tg + 3 * tv), "unit": "iterations/s",
                                        "sample": "one gradient (%.2f s) and one value (%.2f s) evaluation of the oracle, "
                                                  "priced at the steady mix of 2 + 3 per iteration" % (tg, tv)}
    return out


if __name__ == "__main__":
    main()
