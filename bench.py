"""bench.py -- D-optimal design iterations/second on MI355X (BASELINE.json metric).

A "step" is one outer iteration of the solver over one resident instance.  Default workload
(N=1) is BASELINE config 2: D_opt_design(2048, 32768), ABPG_gain(gamma=2), fp64.  With --gpus N
every rank owns an independent instance of the same shape (seeds 1..N) -- instances shard with
no data-path collective ("scaling": "weak"); the whole-job value is N ranks' iterations over
the max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload abpg_gain|abpg|bpg|fw|fw_away|poisson_abpg|poisson_bpg]
                    [--m 2048 --n 32768] [--no-cpu-baseline]

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline     -- dominant kernel (Gram stream-K, fp64 MFMA bound; FW workloads: the V pass, HBM bound)
  cpu_baseline -- the NumPy oracle timed on this box's host cores on a bounded sample (N=1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X vendor figure (fp64 matrix); confirmed on the box by accbpg_mfma_f64_peak
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="abpg_gain", choices=["abpg_gain", "abpg", "bpg", "fw", "fw_away", "poisson_abpg", "poisson_bpg"])
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--mode", default="instances", choices=["instances", "shard"],
                    help="N>1: 'instances' = one independent instance per GPU (weak scaling, default); "
                         "'shard' = ONE instance, design points partitioned over the GPUs with one RCCL "
                         "all-reduce of the Gram matrix per evaluation (strong scaling, BASELINE config 5)")
    ap.add_argument("--instances-per-gpu", type=int, default=1,
                    help="independent instances per GPU driven concurrently from host threads on separate "
                         "streams (BASELINE config 4: 64 x D_opt_design(512,8192) over 8 GPUs = 8 per GPU)")
    ap.add_argument("--linear-gram", action="store_true",
                    help="measure with Gram-matrix reuse through linearity switched on (extension); the "
                         "default run reports it separately as linear_gram_variant")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the overlap / linear-gram variant segments (profiling runs: the kernel trace then "
                         "holds the headline region only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3)
    return ap.parse_args()


def make_instance(m, n, seed, device):
    """np.random.seed(s); randn(m,n) as accbpg/applications.py:47-49, resident in HBM."""
    import torch
    np.random.seed(seed)
    V = np.random.randn(m, n)
    return torch.from_numpy(V).to(device)


class FWStepper:
    """Frank-Wolfe loops as step generators over the same entry points D_opt_FW* use."""

    def __init__(self, acc, f, x0, away, maxitrs):
        from accbpg_and_fw_amd.D_opt_alg import _FWState
        self.st = _FWState(f, x0)
        self.away = away
        self.m = f.m

    def step(self):
        st, m = self.st, self.m
        pr = st.probe(away=1 if self.away else 0, refresh_logdet=1 if self.away else 0)
        w_i, w_j = pr.w_i, pr.w_j
        eps_pos, eps_neg = w_i / m - 1, 1 - w_j / m
        if (not self.away) or eps_pos >= eps_neg:
            t = (w_i / m - 1) / (w_i - 1)
            coef = t / (1 - t + t * w_i) if self.away else t / (1 + t * (w_i - 1))
            st.update(pr.i, 1 - t, t, -coef, 1 - t)
        else:
            t = min((1 - w_j / m) / (w_j - 1), pr.x_j / (1 - pr.x_j))
            coef = t / (1 + t - t * w_j)
            st.update(pr.j, 1 + t, -t, coef, 1 + t)


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


def main():
    args = parse()
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
    # box) the control-plane collectives (barrier, max of the timings) go over gloo; the instances
    # mode has no data-path collective, shard mode requires RCCL.
    use_nccl = (world > 1 and ndev >= world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if use_nccl else "gloo", rank=rank, world_size=world)
        if args.mode == "shard" and not use_nccl:
            raise SystemExit("--mode shard needs one GPU per rank (RCCL)")

    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd import algorithms as alg

    m, n = args.m, args.n
    shard = (args.mode == "shard" and world > 1)
    if shard:
        # synthetic standard-normal design points generated on the device, per shard (the whole
        # matrix of config 5 is 16 GiB and never exists on one host)
        from accbpg_and_fw_amd.sharded import make_sharded, shard_bounds
        lo, hi = shard_bounds(n, world, rank)
        gen = torch.Generator(device=device)
        gen.manual_seed(1000 + rank)
        V = torch.randn(m, hi - lo, dtype=torch.float64, device=device, generator=gen)
        f = make_sharded(V, m, n, rank, world)
        prof_obj = f.local
    else:
        V = make_instance(m, n, 1 + rank, device)
        f = acc.DOptimalObj(V)
        prof_obj = f
    if args.linear_gram and not shard:
        f.linear_gram(True)
    h = acc.BurgEntropySimplex()
    x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device=device)
    total = args.warmup + args.steps

    ipg = max(1, args.instances_per_gpu)
    batch = None
    if ipg > 1:
        if shard or args.workload.startswith("fw"):
            raise SystemExit("--instances-per-gpu applies to the BPG family in instances mode")
        from accbpg_and_fw_amd.batched import BatchStepper
        objs = [f] + [acc.DOptimalObj(make_instance(m, n, 1 + rank + 1000 * j, device)) for j in range(1, ipg)]
        steps_fn = {"abpg_gain": lambda ff: alg.ABPG_gain_steps(ff, acc.BurgEntropySimplex(), 1.0, x0.clone(), 2,
                                                                total + 1, verbose=False),
                    "abpg": lambda ff: alg.ABPG_steps(ff, acc.BurgEntropySimplex(), 1.0, x0.clone(), 2, total + 1,
                                                      verbose=False),
                    "bpg": lambda ff: alg.BPG_steps(ff, acc.BurgEntropySimplex(), 1.0, x0.clone(), total + 1,
                                                    verbose=False)}[args.workload]
        batch = BatchStepper([(lambda ff=ff: steps_fn(ff)) for ff in objs], device, threads=ipg)
        step = batch.step
    elif args.workload == "abpg_gain":
        gen = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        step = lambda: next(gen)
    elif args.workload == "abpg":
        gen = alg.ABPG_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        step = lambda: next(gen)
    elif args.workload == "bpg":
        gen = alg.BPG_steps(f, h, 1.0, x0, total + 1, verbose=False)
        step = lambda: next(gen)
    else:
        fw = FWStepper(acc, f, x0, args.workload == "fw_away", total + 1)
        step = fw.step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            if use_nccl:
                dist.barrier(device_ids=[torch.cuda.current_device()])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    if shard and args.workload.startswith("fw"):
        raise SystemExit("shard mode covers the BPG family (the FW solvers are not a multi-GPU config)")
    if batch is not None:
        batch.step(args.warmup)
    else:
        for _ in range(args.warmup):
            step()
    prof_obj.profile(ipg == 1)
    calls0 = dict(f.calls)
    barrier()
    t0 = time.perf_counter()
    if batch is not None:
        batch.step(args.steps)          # every instance advances args.steps outer iterations
    else:
        for _ in range(args.steps):
            step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = prof_obj.profile_read()
    prof_obj.profile(False)
    calls = {k: f.calls[k] - calls0[k] for k in calls0}

    # same workload once more with F[k] = f(x) overlapped with the gradient evaluation (two streams)
    ovl_variant = None
    if (not args.no_variants) and (not args.linear_gram) and (not shard) and ipg == 1 \
            and args.workload in ("abpg_gain", "abpg"):
        f.overlap_values(True)
        if args.workload == "abpg_gain":
            gen3 = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        else:
            gen3 = alg.ABPG_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        for _ in range(args.warmup):
            next(gen3)
        barrier()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            next(gen3)
        barrier()
        dt = time.perf_counter() - t2
        if world > 1:
            tl = torch.tensor([dt], dtype=torch.float64, device=device if use_nccl else "cpu")
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            dt = float(tl.item())
        ovl_variant = {"value": world * args.steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / args.steps,
                       "note": "F[k] = f(x) evaluated on a second stream beside func_grad(y), which does not depend "
                               "on it: the latency-bound Cholesky of one runs under the MFMA-bound products of the "
                               "other; identical kernels and results (test_overlapped_value_*)"}
        f.overlap_values(False)

    # same workload once more with Gram-matrix reuse through linearity (extension, reported apart)
    lin_variant = None
    if (not args.no_variants) and (not args.linear_gram) and (not shard) and ipg == 1 \
            and args.workload in ("abpg_gain", "abpg"):
        f.linear_gram(True)
        if args.workload == "abpg_gain":
            gen2 = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        else:
            gen2 = alg.ABPG_steps(f, h, 1.0, x0, 2, total + 1, verbose=False)
        for _ in range(args.warmup):
            next(gen2)
        gl0, vh0 = f.gram_launches, f.value_hits
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            next(gen2)
        barrier()
        dt = time.perf_counter() - t1
        if world > 1:
            tl = torch.tensor([dt], dtype=torch.float64, device=device if use_nccl else "cpu")
            dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            dt = float(tl.item())
        lin_variant = {"value": world * args.steps / dt, "unit": "iterations/s", "ms_per_step": 1e3 * dt / args.steps,
                       "gram_launches_per_step": (f.gram_launches - gl0) / args.steps,
                       "repeated_value_lookups_per_step": (f.value_hits - vh0) / args.steps,
                       "note": "V diag(x) V^T is linear in x: Gram matrices at x, z stay resident and the ones at "
                               "y and x+ are O(m^2) combinations; f at a vector object already evaluated (the "
                               "line-search point, re-read as F[k+1]) is looked up; results equal to rounding "
                               "(test_linear_gram_*)"}
        f.linear_gram(False)

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if (world == 1 or use_nccl) else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())

    if rank == 0:
        value = (1 if shard else world * ipg) * args.steps / tmax
        out = {
            "metric": "D-opt iters/sec (m=%d,n=%d)" % (m, n),
            "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * tmax / args.steps, "higher_is_better": True,
            "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "D_opt_design(%d,%d) %s%s fp64, %s"
                                   % (m, n, args.workload, " gamma=2" if "bpg" in args.workload and args.workload != "bpg" else "",
                                      "ONE instance, design points sharded over the GPUs, one RCCL all-reduce of the Gram matrix per evaluation"
                                      if shard else "one independent instance per GPU"),
                       "instances": 1 if shard else world * ipg, "instances_per_gpu": ipg,
                       "seeds": "1..%d" % world,
                       "oracle_calls_per_step": {k: v / args.steps for k, v in calls.items()}},
        }
        gram_ms, gram_cnt = prof["gram"]
        grad_ms, grad_cnt = prof["grad"]
        kern = {k: {"ms_total": v[0], "launches": v[1], "ms_avg": (v[0] / v[1] if v[1] else None)}
                for k, v in prof.items()}
        out["kernels"] = kern
        if ovl_variant is not None:
            out["overlap_variant"] = ovl_variant
        if lin_variant is not None:
            out["linear_gram_variant"] = lin_variant
        out["config"]["linear_gram"] = bool(args.linear_gram)
        if args.workload in ("abpg_gain", "abpg", "bpg") and gram_cnt:
            # dominant kernel: Gram stream-K.  Algorithmic flops per launch = m^2 * n (SURVEY 8(d):
            # the SYRK share of 2 m^2 n + m^3/3 + 2 m n).
            flops = float(m) * m * n
            achieved = flops / (gram_ms / gram_cnt * 1e-3) * 1e-12
            traffic = None
            try:        # HBM bytes per launch from the committed PMC passes (same workload only)
                tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
                if (m, n) == (2048, 32768):
                    traffic = tj["gram_streamk_glds_kernel"]["hbm_bytes"]
            except Exception:
                traffic = None
            out["roofline"] = {"bound": "mfma", "kernel": "gram_streamk_glds_kernel (weighted Gram matrix, stream-K)",
                               "achieved": achieved,
                               "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                               "avg_launch_ms": gram_ms / gram_cnt, "launches": gram_cnt}
            if grad_cnt:
                ga = flops / (grad_ms / grad_cnt * 1e-3) * 1e-12
                out["roofline_grad_kernel"] = {"bound": "mfma", "kernel": "colnorm_glds_kernel (triangular product + column norms)", "achieved": ga,
                                               "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                               "frac": ga / PEAK_FP64_MFMA_TFLOPS,
                                               "avg_launch_ms": grad_ms / grad_cnt, "launches": grad_cnt}
        else:
            # Frank-Wolfe step: HBM bound, algorithmic bytes 8 m n + 24 m^2 + 48 n per step
            bytes_step = 8.0 * m * n + 24.0 * m * m + 48.0 * n
            achieved = bytes_step / (tmax / args.steps) * 1e-9
            out["roofline"] = {"bound": "hbm", "kernel": "fw step (whole step, host-timed)", "achieved": achieved,
                               "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                               "traffic": None}
        # measured fp64 MFMA peak on this device
        import ctypes as C
        from accbpg_and_fw_amd import _lib
        tf = C.c_double(0.0)
        _lib.load().accbpg_mfma_f64_peak(20000, C.byref(tf), None)
        out["mfma_f64_peak_measured_tflops"] = tf.value

        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, m, n)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, m, n):
    """The NumPy oracle (kind "port") on this box's host cores, bounded sample."""
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
    return {"value": iters / dt, "unit": "iterations/s", "cores": threads, "kind": "port",
            "sample": "%d outer iteration(s) of the NumPy oracle %s at (%d,%d), %.1f s wall%s" %
                      (iters, args.workload, m, n, dt,
                       " (includes the one-off O(m^2 n) setup)" if args.workload.startswith("fw") else ""),
            "numpy": np.__version__, "os_cpu_count": os.cpu_count()}


if __name__ == "__main__":
    main()
