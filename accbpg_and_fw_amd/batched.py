"""Batches of independent instances on one GPU (BASELINE config 4: many D_opt_design(512,8192)
instances per device).

Small instances are latency-bound (a 512 x 512 Cholesky is a chain of 512 pivots on a handful of
CUs), so one instance cannot fill the chip.  Instances are independent and take different
line-search / stopping paths, so instead of a masked lock-step batch each instance runs the
ordinary solver loop from its own host thread on its own HIP stream; kernels of different
instances overlap on the device.  ctypes releases the GIL while a call waits on its stream.
Every instance owns its D-optimal handle; the length-n kernels keep their scratch per thread.
"""
from __future__ import annotations

import threading
from concurrent.futures import ThreadPoolExecutor

import torch


def solve_instances(make_problem, num_instances, solver, world=1, rank=0, threads=None, concurrent=True,
                    group=None, **solver_kwargs):
    """BASELINE config 4 end to end: deal `num_instances` independent problems to the ranks
    (``split_instances``: round-robin, so instances that stop early spread out), solve this rank's share
    concurrently on its GPU, and gather every instance's result on every rank, in instance order.

    make_problem(i) -> (f, h, L, x0) builds instance i (only called for the instances of this rank);
    solver(f, h, L, x0, **solver_kwargs) is any solver of this package.  There is no data-path collective:
    the only communication is the result gather (torch.distributed all_gather_object over the process
    group that is already up -- RCCL ranks use their gloo/object path for this small host payload).
    `concurrent=False` runs the share sequentially on the calling thread (no streams; what the CPU tests of
    the dealing and gathering use with stand-in problems)."""
    from .sharded import split_instances
    mine = split_instances(num_instances, world, rank)
    problems = [make_problem(i) for i in mine]
    if concurrent:
        results = solve_batch(problems, solver, threads=threads, **solver_kwargs)
    else:
        results = [solver(*prob, **solver_kwargs) for prob in problems]
    local = {i: _to_host(res) for i, res in zip(mine, results)}
    if world > 1:
        import torch.distributed as dist
        shares = [None] * world
        dist.all_gather_object(shares, local, group=group)
        merged = {}
        for share in shares:
            merged.update(share)
    else:
        merged = local
    return [merged[i] for i in range(num_instances)]


def _to_host(result):
    """Solver results as host data (NumPy / floats) so that they can be shipped between ranks."""
    if isinstance(result, torch.Tensor):
        return result.detach().cpu().numpy()
    if isinstance(result, (tuple, list)):
        return type(result)(_to_host(r) for r in result)
    return result


def solve_batch(problems, solver, threads=None, **solver_kwargs):
    """Run ``solver(f, h, L, x0, **solver_kwargs)`` for every (f, h, L, x0) in `problems`
    concurrently; returns the list of results in order.  `solver` is any of BPG / ABPG /
    ABPG_gain (or D_opt_FW-style callables taking the same leading arguments)."""
    if not problems:
        return []
    threads = min(len(problems), threads or 8)
    device = problems[0][0].device

    def run(prob):
        f, h, L, x0 = prob
        stream = torch.cuda.Stream(device=device)
        with torch.cuda.device(device), torch.cuda.stream(stream):
            out = solver(f, h, L, x0, **solver_kwargs)
            stream.synchronize()
        return out

    with ThreadPoolExecutor(max_workers=threads) as pool:
        return list(pool.map(run, problems))


class BatchStepper:
    """Step generators of several instances advanced concurrently (used by bench.py): every call
    of ``step()`` advances each instance by one outer iteration."""

    def __init__(self, make_generators, device, threads=None):
        self.device = device
        self.n = len(make_generators)
        self.threads = min(self.n, threads or 8)
        self._gens = [None] * self.n
        self._streams = [torch.cuda.Stream(device=device) for _ in range(self.n)]
        self._pool = ThreadPoolExecutor(max_workers=self.threads)
        self._make = make_generators

    def _advance(self, idx, count):
        with torch.cuda.device(self.device), torch.cuda.stream(self._streams[idx]):
            if self._gens[idx] is None:
                self._gens[idx] = self._make[idx]()
            for _ in range(count):
                next(self._gens[idx])
            self._streams[idx].synchronize()
        return idx

    def step(self, count=1):
        list(self._pool.map(lambda i: self._advance(i, count), range(self.n)))

    def close(self):
        self._pool.shutdown()
