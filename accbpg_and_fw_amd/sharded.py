"""Multi-GPU forms of the D-optimal objective (no counterpart in the reference, which is
single-process).

1. Independent instances (BASELINE config 4): nothing to do here -- every rank builds its own
   ``DOptimalObj`` and runs the ordinary solvers; there is no data-path collective.
   ``split_instances`` only deals instance indices to ranks.

2. One large instance (BASELINE config 5): the design points (columns of V) are partitioned over
   the ranks.  ``H = sum_i x_i v_i v_i^T`` is a sum over design points, so per objective
   evaluation each rank forms the Gram contribution of its columns, ONE all-reduce (RCCL over
   xGMI) sums the m x m matrices, every rank factors the (replicated) sum, evaluates the gradient
   entries of its own columns, and one small all-reduce assembles the length-n gradient so that
   the Burg prox, divergences and dots of the solver loop run redundantly on full vectors with no
   further communication.  ``ShardedDOptimalObj`` has the f-protocol of ``DOptimalObj``
   (``__call__``, ``gradient``, ``func_grad``), so BPG / ABPG / ABPG_gain run on it unchanged.

The per-rank compute object is injectable (``local=``): on GPUs it is a ``DOptimalObj`` over the
local columns (HIP kernels); the CPU tests pass a NumPy stand-in so that the collective logic is
exercised with gloo at world_size 2 without a GPU.
"""
from __future__ import annotations

import numpy as np
import torch

from .functions import RSmoothFunction


def shard_bounds(n, world, rank):
    """Columns [lo, hi) of rank `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def split_instances(num_instances, world, rank):
    """Instance indices owned by `rank` (round-robin, so early-stopping instances spread out)."""
    return list(range(rank, num_instances, world))


class _DistSum:
    """sum over ranks through torch.distributed (backend nccl = RCCL on ROCm, or gloo on CPU)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def __call__(self, tensor):
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor


class ShardedDOptimalObj(RSmoothFunction):
    """f(x) = -log det(V diag(x) V^T) with the columns of V partitioned over ranks.

    local   : per-rank compute object over V[:, lo:hi] with the staged interface of
              ``DOptimalObj`` -- gram_into(x_local, gram), factor(gram) -> f, grad_from_factor(g_local)
    n       : total number of design points;  (lo, hi): this rank's columns
    reduce  : callable summing a tensor over ranks in place (default: torch.distributed all-reduce)
    """

    def __init__(self, local, m, n, lo, hi, device, reduce=None):
        self.local = local
        self.m, self.n = int(m), int(n)
        self.lo, self.hi = int(lo), int(hi)
        self.device = device
        self.reduce = reduce if reduce is not None else _DistSum()
        self._gram = torch.zeros(self.m, self.m, dtype=torch.float64, device=device)
        self.calls = {"value": 0, "grad": 0}
        self.H = None           # the full design matrix is not resident on any single rank

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)

    def func_grad(self, x, flag=2):
        assert x.numel() == self.n, "DOptimalObj: x.size not equal to n"
        x_local = x[self.lo:self.hi].contiguous()
        self.local.gram_into(x_local, self._gram)
        self.reduce(self._gram)                                  # the one Gram all-reduce
        fval = self.local.factor(self._gram)                     # replicated Cholesky + log det
        self.calls["value" if flag == 0 else "grad"] += 1
        if flag == 0:
            return fval
        g = torch.zeros(self.n, dtype=torch.float64, device=self.device)
        g_local = torch.empty(self.hi - self.lo, dtype=torch.float64, device=self.device)
        self.local.grad_from_factor(g_local)
        g[self.lo:self.hi] = g_local
        self.reduce(g)                                           # assemble the full gradient
        return g if flag == 1 else (fval, g)


class LogicalShards:
    """P shards of one instance on ONE device, the all-reduce replaced by an in-process sum: the
    test mode for the sharded arithmetic where fewer than two GPUs are visible."""

    def __init__(self, V, parts):
        from .functions import DOptimalObj, to_dev
        Vd, _ = to_dev(V)
        self.m, self.n = Vd.shape
        self.device = Vd.device
        self.bounds = [shard_bounds(self.n, parts, r) for r in range(parts)]
        self.objs = [DOptimalObj(Vd[:, lo:hi].contiguous(), _shard=True) for lo, hi in self.bounds]
        self.grams = [torch.zeros(self.m, self.m, dtype=torch.float64, device=self.device) for _ in self.objs]

    def func_grad(self, x, flag=2):
        from .functions import to_dev, from_dev
        xd, was_np = to_dev(x)
        for (lo, hi), obj, gram in zip(self.bounds, self.objs, self.grams):
            obj.gram_into(xd[lo:hi].contiguous(), gram)
        total = self.grams[0].clone()
        for gram in self.grams[1:]:
            total += gram                                        # stands in for the all-reduce
        fvals = [obj.factor(total) for obj in self.objs]         # every "rank" factors the same sum
        if flag == 0:
            return fvals[0]
        g = torch.empty(self.n, dtype=torch.float64, device=self.device)
        for (lo, hi), obj in zip(self.bounds, self.objs):
            gl = torch.empty(hi - lo, dtype=torch.float64, device=self.device)
            obj.grad_from_factor(gl)
            g[lo:hi] = gl
        g = from_dev(g, was_np)
        return g if flag == 1 else (fvals[0], g)

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)


def make_sharded(V_local, m, n, rank, world, device=None, group=None):
    """GPU construction helper: V_local = V[:, lo:hi] of this rank (NumPy or CUDA tensor)."""
    from .functions import DOptimalObj
    lo, hi = shard_bounds(n, world, rank)
    assert V_local.shape == (m, hi - lo), "V_local must hold this rank's columns"
    local = DOptimalObj(V_local, _shard=True)
    return ShardedDOptimalObj(local, m, n, lo, hi, local.device, reduce=_DistSum(group))
