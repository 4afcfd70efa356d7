"""Multi-GPU forms of the D-optimal objective (no counterpart in the reference, which is
single-process).

1. Independent instances (BASELINE config 4): nothing to do here -- every rank builds its own
   ``DOptimalObj`` and runs the ordinary solvers; there is no data-path collective.
   ``split_instances`` only deals instance indices to ranks.

2. One large instance (BASELINE config 5): the design points (columns of V) are partitioned over
   the ranks.  ``H = sum_i x_i v_i v_i^T`` is a sum over design points, so per objective
   evaluation each rank forms the Gram contribution of its columns, ONE all-reduce (RCCL over
   xGMI) sums them -- as packed lower triangles, m(m+1)/2 doubles, with the count of entries of the
   local x that violate x >= 0 riding as one more element -- every rank factors the (replicated)
   sum, evaluates the gradient entries of its own columns, and one all-gather of the slices
   assembles the length-n gradient so that the Burg prox, divergences and dots of the solver loop
   run redundantly on full vectors with no further communication.  ``ShardedDOptimalObj`` has the f-protocol of ``DOptimalObj``
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
    """sum over ranks through torch.distributed (backend nccl = RCCL on ROCm, or gloo on CPU);
    ``gather(piece, out)`` concatenates equal-length pieces of all ranks into `out`."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group

    def __call__(self, tensor):
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor

    def gather(self, piece, out):
        self.dist.all_gather_into_tensor(out, piece, group=self.group)
        return out


class _DeviceMessages:
    """Message forms of the per-rank compute object on a GPU (libaccbpg_hip.so): triangle packing and the
    x >= 0 violation count, all on the current stream without a host round trip."""

    @staticmethod
    def pack(gram, packed):
        from . import _lib
        from .functions import _ptr, _stream
        with torch.cuda.device(gram.device):
            _lib.check(_lib.load().accbpg_tri_pack(_ptr(gram), gram.shape[0], _ptr(packed), _stream()), "accbpg_tri_pack")

    @staticmethod
    def unpack(packed, gram):
        from . import _lib
        from .functions import _ptr, _stream
        with torch.cuda.device(gram.device):
            _lib.check(_lib.load().accbpg_tri_unpack(_ptr(packed), gram.shape[0], _ptr(gram), _stream()),
                       "accbpg_tri_unpack")

    @staticmethod
    def count_bad(x_local, slot):
        from . import _lib
        from .functions import _ptr, _stream
        with torch.cuda.device(x_local.device):
            _lib.check(_lib.load().accbpg_vec_count_bad(_ptr(x_local), x_local.numel(), _ptr(slot), _stream()),
                       "accbpg_vec_count_bad")


class ShardedDOptimalObj(RSmoothFunction):
    """f(x) = -log det(V diag(x) V^T) with the columns of V partitioned over ranks.

    local   : per-rank compute object over V[:, lo:hi] with the staged interface of
              ``DOptimalObj`` -- gram_into(x_local, gram), factor(gram) -> f, grad_from_factor(g_local)
    n       : total number of design points;  (lo, hi): this rank's columns
    reduce  : callable summing a tensor over ranks in place (default: torch.distributed all-reduce); if it
              has a ``gather(piece, out)`` method the gradient slices are all-gathered, otherwise the
              zero-padded gradient is summed
    messages: pack / unpack / count_bad of the message buffer (default: the HIP kernels; the CPU tests pass
              a NumPy stand-in)
    """

    def __init__(self, local, m, n, lo, hi, device, reduce=None, messages=None, world=None):
        self.local = local
        self.m, self.n = int(m), int(n)
        self.lo, self.hi = int(lo), int(hi)
        self.device = device
        self.reduce = reduce if reduce is not None else _DistSum()
        self.msg = messages if messages is not None else _DeviceMessages
        self._gram = torch.zeros(self.m, self.m, dtype=torch.float64, device=device)
        # the one message per evaluation: packed lower triangle + the x >= 0 violation count (padded to a
        # whole number of 16-byte pieces)
        self._tri = self.m * (self.m + 1) // 2
        self._msg = torch.zeros(self._tri + 2 - (self._tri & 1), dtype=torch.float64, device=device)
        self.world = world
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
        self.msg.pack(self._gram, self._msg)
        self.msg.count_bad(x_local, self._msg[self._tri:self._tri + 1])
        self.reduce(self._msg)                                   # the one Gram all-reduce (packed triangle)
        # accbpg/functions.py:45 on the whole vector: every rank sees the sum of every rank's violations
        # and raises together, before anything is factored
        assert float(self._msg[self._tri]) == 0.0, "DOptimalObj: x needs to be nonnegative"
        self.msg.unpack(self._msg, self._gram)
        fval = self.local.factor(self._gram)                     # replicated Cholesky + log det
        self.calls["value" if flag == 0 else "grad"] += 1
        if flag == 0:
            return fval
        g_local = torch.empty(self.hi - self.lo, dtype=torch.float64, device=self.device)
        self.local.grad_from_factor(g_local)
        g = self._assemble(g_local)
        return g if flag == 1 else (fval, g)

    def _assemble(self, g_local):
        """Full gradient on every rank from the slices: one all-gather (slices padded to the longest, which
        is at most one entry more), or a sum of zero-padded vectors when the reducer cannot gather."""
        world = self.world
        if world is None or not hasattr(self.reduce, "gather"):
            g = torch.zeros(self.n, dtype=torch.float64, device=self.device)
            g[self.lo:self.hi] = g_local
            return self.reduce(g)
        width = -(-self.n // world)
        piece = torch.zeros(width, dtype=torch.float64, device=self.device)
        piece[:self.hi - self.lo] = g_local
        parts = torch.empty(world * width, dtype=torch.float64, device=self.device)
        self.reduce.gather(piece, parts)
        g = torch.empty(self.n, dtype=torch.float64, device=self.device)
        for r in range(world):
            lo, hi = shard_bounds(self.n, world, r)
            g[lo:hi] = parts[r * width:r * width + hi - lo]
        return g


class NativeShardedDOptimalObj(RSmoothFunction):
    """The sharded objective with the collectives inside libaccbpg_hip.so (accbpg_dopt_shard_*, RCCL opened by
    the library): one C call per evaluation.  Same f-protocol and results as ``ShardedDOptimalObj`` on GPUs.

    V_local : this rank's columns V[:, lo:hi], (lo, hi) = shard_bounds(n, world, rank)
    token   : the bytes of ``native_unique_id()`` made on rank 0 and handed to every rank
              (``exchange_token`` does that through torch.distributed)
    """

    def __init__(self, V_local, n, world, rank, token):
        import ctypes as C
        from . import _lib
        from .functions import DOptimalObj
        self._lib = _lib.load()
        self.local = DOptimalObj(V_local, _shard=True)
        self.m, self.n = self.local.m, int(n)
        self.lo, self.hi = shard_bounds(n, world, rank)
        assert self.local.n == self.hi - self.lo, "V_local must hold this rank's columns"
        self.device = self.local.device
        self.world, self.rank = int(world), int(rank)
        self.calls = {"value": 0, "grad": 0}
        self.H = None
        buf = (C.c_ubyte * len(token)).from_buffer_copy(bytes(token))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.accbpg_dopt_shard_create(self.local._h, self.n, self.world, self.rank, buf, None, C.byref(h))
        _lib.check(rc, "accbpg_dopt_shard_create")
        self._s = h

    def __del__(self):
        s, self._s = getattr(self, "_s", None), None
        if s:
            self._lib.accbpg_dopt_shard_destroy(s)

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)

    def func_grad(self, x, flag=2):
        import ctypes as C
        from . import _lib
        from .functions import _ptr, _stream, to_dev, from_dev
        xd, was_np = to_dev(x)
        assert xd.numel() == self.n, "DOptimalObj: x.size not equal to n"
        fval = C.c_double(0.0)
        g = torch.empty(self.n, dtype=torch.float64, device=self.device) if flag != 0 else None
        with torch.cuda.device(self.device):
            self._lib.accbpg_dopt_set_stream(self.local._h, _stream())
            rc = self._lib.accbpg_dopt_shard_func_grad(self._s, _ptr(xd), flag, C.byref(fval),
                                                       _ptr(g) if g is not None else None)
        _lib.check(rc, "accbpg_dopt_shard_func_grad")
        self.calls["value" if flag == 0 else "grad"] += 1
        if flag == 0:
            return fval.value
        g = from_dev(g, was_np)
        return g if flag == 1 else (fval.value, g)


def native_unique_id():
    """Rendezvous token of the library's own RCCL communicator (bytes; make it on rank 0)."""
    import ctypes as C
    from . import _lib
    buf = (C.c_ubyte * 128)()
    _lib.check(_lib.load().accbpg_shard_unique_id(buf), "accbpg_shard_unique_id")
    return bytes(buf)


def exchange_token(rank, group=None):
    """Rank 0 makes the token, torch.distributed hands it to everyone."""
    import torch.distributed as dist
    box = [native_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


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
        m = self.m
        tri = m * (m + 1) // 2
        msgs = []
        for (lo, hi), obj, gram in zip(self.bounds, self.objs, self.grams):
            piece = xd[lo:hi].contiguous()
            obj.gram_into(piece, gram)
            msg = torch.zeros(tri + 2 - (tri & 1), dtype=torch.float64, device=self.device)
            _DeviceMessages.pack(gram, msg)                      # the same message a rank would send
            _DeviceMessages.count_bad(piece, msg[tri:tri + 1])
            msgs.append(msg)
        summed = msgs[0]
        for msg in msgs[1:]:
            summed += msg                                        # stands in for the all-reduce
        assert float(summed[tri]) == 0.0, "DOptimalObj: x needs to be nonnegative"
        total = torch.zeros(m, m, dtype=torch.float64, device=self.device)
        _DeviceMessages.unpack(summed, total)
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
    return ShardedDOptimalObj(local, m, n, lo, hi, local.device, reduce=_DistSum(group), world=world)


def make_sharded_native(V_local, m, n, rank, world, group=None):
    """As make_sharded, with the collectives inside the library (its own RCCL communicator)."""
    lo, hi = shard_bounds(n, world, rank)
    assert V_local.shape == (m, hi - lo), "V_local must hold this rank's columns"
    return NativeShardedDOptimalObj(V_local, n, world, rank, exchange_token(rank, group))
