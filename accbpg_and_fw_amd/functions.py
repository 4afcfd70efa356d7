"""Device-backed mirror of the f / h object protocol of the reference
(accbpg/functions.py:10-59, 199-271, 326-356).

Same class names, method names, argument meaning and exceptions, so the
reference's own solver loops (and this package's) can call them unchanged.
Vectors may be NumPy arrays (copied to the GPU and results copied back, for
drop-in use from NumPy drivers) or fp64 CUDA tensors (kept on the device, the
fast path used by this package's solvers).  All arithmetic runs in
libaccbpg_hip.so; torch only owns memory and streams.
"""
from __future__ import annotations

import ctypes as C
import threading

import numpy as np
import torch

from . import _lib


# ------------------------------------------------------------------ helpers
def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("accbpg_and_fw_amd needs an AMD GPU (torch.cuda is not available); "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(a):
    """-> (fp64 contiguous CUDA tensor, came_from_numpy)."""
    if isinstance(a, torch.Tensor):
        if a.dtype != torch.float64:
            a = a.to(torch.float64)
        if not a.is_cuda:
            return a.to(_device()).contiguous(), False
        return a.contiguous(), False
    arr = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return torch.from_numpy(arr).to(_device()), True


def from_dev(t, as_numpy):
    return t.cpu().numpy() if as_numpy else t


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _Workspace:
    """Scratch for the length-n kernels, one per (host thread, device, n).  Kept per thread (it goes away with
    the thread, so generations of pool workers do not pile up) and bounded per thread: the least recently
    used size is dropped beyond `_KEEP` entries."""
    _local = threading.local()
    _KEEP = 8

    @classmethod
    def get(cls, n, device):
        cache = getattr(cls._local, "cache", None)
        if cache is None:
            cache = cls._local.cache = {}
        key = (device.index, int(n))
        ws = cache.pop(key, None)
        if ws is None:
            size = _lib.load().accbpg_vec_workspace_doubles(int(n))
            ws = torch.empty(size, dtype=torch.float64, device=device)
            while len(cache) >= cls._KEEP:
                cache.pop(next(iter(cache)))
        cache[key] = ws                      # most recently used last
        return ws


# ------------------------------------------------------------------ f
class RSmoothFunction:
    """Relatively-smooth function protocol (accbpg/functions.py:10-24)."""

    def __call__(self, x):
        assert 0, "RSmoothFunction: __call__(x) is not defined"

    def gradient(self, x):
        assert 0, "RSmoothFunction: gradient(x) is not defined"

    def func_grad(self, x, flag):
        assert 0, "RSmoothFunction: func_grad(x, flag) is not defined"


class DOptimalObj(RSmoothFunction):
    """f(x) = -log det(H diag(x) H^T), H m x n with m < n  (accbpg/functions.py:27-59).

    ``H`` may be a NumPy array (copied to the GPU once) or an fp64 CUDA tensor
    (borrowed).  ``self.H``, ``self.m``, ``self.n`` stay readable as in the
    reference (callers use ``f.H``)."""

    def __init__(self, H, _shard=False, _borrowed=None):
        self.H = H
        self.m = H.shape[0]
        self.n = H.shape[1]
        # a column shard of a larger instance may have fewer columns than rows
        assert _shard or self.m < self.n, "DOptimalObj: need m < n"
        self._V, _ = to_dev(H)
        lib = _lib.load()
        if _borrowed is not None:
            # one instance of a DOptimalBatch: the handle belongs to the batch (kept alive through _owner)
            h, self._owner = _borrowed
            self._owned = False
        else:
            h = C.c_void_p()
            with torch.cuda.device(self._V.device):
                rc = lib.accbpg_dopt_create(_ptr(self._V), self.m, self.n, self._V.stride(0), _stream(), C.byref(h),
                                            1 if _shard else 0)
            _lib.check(rc, "accbpg_dopt_create")
            self._owned = True
        self._h = h
        self._lib = lib
        self.calls = {"value": 0, "grad": 0}
        # F[k] = f(x) of the accelerated solvers runs beside the gradient evaluation (see overlap_values())
        self._overlap = True
        self._spec = False
        self.spec_unused = 0
        self._prof = False
        # opt-in reuse of resident Gram matrices through linearity (see linear_gram())
        self._lin = False
        self._gcache = []           # [(vector tensor, Gram tensor, age)], most recent last
        self._gcache_cap = 5
        self._lin_refresh = 50
        self.gram_launches = 0
        self.gram_combos = 0
        self.value_hits = 0

    def __del__(self):
        for name in ("_h", "_h2"):
            h = getattr(self, name, None)
            if name == "_h" and not getattr(self, "_owned", True):
                continue
            if h:
                try:
                    self._lib.accbpg_dopt_destroy(h)
                except Exception:
                    pass
                setattr(self, name, None)

    # ---- a value evaluation that runs beside other work (second handle, second stream) ----
    def overlap_values(self, enable=True):
        """The accelerated solvers evaluate F[k] = f(x) on a side stream beside the gradient evaluation at y,
        which does not depend on it (accbpg/algorithms.py:135/148, :347/371): the latency-bound factorisation
        of one evaluation runs under the MFMA-bound products of the other.  Identical kernels and bit-identical
        results; on by default, ``overlap_values(False)`` puts both evaluations on the solver's stream."""
        self._overlap = bool(enable)
        return self

    def value_async(self, x):
        """Start f(x) on a side stream and return a ticket for ``value_wait``.  The accelerated
        solvers use it for F[k] = f(x), which the gradient evaluation at y does not depend on, so the
        latency-bound factorisation of one evaluation runs under the MFMA-bound products of the
        other.  Same kernels, same results as ``f(x)``."""
        assert x.numel() == self.n, "DOptimalObj: x.size not equal to n"
        self._side_handle()
        with torch.cuda.device(self._V.device):
            self._side.wait_stream(torch.cuda.current_stream())          # x is produced on the caller's stream
            rc = self._lib.accbpg_dopt_func_grad_begin(self._h2, _ptr(x), 0, None)
        _lib.check(rc, "accbpg_dopt_func_grad_begin")
        return x                                                          # the ticket keeps x alive

    def value_wait(self, ticket):
        fval = C.c_double(0.0)
        with torch.cuda.device(self._V.device):
            rc = self._lib.accbpg_dopt_func_grad_end(self._h2, C.byref(fval))
        _lib.check(rc, "accbpg_dopt_func_grad_end", "DOptimalObj: x needs to be nonnegative")
        self.calls["value"] += 1
        self._memo_put(ticket, fval.value)                      # (the ticket is the evaluated tensor)
        return fval.value

    # ---- a gradient evaluation started ahead of the decision that may need it (second handle, second stream) ----
    def speculate(self, enable=True):
        """Opt-in.  ABPG_gain's gain is cut before every search (accbpg/algorithms.py:358) and raised again on
        failure (:390), so once it has settled the search retries about once per iteration.  With this switch on the
        solver starts the gradient evaluation of the NEXT trial point -- which depends on host scalars and the
        previous iterates only -- on the side stream beside the value test of the current one, whenever the previous
        iteration needed that retry too; a trial that passes leaves the extra evaluation unused.  Every evaluation
        that IS used is the one the sequential loop would have made: results are bit-identical
        (test_gradients_started_ahead_change_nothing).  Off by default: the retry counts are irregular (0,1,1,2,...
        around an average of one), about a quarter of the evaluations started ahead go unused, and an unused one
        costs more than a used one saves -- measured 50.5 -> 47.7 it/s at (2048,32768), 562 -> 578 at (512,8192)."""
        self._spec = bool(enable)
        return self

    # ---- f at a vector object that was evaluated a moment ago (extension; off unless enabled) ----
    def memoize_values(self, enable=True):
        """Opt-in.  The reference evaluates f at the accepted line-search point twice: in the test that accepts it
        (accbpg/algorithms.py:387) and again as F[k+1] = f(x) at the top of the next iteration (:347).  With this switch
        on, f(x) at the very tensor object whose value was the last one computed is answered from that value instead of
        a second Gram product and factorisation -- the same number to the bit (the kernels are deterministic), one
        value evaluation fewer per ABPG_gain iteration (3 -> 2 in the steady state).  Off by default because it is not
        how the reference spends its time; bench.py reports it apart (`--memo-values`)."""
        self._memo_on = bool(enable)
        self._memo = None
        return self

    def _memo_get(self, x):
        memo = getattr(self, "_memo", None)
        if getattr(self, "_memo_on", False) and memo is not None and memo[0] is x and memo[1] == x._version:
            self.value_hits += 1
            self.calls["value"] += 1
            return memo[2]
        return None

    def _memo_put(self, x, value):
        if getattr(self, "_memo_on", False) and isinstance(x, torch.Tensor):
            self._memo = (x, x._version, value)                 # (holds x: its storage cannot be reused meanwhile)

    def _side_handle(self):
        if getattr(self, "_h2", None) is None:
            h2 = C.c_void_p()
            with torch.cuda.device(self._V.device):
                self._side = torch.cuda.Stream(device=self._V.device)
                rc = self._lib.accbpg_dopt_create(_ptr(self._V), self.m, self.n, self._V.stride(0),
                                                  C.c_void_p(self._side.cuda_stream), C.byref(h2), 1)
            _lib.check(rc, "accbpg_dopt_create")
            self._h2 = h2
            # its evaluations run beside the gradient evaluation of the solver's own stream: a launch per block
            # column leaves that stream its compute units (bit-identical results)
            self._lib.accbpg_dopt_factor_in_small_launches(h2, 2)
            if self._prof:
                self._lib.accbpg_dopt_profile_enable(self._h2, 1)
        return self._h2

    def grad_async(self, y):
        """Start func_grad(y, 2) on the side stream; returns a ticket for ``grad_wait`` / ``grad_drop``."""
        h2 = self._side_handle()
        g = torch.empty(self.n, dtype=torch.float64, device=self._V.device)
        with torch.cuda.device(self._V.device):
            self._side.wait_stream(torch.cuda.current_stream())          # y is produced on the caller's stream
            y.record_stream(self._side)
            g.record_stream(self._side)
            rc = self._lib.accbpg_dopt_func_grad_begin(h2, _ptr(y), 2, _ptr(g))
        _lib.check(rc, "accbpg_dopt_func_grad_begin")
        return (y, g)

    def grad_wait(self, ticket):
        y, g = ticket
        fval = C.c_double(0.0)
        with torch.cuda.device(self._V.device):
            rc = self._lib.accbpg_dopt_func_grad_end(self._h2, C.byref(fval))
            torch.cuda.current_stream().wait_stream(self._side)          # g is consumed on the caller's stream
        _lib.check(rc, "accbpg_dopt_func_grad_end", "DOptimalObj: x needs to be nonnegative")
        self.calls["grad"] += 1
        return fval.value, g

    def grad_drop(self, ticket):
        """The trial passed: the evaluation started ahead is not needed (it finishes on its own; its buffers are
        returned to the allocator only behind it)."""
        self.spec_unused += 1

    def value_lead_seconds(self):
        """How long before the last evaluation on the solver's stream the last side-stream value was known
        (0 when it was known later): what the solvers subtract so that T[k] is the moment F[k] existed."""
        ms = C.c_double(0.0)
        rc = self._lib.accbpg_dopt_eval_gap_ms(self._h2, self._h, C.byref(ms))
        return max(0.0, ms.value * 1e-3) if rc == _lib.OK else 0.0

    @property
    def device(self):
        return self._V.device

    @property
    def V_dev(self):
        return self._V

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)

    def func_grad(self, x, flag=2):
        """flag=0: function, flag=1: gradient, flag=2: function & gradient."""
        assert (x.numel() if isinstance(x, torch.Tensor) else x.size) == self.n, \
            "DOptimalObj: x.size not equal to n"
        if flag == 0 and isinstance(x, torch.Tensor):
            hit = self._memo_get(x)
            if hit is not None:
                return hit
        xd, was_np = to_dev(x)
        fval = C.c_double(0.0)
        g = None
        with torch.cuda.device(self._V.device):
            self._lib.accbpg_dopt_set_stream(self._h, _stream())
            if flag != 0:
                g = torch.empty(self.n, dtype=torch.float64, device=self._V.device)
            rc = self._lib.accbpg_dopt_func_grad(self._h, _ptr(xd), int(flag), C.byref(fval), _ptr(g))
        _lib.check(rc, "accbpg_dopt_func_grad", "DOptimalObj: x needs to be nonnegative")
        self.calls["value" if flag == 0 else "grad"] += 1
        if flag == 0:
            self._memo_put(x, fval.value)
            return fval.value
        g = from_dev(g, was_np)
        return g if flag == 1 else (fval.value, g)

    # ---- Gram-matrix reuse through linearity (extension; off unless enabled) ----
    def linear_gram(self, enable=True, refresh=50, capacity=5):
        """V diag(x) V^T is linear in x.  When enabled, the accelerated solvers of this package
        tell the objective that a point is a*x1 + b*x2 (``func_grad_combo``); if the Gram matrices at
        x1 and x2 are still resident, the O(m^2 n) Gram launch is replaced by an O(m^2) combination.
        A chain of combinations is cut every `refresh` links by a direct evaluation so rounding does
        not accumulate.  Results agree with direct evaluation to rounding (tests pin 1e-12)."""
        self._lin = bool(enable)
        self._lin_refresh = int(refresh)
        self._gcache_cap = int(capacity)
        self._gcache = []
        return self

    def _gram_lookup(self, vec):
        for idx, ent in enumerate(self._gcache):
            if ent[0] is vec:
                self._gcache.append(self._gcache.pop(idx))      # least recently used first
                return ent
        return None

    def ensure_gram(self, x):
        """Make the Gram matrix at the (fresh) device vector x resident."""
        with torch.cuda.device(self._V.device):
            self._lib.accbpg_dopt_set_stream(self._h, _stream())
            if self._gram_lookup(x) is None:
                self._gram_direct(x)

    def _gram_store(self, vec, gram, age):
        self._gcache = [e for e in self._gcache if e[0] is not vec]
        self._gcache.append([vec, gram, age, None])             # [vector, Gram, chain length, f value]
        if len(self._gcache) > self._gcache_cap:
            self._gcache.pop(0)

    def _gram_buffer(self):
        # recycle the buffer that is about to fall out of the cache
        if len(self._gcache) >= self._gcache_cap:
            return self._gcache.pop(0)[1]
        return torch.empty(self.m, self.m, dtype=torch.float64, device=self._V.device)

    def _gram_direct(self, xd):
        gram = self._gram_buffer()
        rc = self._lib.accbpg_dopt_gram(self._h, _ptr(xd), _ptr(gram))
        _lib.check(rc, "accbpg_dopt_gram")
        self.gram_launches += 1
        self._gram_store(xd, gram, 0)
        return gram

    def func_grad_combo(self, x, combo, flag=2):
        """func_grad(x, flag) where the caller states x = a*x1 + b*x2 (combo = (a, x1, b, x2)), or
        combo=None for a fresh point.  Device tensors only."""
        assert x.numel() == self.n, "DOptimalObj: x.size not equal to n"
        fval = C.c_double(0.0)
        g = None
        with torch.cuda.device(self._V.device):
            self._lib.accbpg_dopt_set_stream(self._h, _stream())
            ent = self._gram_lookup(x)
            if ent is not None:
                gram = ent[1]
            else:
                gram = None
                if combo is not None:
                    a, x1, b, x2 = combo
                    e1, e2 = self._gram_lookup(x1), self._gram_lookup(x2)
                    if e1 is not None and e2 is not None and max(e1[2], e2[2]) < self._lin_refresh:
                        g1, g2, age = e1[1], e2[1], max(e1[2], e2[2]) + 1   # read before a buffer is recycled
                        gram = self._gram_buffer()
                        rc = self._lib.accbpg_dopt_gram_lincomb(self._h, float(a), _ptr(g1), float(b), _ptr(g2),
                                                                _ptr(gram))
                        _lib.check(rc, "accbpg_dopt_gram_lincomb")
                        self.gram_combos += 1
                        self._gram_store(x, gram, age)
                if gram is None:
                    gram = self._gram_direct(x)
            ent = self._gram_lookup(x)
            if flag == 0 and ent is not None and ent[3] is not None:
                # the same (immutable) vector object was already evaluated: the solvers test f(x+) in
                # the line search and record F[k+1] = f(x+) at the top of the next iteration
                self.calls["value"] += 1
                self.value_hits += 1
                return ent[3]
            if flag != 0:
                g = torch.empty(self.n, dtype=torch.float64, device=self._V.device)
            # precondition of functions.py:45 (the staged path does not run the fused check)
            mn, _ = vec_min_sum(x)
            assert mn >= 0, "DOptimalObj: x needs to be nonnegative"
            rc = self._lib.accbpg_dopt_eval_gram(self._h, _ptr(gram), int(flag), C.byref(fval), _ptr(g))
        _lib.check(rc, "accbpg_dopt_eval_gram")
        self.calls["value" if flag == 0 else "grad"] += 1
        if ent is not None:
            ent[3] = fval.value
        if flag == 0:
            return fval.value
        return g if flag == 1 else (fval.value, g)

    # ---- staged evaluation for design-point sharding ----
    def gram_into(self, x_dev, gram_dev):
        with torch.cuda.device(self._V.device):
            self._lib.accbpg_dopt_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_gram(self._h, _ptr(x_dev), _ptr(gram_dev))
        _lib.check(rc, "accbpg_dopt_gram")

    def factor(self, gram_dev):
        fval = C.c_double(0.0)
        with torch.cuda.device(self._V.device):
            rc = self._lib.accbpg_dopt_factor(self._h, _ptr(gram_dev), C.byref(fval))
        _lib.check(rc, "accbpg_dopt_factor")
        return fval.value

    def grad_from_factor(self, g_dev):
        with torch.cuda.device(self._V.device):
            rc = self._lib.accbpg_dopt_grad(self._h, _ptr(g_dev))
        _lib.check(rc, "accbpg_dopt_grad")

    # ---- pieces used by D_opt_KYinit (accbpg/applications.py:79,84) ----
    def vt_times(self, q):
        """u = V^T q (np.dot(q, V)) for a length-m vector q -> length-n device vector."""
        qd, _ = to_dev(q)
        u = torch.empty(self.n, dtype=torch.float64, device=self._V.device)
        with torch.cuda.device(self._V.device):
            self._lib.accbpg_dopt_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_vt_times(self._h, _ptr(qd), _ptr(u))
        _lib.check(rc, "accbpg_dopt_vt_times")
        return u

    def column(self, j):
        """V[:, j] as a NumPy vector."""
        out = torch.empty(self.m, dtype=torch.float64, device=self._V.device)
        with torch.cuda.device(self._V.device):
            rc = self._lib.accbpg_dopt_get_column(self._h, int(j), _ptr(out))
        _lib.check(rc, "accbpg_dopt_get_column")
        return out.cpu().numpy()

    # ---- kernel-time accounting used by bench.py ----
    def _handles(self):
        return [h for h in (self._h, getattr(self, "_h2", None)) if h]

    def profile(self, enable=True):
        """HIP-event timing of every kernel family on the stream it is launched on.  While it is on, the
        solvers keep F[k] = f(x) on their own stream (no side-stream overlap), so that the durations are those
        of kernels that have the chip to themselves."""
        self._prof = bool(enable)
        for h in self._handles():
            self._lib.accbpg_dopt_profile_enable(h, 1 if enable else 0)
            self._lib.accbpg_dopt_profile_reset(h)

    def profile_read(self):
        out = {}
        for idx, name in enumerate(["gram", "cholesky", "trtri", "grad", "gram_fixup", "fw_vpass"]):
            tot, num = 0.0, 0
            for h in self._handles():
                ms, cnt = C.c_double(0.0), C.c_int64(0)
                self._lib.accbpg_dopt_profile_read(h, idx, C.byref(ms), C.byref(cnt))
                tot, num = tot + ms.value, num + cnt.value
            out[name] = (tot, num)
        return out


class PoissonRegression(RSmoothFunction):
    """f(x) = D_KL(b, Ax) for the linear inverse problem A x = b  (accbpg/functions.py:85-120).

    ``A`` (m x n) and ``b`` may be NumPy arrays (copied to the GPU once) or fp64 CUDA tensors
    (borrowed); ``self.A``, ``self.b``, ``self.m``, ``self.n`` stay readable as in the reference.
    Each func_grad is two passes over A: A x with the KL terms fused into its epilogue, then A^T r."""

    def __init__(self, A, b):
        assert A.shape[0] == b.shape[0], "A and b sizes not matching"
        self.A = A
        self.b = b
        self.m = A.shape[0]
        self.n = A.shape[1]
        self._A, _ = to_dev(A)
        self._b, _ = to_dev(b)
        lib = _lib.load()
        h = C.c_void_p()
        with torch.cuda.device(self._A.device):
            rc = lib.accbpg_poisson_create(_ptr(self._A), self.m, self.n, self._A.stride(0), _ptr(self._b),
                                           _stream(), C.byref(h))
        _lib.check(rc, "accbpg_poisson_create")
        self._h = h
        self._lib = lib

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.accbpg_poisson_destroy(h)
            except Exception:
                pass
            self._h = None

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)

    def func_grad(self, x, flag=2):
        size = x.numel() if isinstance(x, torch.Tensor) else x.size
        assert size == self.n, "PoissonRegression: x.size not equal to n."
        xd, was_np = to_dev(x)
        g = torch.empty(self.n, dtype=torch.float64, device=self._A.device) if flag != 0 else None
        fval = C.c_double(0.0)
        with torch.cuda.device(self._A.device):
            self._lib.accbpg_poisson_set_stream(self._h, _stream())
            rc = self._lib.accbpg_poisson_func_grad(self._h, _ptr(xd), int(flag), C.byref(fval), _ptr(g))
        _lib.check(rc, "accbpg_poisson_func_grad")
        if flag == 0:
            return fval.value
        if flag == 1:
            return from_dev(g, was_np)
        return fval.value, from_dev(g, was_np)

    def fitted(self):
        """A x of the last evaluation (length m) as a NumPy vector."""
        out = torch.empty(self.m, dtype=torch.float64, device=self._A.device)
        with torch.cuda.device(self._A.device):
            rc = self._lib.accbpg_poisson_get_ax(self._h, _ptr(out))
        _lib.check(rc, "accbpg_poisson_get_ax")
        return out.cpu().numpy()


# ------------------------------------------------------------------ h
class LegendreFunction:
    """Legendre kernel protocol (accbpg/functions.py:199-235)."""

    def __call__(self, x):
        assert 0, "LegendreFunction: __call__(x) is not defined."

    def extra_Psi(self, x):
        return 0

    def gradient(self, x):
        assert 0, "LegendreFunction: gradient(x) is not defined."

    def divergence(self, x, y):
        assert 0, "LegendreFunction: divergence(x,y) is not defined."

    def prox_map(self, g, L):
        assert 0, "LegendreFunction: prox_map(x, L) is not defined."

    def div_prox_map(self, y, g, L):
        assert y.shape == g.shape, "Vectors y and g should have same size."
        assert L > 0, "Relative smoothness constant L should be positive."
        return self.prox_map(g - L * self.gradient(y), L)


class BurgEntropy(LegendreFunction):
    """h(x) = -sum log x_i  (accbpg/functions.py:238-271)."""

    def __call__(self, x):
        # off the solver path (no solver calls h(x)); evaluated with torch on the device
        xd, _ = to_dev(x)
        assert float(xd.min()) > 0, "BurgEntropy only takes positive arguments."
        return float(-torch.log(xd).sum())

    def gradient(self, x):
        xd, was_np = to_dev(x)
        assert float(xd.min()) > 0, "BurgEntropy only takes positive arguments."
        return from_dev(-1.0 / xd, was_np)

    def divergence(self, x, y):
        assert x.shape == y.shape, "Vectors x and y are of different sizes."
        xd, _ = to_dev(x)
        yd, _ = to_dev(y)
        n = xd.numel()
        out = C.c_double(0.0)
        with torch.cuda.device(xd.device):
            ws = _Workspace.get(n, xd.device)
            rc = _lib.load().accbpg_burg_divergence(_ptr(xd), _ptr(yd), n, C.byref(out), _ptr(ws), _stream())
        _lib.check(rc, "accbpg_burg_divergence", "Entries of x or y not positive.")
        return np.float64(out.value)                            # a NumPy scalar, as the reference's sum is (:253)

    _kind = 0       # closed-form variant of accbpg_burg_reg_div_prox
    lamda = 0

    def _reg_prox(self, y, g, L):
        gd, was_np = to_dev(g)
        yd = None
        if y is not None:
            yd, _ = to_dev(y)
        out = torch.empty_like(gd)
        with torch.cuda.device(gd.device):
            rc = _lib.load().accbpg_burg_reg_div_prox(self._kind, _ptr(yd), _ptr(gd), float(L), float(self.lamda),
                                                      gd.numel(), _ptr(out), _stream())
        _lib.check(rc, "accbpg_burg_reg_div_prox")
        return from_dev(out, was_np)

    def prox_map(self, g, L):
        assert L > 0, "BurgEntropy prox_map only takes positive L value."
        return self._reg_prox(None, g, L)       # g.min() > 0 is checked on the device (functions.py:261)

    def div_prox_map(self, y, g, L):
        assert y.shape == g.shape, "Vectors y and g are of different sizes."
        assert L > 0, "Either y or L is not positive."
        return self._reg_prox(y, g, L)          # y.min() > 0 is checked on the device (functions.py:270)


class BurgEntropyL1(BurgEntropy):
    """Burg entropy for min_{x>0} f(x) + lamda*||x||_1  (accbpg/functions.py:274-298)."""
    _kind = 1

    def __init__(self, lamda=0, x_max=1e4):
        assert lamda >= 0, "BurgEntropyL1: lambda should be nonnegative."
        self.lamda = lamda
        self.x_max = x_max

    def extra_Psi(self, x):
        if self.lamda == 0:
            return 0.0
        xd, _ = to_dev(x)
        return self.lamda * vec_min_sum(xd)[1]

    def prox_map(self, g, L):
        assert L > 0, "BurgEntropyL1: prox_map only takes positive L."
        return self._reg_prox(None, g, L)       # g.min() > -lamda is checked on the device (:296)


class BurgEntropyL2(BurgEntropy):
    """Burg entropy for min_{x>0} f(x) + (lamda/2)*||x||_2^2  (accbpg/functions.py:301-323)."""
    _kind = 2

    def __init__(self, lamda=0):
        assert lamda >= 0, "BurgEntropyL2: lamda should be nonnegative."
        self.lamda = lamda

    def extra_Psi(self, x):
        if self.lamda == 0:
            return 0.0
        xd, _ = to_dev(x)
        return (self.lamda / 2) * vec_dot(xd, xd)

    def prox_map(self, g, L):
        assert L > 0, "BurgEntropyL2: prox_map only takes positive L value."
        return self._reg_prox(None, g, L)


class BurgEntropySimplex(BurgEntropy):
    r"""Burg entropy with the unit-simplex constraint (accbpg/functions.py:326-356)."""

    def __init__(self, eps=1e-8):
        assert eps > 0, "BurgEntropySimplex: eps should be positive."
        self.eps = eps
        self.last_info = (0, 0)     # (bisection steps, Newton steps) of the last prox

    def _prox(self, y, g, L):
        gd, was_np = to_dev(g)
        yd = None
        if y is not None:
            yd, _ = to_dev(y)
        n = gd.numel()
        out = torch.empty(n, dtype=torch.float64, device=gd.device)
        info = (C.c_int * 2)(0, 0)
        with torch.cuda.device(gd.device):
            ws = _Workspace.get(n, gd.device)
            rc = _lib.load().accbpg_burg_simplex_div_prox(_ptr(yd), _ptr(gd), float(L), float(self.eps), n,
                                                          _ptr(out), _ptr(ws), info, _stream())
        _lib.check(rc, "accbpg_burg_simplex_div_prox", "Either y or L is not positive.")
        self.last_info = (info[0], info[1])
        return from_dev(out, was_np)

    def prox_map(self, g, L):
        assert L > 0, "BergEntropySimplex prox_map only takes positive L."
        return self._prox(None, g, L)

    def div_prox_map(self, y, g, L):
        assert y.shape == g.shape, "Vectors y and g are of different sizes."
        assert L > 0, "Either y or L is not positive."
        return self._prox(y, g, L)


# ------------------------------------------------------------------ vector helpers for the solver loops
def vec_axpby(a, x, b, z):
    """a*x + b*z with NumPy's rounding (two products, one sum)."""
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().accbpg_vec_axpby(float(a), _ptr(x), float(b), _ptr(z), x.numel(), _ptr(out), _stream())
    _lib.check(rc, "accbpg_vec_axpby")
    return out


def vec_dot_diff(g, x, y):
    """<g, x - y>  (np.dot(g, x1-x), algorithms.py:53)."""
    out = C.c_double(0.0)
    with torch.cuda.device(x.device):
        ws = _Workspace.get(x.numel(), x.device)
        rc = _lib.load().accbpg_vec_dot_diff(_ptr(g), _ptr(x), _ptr(y), x.numel(), C.byref(out), _ptr(ws), _stream())
    _lib.check(rc, "accbpg_vec_dot_diff")
    return out.value


def vec_dot(x, y):
    """<x, y>  (np.dot(x, x) of BurgEntropyL2.extra_Psi, functions.py:314)."""
    out = C.c_double(0.0)
    with torch.cuda.device(x.device):
        ws = _Workspace.get(x.numel(), x.device)
        rc = _lib.load().accbpg_vec_dot(_ptr(x), _ptr(y), x.numel(), C.byref(out), _ptr(ws), _stream())
    _lib.check(rc, "accbpg_vec_dot")
    return out.value


def ls_terms(g, x, y, z=None, z1=None):
    """(<g,x-y>, D(x,y), D(z,z1)) in one launch and one readback."""
    out = (C.c_double * 3)(0.0, 0.0, 0.0)
    with torch.cuda.device(x.device):
        ws = _Workspace.get(x.numel(), x.device)
        rc = _lib.load().accbpg_ls_terms(_ptr(g), _ptr(x), _ptr(y), _ptr(z), _ptr(z1), x.numel(), out, _ptr(ws),
                                         _stream())
    _lib.check(rc, "accbpg_ls_terms", "Entries of x or y not positive.")
    # NumPy scalars, as the reference's sums are (accbpg/functions.py:253): D(x+,y) / D(z+,z) with D(z+,z) == 0 --
    # an iterate that has stopped moving -- is then inf or nan with a warning, as in the reference, not an exception
    # (the stopping rule dzz < epsilon right behind it ends the run, accbpg/algorithms.py:155,174)
    return np.float64(out[0]), np.float64(out[1]), np.float64(out[2])


def vec_min_sum(x):
    out = (C.c_double * 2)(0.0, 0.0)
    with torch.cuda.device(x.device):
        ws = _Workspace.get(x.numel(), x.device)
        rc = _lib.load().accbpg_vec_min_sum(_ptr(x), x.numel(), out, _ptr(ws), _stream())
    _lib.check(rc, "accbpg_vec_min_sum")
    return out[0], out[1]


def vec_div_scalar(x, d):
    """x / d elementwise (NumPy true division)."""
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        rc = _lib.load().accbpg_vec_div_scalar(_ptr(x), float(d), x.numel(), _ptr(out), _stream())
    _lib.check(rc, "accbpg_vec_div_scalar")
    return out


def vec_argminmax(x):
    """(first argmin, first argmax, min, max) of a device vector."""
    idx = (C.c_int64 * 2)(0, 0)
    val = (C.c_double * 2)(0.0, 0.0)
    with torch.cuda.device(x.device):
        ws = _Workspace.get(x.numel(), x.device)
        rc = _lib.load().accbpg_vec_argminmax(_ptr(x), x.numel(), idx, val, _ptr(ws), _stream())
    _lib.check(rc, "accbpg_vec_argminmax")
    return idx[0], idx[1], val[0], val[1]


def vec_vertex(idx, value, fill, n, device):
    out = torch.empty(n, dtype=torch.float64, device=device)
    with torch.cuda.device(device):
        rc = _lib.load().accbpg_vec_vertex(int(idx), float(value), float(fill), int(n), _ptr(out), _stream())
    _lib.check(rc, "accbpg_vec_vertex")
    return out
