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

import ctypes as C
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from .functions import DOptimalObj, _ptr, _stream, to_dev


class DOptimalBatch:
    """K D-optimal objectives of ONE shape evaluated together (C-ABI ``accbpg_dopt_batch_*``): every call covers the
    active instances with one launch per kernel family and one readback.  Vectors are K x n CUDA tensors, row i =
    instance i.  The arithmetic of instance i is bit for bit that of ``self.instance(i)`` (a ``DOptimalObj`` on the
    same handle), so a lock-step batch and a loop over the instances give identical results."""

    def __init__(self, matrices):
        self._Vs = [to_dev(V)[0] for V in matrices]
        assert self._Vs, "DOptimalBatch: no instances"
        self.K = len(self._Vs)
        self.m, self.n = self._Vs[0].shape
        for V in self._Vs:
            assert V.shape == (self.m, self.n) and V.stride(0) == self._Vs[0].stride(0), \
                "DOptimalBatch: instances must have one shape and one row stride"
        self.device = self._Vs[0].device
        self._lib = _lib.load()
        ptrs = (C.c_void_p * self.K)(*[V.data_ptr() for V in self._Vs])
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self._lib.accbpg_dopt_batch_create(ptrs, self.K, self.m, self.n, self._Vs[0].stride(0), _stream(),
                                                    C.byref(h))
        _lib.check(rc, "accbpg_dopt_batch_create")
        self._h = h
        self.fused = bool(self._lib.accbpg_dopt_batch_is_fused(h))
        self.chunk = int(self._lib.accbpg_dopt_batch_chunk(h))      # instances one launch covers
        self.calls = {"value": 0, "grad": 0}        # per instance-evaluation, as DOptimalObj counts them
        self._views = {}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self._lib.accbpg_dopt_batch_destroy(h)
            except Exception:
                pass
            self._h = None

    def instance(self, i):
        """Instance i as an ordinary objective on the batch's own handle (same plans, same kernels)."""
        if i not in self._views:
            sub = C.c_void_p(self._lib.accbpg_dopt_batch_instance(self._h, int(i)))
            view = DOptimalObj(self._Vs[i], _borrowed=(sub, self))
            view.overlap_values(False)
            self._views[i] = view
        return self._views[i]

    # ---- kernel-time accounting used by bench.py (rides on instance 0's slots of this batch and of its twin) ----
    def _handles0(self):
        hs = [C.c_void_p(self._lib.accbpg_dopt_batch_instance(self._h, 0))]
        twin = getattr(self, "_twin", None)
        if twin is not None:
            hs.append(C.c_void_p(self._lib.accbpg_dopt_batch_instance(twin._h, 0)))
        return hs

    def profile(self, enable=True):
        """HIP-event timing of every batched launch on the stream it is launched on (one event pair per kernel family
        and launch)."""
        self._prof = bool(enable)
        for h in self._handles0():
            self._lib.accbpg_dopt_profile_enable(h, 1 if enable else 0)
            self._lib.accbpg_dopt_profile_reset(h)

    def profile_read(self):
        out = {}
        for idx, name in enumerate(["gram", "cholesky", "trtri", "grad", "gram_fixup", "fw_vpass"]):
            tot, num = 0.0, 0
            for h in self._handles0():
                ms, cnt = C.c_double(0.0), C.c_int64(0)
                self._lib.accbpg_dopt_profile_read(h, idx, C.byref(ms), C.byref(cnt))
                tot, num = tot + ms.value, num + cnt.value
            out[name] = (tot, num)
        return out

    def _mask(self, active):
        if active is None:
            return None, list(range(self.K))
        arr = (C.c_int * self.K)(*[1 if a else 0 for a in active])
        return arr, [i for i in range(self.K) if active[i]]

    @staticmethod
    def _raise(status, idx, what, assert_msg):
        for i in idx:
            if status[i] != _lib.OK:
                _lib.check(status[i], "%s (instance %d)" % (what, i), assert_msg)

    def func_grad(self, X, flag=2, active=None, out=None):
        """(f, G): f a length-K NumPy array (entries of inactive instances are nan), G a K x n tensor (rows of
        inactive instances are left unwritten; `out`: an existing tensor for them); flag 0 returns f only, flag 1 G
        only."""
        assert X.shape == (self.K, self.n) and X.is_cuda and X.dtype == torch.float64 and X.stride(1) == 1
        mask, idx = self._mask(active)
        f = (C.c_double * self.K)(*([float("nan")] * self.K))
        st = (C.c_int * self.K)()
        G = None
        if flag != 0:
            G = out if out is not None else torch.empty(self.K, self.n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._lib.accbpg_dopt_batch_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_batch_func_grad(self._h, _ptr(X), X.stride(0), mask, int(flag), f, _ptr(G),
                                                       self.n, st)
        _lib.check(rc, "accbpg_dopt_batch_func_grad")
        self._raise(st, idx, "accbpg_dopt_batch_func_grad", "DOptimalObj: x needs to be nonnegative")
        self.calls["value" if flag == 0 else "grad"] += len(idx)
        fv = np.array(f[:], dtype=np.float64)
        if flag == 0:
            return fv
        return G if flag == 1 else (fv, G)

    # ---- value evaluations beside the gradient evaluations: a second batch over the same matrices, own stream ----
    def value_async(self, X, active=None):
        """Start f(X[i]) for the active instances on a side stream (a twin batch over the same matrices) and
        return a ticket for ``value_wait``.  Same kernels, same plans: the values are those of ``func_grad(X, 0)``."""
        twin = getattr(self, "_twin", None)
        if twin is None:
            twin = self._twin = DOptimalBatch(self._Vs)
            with torch.cuda.device(self.device):
                self._side = torch.cuda.Stream(device=self.device)
        mask, idx = self._mask(active)
        with torch.cuda.device(self.device):
            self._side.wait_stream(torch.cuda.current_stream())          # X is produced on the caller's stream
            self._lib.accbpg_dopt_batch_set_stream(twin._h, C.c_void_p(self._side.cuda_stream))
            rc = self._lib.accbpg_dopt_batch_func_grad_begin(twin._h, _ptr(X), X.stride(0), mask, 0, None, self.n)
        _lib.check(rc, "accbpg_dopt_batch_func_grad_begin")
        return (X, idx)                                                   # the ticket keeps X alive

    def value_wait(self, ticket):
        _, idx = ticket
        f = (C.c_double * self.K)(*([float("nan")] * self.K))
        st = (C.c_int * self.K)()
        with torch.cuda.device(self.device):
            rc = self._lib.accbpg_dopt_batch_func_grad_end(self._twin._h, f, st)
        _lib.check(rc, "accbpg_dopt_batch_func_grad_end")
        self._raise(st, idx, "accbpg_dopt_batch_func_grad", "DOptimalObj: x needs to be nonnegative")
        self.calls["value"] += len(idx)
        return np.array(f[:], dtype=np.float64)

    def prox(self, Y, G, Ls, eps, active=None, out=None):
        """Row i: BurgEntropySimplex(eps).div_prox_map(Y[i], G[i], Ls[i]) (Y None: prox_map).  Only the rows of the
        active instances are written (`out`: an existing K x n tensor to write them into)."""
        mask, idx = self._mask(active)
        Lc = (C.c_double * self.K)(*[float(v) for v in Ls])
        st = (C.c_int * self.K)()
        if out is None:
            out = torch.empty(self.K, self.n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._lib.accbpg_dopt_batch_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_batch_burg_simplex_div_prox(self._h, _ptr(Y), _ptr(G), self.n, Lc, float(eps),
                                                                   _ptr(out), mask, st, None)
        _lib.check(rc, "accbpg_dopt_batch_burg_simplex_div_prox")
        self._raise(st, idx, "accbpg_dopt_batch_burg_simplex_div_prox", "Either y or L is not positive.")
        return out

    def ls_terms(self, G, X, Y, Z=None, Z1=None, active=None):
        """K x 3 NumPy array: (<g, x-y>, D(x,y), D(z,z1)) per instance."""
        mask, idx = self._mask(active)
        out = (C.c_double * (3 * self.K))()
        st = (C.c_int * self.K)()
        with torch.cuda.device(self.device):
            self._lib.accbpg_dopt_batch_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_batch_ls_terms(self._h, _ptr(G), _ptr(X), _ptr(Y), _ptr(Z), _ptr(Z1), self.n, mask,
                                                      out, st)
        _lib.check(rc, "accbpg_dopt_batch_ls_terms")
        self._raise(st, idx, "accbpg_dopt_batch_ls_terms", "Entries of x or y not positive.")
        return np.array(out[:], dtype=np.float64).reshape(self.K, 3)

    def axpby(self, a, X, b, Z, active=None, out=None):
        """Row i: a[i]*X[i] + b[i]*Z[i] with NumPy's rounding; only the rows of the active instances are written."""
        mask, _ = self._mask(active)
        av = (C.c_double * self.K)(*[float(v) for v in a])
        bv = (C.c_double * self.K)(*[float(v) for v in b])
        if out is None:
            out = torch.empty(self.K, self.n, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            self._lib.accbpg_dopt_batch_set_stream(self._h, _stream())
            rc = self._lib.accbpg_dopt_batch_axpby(self._h, av, _ptr(X), bv, _ptr(Z), self.n, mask, _ptr(out))
        _lib.check(rc, "accbpg_dopt_batch_axpby")
        return out


def ABPG_batch_steps(batch, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=False, restart=False, restart_rule='g',
                     overlap=True):
    """ABPG (accbpg/algorithms.py:94-180) on the K instances of a ``DOptimalBatch`` in lock-step: every oracle call,
    prox and vector pass covers all instances that are still running.  theta, the restart state and the stopping test
    are kept per instance exactly as the sequential solver keeps them; an instance that stops (D(z+,z) < epsilon)
    drops out of the later launches.  Yields k after every outer iteration; returns, per instance, ABPG's
    (x, F, G, T) -- bit-identical to ``ABPG(batch.instance(i), h, L, x0, ...)``.  With `overlap` (default) the values
    F[k] = f(x_i) run on a second stream beside the gradient evaluations at y_i, which do not depend on them."""
    from .algorithms import solve_theta
    K, n = batch.K, batch.n
    t_start = time.time()
    x0d, as_numpy = to_dev(x0)
    X = x0d.reshape(1, -1).repeat(K, 1).contiguous() if x0d.dim() == 1 else x0d.clone().contiguous()
    assert X.shape == (K, n)
    Z = X.clone()
    F = np.zeros((K, maxitrs)); G = np.zeros((K, maxitrs)); T = np.zeros((K, maxitrs))
    theta = [1.0] * K
    kk = [0] * K
    active = [True] * K
    last = [-1] * K                      # last iteration each instance ran
    result_x = [None] * K
    eps_prox = getattr(h, "eps", 1e-8)
    for k in range(maxitrs):
        if not any(active):
            break
        side = overlap and not getattr(batch, "_prof", False)                # (kernel timing keeps everything on one stream)
        ticket = batch.value_async(X, active) if side else None             # :135 (beside the gradients below)
        fx = None if side else batch.func_grad(X, 0, active)
        now = time.time() - t_start
        for i in range(K):
            if active[i]:
                if theta_eq and kk[i] > 0:                                  # :142-145
                    theta[i] = solve_theta(theta[i], gamma)
                else:
                    theta[i] = gamma / (kk[i] + gamma)
        one_m = [1 - t for t in theta]
        Y = batch.axpby(one_m, X, theta, Z, active)                         # :147
        Gr = batch.func_grad(Y, 1, active)                                  # :148
        if side:
            fx = batch.value_wait(ticket)
            now = time.time() - t_start
        for i in range(K):
            if active[i]:
                F[i, k] = fx[i] + h.extra_Psi(None)
                T[i, k] = now
        Zn = batch.prox(Z, Gr, [t ** (gamma - 1) * L for t in theta], eps_prox, active)    # :149
        Xn = batch.axpby(one_m, X, theta, Zn, active)                       # :150
        terms = batch.ls_terms(None, Xn, Y, Zn, Z, active)                  # :153-154
        rest = None
        if restart and restart_rule == 'g' and k > 0:
            rest = batch.ls_terms(Gr, Xn, X, None, None, active)[:, 0]      # <g, x - x_1>, :168
        Znext = Zn
        for i in range(K):
            if not active[i]:
                continue
            dxy, dzz = terms[i, 1], terms[i, 2]
            G[i, k] = dxy / dzz / theta[i] ** gamma                         # :155
            last[i] = k
            kk[i] += 1
            if restart and k > 0:                                           # :165-171
                if (restart_rule == 'f' and F[i, k] > F[i, k - 1]) or (restart_rule == 'g' and rest[i] > 0):
                    theta[i] = 1.0
                    kk[i] = 0
                    if Znext is Zn:
                        Znext = Zn.clone()
                    Znext[i] = Xn[i]
            if dzz < epsilon:                                               # :174
                active[i] = False
                result_x[i] = Xn[i].clone()
        X, Z = Xn, Znext
        yield k
    out = []
    for i in range(K):
        xi = result_x[i] if result_x[i] is not None else X[i].clone()
        xi = xi.cpu().numpy() if as_numpy else xi
        out.append((xi, F[i, :last[i] + 1].copy(), G[i, :last[i] + 1].copy(), T[i, :last[i] + 1].copy()))
    return out


def ABPG_batch(batch, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=False, restart=False, restart_rule='g',
               overlap=True):
    """Drain ``ABPG_batch_steps``: list of (x, F, G, T), one per instance."""
    gen = ABPG_batch_steps(batch, h, L, x0, gamma, maxitrs, epsilon, theta_eq, restart, restart_rule, overlap)
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value


def BPG_batch_steps(batch, h, L, x0, maxitrs, epsilon=1e-14, linesearch=True, ls_ratio=1.2):
    """BPG (accbpg/algorithms.py:11-72) on the K instances of a ``DOptimalBatch`` in lock-step per ORACLE PASS: one
    evaluation of (f, grad) at the iterates of all running instances, then one prox + one trial value per pass of the
    backtracking search, for the instances that are still searching.  L is carried per instance (divided by ls_ratio
    before each search, :51; multiplied after each failed test, :54); an instance stops when |F[k] - F[k-1]| <
    epsilon (:66) and drops out of the later launches.  Yields k after every outer iteration; returns, per instance,
    BPG's (x, F, Ls, T) -- bit-identical to ``BPG(batch.instance(i), h, L, x0, ...)``."""
    K, n = batch.K, batch.n
    t_start = time.time()
    x0d, as_numpy = to_dev(x0)
    X = x0d.reshape(1, -1).repeat(K, 1).contiguous() if x0d.dim() == 1 else x0d.clone().contiguous()
    assert X.shape == (K, n)
    F = np.zeros((K, maxitrs)); Ls = np.ones((K, maxitrs)) * L; T = np.zeros((K, maxitrs))
    Lc = [L] * K
    active = [True] * K
    last = [-1] * K
    result_x = [None] * K
    eps_prox = getattr(h, "eps", 1e-8)
    for k in range(maxitrs):
        if not any(active):
            break
        fx, Gr = batch.func_grad(X, 2, active)                              # :46
        now = time.time() - t_start
        for i in range(K):
            if active[i]:
                F[i, k] = fx[i] + h.extra_Psi(None)                         # :47
                T[i, k] = now
        Xt = torch.empty(K, n, dtype=torch.float64, device=batch.device)    # rows are written by the passes that need them
        if linesearch:
            for i in range(K):
                if active[i]:
                    Lc[i] = Lc[i] / ls_ratio                                # :51
            search = list(active)
            while any(search):                                              # one pass = one trial per searching instance
                batch.prox(X, Gr, Lc, eps_prox, search, out=Xt)             # :52 / :55
                terms = batch.ls_terms(Gr, Xt, X, None, None, search)       # <g, x+ - x>, D(x+, x)
                ft = batch.func_grad(Xt, 0, search)                         # :53
                for i in range(K):
                    if search[i]:
                        if ft[i] > fx[i] + terms[i, 0] + Lc[i] * terms[i, 1]:
                            Lc[i] = Lc[i] * ls_ratio                        # :54
                        else:
                            search[i] = False
        else:
            batch.prox(X, Gr, Lc, eps_prox, active, out=Xt)                 # :58
        for i in range(K):
            if not active[i]:
                continue
            Ls[i, k] = Lc[i]
            last[i] = k
            if k > 0 and abs(F[i, k] - F[i, k - 1]) < epsilon:              # :66
                active[i] = False
                result_x[i] = Xt[i].clone()
        X = Xt
        yield k
    out = []
    for i in range(K):
        xi = result_x[i] if result_x[i] is not None else X[i].clone()
        xi = xi.cpu().numpy() if as_numpy else xi
        e = last[i] + 1
        out.append((xi, F[i, :e].copy(), Ls[i, :e].copy(), T[i, :e].copy()))
    return out


def BPG_batch(batch, h, L, x0, maxitrs, epsilon=1e-14, linesearch=True, ls_ratio=1.2):
    """Drain ``BPG_batch_steps``: list of (x, F, Ls, T), one per instance."""
    gen = BPG_batch_steps(batch, h, L, x0, maxitrs, epsilon, linesearch, ls_ratio)
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value


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


def ABPG_gain_batch_steps(batch, h, L, x0, gamma, maxitrs, epsilon=1e-14, G0=1, ls_inc=1.2, ls_dec=1.2, theta_eq=True,
                          checkdiv=False, restart=False, restart_rule='g', overlap=True):
    """ABPG_gain (accbpg/algorithms.py:295-420) on the K instances of a ``DOptimalBatch`` in lock-step per ORACLE
    PASS: every gradient evaluation, prox, vector pass and trial value covers the instances that need it.  The line
    search is per instance -- each keeps its own gain, theta and retry count on the host; after a pass the instances
    whose test failed raise their gain and take part in the next pass, the others have accepted their point and
    wait (their rows are not touched again in this outer iteration).  Yields k after every outer iteration; returns,
    per instance, ABPG_gain's (x, F, Gain, Gdiv, Gavg, T) -- bit-identical to ``ABPG_gain(batch.instance(i), ...)``."""
    from .algorithms import solve_theta
    K, n = batch.K, batch.n
    t_start = time.time()
    x0d, as_numpy = to_dev(x0)
    X = x0d.reshape(1, -1).repeat(K, 1).contiguous() if x0d.dim() == 1 else x0d.clone().contiguous()
    Z = X.clone()
    F = np.zeros((K, maxitrs)); Gain = np.ones((K, maxitrs)) * G0; Gdiv = np.zeros((K, maxitrs))
    Gavg = np.zeros((K, maxitrs)); T = np.zeros((K, maxitrs))
    Gs = [float(G0)] * K
    sumlog = [gamma * np.log(G0)] * K                                   # :342
    theta = [1.0] * K
    kk = [0] * K
    Gdr = [0.0] * K
    active = [True] * K
    last = [-1] * K
    result_x = [None] * K
    eps_prox = getattr(h, "eps", 1e-8)
    new = lambda: torch.empty(K, n, dtype=torch.float64, device=batch.device)
    for k in range(maxitrs):
        if not any(active):
            break
        side = overlap and not getattr(batch, "_prof", False)                # (kernel timing keeps everything on one stream)
        ticket = batch.value_async(X, active) if side else None             # :347 (beside the first gradients below)
        fx = None if side else batch.func_grad(X, 0, active)
        now = time.time() - t_start
        G_prev, theta_prev = list(Gs), list(theta)
        for i in range(K):
            if active[i]:
                Gs[i] = Gs[i] / ls_dec                                      # :358
        search = list(active)
        Y, Gr, Zt, Xt = new(), new(), new(), new()                          # rows are written by the passes that need them
        dzz = [0.0] * K
        first_pass = True
        while any(search):                                                  # :361, one pass = one trial per searching instance
            for i in range(K):
                if search[i] and kk[i] > 0:
                    if theta_eq:
                        theta[i] = solve_theta(theta_prev[i], gamma, Gs[i] / G_prev[i])
                    else:
                        alpha = Gs[i] / G_prev[i]
                        theta[i] = theta_prev[i] * ((1 + alpha * (gamma - 1)) / (gamma * alpha + theta_prev[i]))
            one_m = [1 - t for t in theta]
            batch.axpby(one_m, X, theta, Z, search, out=Y)                  # :369
            fy, _ = batch.func_grad(Y, 2, search, out=Gr)                   # :371
            if first_pass:
                if side:
                    fx = batch.value_wait(ticket)
                    now = time.time() - t_start
                for i in range(K):
                    if active[i]:
                        F[i, k] = fx[i] + h.extra_Psi(None)                 # :348
                        T[i, k] = now
                first_pass = False
            batch.prox(Z, Gr, [theta[i] ** (gamma - 1) * Gs[i] * L for i in range(K)], eps_prox, search, out=Zt)   # :373
            batch.axpby(one_m, X, theta, Zt, search, out=Xt)                # :374
            terms = batch.ls_terms(Gr, Xt, Y, Zt, Z, search)                # :377-378 and the dot of :387
            need_value = [False] * K
            for i in range(K):
                if not search[i]:
                    continue
                lin, dxy, dzz[i] = terms[i]
                if dzz[i] < epsilon:                                        # :379-380
                    search[i] = False
                    continue
                Gdr[i] = dxy / dzz[i] / theta[i] ** gamma                   # :382
                if checkdiv:
                    if Gdr[i] > Gs[i]:                                      # :385
                        Gs[i] = Gs[i] * ls_inc
                    else:
                        search[i] = False
                else:
                    need_value[i] = True
            if any(need_value):
                ft = batch.func_grad(Xt, 0, need_value)                     # :387
                for i in range(K):
                    if need_value[i]:
                        lin = terms[i, 0]
                        if ft[i] > fy[i] + lin + theta[i] ** gamma * Gs[i] * L * dzz[i]:
                            Gs[i] = Gs[i] * ls_inc                          # :390
                        else:
                            search[i] = False
        rest = None
        if restart and restart_rule == 'g':
            rest = batch.ls_terms(Gr, Xt, X, None, None, active)[:, 0]      # <g, x - x_1>, :406
        Znext = Zt
        for i in range(K):
            if not active[i]:
                continue
            Gain[i, k] = Gs[i]
            Gdiv[i, k] = Gdr[i]
            sumlog[i] += np.log(Gs[i])
            Gavg[i, k] = np.exp(sumlog[i] / (gamma + k))                    # :395-396
            last[i] = k
            kk[i] += 1
            if restart:                                                     # :403-409 (no k > 0 guard)
                if (restart_rule == 'f' and F[i, k] > F[i, k - 1]) or (restart_rule == 'g' and rest[i] > 0):
                    theta[i] = 1.0
                    kk[i] = 0
                    if Znext is Zt:
                        Znext = Zt.clone()
                    Znext[i] = Xt[i]
            if dzz[i] < epsilon:                                            # :412
                active[i] = False
                result_x[i] = Xt[i].clone()
        X, Z = Xt, Znext
        yield k
    out = []
    for i in range(K):
        xi = result_x[i] if result_x[i] is not None else X[i].clone()
        xi = xi.cpu().numpy() if as_numpy else xi
        e = last[i] + 1
        out.append((xi, F[i, :e].copy(), Gain[i, :e].copy(), Gdiv[i, :e].copy(), Gavg[i, :e].copy(), T[i, :e].copy()))
    return out


def ABPG_gain_batch(batch, h, L, x0, gamma, maxitrs, **kwargs):
    """Drain ``ABPG_gain_batch_steps``: list of (x, F, Gain, Gdiv, Gavg, T), one per instance."""
    gen = ABPG_gain_batch_steps(batch, h, L, x0, gamma, maxitrs, **kwargs)
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value
