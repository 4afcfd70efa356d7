"""Frank-Wolfe and Wolfe-Atwood (away-step) solvers for D-optimal design with the
reference's signatures and return tuples (accbpg/D_opt_alg.py:9-88, 91-185).

State (x, the inverse H = (V X V^T)^-1, w_i = v_i^T H v_i) lives on the GPU inside
the D-optimal handle; per iteration the host reads back one small probe record
(argmax / away index and their values), takes the scalar decisions exactly as the
reference writes them, and issues one rank-one update.
"""
from __future__ import annotations

import ctypes as C
import math
import time

import numpy as np
import torch

from . import _lib
from .functions import DOptimalObj, _ptr, _stream, from_dev, to_dev


class _FWState:
    """Owns the handle-side Frank-Wolfe state for one run."""

    def __init__(self, V, x0):
        self.obj = V if isinstance(V, DOptimalObj) else DOptimalObj(V)
        self.lib = _lib.load()
        self.h = self.obj._h
        self.m, self.n = self.obj.m, self.obj.n
        x0d, self.as_numpy = to_dev(x0)
        logdet = C.c_double(0.0)
        with torch.cuda.device(self.obj.device):
            self.lib.accbpg_dopt_set_stream(self.h, _stream())
            rc = self.lib.accbpg_fw_init(self.h, _ptr(x0d), C.byref(logdet))
        _lib.check(rc, "accbpg_fw_init")
        self.logdet_gram = logdet.value

    def _on_device(self):
        """True when the objective's device is already the current one (then the two calls per iteration skip the
        device context manager: a few microseconds each, against a step of 0.12-0.15 ms)."""
        return torch.cuda.current_device() == self.obj.device.index

    def probe(self, away, refresh_logdet):
        pr = _lib.FwProbe()
        if self._on_device():
            rc = self.lib.accbpg_fw_probe_step(self.h, int(away), int(refresh_logdet), C.byref(pr))
        else:
            with torch.cuda.device(self.obj.device):
                rc = self.lib.accbpg_fw_probe_step(self.h, int(away), int(refresh_logdet), C.byref(pr))
        if rc:
            _lib.check(rc, "accbpg_fw_probe_step")
        return pr

    def logdet_ring(self, depth, small_launches=2):
        """How many side factorisations of ``probe(refresh_logdet=2)`` may be in flight, and how they run."""
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_logdet_ring(self.h, int(depth), int(small_launches))
        _lib.check(rc, "accbpg_fw_logdet_ring")

    def flush_logdet(self):
        """log det(H) of the oldest ``probe(refresh_logdet=2)`` call still in flight (see accbpg_fw_logdet_flush)."""
        out = C.c_double(0.0)
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_logdet_flush(self.h, C.byref(out))
        _lib.check(rc, "accbpg_fw_logdet_flush")
        return out.value

    def update(self, p, xscale, xadd, hcoef, hdiv):
        if self._on_device():
            rc = self.lib.accbpg_fw_update(self.h, int(p), float(xscale), float(xadd), float(hcoef), float(hdiv))
        else:
            with torch.cuda.device(self.obj.device):
                rc = self.lib.accbpg_fw_update(self.h, int(p), float(xscale), float(xadd), float(hcoef), float(hdiv))
        if rc:
            _lib.check(rc, "accbpg_fw_update")

    def x(self):
        out = torch.empty(self.n, dtype=torch.float64, device=self.obj.device)
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_get_state(self.h, _ptr(out), None, None)
        _lib.check(rc, "accbpg_fw_get_state")
        return from_dev(out, self.as_numpy)

    def state(self):
        x = torch.empty(self.n, dtype=torch.float64, device=self.obj.device)
        w = torch.empty(self.n, dtype=torch.float64, device=self.obj.device)
        H = torch.empty(self.m, self.m, dtype=torch.float64, device=self.obj.device)
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_get_state(self.h, _ptr(x), _ptr(w), _ptr(H))
        _lib.check(rc, "accbpg_fw_get_state")
        return x, w, H


def _drain(gen):
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value


def D_opt_FW(V, x0, eps, maxitrs, verbose=True, verbskip=1):
    """Frank-Wolfe with exact line search (accbpg/D_opt_alg.py:9-88).
    Returns (x, F, SP, SN, T).  F[k] = -log(detVXVT) with the determinant tracked by
    the rank-one formula (:52,:80); w is never refreshed; the stop test precedes the
    update so x matches F[-1] (:72).  ``V`` may be a matrix or a DOptimalObj."""
    return _drain(D_opt_FW_steps(V, x0, eps, maxitrs, verbose, verbskip))


def D_opt_FW_steps(V, x0, eps, maxitrs, verbose=True, verbskip=1):
    """Generator form of D_opt_FW: yields k after each update, returns D_opt_FW's tuple."""
    start_time = time.time()
    st = _FWState(V, x0)
    m = st.m
    F = np.zeros(maxitrs)
    SP = np.zeros(maxitrs)
    SN = np.zeros(maxitrs)
    T = np.zeros(maxitrs)
    detVXVT = np.exp(st.logdet_gram)                            # :41

    if verbose:
        print("\nSolving D-opt design problem using Frank-Wolfe method")
        print("     k      F(x)     pos_slack   neg_slack    time")

    k = -1
    for k in range(maxitrs):
        F[k] = - np.log(detVXVT)                                # :52
        T[k] = time.time() - start_time
        pr = st.probe(away=0, refresh_logdet=0)                 # :59-61
        w_i = pr.w_i
        eps_pos = w_i / m - 1                                   # :63
        eps_neg = 1 - pr.w_j / m                                # :64
        SP[k] = eps_pos
        SN[k] = eps_neg

        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(
                k, F[k], eps_pos, eps_neg, T[k]))

        if eps_pos <= eps and eps_neg <= eps:                   # :72
            break

        t = (w_i / m - 1) / (w_i - 1)                           # :75
        coef = t / (1 + t * (w_i - 1))                          # :79,:82
        st.update(pr.i, 1 - t, t, -coef, 1 - t)                 # :76-79,:82
        detVXVT *= np.power(1 - t, m - 1) * (1 + t * (w_i - 1))  # :80
        yield k

    return st.x(), F[0:k + 1], SP[0:k + 1], SN[0:k + 1], T[0:k + 1]


# How often D_opt_FW_away refactors the maintained inverse for F[k] = log det(H_k) when the caller does not say
# (``logdet_refresh=None``), and how many factorisations are in flight when it does so every iteration.  Decided with
# numbers (tools/fw_away_modes.py, profiles/r03_fw_away_modes.json; D_opt_design(2048,32768), one MI355X): anchoring
# every 16th iteration leaves F[k] within 1.1e-13 (absolute; |F| ~ 50) of the every-iteration factorisation over 1000
# and over 20000 iterations -- less than that run's own distance from the reference's trace (1e-14 relative) -- with
# bit-identical iterates and gaps, at 5600-5900 iterations/s against 2800-2900 for the every-iteration form with three
# factorisations in flight (1520 with one).  The every-iteration form stays one keyword away (logdet_refresh=1).
LOGDET_REFRESH_DEFAULT = 16
LOGDET_RING_DEFAULT = 3


def D_opt_FW_away(V, x0, eps, maxitrs, verbose=True, verbskip=1, logdet_refresh=None, logdet_ring=None):
    """Frank-Wolfe with Wolfe's away steps (accbpg/D_opt_alg.py:91-185).
    Returns (x, F, SP, SN, T).  F[k] = log det(H_k) of the maintained inverse (:136), a logged value that no decision
    of the iteration reads; iterates, gaps and step choices do not depend on how it is formed.

    ``logdet_refresh`` (extension): 1 = the reference's computation, a fresh factorisation of H_k for every k (formed
    beside the steps, ``logdet_ring`` of them in flight, F[k] filled in that many iterations late).  R > 1 = a fresh
    factorisation of H_k for every k that is a multiple of R (beside the steps as well); in between log det(H) is
    advanced in log space by the matrix determinant lemma for the very rank-one update the step applies,
    log det(H+) = log det(H) + log(1 + c q) - m log(d) with q = v^T H v of the pivot column in the inverse as
    maintained (computed on the device from H itself, not the tracked w), so the error of F[k] against a fresh
    factorisation is the rounding of at most R - 1 such terms.  0 = never refactor (lemma from the start).
    None = ``LOGDET_REFRESH_DEFAULT``."""
    return _drain(D_opt_FW_away_steps(V, x0, eps, maxitrs, verbose, verbskip, logdet_refresh, logdet_ring))


def D_opt_FW_away_steps(V, x0, eps, maxitrs, verbose=True, verbskip=1, logdet_refresh=None, logdet_ring=None):
    """Generator form of D_opt_FW_away: yields k after each update, returns D_opt_FW_away's tuple."""
    start_time = time.time()
    st = _FWState(V, x0)
    m = st.m
    F = np.zeros(maxitrs)
    SP = np.zeros(maxitrs)
    SN = np.zeros(maxitrs)
    T = np.zeros(maxitrs)
    R = LOGDET_REFRESH_DEFAULT if logdet_refresh is None else int(logdet_refresh)
    depth = LOGDET_RING_DEFAULT if logdet_ring is None else int(logdet_ring)
    if R != 1:
        depth = 1                                               # anchors are R iterations apart: one in flight is enough
    st.logdet_ring(depth)

    if verbose:
        print("\nSolving D-opt design problem using Frank-Wolfe method with away steps")
        print("     k      F(x)     pos_slack   neg_slack    time")

    # F[k] is filled in (and its table row printed) when its value is in: an anchor -- a fresh factorisation of H_a
    # started at iteration a on a side stream -- arrives `depth` refreshing iterations later; the iterations between
    # two anchors follow from the first by the log-space steps, each known one probe after its update (q_prev).
    anchors = []            # iterations whose factorisation is in flight, oldest first
    delta = np.zeros(maxitrs)   # delta[k] = log det(H_{k+1}) - log det(H_k) by the determinant lemma
    filled = 0              # F[0:filled] is final
    step = None             # (hcoef, hdiv) of the update applied at the previous iteration

    def fill(upto):
        """F[filled:upto] from F[filled-1] by the log-space steps (upto exclusive), and print their rows."""
        nonlocal filled
        for j in range(filled, upto):
            F[j] = F[j - 1] + delta[j - 1]
            row(j)
        filled = max(filled, upto)

    def row(j):
        if verbose and j % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(j, F[j], SP[j], SN[j], T[j]))

    def settle(a, value):
        """The anchor of iteration a is in."""
        nonlocal filled
        fill(a)
        F[a] = value
        row(a)
        filled = a + 1

    k = -1
    for k in range(maxitrs):
        refresh = (R > 0) and (k % R == 0)
        pr = st.probe(away=1, refresh_logdet=2 if refresh else 0)   # :136, :145-147
        T[k] = time.time() - start_time
        if step is not None:
            hcoef, hdiv = step
            arg = hcoef * pr.q_prev
            delta[k - 1] = (math.log1p(arg) - m * math.log(hdiv)) if (arg > -1.0 and hdiv > 0.0) else float("nan")
        if refresh:
            if len(anchors) >= depth:
                settle(anchors.pop(0), pr.logdet_H)
            anchors.append(k)
        elif R == 0 and k == 0:
            F[0] = -st.logdet_gram
            filled = 1
        w_i, w_j = pr.w_i, pr.w_j
        eps_pos = w_i / m - 1                                   # :150
        eps_neg = 1 - w_j / m                                   # :151
        SP[k] = eps_pos
        SN[k] = eps_neg
        if R == 0 and k == 0:
            row(0)

        if eps_pos <= eps and eps_neg <= eps:                   # :159
            break

        if eps_pos >= eps_neg:                                  # :162-170
            t = (w_i / m - 1) / (w_i - 1)
            coef = t / (1 - t + t * w_i)
            step = (-coef, 1 - t)
            st.update(pr.i, 1 - t, t, -coef, 1 - t)
        else:                                                   # :171-179
            x_j = pr.x_j
            t = min((1 - w_j / m) / (w_j - 1), x_j / (1 - x_j))
            coef = t / (1 + t - t * w_j)
            step = (coef, 1 + t)
            st.update(pr.j, 1 + t, -t, coef, 1 + t)
        yield k

    while anchors:
        settle(anchors.pop(0), st.flush_logdet())
    fill(k + 1)
    return st.x(), F[0:k + 1], SP[0:k + 1], SN[0:k + 1], T[0:k + 1]
