"""Frank-Wolfe and Wolfe-Atwood (away-step) solvers for D-optimal design with the
reference's signatures and return tuples (accbpg/D_opt_alg.py:9-88, 91-185).

State (x, the inverse H = (V X V^T)^-1, w_i = v_i^T H v_i) lives on the GPU inside
the D-optimal handle; per iteration the host reads back one small probe record
(argmax / away index and their values), takes the scalar decisions exactly as the
reference writes them, and issues one rank-one update.
"""
from __future__ import annotations

import ctypes as C
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

    def probe(self, away, refresh_logdet):
        pr = _lib.FwProbe()
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_probe_step(self.h, int(away), int(refresh_logdet), C.byref(pr))
        _lib.check(rc, "accbpg_fw_probe_step")
        return pr

    def flush_logdet(self):
        """log det(H) of the last ``probe(refresh_logdet=2)`` call (see accbpg_fw_logdet_flush)."""
        out = C.c_double(0.0)
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_logdet_flush(self.h, C.byref(out))
        _lib.check(rc, "accbpg_fw_logdet_flush")
        return out.value

    def update(self, p, xscale, xadd, hcoef, hdiv):
        with torch.cuda.device(self.obj.device):
            rc = self.lib.accbpg_fw_update(self.h, int(p), float(xscale), float(xadd), float(hcoef), float(hdiv))
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


def D_opt_FW(V, x0, eps, maxitrs, verbose=True, verbskip=1):
    """Frank-Wolfe with exact line search (accbpg/D_opt_alg.py:9-88).
    Returns (x, F, SP, SN, T).  F[k] = -log(detVXVT) with the determinant tracked by
    the rank-one formula (:52,:80); w is never refreshed; the stop test precedes the
    update so x matches F[-1] (:72).  ``V`` may be a matrix or a DOptimalObj."""
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

    return st.x(), F[0:k + 1], SP[0:k + 1], SN[0:k + 1], T[0:k + 1]


def D_opt_FW_away(V, x0, eps, maxitrs, verbose=True, verbskip=1, logdet_refresh=1):
    """Frank-Wolfe with Wolfe's away steps (accbpg/D_opt_alg.py:91-185).
    Returns (x, F, SP, SN, T).  F[k] = log det(H) of the maintained inverse (:136).

    ``logdet_refresh`` (extension, default 1 = the reference's behaviour): refactor H
    for log det every that many iterations; in between, log det(H) is advanced in
    log-space by the matrix determinant lemma for the same rank-one update."""
    start_time = time.time()
    st = _FWState(V, x0)
    m = st.m
    F = np.zeros(maxitrs)
    SP = np.zeros(maxitrs)
    SN = np.zeros(maxitrs)
    T = np.zeros(maxitrs)

    if verbose:
        print("\nSolving D-opt design problem using Frank-Wolfe method with away steps")
        print("     k      F(x)     pos_slack   neg_slack    time")

    # F[k] = log det(H_k) is only logged (no decision reads it), so with the reference's behaviour (a fresh
    # factorisation every iteration) it is formed on a side stream while this loop already probes, decides and
    # updates, and lands in F one iteration later -- same kernels, same numbers (accbpg_fw_probe_step, form 2);
    # a table row is printed when its F value is in.
    piped = (logdet_refresh == 1)
    logdet_H = -st.logdet_gram
    owed = None                                                 # iteration whose F (and table row) is still outstanding

    def settle(value):
        F[owed] = value
        if verbose and owed % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(
                owed, F[owed], SP[owed], SN[owed], T[owed]))

    k = -1
    for k in range(maxitrs):
        refresh = (logdet_refresh > 0) and (k % logdet_refresh == 0)
        pr = st.probe(away=1, refresh_logdet=(2 if piped else 1) if refresh else 0)   # :136, :145-147
        if piped:
            if owed is not None:
                settle(pr.logdet_H)
            owed = k
        else:
            if refresh:
                logdet_H = pr.logdet_H
            F[k] = logdet_H
        T[k] = time.time() - start_time
        w_i, w_j = pr.w_i, pr.w_j
        eps_pos = w_i / m - 1                                   # :150
        eps_neg = 1 - w_j / m                                   # :151
        SP[k] = eps_pos
        SN[k] = eps_neg

        if verbose and not piped and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(
                k, F[k], eps_pos, eps_neg, T[k]))

        if eps_pos <= eps and eps_neg <= eps:                   # :159
            break

        if eps_pos >= eps_neg:                                  # :162-170
            t = (w_i / m - 1) / (w_i - 1)
            coef = t / (1 - t + t * w_i)
            st.update(pr.i, 1 - t, t, -coef, 1 - t)
            # det(H+) = det(H) * (1 - coef*w_i) / (1-t)^m
            logdet_H += np.log1p(-coef * w_i) - m * np.log1p(-t)
        else:                                                   # :171-179
            x_j = pr.x_j
            t = min((1 - w_j / m) / (w_j - 1), x_j / (1 - x_j))
            coef = t / (1 + t - t * w_j)
            st.update(pr.j, 1 + t, -t, coef, 1 + t)
            logdet_H += np.log1p(coef * w_j) - m * np.log1p(t)

    if piped and owed is not None:
        settle(st.flush_logdet())
    return st.x(), F[0:k + 1], SP[0:k + 1], SN[0:k + 1], T[0:k + 1]
