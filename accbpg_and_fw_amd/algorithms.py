"""Outer loops of BPG / ABPG / ABPG_gain with the reference's signatures, defaults,
return tuples, print tables and quirks (accbpg/algorithms.py:11-180, 295-420), running
on device vectors.  The host keeps the iteration and every scalar decision; all
length-n and matrix work goes through libaccbpg_hip.so via the f / h objects of
``functions.py`` and the fused vector helpers there.

``x0`` may be a NumPy array (results come back as NumPy) or an fp64 CUDA tensor
(results stay on the device).  ``T[k]`` is the time at which ``F[k]`` was known on the
host, as the reference stamps it right after computing ``F[k]``: read off the host clock
when f(x) runs on the solver's stream, and corrected by the device-reported lead when it
ran on the side stream beside the gradient evaluation (``DOptimalObj.overlap_values``,
the default) -- see ``_stamp``.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from .functions import BurgEntropy, from_dev, ls_terms, to_dev, vec_axpby, vec_div_scalar, vec_dot_diff


def _divergences(h, g, x, y, z, z_prev):
    """(<g, x-y>, D(x,y), D(z,z_prev)); one fused launch when h is this package's Burg kernel."""
    if isinstance(h, BurgEntropy):
        return ls_terms(g, x, y, z, z_prev)
    return vec_dot_diff(g, x, y), h.divergence(x, y), h.divergence(z, z_prev)


def _lin(f):
    """True when f is this package's D-optimal objective with Gram-matrix reuse switched on."""
    return bool(getattr(f, "_lin", False))


def _value(f, x, combo=None):
    return f.func_grad_combo(x, combo, 0) if _lin(f) else f(x)


def _value_begin(f, x, combo):
    """F[k] = f(x) does not feed the gradient evaluation at y: it is started on the objective's side stream
    (DOptimalObj.overlap_values, the default) unless that is switched off or kernel timing is on
    (DOptimalObj.profile), in which case it is evaluated now."""
    if (not _lin(f)) and getattr(f, "_overlap", False) and not getattr(f, "_prof", False) \
            and isinstance(x, torch.Tensor) and x.is_cuda:
        if getattr(f, "_memo_on", False):                       # opt-in: f at this very tensor was the last value computed
            hit = f._memo_get(x)
            if hit is not None:
                return ("value", hit)
        return ("ticket", f.value_async(x))
    return ("value", _value(f, x, combo))


def _can_speculate(f, x, checkdiv):
    """Gradient evaluations may be started ahead of the line-search decision on this package's D-optimal objective
    when the caller has switched that on (DOptimalObj.speculate -- opt-in: measured slower at (2048,32768), 47.7
    against 50.5 it/s, and 3 % faster at (512,8192)) and it evaluates directly, on the device, with the side stream
    in use."""
    return (not checkdiv) and (not _lin(f)) and getattr(f, "_spec", False) and getattr(f, "_overlap", False) \
        and not getattr(f, "_prof", False) and hasattr(f, "grad_async") and isinstance(x, torch.Tensor) and x.is_cuda


def _value_end(f, pending):
    kind, payload = pending
    return f.value_wait(payload) if kind == "ticket" else payload


def _stamp(f, pending, t_start, t_known):
    """T[k]: seconds from the start of the run to the moment F[k] = f(x) was known, as the reference stamps
    it right after that evaluation (accbpg/algorithms.py:135-137, 347-349).  `t_known` is the host clock when
    the value came back on the solver's stream; a value that ran beside the gradient evaluation was known
    earlier than the clock read that follows both by the lead the device reports."""
    if pending[0] == "ticket":
        return time.time() - t_start - f.value_lead_seconds()
    return t_known - t_start


def _drain(gen):
    """Run a step generator to completion and return its result."""
    while True:
        try:
            next(gen)
        except StopIteration as stop:
            return stop.value


def BPG(f, h, L, x0, maxitrs, epsilon=1e-14, linesearch=True, ls_ratio=1.2,
        verbose=True, verbskip=1):
    """Bregman proximal gradient method (accbpg/algorithms.py:11-72).

    Returns (x, F, Ls, T).  F[k] is the objective at the iterate *before* update k
    (:47), so the returned x is one step past F[-1]; L is carried between iterations
    and divided by ls_ratio before each backtracking search (:51)."""
    return _drain(BPG_steps(f, h, L, x0, maxitrs, epsilon, linesearch, ls_ratio, verbose, verbskip))


def BPG_steps(f, h, L, x0, maxitrs, epsilon=1e-14, linesearch=True, ls_ratio=1.2,
              verbose=True, verbskip=1):
    """Generator form of BPG: yields k after each outer iteration, returns BPG's tuple."""
    if verbose:
        print("\nBPG_LS method for min_{x in C} F(x) = f(x) + Psi(x)")
        print("     k      F(x)         Lk       time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    Ls = np.ones(maxitrs) * L
    T = np.zeros(maxitrs)

    x, as_numpy = to_dev(x0)
    x = x.clone()
    k = -1
    for k in range(maxitrs):
        fx, g = f.func_grad(x)
        F[k] = fx + h.extra_Psi(x)
        T[k] = time.time() - t_start

        if linesearch:
            L = L / ls_ratio
            trial = h.div_prox_map(x, g, L)
            while True:
                lin, dist, _ = _divergences(h, g, trial, x, None, None)
                if not (f(trial) > fx + lin + L * dist):        # algorithms.py:53
                    break
                L = L * ls_ratio
                trial = h.div_prox_map(x, g, L)
            x = trial
        else:
            x = h.div_prox_map(x, g, L)

        Ls[k] = L
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:6.1f}".format(k, F[k], L, T[k]))

        if k > 0 and abs(F[k] - F[k - 1]) < epsilon:            # algorithms.py:66
            break
        yield k

    return from_dev(x, as_numpy), F[0:k + 1], Ls[0:k + 1], T[0:k + 1]


def solve_theta(theta, gamma, gainratio=1):
    """Newton solve of (1-t)/t^gamma = gainratio/theta^gamma from t = theta, tolerance
    1e-6*theta (accbpg/algorithms.py:75-91).  Pure host scalar arithmetic."""
    ckg = theta ** gamma / gainratio
    cta = theta
    eps = 1e-6 * theta
    phi = cta ** gamma - ckg * (1 - cta)
    while abs(phi) > eps:
        drv = gamma * cta ** (gamma - 1) + ckg
        cta = cta - phi / drv
        phi = cta ** gamma - ckg * (1 - cta)
    return cta


def ABPG(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=False,
         restart=False, restart_rule='g', verbose=True, verbskip=1):
    """Accelerated BPG with fixed triangle-scaling exponent (accbpg/algorithms.py:94-180).
    Returns (x, F, G, T)."""
    return _drain(ABPG_steps(f, h, L, x0, gamma, maxitrs, epsilon, theta_eq, restart, restart_rule,
                             verbose, verbskip))


def ABPG_steps(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=False,
               restart=False, restart_rule='g', verbose=True, verbskip=1):
    """Generator form of ABPG: yields k after each outer iteration, returns ABPG's tuple."""
    if verbose:
        print("\nABPG method for minimize_{x in C} F(x) = f(x) + Psi(x)")
        print("     k      F(x)       theta" +
              "        TSG       D(x+,y)     D(z+,z)     time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    G = np.zeros(maxitrs)
    T = np.zeros(maxitrs)

    x, as_numpy = to_dev(x0)
    x = x.clone()
    z = x.clone()
    theta = 1.0
    kk = 0
    xcombo = None
    k = -1
    for k in range(maxitrs):
        pending = _value_begin(f, x, xcombo)                    # :135 (runs beside the gradient below)
        t_known = time.time()

        z_prev, x_prev = z, x
        if theta_eq and kk > 0:                                 # :142-145
            theta = solve_theta(theta, gamma)
        else:
            theta = gamma / (kk + gamma)

        y = vec_axpby(1 - theta, x, theta, z_prev)              # :147
        if _lin(f):
            g = f.func_grad_combo(y, (1 - theta, x_prev, theta, z_prev), 1)
        else:
            g = f.gradient(y)                                   # :148
        F[k] = _value_end(f, pending) + h.extra_Psi(x_prev)     # :136
        T[k] = _stamp(f, pending, t_start, t_known)             # :137
        z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * L)  # :149
        x = vec_axpby(1 - theta, x, theta, z)                   # :150
        if _lin(f):
            f.ensure_gram(z)                                    # the one O(m^2 n) product of this iteration
            xcombo = (1 - theta, x_prev, theta, z)

        _, dxy, dzz = _divergences(h, None, x, y, z, z_prev)    # :153-154
        Gdr = dxy / dzz / theta ** gamma                        # :155

        G[k] = Gdr
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:6.1f}".format(
                k, F[k], theta, Gdr, dxy, dzz, T[k]))

        kk += 1
        if restart and k > 0:                                   # :165-171
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and vec_dot_diff(g, x, x_prev) > 0):
                theta = 1.0
                kk = 0
                z = x

        if dzz < epsilon:                                       # :174
            break
        yield k

    return from_dev(x, as_numpy), F[0:k + 1], G[0:k + 1], T[0:k + 1]


def ABPG_gain(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, G0=1,
              ls_inc=1.2, ls_dec=1.2, theta_eq=True, checkdiv=False,
              restart=False, restart_rule='g', verbose=True, verbskip=1):
    """Accelerated BPG with gain adaption (accbpg/algorithms.py:295-420).
    Returns (x, F, Gain, Gdiv, Gavg, T).

    Quirks kept: the gain is cut by ls_dec before every search (:358); an inner break
    on D(z+,z) < epsilon records the previous Gdr (:379-382, NameError in the reference
    if that happens at k = 0 -- here it records 0.0); the restart test has no k > 0
    guard, so rule 'f' at k = 0 compares with F[-1] (:403-406)."""
    return _drain(ABPG_gain_steps(f, h, L, x0, gamma, maxitrs, epsilon, G0, ls_inc, ls_dec, theta_eq,
                                  checkdiv, restart, restart_rule, verbose, verbskip))


def ABPG_gain_steps(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, G0=1,
                    ls_inc=1.2, ls_dec=1.2, theta_eq=True, checkdiv=False,
                    restart=False, restart_rule='g', verbose=True, verbskip=1):
    """Generator form of ABPG_gain: yields k after each outer iteration, returns its tuple."""
    if verbose:
        print("\nABPG_gain method for min_{x in C} F(x) = f(x) + Psi(x)")
        print("     k      F(x)       theta         Gk" +
              "         TSG       D(x+,y)     D(z+,z)      Gavg       time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    Gain = np.ones(maxitrs) * G0
    Gdiv = np.zeros(maxitrs)
    Gavg = np.zeros(maxitrs)
    T = np.zeros(maxitrs)

    x, as_numpy = to_dev(x0)
    x = x.clone()
    z = x.clone()
    G = G0
    sumlogG = gamma * np.log(G)                                 # :342
    theta = 1.0
    kk = 0
    Gdr = 0.0
    xcombo = None
    retries_before = 0      # failed trials of the previous iteration: how far ahead the next one starts gradients
    k = -1
    for k in range(maxitrs):
        pending = _value_begin(f, x, xcombo)                    # :347 (runs beside the first gradient below)
        t_known = time.time()

        z_prev, x_prev = z, x
        G_prev, theta_prev = G, theta
        G = G / ls_dec                                          # :358

        def theta_for(Gt):                                      # :362-367 for a trial gain Gt
            if kk == 0:
                return theta
            if theta_eq:
                return solve_theta(theta_prev, gamma, Gt / G_prev)
            alpha = Gt / G_prev
            return theta_prev * ((1 + alpha * (gamma - 1)) / (gamma * alpha + theta_prev))

        ahead = _can_speculate(f, x_prev, checkdiv)
        started = None                                          # (y, ticket) of a gradient evaluation started ahead
        failed = 0
        searching = True
        while searching:                                        # :361
            theta = theta_for(G)
            if started is not None:
                y, ticket = started                             # the point and the evaluation of this very trial
                started = None
                fy, g = f.grad_wait(ticket)
            else:
                y = vec_axpby(1 - theta, x_prev, theta, z_prev)  # :369
                if _lin(f):
                    fy, g = f.func_grad_combo(y, (1 - theta, x_prev, theta, z_prev), 2)
                else:
                    fy, g = f.func_grad(y)                      # :371
            if pending is not None:
                F[k] = _value_end(f, pending) + h.extra_Psi(x_prev)   # :348
                T[k] = _stamp(f, pending, t_start, t_known)           # :349
                pending = None
            z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * G * L)   # :373
            x = vec_axpby(1 - theta, x_prev, theta, z)          # :374
            if _lin(f):
                f.ensure_gram(z)                                # the one O(m^2 n) product of this pass
                xcombo = (1 - theta, x_prev, theta, z)

            lin, dxy, dzz = _divergences(h, g, x, y, z, z_prev)  # :377-378 and the dot of :387
            if dzz < epsilon:                                   # :379-380
                break

            Gdr = dxy / dzz / theta ** gamma                    # :382

            if checkdiv:
                searching = (Gdr > G)                           # :385
            else:
                if ahead and failed < retries_before:
                    # the previous iteration failed this trial too: the next trial's gradient (a function of host
                    # scalars and of x_1, z_1 only) goes out on the side stream beside the value test below
                    theta_next = theta_for(G * ls_inc)
                    y_next = vec_axpby(1 - theta_next, x_prev, theta_next, z_prev)
                    started = (y_next, f.grad_async(y_next))
                searching = (_value(f, x, xcombo) > fy + lin + theta ** gamma * G * L * dzz)   # :387

            if searching:
                G = G * ls_inc                                  # :390
                failed += 1
            elif started is not None:
                f.grad_drop(started[1])                         # the trial passed: not needed
                started = None
        retries_before = failed

        Gain[k] = G
        Gdiv[k] = Gdr
        sumlogG += np.log(G)
        Gavg[k] = np.exp(sumlogG / (gamma + k))                 # :395-396
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:10.3e}  {7:10.3e}  {8:6.1f}".format(
                k, F[k], theta, G, Gdr, dxy, dzz, Gavg[k], T[k]))

        kk += 1
        if restart:                                             # :403-409
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and vec_dot_diff(g, x, x_prev) > 0):
                theta = 1.0
                kk = 0
                z = x

        if dzz < epsilon:                                       # :412
            break
        yield k

    return (from_dev(x, as_numpy), F[0:k + 1], Gain[0:k + 1], Gdiv[0:k + 1],
            Gavg[0:k + 1], T[0:k + 1])


def ABPG_expo(f, h, L, x0, gamma0, maxitrs, epsilon=1e-14, delta=0.2,
              theta_eq=True, checkdiv=False, Gmargin=10, restart=False,
              restart_rule='g', verbose=True, verbskip=1):
    """Accelerated BPG with exponent adaption (accbpg/algorithms.py:183-292).
    Returns (x, F, Gamma, G, T).  One gradient at y per outer iteration (:245); the inner loop
    lowers gamma by delta while its test fails and gamma > 1 (:262-265); the restart test has no
    k > 0 guard (:276-282)."""
    return _drain(ABPG_expo_steps(f, h, L, x0, gamma0, maxitrs, epsilon, delta, theta_eq, checkdiv, Gmargin,
                                  restart, restart_rule, verbose, verbskip))


def ABPG_expo_steps(f, h, L, x0, gamma0, maxitrs, epsilon=1e-14, delta=0.2,
                    theta_eq=True, checkdiv=False, Gmargin=10, restart=False,
                    restart_rule='g', verbose=True, verbskip=1):
    """Generator form of ABPG_expo: yields k after each outer iteration, returns its tuple."""
    if verbose:
        print("\nABPG_expo method for min_{x in C} F(x) = f(x) + Psi(x)")
        print("     k      F(x)       theta       gamma" +
              "        TSG       D(x+,y)     D(z+,z)     time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    G = np.zeros(maxitrs)
    Gamma = np.ones(maxitrs) * gamma0
    T = np.zeros(maxitrs)

    gamma = gamma0
    x, as_numpy = to_dev(x0)
    x = x.clone()
    z = x.clone()
    theta = 1.0
    kk = 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)                            # :231-232
        T[k] = time.time() - t_start

        z_prev, x_prev = z, x
        if theta_eq and kk > 0:                                 # :238-241
            theta = solve_theta(theta, gamma)
        else:
            theta = gamma / (kk + gamma)

        y = vec_axpby(1 - theta, x_prev, theta, z_prev)         # :243
        fy, g = f.func_grad(y)                                  # :245

        trying = True
        while trying:                                           # :248
            z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * L)   # :249
            x = vec_axpby(1 - theta, x_prev, theta, z)          # :250
            lin, dxy, dzz = _divergences(h, g, x, y, z, z_prev)  # :253-254 and the dot of :260
            Gdr = dxy / dzz / theta ** gamma                    # :255

            if checkdiv:
                trying = (dxy > Gmargin * (theta ** gamma) * dzz)    # :258
            else:
                trying = (f(x) > fy + lin + theta ** gamma * L * dzz)   # :260

            if trying and gamma > 1:                            # :262-265
                gamma = max(gamma - delta, 1)
            else:
                trying = False

        G[k] = Gdr
        Gamma[k] = gamma
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:10.3e}  {7:6.1f}".format(
                k, F[k], theta, gamma, Gdr, dxy, dzz, T[k]))

        kk += 1
        if restart:                                             # :276-282
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and vec_dot_diff(g, x, x_prev) > 0):
                theta = 1.0
                kk = 0
                z = x

        if dzz < epsilon:                                       # :285
            break
        yield k

    return from_dev(x, as_numpy), F[0:k + 1], Gamma[0:k + 1], G[0:k + 1], T[0:k + 1]


def ABDA(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=True,
         verbose=True, verbskip=1):
    """Accelerated Bregman dual averaging (accbpg/algorithms.py:423-514).  Returns (x, F, G, T).
    gavg accumulates theta^(1-gamma) * g (:483), z = prox_map(gavg/csum, L/csum) (:485)."""
    return _drain(ABDA_steps(f, h, L, x0, gamma, maxitrs, epsilon, theta_eq, verbose, verbskip))


def ABDA_steps(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=True,
               verbose=True, verbskip=1):
    """Generator form of ABDA: yields k after each outer iteration, returns ABDA's tuple."""
    if verbose:
        print("\nABDA method for min_{x in C} F(x) = f(x) + Psi(x)")
        print("     k      F(x)       theta" +
              "        TSG       D(x+,y)     D(z+,z)     time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    G = np.zeros(maxitrs)
    T = np.zeros(maxitrs)

    x, as_numpy = to_dev(x0)
    x = x.clone()
    z = x.clone()
    theta = 1.0
    kk = 0
    gavg = torch.zeros_like(x)
    csum = 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)                            # :465-466
        T[k] = time.time() - t_start

        z_prev, x_prev = z, x
        if theta_eq and kk > 0:                                 # :472-475
            theta = solve_theta(theta, gamma)
        else:
            theta = gamma / (kk + gamma)

        y = vec_axpby(1 - theta, x_prev, theta, z_prev)         # :477
        g = f.gradient(y)                                       # :478
        wgt = theta ** (1 - gamma)
        gavg = vec_axpby(1.0, gavg, wgt, g)                     # :479  (1.0*gavg is exact)
        csum = csum + wgt                                       # :480
        z = h.prox_map(vec_div_scalar(gavg, csum), L / csum)    # :481
        x = vec_axpby(1 - theta, x_prev, theta, z)              # :482

        _, dxy, dzz = _divergences(h, None, x, y, z, z_prev)    # :485-486
        Gdr = dxy / dzz / theta ** gamma

        G[k] = Gdr
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:6.1f}".format(
                k, F[k], theta, Gdr, dxy, dzz, T[k]))

        kk += 1
        if dzz < epsilon:                                       # :508
            break
        yield k

    return from_dev(x, as_numpy), F[0:k + 1], G[0:k + 1], T[0:k + 1]
