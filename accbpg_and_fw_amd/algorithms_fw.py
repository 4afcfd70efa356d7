"""Frank-Wolfe with a step length taken from the Bregman divergence to the LMO vertex, on device
vectors (behaviour of accbpg/algorithms_fw.py:6-75; used with ``lmo_simplex`` on the D-optimal
objective as in frank_wolfe_wtih_rs/ex_Dopt_design.py:17-18).

Written like the other solvers of this package: a step generator that the public function drains,
preallocated traces cut to the iterations that ran, and one fused launch for the slope <g, s - x> and
the divergence D(s, x) when h is the Burg kernel.
"""
from __future__ import annotations

import math
import time

import numpy as np

from .algorithms import _divergences, _drain
from .functions import from_dev, to_dev, vec_axpby

_SLOPE_FLOOR = 1e-6     # stand-in for a vanishing divergence, and the band of slopes treated as zero


def _check_options(L, epsilon, ls_ratio):
    for bad, message in ((ls_ratio < 1, "ls_ratio must be >= 1"),
                         (L <= 0, "Initial L must be positive"),
                         (epsilon <= 0, "epsilon must be positive")):
        if bad:
            raise ValueError(message)


def _step_length(slope, L, dist, gamma):
    """min((-slope / (2 L D))^(1/(gamma-1)), 1): minimiser of the upper model along s - x."""
    return min((-slope / (2 * L * dist)) ** (1 / (gamma - 1)), 1.0)


def FW_alg_div_step(f, h, L, x0, maxitrs, gamma, lmo, epsilon=1e-14, linesearch=True, ls_ratio=2,
                    verbose=True, verbskip=1):
    """Returns (x, F, Ls, T).  Each iteration moves from x towards the vertex s = lmo(grad f(x)) by
    the step length above; with `linesearch` the constant L is first divided by ls_ratio and then
    multiplied back until f(x+) <= f(x) + a*slope + a^gamma * L * D(s, x)."""
    return _drain(FW_alg_div_step_steps(f, h, L, x0, maxitrs, gamma, lmo, epsilon, linesearch, ls_ratio,
                                        verbose, verbskip))


def FW_alg_div_step_steps(f, h, L, x0, maxitrs, gamma, lmo, epsilon=1e-14, linesearch=True, ls_ratio=2,
                          verbose=True, verbskip=1):
    """Generator form: yields k after each outer iteration, returns FW_alg_div_step's tuple."""
    _check_options(L, epsilon, ls_ratio)
    if verbose:
        print("\nFW adaptive algorithm")
        print("     k      F(x)         Lk       time")

    t_start = time.time()
    F = np.zeros(maxitrs)
    Ls = np.zeros(maxitrs)
    T = np.zeros(maxitrs)

    point, as_numpy = to_dev(x0)
    point = point.clone()
    done = 0
    for k in range(maxitrs):
        value, grad = f.func_grad(point)
        F[k] = value + h.extra_Psi(point)
        T[k] = time.time() - t_start

        vertex = lmo(grad)
        slope, dist, _ = _divergences(h, grad, vertex, point, None, None)     # <g, s - x> and D(s, x)
        dist = dist if dist != 0 else _SLOPE_FLOOR
        if 0 < slope <= _SLOPE_FLOOR:
            slope = 0.0
        if slope > 0:
            raise ValueError("grad_d_prod must be non-positive")
        towards = vec_axpby(1.0, vertex, -1.0, point)                          # s - x, one rounding

        if linesearch:
            L = L / ls_ratio
        while True:
            a = _step_length(slope, L, dist, gamma)
            trial = vec_axpby(1.0, point, a, towards)                          # x + a*(s - x)
            if not linesearch:
                break
            assert not math.isinf(L), "L is infinite"
            if f.func_grad(trial, flag=0) <= value + a * slope + a ** gamma * L * dist:
                break
            L = L * ls_ratio

        point = trial
        Ls[k] = L
        done = k + 1
        if verbose and k % verbskip == 0:
            print(f"{k:6d}  {F[k]:10.3e}  {L:10.3e}  {T[k]:6.1f}")
        if k > 0 and abs(F[k] - F[k - 1]) < epsilon:
            break
        yield k

    return from_dev(point, as_numpy), F[:done], Ls[:done], T[:done]
