"""Frank-Wolfe with a Bregman-divergence step size (accbpg/algorithms_fw.py:6-75), on device
vectors; used with ``lmo_simplex`` on the D-optimal objective
(frank_wolfe_wtih_rs/ex_Dopt_design.py:17-18)."""
from __future__ import annotations

import math
import time

import numpy as np

from .functions import from_dev, to_dev, vec_axpby, vec_dot_diff


def FW_alg_div_step(f, h, L, x0, maxitrs, gamma, lmo, epsilon=1e-14, linesearch=True, ls_ratio=2,
                    verbose=True, verbskip=1):
    """Returns (x, F, Ls, T).  alpha_k = min((-<g,d> / (2 L D(s,x)))^(1/(gamma-1)), 1) with
    backtracking on L (algorithms_fw.py:49-64); a zero divergence is replaced by 1e-6 (:37-38) and a
    slope in (0, 1e-6] by 0 (:41-42)."""
    if ls_ratio < 1:
        raise ValueError("ls_ratio must be >= 1")
    if L <= 0:
        raise ValueError("Initial L must be positive")
    if epsilon <= 0:
        raise ValueError("epsilon must be positive")

    if verbose:
        print("\nFW adaptive algorithm")
        print("     k      F(x)         Lk       time")

    t_start = time.time()
    F, Ls, T = [], [], []
    delta = 1e-6

    x, as_numpy = to_dev(x0)
    x = x.clone()
    for k in range(maxitrs):
        fx, g = f.func_grad(x)                                  # :30
        F.append(fx + h.extra_Psi(x))
        T.append(time.time() - t_start)

        s_k = lmo(g)                                            # :34
        d_k = vec_axpby(1.0, s_k, -1.0, x)                      # :35  s - x
        div = h.divergence(s_k, x)                              # :36
        if div == 0:
            div = delta

        grad_d_prod = vec_dot_diff(g, s_k, x)                   # :40  <g, s - x>
        if 0 < grad_d_prod <= delta:
            grad_d_prod = 0.0
        if grad_d_prod > 0:
            raise ValueError("grad_d_prod must be non-positive")

        if linesearch:
            L = L / ls_ratio                                    # :47

        while True:
            alpha_k = min((-grad_d_prod / (2 * L * div)) ** (1 / (gamma - 1)), 1.0)   # :50-53
            x1 = vec_axpby(1.0, x, alpha_k, d_k)                # :54
            if not linesearch:
                break
            assert not math.isinf(L), "L is infinite"
            if f.func_grad(x1, flag=0) <= fx + alpha_k * grad_d_prod + alpha_k ** gamma * L * div:   # :61
                break
            L = L * ls_ratio

        x = x1
        Ls.append(L)
        if verbose and k % verbskip == 0:
            print(f"{k:6d}  {F[k]:10.3e}  {L:10.3e}  {T[k]:6.1f}")

        if k > 0 and abs(F[k] - F[k - 1]) < epsilon:            # :72
            break

    return from_dev(x, as_numpy), np.array(F), np.array(Ls), np.array(T)
