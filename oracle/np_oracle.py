"""CPU oracle for the accbpg D-optimal-design hot path.  TEST INFRASTRUCTURE ONLY.

This module is a clean-room NumPy restatement of the arithmetic that the
reference package performs on the path named by BASELINE.json.  It exists so
that the HIP implementation can be checked against something that runs on the
GPU box (where /root/reference does not exist).  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  The product package (``accbpg_and_fw_amd``) never imports it.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference in
the build container and writes ``tests/golden/*.npz``; ``tests/test_oracle.py``
checks this file against those vectors and against the stored notebook rows
listed in SURVEY.md section 4.

Every routine cites the reference lines whose arithmetic (operation order,
stopping rules, quirks) it follows.  Paths are relative to /root/reference.
"""
from __future__ import annotations

import math
import time

import numpy as np


# --------------------------------------------------------------------------
# D-optimal objective                      accbpg/functions.py:27-59
# --------------------------------------------------------------------------
class DOptOracle:
    """f(x) = -log det(V diag(x) V^T); grad_i = -v_i^T (V X V^T)^-1 v_i.

    Follows accbpg/functions.py:43-59: the weighted Gram matrix is formed as
    dot(V*x, V.T) (:46), its log-determinant comes from an LU ``slogdet``
    (:48), a non-positive sign raises ValueError (:49-50), flag 0 returns
    before the solve (:53-54), and the gradient is the negated column sum of
    V * solve(G, V) (:57-58).
    """

    def __init__(self, V):
        self.H = V                      # attribute name callers read (:32)
        self.m, self.n = V.shape
        assert self.m < self.n, "DOptimalObj: need m < n"

    def gram(self, x):
        return np.dot(self.H * x, self.H.T)

    def func_grad(self, x, flag=2):
        assert x.size == self.n, "DOptimalObj: x.size not equal to n"
        assert x.min() >= 0, "DOptimalObj: x needs to be nonnegative"
        G = self.gram(x)
        sign, logdet = np.linalg.slogdet(G)
        if sign <= 0:
            raise ValueError("HXHT is singular or not positive definite")
        fval = -logdet
        if flag == 0:
            return fval
        sol = np.linalg.solve(G, self.H)
        grad = -np.sum(self.H * sol, axis=0)
        return grad if flag == 1 else (fval, grad)

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)


# --------------------------------------------------------------------------
# Burg entropy on the simplex              accbpg/functions.py:238-271,326-356
# --------------------------------------------------------------------------
def _seq_sum(a):
    """Left-to-right fp64 sum, the order of Python's builtin ``sum`` used at
    accbpg/functions.py:253,345,347,350,354."""
    return sum(a)


class BurgSimplexOracle:
    def __init__(self, eps=1e-8):
        assert eps > 0, "BurgEntropySimplex: eps should be positive."
        self.eps = eps
        self.last_newton_steps = 0
        self.last_bisect_steps = 0

    def extra_Psi(self, x):             # functions.py:210-211
        return 0

    def __call__(self, x):              # functions.py:242-244
        assert x.min() > 0, "BurgEntropy only takes positive arguments."
        return -_seq_sum(np.log(x))

    def gradient(self, x):              # functions.py:246-248
        assert x.min() > 0, "BurgEntropy only takes positive arguments."
        return -1 / x

    def divergence(self, x, y):         # functions.py:250-253
        assert x.shape == y.shape, "Vectors x and y are of different sizes."
        assert x.min() > 0 and y.min() > 0, "Entries of x or y not positive."
        r = x / y
        return _seq_sum(r - np.log(r) - 1)

    def prox_map(self, g, L):           # functions.py:336-356
        assert L > 0, "BergEntropySimplex prox_map only takes positive L."
        gg = g / L
        cmin = -gg.min()
        c = cmin + 1
        nb = 0
        while _seq_sum(1 / (gg + c)) - 1 < 0:       # :345-346
            c = (cmin + c) / 2.0
            nb += 1
        phi = _seq_sum(1 / (gg + c)) - 1            # :347
        nn = 0
        while abs(phi) > self.eps:                  # :349
            dphi = _seq_sum(-1.0 / (gg + c) ** 2)   # :350
            step = phi / dphi
            if (c - (c - step)) == 0:               # :351-352 stall test
                break
            c = c - step
            phi = _seq_sum(1 / (gg + c)) - 1        # :354
            nn += 1
        self.last_newton_steps, self.last_bisect_steps = nn, nb
        return 1.0 / (gg + c)                       # :355 (not renormalised)

    def div_prox_map(self, y, g, L):    # functions.py:264-271
        assert y.shape == g.shape, "Vectors y and g are of different sizes."
        assert y.min() > 0 and L > 0, "Either y or L is not positive."
        return self.prox_map(g - L * self.gradient(y), L)


# --------------------------------------------------------------------------
# problem factory                          accbpg/applications.py:36-56
# --------------------------------------------------------------------------
def design_matrix(m, n, randseed=-1):
    """The Gaussian design matrix of D_opt_design: the legacy global RNG is
    seeded only when randseed > 0 (applications.py:47-49)."""
    if randseed > 0:
        np.random.seed(randseed)
    return np.random.randn(m, n)


def D_opt_design(m, n, randseed=-1):
    V = design_matrix(m, n, randseed)
    return DOptOracle(V), BurgSimplexOracle(), 1.0, (1.0 / n) * np.ones(n)


# --------------------------------------------------------------------------
# solvers                                  accbpg/algorithms.py
# --------------------------------------------------------------------------
def solve_theta(theta, gamma, gainratio=1):
    """Scalar Newton for (1-t)/t^gamma = gainratio/theta^gamma
    (algorithms.py:75-91): tolerance 1e-6*theta, start at theta."""
    ckg = theta ** gamma / gainratio
    t = theta
    tol = 1e-6 * theta
    phi = t ** gamma - ckg * (1 - t)
    while abs(phi) > tol:
        t = t - phi / (gamma * t ** (gamma - 1) + ckg)
        phi = t ** gamma - ckg * (1 - t)
    return t


class _Counter:
    """Counts oracle calls per solver run (for the bench call-mix report)."""

    def __init__(self):
        self.value = self.grad = self.prox = self.div = 0


def BPG(f, h, L, x0, maxitrs, epsilon=1e-14, linesearch=True, ls_ratio=1.2,
        verbose=False, verbskip=1):
    """algorithms.py:11-72.  F[k] is recorded at the pre-update point (:47),
    L is carried across iterations and divided by ls_ratio first (:51), the
    stop test runs after the update (:66) and the arrays are cut to k+1."""
    t0 = time.time()
    F = np.zeros(maxitrs)
    Ls = np.ones(maxitrs) * L
    T = np.zeros(maxitrs)
    x = np.copy(x0)
    k = -1
    for k in range(maxitrs):
        fx, g = f.func_grad(x)
        F[k] = fx + h.extra_Psi(x)
        T[k] = time.time() - t0
        if linesearch:
            L = L / ls_ratio
            cand = h.div_prox_map(x, g, L)
            while f(cand) > fx + np.dot(g, cand - x) + L * h.divergence(cand, x):
                L = L * ls_ratio
                cand = h.div_prox_map(x, g, L)
            x = cand
        else:
            x = h.div_prox_map(x, g, L)
        Ls[k] = L
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:6.1f}".format(k, F[k], L, T[k]))
        if k > 0 and abs(F[k] - F[k - 1]) < epsilon:
            break
    return x, F[:k + 1], Ls[:k + 1], T[:k + 1]


def ABPG(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=False,
         restart=False, restart_rule='g', verbose=False, verbskip=1):
    """algorithms.py:94-180."""
    t0 = time.time()
    F = np.zeros(maxitrs)
    G = np.zeros(maxitrs)
    T = np.zeros(maxitrs)
    x = np.copy(x0)
    z = np.copy(x0)
    theta, kk = 1.0, 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)                        # :135-136
        T[k] = time.time() - t0
        z_prev, x_prev = z, x
        if theta_eq and kk > 0:                             # :142-145
            theta = solve_theta(theta, gamma)
        else:
            theta = gamma / (kk + gamma)
        y = (1 - theta) * x + theta * z_prev                # :147
        g = f.gradient(y)                                   # :148
        z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * L)   # :149
        x = (1 - theta) * x + theta * z                     # :150
        dxy = h.divergence(x, y)                            # :153
        dzz = h.divergence(z, z_prev)                       # :154
        Gdr = dxy / dzz / theta ** gamma                    # :155
        G[k] = Gdr
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:6.1f}".format(
                k, F[k], theta, Gdr, dxy, dzz, T[k]))
        kk += 1
        if restart and k > 0:                               # :165-171
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and np.dot(g, x - x_prev) > 0):
                theta, kk, z = 1.0, 0, x
        if dzz < epsilon:                                   # :174
            break
    return x, F[:k + 1], G[:k + 1], T[:k + 1]


def ABPG_gain(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, G0=1,
              ls_inc=1.2, ls_dec=1.2, theta_eq=True, checkdiv=False,
              restart=False, restart_rule='g', verbose=False, verbskip=1):
    """algorithms.py:295-420, including the quirk that an inner ``break`` on
    dzz < epsilon leaves Gdr at its previous value (:379-382) and that the
    restart test has no k > 0 guard (:403-406)."""
    t0 = time.time()
    F = np.zeros(maxitrs)
    Gain = np.ones(maxitrs) * G0
    Gdiv = np.zeros(maxitrs)
    Gavg = np.zeros(maxitrs)
    T = np.zeros(maxitrs)
    x = np.copy(x0)
    z = np.copy(x0)
    G = G0
    sumlogG = gamma * np.log(G)                             # :342
    theta, kk = 1.0, 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)                        # :347-348
        T[k] = time.time() - t0
        z_prev, x_prev = z, x
        G_prev, theta_prev = G, theta
        G = G / ls_dec                                      # :358
        retry = True
        while retry:                                        # :361
            if kk > 0:
                if theta_eq:
                    theta = solve_theta(theta_prev, gamma, G / G_prev)
                else:
                    alpha = G / G_prev
                    theta = theta_prev * ((1 + alpha * (gamma - 1)) / (gamma * alpha + theta_prev))
            y = (1 - theta) * x_prev + theta * z_prev       # :369
            fy, g = f.func_grad(y)                          # :371
            z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * G * L)   # :373
            x = (1 - theta) * x_prev + theta * z            # :374
            dxy = h.divergence(x, y)
            dzz = h.divergence(z, z_prev)
            if dzz < epsilon:                               # :379-380
                break
            Gdr = dxy / dzz / theta ** gamma                # :382
            if checkdiv:
                retry = (Gdr > G)
            else:
                retry = (f(x) > fy + np.dot(g, x - y) + theta ** gamma * G * L * dzz)   # :387
            if retry:
                G = G * ls_inc
        Gain[k] = G
        Gdiv[k] = Gdr
        sumlogG += np.log(G)
        Gavg[k] = np.exp(sumlogG / (gamma + k))             # :395-396
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:10.3e}  {5:10.3e}  {6:10.3e}  {7:10.3e}  {8:6.1f}".format(
                k, F[k], theta, G, Gdr, dxy, dzz, Gavg[k], T[k]))
        kk += 1
        if restart:                                         # :403-409
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and np.dot(g, x - x_prev) > 0):
                theta, kk, z = 1.0, 0, x
        if dzz < epsilon:                                   # :412
            break
    return x, F[:k + 1], Gain[:k + 1], Gdiv[:k + 1], Gavg[:k + 1], T[:k + 1]


# --------------------------------------------------------------------------
# Frank-Wolfe solvers                      accbpg/D_opt_alg.py
# --------------------------------------------------------------------------
def _fw_setup(V, x0):
    """D_opt_alg.py:39-45 / :123-129: explicit inverse of the Gram matrix and
    w_i = v_i^T H v_i for every design point."""
    x = np.copy(x0)
    gram = np.dot(V * x, V.T)
    det = np.linalg.det(gram)
    H = np.linalg.inv(gram)
    w = np.sum(V * np.dot(H, V), axis=0)
    return x, det, H, w


def D_opt_FW(V, x0, eps, maxitrs, verbose=False, verbskip=1):
    """D_opt_alg.py:9-88 (Fedorov-Wynn step with exact line search)."""
    t0 = time.time()
    m, n = V.shape
    F = np.zeros(maxitrs); SP = np.zeros(maxitrs)
    SN = np.zeros(maxitrs); T = np.zeros(maxitrs)
    x, det, H, w = _fw_setup(V, x0)
    k = -1
    for k in range(maxitrs):
        F[k] = -np.log(det)                                 # :52
        T[k] = time.time() - t0
        i = np.argmax(w)                                    # :59
        w_support = w[x > 0]                                # :60
        j = np.argmin(w_support)                            # :61
        eps_pos = w[i] / m - 1
        eps_neg = 1 - w_support[j] / m
        SP[k], SN[k] = eps_pos, eps_neg
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(
                k, F[k], eps_pos, eps_neg, T[k]))
        if eps_pos <= eps and eps_neg <= eps:               # :72
            break
        t = (w[i] / m - 1) / (w[i] - 1)                     # :75
        x *= (1 - t)
        x[i] += t
        Hv = np.dot(H, V[:, i])
        coef = t / (1 + t * (w[i] - 1))                     # :79,:82
        H = (H - coef * np.outer(Hv, Hv)) / (1 - t)
        det *= np.power(1 - t, m - 1) * (1 + t * (w[i] - 1))    # :80
        w = (w - coef * np.dot(Hv, V) ** 2) / (1 - t)       # :82
    return x, F[:k + 1], SP[:k + 1], SN[:k + 1], T[:k + 1]


def D_opt_FW_away(V, x0, eps, maxitrs, verbose=False, verbskip=1):
    """D_opt_alg.py:91-185 (Wolfe-Atwood).  F[k] is log det of the maintained
    inverse, recomputed every iteration (:136); the away index is the flat
    argmin of (w - w[i]) * [x > 1e-8] (:146-147)."""
    t0 = time.time()
    m, n = V.shape
    F = np.zeros(maxitrs); SP = np.zeros(maxitrs)
    SN = np.zeros(maxitrs); T = np.zeros(maxitrs)
    x, det, H, w = _fw_setup(V, x0)
    k = -1
    for k in range(maxitrs):
        F[k] = np.log(np.linalg.det(H))                     # :136
        T[k] = time.time() - t0
        i = np.argmax(w)
        shifted = w - w[i]
        j = np.argmin(shifted * [x > 1.0e-8])               # :147
        eps_pos = w[i] / m - 1
        eps_neg = 1 - w[j] / m
        SP[k], SN[k] = eps_pos, eps_neg
        if verbose and k % verbskip == 0:
            print("{0:6d}  {1:10.3e}  {2:10.3e}  {3:10.3e}  {4:6.1f}".format(
                k, F[k], eps_pos, eps_neg, T[k]))
        if eps_pos <= eps and eps_neg <= eps:
            break
        if eps_pos >= eps_neg:                              # :162-170
            t = (w[i] / m - 1) / (w[i] - 1)
            x *= (1 - t)
            x[i] += t
            Hv = np.dot(H, V[:, i])
            coef = t / (1 - t + t * w[i])
            H = (H - coef * np.outer(Hv, Hv)) / (1 - t)
            w = (w - coef * np.dot(Hv, V) ** 2) / (1 - t)
        else:                                               # :171-179
            t = min((1 - w[j] / m) / (w[j] - 1), x[j] / (1 - x[j]))
            x *= (1 + t)
            x[j] -= t
            Hv = np.dot(H, V[:, j])
            coef = t / (1 + t - t * w[j])
            H = (H + coef * np.outer(Hv, Hv)) / (1 + t)
            w = (w + coef * np.dot(Hv, V) ** 2) / (1 + t)
    return x, F[:k + 1], SP[:k + 1], SN[:k + 1], T[:k + 1]


# ==========================================================================
# callers and data formats either side of the path
# ==========================================================================
def ABPG_expo(f, h, L, x0, gamma0, maxitrs, epsilon=1e-14, delta=0.2, theta_eq=True,
              checkdiv=False, Gmargin=10, restart=False, restart_rule='g', verbose=False, verbskip=1):
    """Exponent-adaptive ABPG (algorithms.py:183-292): the gradient at y is taken once per outer
    iteration (:245); the inner loop lowers gamma by delta while the test fails and gamma > 1
    (:262-265); restart has no k > 0 guard (:276-282)."""
    t0 = time.time()
    F = np.zeros(maxitrs); G = np.zeros(maxitrs)
    Gamma = np.ones(maxitrs) * gamma0; T = np.zeros(maxitrs)
    gamma = gamma0
    x = np.copy(x0); z = np.copy(x0)
    theta, kk = 1.0, 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)
        T[k] = time.time() - t0
        z_prev, x_prev = z, x
        theta = solve_theta(theta, gamma) if (theta_eq and kk > 0) else gamma / (kk + gamma)
        y = (1 - theta) * x_prev + theta * z_prev
        fy, g = f.func_grad(y)
        again = True
        while again:
            z = h.div_prox_map(z_prev, g, theta ** (gamma - 1) * L)
            x = (1 - theta) * x_prev + theta * z
            dxy = h.divergence(x, y)
            dzz = h.divergence(z, z_prev)
            Gdr = dxy / dzz / theta ** gamma
            if checkdiv:
                again = (dxy > Gmargin * (theta ** gamma) * dzz)
            else:
                again = (f(x) > fy + np.dot(g, x - y) + theta ** gamma * L * dzz)
            if again and gamma > 1:
                gamma = max(gamma - delta, 1)
            else:
                again = False
        G[k] = Gdr
        Gamma[k] = gamma
        kk += 1
        if restart:
            if (restart_rule == 'f' and F[k] > F[k - 1]) or \
               (restart_rule == 'g' and np.dot(g, x - x_prev) > 0):
                theta, kk, z = 1.0, 0, x
        if dzz < epsilon:
            break
    return x, F[:k + 1], Gamma[:k + 1], G[:k + 1], T[:k + 1]


def ABDA(f, h, L, x0, gamma, maxitrs, epsilon=1e-14, theta_eq=True, verbose=False, verbskip=1):
    """Accelerated Bregman dual averaging (algorithms.py:423-514): weighted gradient average
    gavg += theta^(1-gamma) g (:483), z = prox_map(gavg/csum, L/csum) (:485)."""
    t0 = time.time()
    F = np.zeros(maxitrs); G = np.zeros(maxitrs); T = np.zeros(maxitrs)
    x = np.copy(x0); z = np.copy(x0)
    theta, kk = 1.0, 0
    gavg = np.zeros(x.size)
    csum = 0
    k = -1
    for k in range(maxitrs):
        F[k] = f(x) + h.extra_Psi(x)
        T[k] = time.time() - t0
        z_prev, x_prev = z, x
        theta = solve_theta(theta, gamma) if (theta_eq and kk > 0) else gamma / (kk + gamma)
        y = (1 - theta) * x_prev + theta * z_prev
        g = f.gradient(y)
        gavg = gavg + theta ** (1 - gamma) * g
        csum = csum + theta ** (1 - gamma)
        z = h.prox_map(gavg / csum, L / csum)
        x = (1 - theta) * x_prev + theta * z
        dxy = h.divergence(x, y)
        dzz = h.divergence(z, z_prev)
        G[k] = dxy / dzz / theta ** gamma
        kk += 1
        if dzz < epsilon:
            break
    return x, F[:k + 1], G[:k + 1], T[:k + 1]


def lmo_simplex(radius=1):
    """Simplex LMO (functions_lmo.py:137-160): 1e-15 everywhere, `radius` at the first argmin."""
    def vertex(g):
        s = np.zeros(g.shape)
        s += 1e-15
        s[np.where(g == np.min(g))[0][0]] = radius
        return s
    return vertex


def FW_alg_div_step(f, h, L, x0, maxitrs, gamma, lmo, epsilon=1e-14, linesearch=True, ls_ratio=2,
                    verbose=False, verbskip=1):
    """Frank-Wolfe with a Bregman step size (algorithms_fw.py:6-75)."""
    if ls_ratio < 1:
        raise ValueError("ls_ratio must be >= 1")
    if L <= 0:
        raise ValueError("Initial L must be positive")
    if epsilon <= 0:
        raise ValueError("epsilon must be positive")
    t0 = time.time()
    F, Ls, T = [], [], []
    tiny = 1e-6
    x = np.copy(x0)
    for k in range(maxitrs):
        fx, g = f.func_grad(x)
        F.append(fx + h.extra_Psi(x))
        T.append(time.time() - t0)
        s = lmo(g)
        d = s - x
        div = h.divergence(s, x)
        if div == 0:
            div = tiny
        slope = np.dot(g.ravel(), d.ravel())
        if 0 < slope <= tiny:
            slope = 0.0
        if slope > 0:
            raise ValueError("grad_d_prod must be non-positive")
        if linesearch:
            L = L / ls_ratio
        while True:
            alpha = min((-slope / (2 * L * div)) ** (1 / (gamma - 1)), 1.0)
            x1 = x + alpha * d
            if not linesearch:
                break
            assert not math.isinf(L), "L is infinite"
            if f.func_grad(x1, flag=0) <= fx + alpha * slope + alpha ** gamma * L * div:
                break
            L = L * ls_ratio
        x = x1
        Ls.append(L)
        if k > 0 and abs(F[k] - F[k - 1]) < epsilon:
            break
    return x, np.array(F), np.array(Ls), np.array(T)


def D_opt_KYinit(V):
    """Kumar-Yildirim sparse starting point (applications.py:59-95).  Uses the legacy global RNG
    (np.random.rand(m) per direction) and a Gram-Schmidt whose coefficients come from the
    un-deflated vector (:75-78, :86-89)."""
    m, n = V.shape
    if n <= 2 * m:
        return (1.0 / n) * np.ones(n)
    picked = []
    Q = np.zeros((m, m))
    for i in range(m):
        b = np.random.rand(m)
        q = np.copy(b)
        for j in range(i):
            q = q - np.dot(Q[:, j], b) * Q[:, j]
        proj = np.dot(q, V)
        kmax = np.argmax(proj)
        kmin = np.argmin(proj)
        picked.append(kmax)
        picked.append(kmin)
        v = V[:, kmin] - V[:, kmax]
        q = np.copy(v)
        for j in range(i):
            q = q - np.dot(Q[:, j], v) * Q[:, j]
        Q[:, i] = q / np.linalg.norm(q)
    x0 = np.zeros(n)
    x0[picked] = np.ones(len(picked)) / len(picked)
    x0 /= x0.sum()
    return x0


def load_libsvm_file(filename):
    """LIBSVM text -> (dense samples x features array, labels); index base detected as in
    utils.py:22-95 (indices shifted down when the smallest one is > 0)."""
    labels, rows = [], []
    with open(filename, "r") as fh:
        for line in fh:
            cut = line.find('#')
            if cut >= 0:
                line = line[:cut]
            parts = line.split()
            if not parts:
                continue
            labels.append(float(parts[0]))
            entry, prev = [], -1
            for tok in parts[1:]:
                idx_s, val = tok.split(':', 1)
                idx = int(idx_s)
                if idx < 0:
                    raise ValueError("Invalid index {0:d} in LibSVM data file.".format(idx))
                if idx <= prev:
                    raise ValueError("Feature indices in LibSVM data fileshould be sorted and unique.")
                entry.append((idx, float(val)))
                prev = idx
            rows.append(entry)
    all_idx = [i for r in rows for i, _ in r]
    shift = 1 if min(all_idx) > 0 else 0
    nfeat = max(all_idx) - shift + 1
    X = np.zeros((len(rows), nfeat))
    for r, entry in enumerate(rows):
        for i, v in entry:
            X[r, i - shift] = v
    return X, np.array(labels)


def D_opt_libsvm(filename):
    """applications.py:17-33: design matrix = the data matrix, transposed if it is tall."""
    X, y = load_libsvm_file(filename)
    H = np.ascontiguousarray(X.T) if X.shape[0] > X.shape[1] else np.ascontiguousarray(X)
    n = H.shape[1]
    return DOptOracle(H), BurgSimplexOracle(), 1.0, (1.0 / n) * np.ones(n)


# --------------------------------------------------------------------------
# Poisson linear inverse problem with Burg L1 / L2 kernels
#                                          accbpg/functions.py:85-120, 274-323
#                                          accbpg/applications.py:98-175
# --------------------------------------------------------------------------
class PoissonOracle:
    """f(x) = D_KL(b, Ax) = sum_i b_i log(b_i/(Ax)_i) + (Ax)_i - b_i.

    Follows functions.py:103-120: one matrix-vector product (:105), the value is
    a left-to-right builtin ``sum`` (:107, :119), the gradient is the row sum over
    axis 0 of (1 - b/Ax)[:, None] * A (:111), flag 0 returns before the gradient."""

    def __init__(self, A, b):
        assert A.shape[0] == b.shape[0], "A and b sizes not matching"
        self.A, self.b = A, b
        self.m, self.n = A.shape

    def func_grad(self, x, flag=2):
        assert x.size == self.n, "PoissonRegression: x.size not equal to n."
        Ax = np.dot(self.A, x)
        if flag == 0:
            return _seq_sum(self.b * np.log(self.b / Ax) + Ax - self.b)
        grad = ((1 - self.b / Ax).reshape(self.m, 1) * self.A).sum(axis=0)
        if flag == 1:
            return grad
        return _seq_sum(self.b * np.log(self.b / Ax) + Ax - self.b), grad

    def __call__(self, x):
        return self.func_grad(x, flag=0)

    def gradient(self, x):
        return self.func_grad(x, flag=1)


class BurgOracle(BurgSimplexOracle):
    """Unconstrained Burg entropy (functions.py:238-271): prox_map is L/g for g > 0."""

    def __init__(self):
        pass

    def prox_map(self, g, L):           # functions.py:255-262
        assert L > 0, "BurgEntropy prox_map only takes positive L value."
        assert g.min() > 0, "BurgEntropy prox_map only takes positive value."
        return L / g


class BurgL1Oracle(BurgOracle):
    """Burg entropy with Psi(x) = lamda*||x||_1 on x > 0 (functions.py:274-298)."""

    def __init__(self, lamda=0, x_max=1e4):
        assert lamda >= 0, "BurgEntropyL1: lambda should be nonnegative."
        self.lamda, self.x_max = lamda, x_max

    def extra_Psi(self, x):             # :284-288
        return self.lamda * x.sum()

    def prox_map(self, g, L):           # :290-298
        assert L > 0, "BurgEntropyL1: prox_map only takes positive L."
        assert g.min() > -self.lamda, "Not getting positive solution."
        return L / (self.lamda + g)


class BurgL2Oracle(BurgOracle):
    """Burg entropy with Psi(x) = (lamda/2)*||x||_2^2 (functions.py:301-323)."""

    def __init__(self, lamda=0):
        assert lamda >= 0, "BurgEntropyL2: lamda should be nonnegative."
        self.lamda = lamda

    def extra_Psi(self, x):             # :310-314
        return (self.lamda / 2) * np.dot(x, x)

    def prox_map(self, g, L):           # :316-323 (positive root of lamda_L*t^2 + gg*t - 1)
        assert L > 0, "BurgEntropyL2: prox_map only takes positive L value."
        gg = g / L
        lamda_L = self.lamda / L
        return (np.sqrt(gg * gg + 4 * lamda_L) - gg) / (2 * lamda_L)


def poisson_instance(m, n, noise=0.01, randseed=-1, normalizeA=True):
    """(A, b) of Poisson_regrL1/L2 (applications.py:114-123 == :153-162): uniform A with unit column
    sums, a sparse nonnegative x, b = A x + centred uniform noise; the legacy global RNG is drawn in the
    order A, x, noise."""
    if randseed > 0:
        np.random.seed(randseed)
    A = np.random.rand(m, n)
    if normalizeA:
        A = A / A.sum(axis=0)
    x = np.random.rand(n) / n
    xavg = x.sum() / x.size
    x = np.maximum(x - xavg, 0) * 10
    b = np.dot(A, x) + noise * (np.random.rand(m) - 0.5)
    assert b.min() > 0, "need b > 0 for nonnegative regression."
    return A, b


def Poisson_regrL1(m, n, noise=0.01, lamda=0, randseed=-1, normalizeA=True):
    A, b = poisson_instance(m, n, noise, randseed, normalizeA)
    return PoissonOracle(A, b), BurgL1Oracle(lamda), b.sum(), (1.0 / n) * np.ones(n) * 10   # :125-131


def Poisson_regrL2(m, n, noise=0.01, lamda=0, randseed=-1, normalizeA=True):
    A, b = poisson_instance(m, n, noise, randseed, normalizeA)
    return PoissonOracle(A, b), BurgL2Oracle(lamda), b.sum(), (1.0 / n) * np.ones(n)        # :164-170
