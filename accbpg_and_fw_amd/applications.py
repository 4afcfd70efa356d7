"""Problem factory for the D-optimal design benchmark (accbpg/applications.py:36-56)."""
from __future__ import annotations

import numpy as np

from .functions import (BurgEntropyL1, BurgEntropyL2, BurgEntropySimplex, DOptimalObj, PoissonRegression,
                        vec_argminmax)
from .utils import load_libsvm_file


def D_opt_libsvm(filename):
    """D-optimal design instance from a LIBSVM data file (accbpg/applications.py:17-33): the data
    matrix, transposed when it has more rows than columns, is the m x n design matrix."""
    X, y = load_libsvm_file(filename)
    if X.shape[0] > X.shape[1]:
        H = X.T.toarray('C')
    else:
        H = X.toarray('C')
    n = H.shape[1]
    f = DOptimalObj(H)
    h = BurgEntropySimplex()
    L = 1.0
    x0 = (1.0 / n) * np.ones(n)
    return f, h, L, x0


def D_opt_KYinit(V):
    """Sparse Kumar-Yildirim starting point (accbpg/applications.py:59-95).  ``V`` is the design
    matrix or a DOptimalObj over it.  The m passes over V (q^T V, argmax / argmin, two column reads)
    run on the GPU; the length-m Gram-Schmidt recurrences stay on the host in the reference's order
    (coefficients from the un-deflated vector, :75-78 and :86-89), with the same legacy-RNG draws
    (np.random.rand(m) per direction, :74)."""
    obj = V if isinstance(V, DOptimalObj) else None
    m, n = (obj.m, obj.n) if obj is not None else V.shape
    if n <= 2 * m:
        return (1.0 / n) * np.ones(n)
    if obj is None:
        obj = DOptimalObj(V)

    picked = []
    Q = np.zeros((m, m))
    for i in range(m):
        b = np.random.rand(m)
        q = np.copy(b)
        for j in range(i):
            q = q - np.dot(Q[:, j], b) * Q[:, j]
        kmin, kmax, _, _ = vec_argminmax(obj.vt_times(q))       # :79-81
        picked.append(kmax)
        picked.append(kmin)
        v = obj.column(kmin) - obj.column(kmax)                 # :84
        q = np.copy(v)
        for j in range(i):
            q = q - np.dot(Q[:, j], v) * Q[:, j]
        Q[:, i] = q / np.linalg.norm(q)

    x0 = np.zeros(n)
    x0[picked] = np.ones(len(picked)) / len(picked)
    x0 /= x0.sum()                                              # repeated indices: rescale to sum 1 (:93-94)
    return x0


def D_opt_design(m, n, randseed=-1):
    """Random Gaussian instance: returns (f, h, L, x0) with f = DOptimalObj(H),
    h = BurgEntropySimplex(), L = 1, x0 = centre of the simplex.  As in the reference
    the legacy global NumPy generator is seeded only if randseed > 0
    (applications.py:47-49), so the same seed gives the same H on both sides."""
    if randseed > 0:
        np.random.seed(randseed)
    H = np.random.randn(m, n)
    f = DOptimalObj(H)
    h = BurgEntropySimplex()
    L = 1.0
    x0 = (1.0 / n) * np.ones(n)
    return f, h, L, x0


def _poisson_instance(m, n, noise, randseed, normalizeA):
    """(A, b) of the random Poisson linear inverse problem (accbpg/applications.py:114-123, 153-162):
    legacy global RNG drawn in the order A, x, noise; generated on the host like the reference's."""
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
    """minimize_{x >= 0} D_KL(b, Ax) + lamda*||x||_1  (accbpg/applications.py:98-133).
    Returns f, h, L = ||b||_1, x0 = (10/n)*ones."""
    A, b = _poisson_instance(m, n, noise, randseed, normalizeA)
    return PoissonRegression(A, b), BurgEntropyL1(lamda), b.sum(), (1.0 / n) * np.ones(n) * 10


def Poisson_regrL2(m, n, noise=0.01, lamda=0, randseed=-1, normalizeA=True):
    """minimize_{x >= 0} D_KL(b, Ax) + (lamda/2)*||x||_2^2  (accbpg/applications.py:136-172).
    Returns f, h, L = ||b||_1, x0 = (1/n)*ones."""
    A, b = _poisson_instance(m, n, noise, randseed, normalizeA)
    return PoissonRegression(A, b), BurgEntropyL2(lamda), b.sum(), (1.0 / n) * np.ones(n)
