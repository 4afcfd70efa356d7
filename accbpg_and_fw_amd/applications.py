"""Problem factory for the D-optimal design benchmark (accbpg/applications.py:36-56)."""
from __future__ import annotations

import numpy as np

from .functions import BurgEntropySimplex, DOptimalObj


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
