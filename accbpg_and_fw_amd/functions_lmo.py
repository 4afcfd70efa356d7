"""Linear minimisation oracle for the simplex (accbpg/functions_lmo.py:137-160): the vertex that
minimises <g, s>, returned with 1e-15 in every other entry as the reference does (so that Burg
divergences from it stay finite)."""
from __future__ import annotations

import numpy as np
import torch

from .functions import from_dev, to_dev, vec_argminmax, vec_vertex


def lmo_simplex(radius=1):
    """Returns g -> s with s[i] = 1e-15 and s[first argmin g] = radius (NumPy in, NumPy out; CUDA
    tensor in, CUDA tensor out: first-index argmin and the fill run on the device)."""
    def vertex(g):
        gd, was_np = to_dev(g)
        imin, _, _, _ = vec_argminmax(gd)
        return from_dev(vec_vertex(imin, radius, 1e-15, gd.numel(), gd.device), was_np)
    return vertex
