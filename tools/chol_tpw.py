"""Cholesky time against the number of trailing-update tiles one workgroup takes (development aid).
f(x) must not change with the setting: a tile's arithmetic does not depend on which workgroup runs it."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import _lib
lib = _lib.load()
for m, n in [(2048, 8192), (4096, 8192), (4160, 8192), (8192, 16384)]:
    V = torch.randn(m, n, dtype=torch.float64, device="cuda")
    f = acc.DOptimalObj(V)
    x = torch.rand(n, dtype=torch.float64, device="cuda") + 0.5
    x /= x.sum()
    ref = None
    for tpw in [1, 2, 4, 8, 16]:
        lib.accbpg_debug_chol_variant(f._h, 2048 | (tpw << 12))
        f.profile(True)
        for _ in range(4):
            fx = f(x)
        p = f.profile_read()
        ref = fx if ref is None else ref
        print("m %5d  tiles/workgroup <= %2d  cholesky %.3f ms   f %.17g %s" % (
            m, tpw, p["cholesky"][0] / p["cholesky"][1], fx, "same" if fx == ref else "DIFFERENT"), flush=True)
    del f, V
