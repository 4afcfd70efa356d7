"""Time the Cholesky with parts of chol_step_kernel switched off (development aid)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import _lib
m, n = 2048, 32768
V = torch.randn(m, n, dtype=torch.float64, device="cuda")
f = acc.DOptimalObj(V)
x = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
lib = _lib.load()
names = {0: "full", 1: "no potrf", 2: "no trsm", 3: "no potrf, no trsm", 4: "no mfma", 7: "loads/stores only",
         8: "potrf: no row solves", 16: "potrf: no MFMA updates", 32: "potrf: no 16x16 factor", 56: "potrf: barriers only"}
for bits in [0, 8, 16, 32, 56, 1, 2, 3, 4, 7, 0]:
    lib.accbpg_debug_chol_variant(f._h, bits)
    f.profile(True)
    for _ in range(5):
        try:
            f(x)
        except Exception:
            pass
    p = f.profile_read()
    if bits == 0:
        fx = f(x)
        print("   f(x) = %.15g" % fx)
    print("bits %d (%-18s) cholesky %.3f ms" % (bits, names[bits], p["cholesky"][0] / p["cholesky"][1]), flush=True)
lib.accbpg_debug_chol_variant(f._h, 0)
