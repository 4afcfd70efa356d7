"""Time the Gram kernel ablation variants on the GPU (development aid)."""
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
names = {10: "glds full", 11: "glds no loads", 12: "glds no wait+barrier", 13: "glds no loads no barrier",
         14: "glds no fragment reads", 15: "glds barrier w/o vmcnt wait", 16: "glds MFMA only", 17: "glds half of the fragment reads"}
for rep in range(2):
    for var in [10, 11, 12, 13, 14, 15, 16, 17]:
        ms = C.c_double(0.0)
        rc = lib.accbpg_debug_gram_variant(f._h, C.c_void_p(x.data_ptr()), var, 10, C.byref(ms))
        print("variant %d (%-24s) rc=%d  %.3f ms" % (var, names[var], rc, ms.value), flush=True)
