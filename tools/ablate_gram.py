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
names = {0: "reg-staged full", 3: "reg: no loads/stage/barrier", 4: "MFMA only", 10: "glds full", 11: "glds no loads", 12: "glds no barrier", 13: "glds no loads no barrier"}
for rep in range(2):
    for var in [0, 3, 4, 10, 11, 12, 13]:
        ms = C.c_double(0.0)
        rc = lib.accbpg_debug_gram_variant(f._h, C.c_void_p(x.data_ptr()), var, 10, C.byref(ms))
        print("variant %d (%-24s) rc=%d  %.3f ms" % (var, names[var], rc, ms.value), flush=True)
