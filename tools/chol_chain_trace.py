"""Stage times of the one-launch Cholesky's critical chain (development aid; run on the GPU box).
    python tools/chol_chain_trace.py [m] [with_inverse]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import _lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
winv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = 4 * m
torch.manual_seed(0)
V = torch.randn(m, n, dtype=torch.float64, device="cuda")
f = acc.DOptimalObj(V)
x = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
gram = torch.empty(m, m, dtype=torch.float64, device="cuda")
f.gram_into(x, gram)
lib = _lib.load()
T = (m + 63) // 64
NS = 32
acc_d = np.zeros((T, NS))
reps = 20
for r in range(reps + 2):
    st = (C.c_int64 * (NS * T))()
    rc = lib.accbpg_debug_chol_trace(f._h, C.c_void_p(gram.data_ptr()), winv, st)
    assert rc == 0, _lib.last_error()
    a = np.array(st, dtype=np.int64).reshape(T, NS).astype(np.float64) * 0.01     # microseconds
    if r >= 2:
        acc_d += a - a[0, 0]
a = acc_d / reps
print("m=%d T=%d: kernel start -> last factor published %.1f us  (%.2f us per block column)" % (m, T, a[-1, 7], a[-1, 7] / T))
print("  per block column (publish to publish): mean %.2f us" % np.diff(a[:, 7]).mean())
# everything relative to T0 = the moment the PREVIOUS chain workgroup starts its factorisation (its stamp 5)
rows = range(2, T)
rel = lambda col: np.mean([a[d, col] - a[d - 1, 5] for d in rows])
print("  relative to the previous chain workgroup's potrf start (T0):")
print("    its potrf ends %.2f, its factor is published %.2f" % (np.mean([a[d - 1, 6] - a[d - 1, 5] for d in rows]),
                                                                 np.mean([a[d - 1, 7] - a[d - 1, 5] for d in rows])))
print("    this workgroup: left updates in %.2f" % rel(1))
for p in range(4):
    print("    piece %d: asked %.2f  in LDS %.2f  solved %.2f  folded %.2f" % (p, rel(8 + 4 * p), rel(9 + 4 * p), rel(10 + 4 * p), rel(11 + 4 * p)))
print("    diagonal tile staged (own potrf starts) %.2f" % rel(5))
