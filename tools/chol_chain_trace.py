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
names = ["start", "left updates in", "prev factor seen", "fetched+staged", "panel solve", "publish+diag update", "potrf", "publish factor"]
acc_d = np.zeros((T, 8))
reps = 20
for r in range(reps + 2):
    st = (C.c_int64 * (8 * T))()
    rc = lib.accbpg_debug_chol_trace(f._h, C.c_void_p(gram.data_ptr()), winv, st)
    assert rc == 0, _lib.last_error()
    a = np.array(st, dtype=np.int64).reshape(T, 8).astype(np.float64) * 0.01     # microseconds
    if r >= 2:
        acc_d += a - a[0, 0]
a = acc_d / reps
print("m=%d T=%d: kernel start -> last factor published %.1f us  (%.2f us per block column)" % (m, T, a[-1, 7], a[-1, 7] / T))
d = np.diff(a[1:], axis=1)          # stages within a chain workgroup (d >= 1)
for k in range(1, 7):
    print("  %-22s mean %6.2f us   (min %.2f max %.2f)" % (names[k + 1], d[:, k].mean(), d[:, k].min(), d[:, k].max()))
hop = a[1:, 2] - a[:-1, 7]
print("  flag set -> seen by the next chain workgroup: mean %.2f us (min %.2f max %.2f)" % (hop.mean(), hop.min(), hop.max()))
wait = a[1:, 2] - a[1:, 1]
print("  chain workgroup idle before the factor arrives: mean %.2f us" % wait.mean())
print("  per block column (publish to publish): mean %.2f us" % np.diff(a[:, 7]).mean())
