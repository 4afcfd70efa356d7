"""One-level against two-level Cholesky (development aid): agreement of f and the gradient at small sizes with
the two-level scheme forced on, then the time of either at the sizes where the choice matters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import _lib
lib = _lib.load()


def setting(f, T):
    lib.accbpg_debug_chol_variant(f._h, 2048 | (T << 12))


for m, n in [(64, 256), (300, 900), (1100, 3000), (2048, 8192), (2080, 4200)]:
    V = torch.randn(m, n, dtype=torch.float64, device="cuda")
    f = acc.DOptimalObj(V)
    x = torch.rand(n, dtype=torch.float64, device="cuda") + 0.5
    x /= x.sum()
    setting(f, 4000)
    f1, g1 = f.func_grad(x, 2)
    g1 = g1.clone()
    setting(f, 1)
    f2, g2 = f.func_grad(x, 2)
    print("m %5d  f one-level %.15g two-level %.15g  rel %.2e   grad rel %.2e   sum x(-g) - m = %.2e" % (
        m, f1, f2, abs(f1 - f2) / abs(f1), float((g1 - g2).abs().max() / g1.abs().max()),
        float(-(x * g2).sum()) - m), flush=True)
    del f, V

for m, n in [(3072, 8192), (4096, 8192), (6144, 8192), (8192, 16384), (8256, 16384)]:
    V = torch.randn(m, n, dtype=torch.float64, device="cuda")
    f = acc.DOptimalObj(V)
    x = torch.rand(n, dtype=torch.float64, device="cuda") + 0.5
    x /= x.sum()
    for T, nk in [(4000, 4), (1, 2), (1, 3), (1, 4), (1, 6), (1, 8)]:
        lib.accbpg_debug_chol_variant(f._h, 2048 | (T << 12) | (nk << 24))
        f.profile(True)
        for _ in range(4):
            fx = f(x)
        p = f.profile_read()
        print("m %5d  %s  cholesky %.3f ms   f %.15g" % (m, "one-level      " if T > 1 else "two-level nk=%d" % nk,
                                                        p["cholesky"][0] / p["cholesky"][1], fx), flush=True)
    del f, V
