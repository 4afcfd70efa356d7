"""Repeat one evaluation many times and compare bitwise: which stage is not reproducible? (development aid)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import _lib
lib = _lib.load()
for (m, n) in [(300, 3000), (300, 1111), (80, 200), (256, 4096)]:
    f, h, L, x0 = acc.D_opt_design(m, n, randseed=21)
    rng = np.random.RandomState(5)
    x = rng.rand(n) + 0.01; x /= x.sum()
    xd = torch.from_numpy(x).cuda()
    G0 = torch.empty(m, m, dtype=torch.float64, device="cuda")
    G1 = torch.empty_like(G0)
    lib.accbpg_dopt_gram(f._h, C.c_void_p(xd.data_ptr()), C.c_void_p(G0.data_ptr()))
    torch.cuda.synchronize()
    low = torch.tril(torch.ones(m, m, dtype=torch.bool, device="cuda"))
    nbad_gram = 0
    for it in range(3000):
        lib.accbpg_dopt_gram(f._h, C.c_void_p(xd.data_ptr()), C.c_void_p(G1.data_ptr()))
        if not torch.equal(G1[low], G0[low]):
            nbad_gram += 1
            d = (G1 - G0).abs() * low
            idx = torch.nonzero(d > 0)
            print("   gram (%d,%d) it %d: %d entries differ, max %.3e, first at %s" % (m, n, it, idx.shape[0], float(d.max()), idx[0].tolist()), flush=True)
            if nbad_gram > 3: break
    f0, g0 = f.func_grad(xd, 2)
    nbad_f = nbad_g = 0
    for it in range(3000):
        f1, g1 = f.func_grad(xd, 2)
        if f1 != f0:
            nbad_f += 1
            print("   value (%d,%d) it %d: f differs by %.3e" % (m, n, it, f1 - f0), flush=True)
        if not torch.equal(g1, g0):
            nbad_g += 1
            d = (g1 - g0).abs()
            print("   grad  (%d,%d) it %d: %d entries differ, max rel %.3e" % (m, n, it, int((d > 0).sum()), float((d / g0.abs()).max())), flush=True)
        if nbad_f + nbad_g > 6: break
    print("(%d,%d): gram mismatches %d/3000, value %d/3000, gradient %d/3000" % (m, n, nbad_gram, nbad_f, nbad_g), flush=True)
