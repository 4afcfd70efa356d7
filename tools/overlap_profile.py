"""Kernel event timings with the two-stream value overlap switched on (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import algorithms as alg
m, n = 2048, 32768
np.random.seed(1)
V = torch.from_numpy(np.random.randn(m, n)).cuda()
f = acc.DOptimalObj(V); h = acc.BurgEntropySimplex()
x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
for ovl in (False, True):
    f.overlap_values(ovl)
    gen = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, 100, verbose=False)
    for _ in range(3): next(gen)
    f.profile(True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): next(gen)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    p = f.profile_read(); f.profile(False)
    print("overlap=%s  ms/step %.3f  " % (ovl, dt / 20 * 1e3), {k: (round(v[0] / v[1], 4), v[1]) for k, v in p.items() if v[1]}, flush=True)
