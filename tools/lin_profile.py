import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import algorithms as alg
m, n = 2048, 32768
np.random.seed(1); V = torch.from_numpy(np.random.randn(m, n)).cuda()
f = acc.DOptimalObj(V); h = acc.BurgEntropySimplex()
x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
for lin in (False, True):
    f.linear_gram(lin)
    gen = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, 100, verbose=False)
    for _ in range(3): next(gen)
    f.profile(True); c0 = dict(f.calls); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): next(gen)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = f.profile_read(); f.profile(False)
    print("lin=%s  %.2f ms/step  calls/step=%s" % (lin, 1e3 * dt / 10, {k: (f.calls[k] - c0[k]) / 10 for k in c0}))
    print("   per-step kernel ms:", {k: round(v[0] / 10, 3) for k, v in p.items()}, " launches/step:", {k: v[1] / 10 for k, v in p.items()})
