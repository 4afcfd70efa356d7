"""Host-side profile of the ABPG_gain loop at the bench shape (development aid)."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import algorithms as alg
m, n = 2048, 32768
np.random.seed(1)
V = torch.from_numpy(np.random.randn(m, n)).cuda()
f = acc.DOptimalObj(V); h = acc.BurgEntropySimplex()
x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
gen = alg.ABPG_gain_steps(f, h, 1.0, x0, 2, 100, verbose=False)
for _ in range(3): next(gen)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); t = time.perf_counter()
for _ in range(20): next(gen)
torch.cuda.synchronize(); dt = time.perf_counter() - t; pr.disable()
print("ms/step %.3f" % (dt / 20 * 1e3))
pstats.Stats(pr).sort_stats("tottime").print_stats(25)
