import sys, time; sys.path.insert(0, ".")
import numpy as np, torch
import accbpg_and_fw_amd as acc
f, h, L, x0 = acc.D_opt_design(2048, 32768, randseed=10)
t = time.time()
x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, torch.from_numpy(x0).cuda(), gamma=2, maxitrs=1000, verbose=False)
dt = time.time() - t
xs = x.cpu().numpy()
print("1000 its %.1f s (%.1f it/s); F[0]=%.6f F[-1]=%.6f min dF=%.3e; x min %.3e sum-1 %.3e; calls %s; T[-1]=%.1f" % (
    dt, len(F) / dt, F[0], F[-1], np.min(F[:-1] - F[1:]), xs.min(), xs.sum() - 1, f.calls, T[-1]))
print("monotone F:", bool(np.all(np.diff(F) <= 1e-9)), " iterations", len(F))
