"""Single-object reproducibility at the shard shape (300,1000) and at (300,3000) (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
h = acc.BurgEntropySimplex()
objs = []
for (m, n) in [(300, 1000), (300, 1008), (300, 3000)]:
    np.random.seed(11); V = np.random.randn(m, n)
    f = acc.DOptimalObj(V); x0 = np.ones(n) / n
    objs.append((m, n, f, x0, acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False), [0, 0]))
t0 = time.time()
while time.time() - t0 < budget:
    for (m, n, f, x0, ref, st) in objs:
        for _ in range(4):
            r = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False); st[1] += 1
            if not all(np.array_equal(p, q) for p, q in zip(ref[:-1], r[:-1])):
                st[0] += 1
                dF = np.abs(ref[1] - r[1]); k = int(np.argmax(dF > 0)) if np.any(dF > 0) else -1
                print("(%d,%d): first F mismatch k=%d max|dx| %.3e" % (m, n, k, np.max(np.abs(ref[0] - r[0]))), flush=True)
for (m, n, f, x0, ref, st) in objs:
    print("(%d,%d): %d of %d runs differ" % (m, n, st[0], st[1]), flush=True)
