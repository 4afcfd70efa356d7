"""Repeat short solver runs at ragged sizes and compare bitwise with the first run (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd.sharded import LogicalShards
t0 = time.time()
for (m, n, parts) in [(300, 1111, 4), (300, 3000, 3)]:
    np.random.seed(11); V = np.random.randn(m, n)
    f = acc.DOptimalObj(V); fs = LogicalShards(V, parts); h = acc.BurgEntropySimplex()
    x0 = np.ones(n) / n
    ref = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    refs = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    refg = acc.ABPG_gain(f, h, 1.0, x0, gamma=2, maxitrs=30, verbose=False)
    bad = [0, 0, 0]
    for rep in range(250):
        r = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
        rs = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
        rg = acc.ABPG_gain(f, h, 1.0, x0, gamma=2, maxitrs=30, verbose=False)
        for idx, (a, b, name) in enumerate([(ref, r, "abpg single"), (refs, rs, "abpg shards"), (refg, rg, "gain single")]):
            if not all(np.array_equal(p, q) for p, q in zip(a[:-1], b[:-1])):
                bad[idx] += 1
                dF = np.abs(a[1] - b[1]) if len(a[1]) == len(b[1]) else np.array([np.nan])
                k = int(np.argmax(dF > 0)) if np.any(dF > 0) else -1
                print("   (%d,%d) rep %d %s: first F mismatch at k=%d (|dF|=%.3e), max|dx|=%.3e" % (m, n, rep, name, k, dF[k] if k >= 0 else 0.0, np.max(np.abs(a[0] - b[0]))), flush=True)
    print("(%d,%d): mismatching runs single/shards/gain = %s of 250   [%.0f s]" % (m, n, bad, time.time() - t0), flush=True)
