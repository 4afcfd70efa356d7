"""Rate of non-reproducible short solver runs on logical shards at a ragged size (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd.sharded import LogicalShards
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
m, n, parts = 300, 3000, 3
np.random.seed(11); V = np.random.randn(m, n)
fs = LogicalShards(V, parts); f = acc.DOptimalObj(V); h = acc.BurgEntropySimplex()
x0 = np.ones(n) / n
refs = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
ref = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
bad = bad1 = 0
t0 = time.time()
for rep in range(reps):
    rs = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    if not all(np.array_equal(p, q) for p, q in zip(refs[:-1], rs[:-1])):
        bad += 1
        dF = np.abs(refs[1] - rs[1]); k = int(np.argmax(dF > 0)) if np.any(dF > 0) else -1
        dG = np.abs(refs[2] - rs[2]); kg = int(np.argmax(dG > 0)) if np.any(dG > 0) else -1
        print("rep %d shards: first F mismatch k=%d (%.3e), first G mismatch k=%d, max|dx|=%.3e" % (rep, k, dF[k] if k >= 0 else 0, kg, np.max(np.abs(refs[0] - rs[0]))), flush=True)
    r = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    if not all(np.array_equal(p, q) for p, q in zip(ref[:-1], r[:-1])):
        bad1 += 1
        dF = np.abs(ref[1] - r[1]); k = int(np.argmax(dF > 0)) if np.any(dF > 0) else -1
        print("rep %d single: first F mismatch k=%d (%.3e), max|dx|=%.3e" % (rep, k, dF[k] if k >= 0 else 0, np.max(np.abs(ref[0] - r[0]))), flush=True)
print("shards: %d of %d runs differ; single: %d of %d   [%.0f s]" % (bad, reps, bad1, reps, time.time() - t0), flush=True)
