"""Reproducibility of short solver runs under a changing allocator layout (development aid)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd.sharded import LogicalShards
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
only_shards = len(sys.argv) > 2 and sys.argv[2] == "shards"
m, n, parts = 300, 3000, 3
np.random.seed(11); V = np.random.randn(m, n)
x0 = np.ones(n) / n
h = acc.BurgEntropySimplex()
f0 = acc.DOptimalObj(V)
ref = acc.ABPG(f0, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
refg = acc.ABPG_gain(f0, h, 1.0, x0, gamma=2, maxitrs=30, verbose=False)
refs = None
random.seed(3)
junk = []
stats = {"single": [0, 0], "gain": [0, 0], "shards": [0, 0], "overlap": [0, 0]}
t0 = time.time(); rnd = 0
while time.time() - t0 < budget:
    rnd += 1
    # perturb the allocator: random small tensors come and go
    for _ in range(random.randint(0, 6)):
        junk.append(torch.empty(random.choice([3000, 1000, 90000, 257, 8192, 24000]), dtype=torch.float64, device="cuda").normal_())
    while len(junk) > 12: junk.pop(random.randrange(len(junk)))
    f = acc.DOptimalObj(V) if rnd % 3 == 0 else f0
    fs = LogicalShards(V, parts)
    if refs is None: refs = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    for rep in range(6):
        for name, fn, rf in ([("shards", lambda: acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False), refs)] * 4 if only_shards else [("single", lambda: acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False), ref),
                             ("shards", lambda: acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False), refs),
                             ("gain", lambda: acc.ABPG_gain(f, h, 1.0, x0, gamma=2, maxitrs=30, verbose=False), refg)]):
            r = fn(); stats[name][1] += 1
            if not all(np.array_equal(p, q) for p, q in zip(rf[:-1], r[:-1])):
                stats[name][0] += 1
                dF = np.abs(rf[1] - r[1]) if len(rf[1]) == len(r[1]) else np.array([1.0]); k = int(np.argmax(dF > 0))
                print("round %d rep %d %s: first F mismatch k=%d (%.3e) max|dx| %.3e" % (rnd, rep, name, k, dF[k], np.max(np.abs(rf[0] - r[0]))), flush=True)
        if only_shards: continue
        f.overlap_values(True)
        r = acc.ABPG_gain(f, h, 1.0, x0, gamma=2, maxitrs=30, verbose=False); stats["overlap"][1] += 1
        f.overlap_values(False)
        if not all(np.array_equal(p, q) for p, q in zip(refg[:-1], r[:-1])):
            stats["overlap"][0] += 1
            dF = np.abs(refg[1] - r[1]); k = int(np.argmax(dF > 0)) if np.any(dF > 0) else -1
            print("round %d rep %d overlap: first F mismatch k=%d max|dx| %.3e" % (rnd, rep, k, np.max(np.abs(refg[0] - r[0]))), flush=True)
    del fs
print("mismatching/total:", stats, "rounds", rnd, flush=True)
