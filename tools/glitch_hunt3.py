"""Stage-level reproducibility with several handles alternating (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd.sharded import LogicalShards
m, n, parts = 300, 3000, 3
np.random.seed(11); V = np.random.randn(m, n)
fs = LogicalShards(V, parts)
rng = np.random.RandomState(1)
low = torch.tril(torch.ones(m, m, dtype=torch.bool, device="cuda"))
bad = {"gram": 0, "factor": 0, "grad": 0}
t0 = time.time()
for it in range(4000):
    x = rng.rand(n) + 0.01; x /= x.sum()
    xd = torch.from_numpy(x).cuda()
    grams = []
    for (lo, hi), obj in zip(fs.bounds, fs.objs):
        g1 = torch.zeros(m, m, dtype=torch.float64, device="cuda"); g2 = torch.zeros_like(g1)
        obj.gram_into(xd[lo:hi].contiguous(), g1)
        obj.gram_into(xd[lo:hi].contiguous(), g2)
        if not torch.equal(g1[low], g2[low]):
            bad["gram"] += 1
            d = ((g1 - g2).abs() * low); idx = torch.nonzero(d > 0)
            print("it %d gram shard [%d,%d): %d entries differ, max %.3e, rows %d..%d cols %d..%d" % (it, lo, hi, idx.shape[0], float(d.max()), int(idx[:,0].min()), int(idx[:,0].max()), int(idx[:,1].min()), int(idx[:,1].max())), flush=True)
        grams.append(g1)
    total = grams[0].clone()
    for g in grams[1:]: total += g
    fv = [obj.factor(total) for obj in fs.objs]
    fv2 = [obj.factor(total) for obj in fs.objs]
    if fv != fv2 or len(set(fv)) != 1:
        bad["factor"] += 1
        print("it %d factor: %s vs %s" % (it, fv, fv2), flush=True)
    gl = []
    for (lo, hi), obj in zip(fs.bounds, fs.objs):
        a = torch.empty(hi - lo, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
        obj.grad_from_factor(a); obj.grad_from_factor(b)
        if not torch.equal(a, b):
            bad["grad"] += 1
            print("it %d grad shard [%d,%d): max rel %.3e" % (it, lo, hi, float(((a - b).abs() / a.abs()).max())), flush=True)
    if sum(bad.values()) > 8: break
print("mismatches:", bad, "in", it + 1, "iterations, %.0f s" % (time.time() - t0))
