"""Locate the non-reproducible stage of the logical-shards path with per-call checksums (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd.sharded import LogicalShards
from accbpg_and_fw_amd.functions import to_dev, from_dev
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0

def bits(t):   # order-independent exact fingerprint: xor of the 64-bit patterns
    v = t.contiguous().view(torch.int64)
    return int(torch.bitwise_xor(v[: v.numel() // 2 * 2].view(-1, 2)[:, 0], v[: v.numel() // 2 * 2].view(-1, 2)[:, 1]).sum().item())

class Traced(LogicalShards):
    def __init__(self, V, parts):
        super().__init__(V, parts); self.log = []
    def func_grad(self, x, flag=2):
        xd, was_np = to_dev(x)
        rec = {"x": bits(xd)}
        for i, ((lo, hi), obj, gram) in enumerate(zip(self.bounds, self.objs, self.grams)):
            obj.gram_into(xd[lo:hi].contiguous(), gram)
            rec["gram%d" % i] = bits(gram)
        total = self.grams[0].clone()
        for gram in self.grams[1:]: total += gram
        rec["total"] = bits(total)
        fvals = [obj.factor(total) for obj in self.objs]
        rec["f"] = tuple(fvals)
        if flag == 0:
            self.log.append(rec); return fvals[0]
        g = torch.empty(self.n, dtype=torch.float64, device=self.device)
        for i, ((lo, hi), obj) in enumerate(zip(self.bounds, self.objs)):
            gl = torch.empty(hi - lo, dtype=torch.float64, device=self.device)
            obj.grad_from_factor(gl)
            rec["g%d" % i] = bits(gl)
            g[lo:hi] = gl
        rec["g"] = bits(g)
        self.log.append(rec)
        g = from_dev(g, was_np)
        return g if flag == 1 else (fvals[0], g)

m, n, parts = 300, 3000, 3
np.random.seed(11); V = np.random.randn(m, n)
x0 = np.ones(n) / n; h = acc.BurgEntropySimplex()
fs = Traced(V, parts)
if os.environ.get("HUNT_STEP_KERNELS", "1") == "1":
    # the launch-per-block-column Cholesky (the path the events were seen on), not the one-launch kernel
    from accbpg_and_fw_amd import _lib
    for obj in fs.objs:
        _lib.load().accbpg_debug_chol_variant(obj._h, 64)
ref = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False); reflog = fs.log
t0 = time.time(); runs = 0; events = 0; tick = t0
while time.time() - t0 < budget:
    if time.time() - tick > 60:
        tick = time.time(); print("... %d runs, %d events" % (runs, events), flush=True)
    fs.log = []
    r = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False); runs += 1
    if fs.log != reflog:
        events += 1
        for ci, (a, b) in enumerate(zip(reflog, fs.log)):
            diff = [k for k in a if a[k] != b.get(k)]
            if diff:
                print("run %d: first divergence at call %d, fields %s  (f ref %r now %r)" % (runs, ci, diff, a["f"], b["f"]), flush=True)
                break
print("runs %d, events %d" % (runs, events), flush=True)
