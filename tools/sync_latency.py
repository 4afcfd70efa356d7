"""Round-trip latency of the small synchronous entry points (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accbpg_and_fw_amd as acc
from accbpg_and_fw_amd import functions as F
x = torch.rand(64, dtype=torch.float64, device="cuda") + 0.1
y = torch.rand(64, dtype=torch.float64, device="cuda") + 0.1
h = acc.BurgEntropySimplex()
for name, fn in [("vec_min_sum", lambda: F.vec_min_sum(x)), ("ls_terms", lambda: F.ls_terms(x, x, y, x, y)),
                 ("prox n=64", lambda: h.div_prox_map(y, x, 1.0)), ("axpby (async)", lambda: F.vec_axpby(0.5, x, 0.5, y)),
                 ("torch sync only", lambda: torch.cuda.synchronize())]:
    for _ in range(50): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(2000): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 2000
    print("%-16s %.1f us per call" % (name, dt * 1e6), flush=True)
