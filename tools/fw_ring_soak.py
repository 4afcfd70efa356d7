"""Soak of the side factorisations of D_opt_FW_away (a ring of auxiliary handles, snapshot copies in stream order, results
collected `depth` calls later): many repetitions of the same run must give the same bits every time -- F (which comes
through the ring) and the iterates.  Shapes on either side of the size switch of the slots' Cholesky (one launch /
launch per block column), exact and anchored forms."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import accbpg_and_fw_amd as acc
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    out = []
    for (m, n, its) in [(512, 2048, 200), (1536, 4096, 200), (2048, 32768, 120)]:
        gen = torch.Generator(device="cuda").manual_seed(m)
        V = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=gen)
        x0 = torch.full((n,), 1.0 / n, dtype=torch.float64, device="cuda")
        f = acc.DOptimalObj(V)
        for kw in (dict(logdet_refresh=1, logdet_ring=3), dict(logdet_refresh=1, logdet_ring=2), dict(logdet_refresh=4), dict()):
            ref = acc.D_opt_FW_away(f, x0, -1.0, its, verbose=False, **kw)
            bad = 0
            t0 = time.time()
            n_runs = reps if m < 2048 else max(20, reps // 6)
            for r in range(n_runs):
                if r % 7 == 3:
                    f.func_grad(x0, 2)                          # an evaluation on the main handle in between
                res = acc.D_opt_FW_away(f, x0, -1.0, its, verbose=False, **kw)
                same = torch.equal(res[0], ref[0]) and np.array_equal(res[1], ref[1]) and np.array_equal(res[2], ref[2])
                bad += 0 if same else 1
            rec = {"shape": [m, n], "iterations": its, "form": kw or "default", "runs": n_runs, "runs_that_differ": bad,
                   "seconds": time.time() - t0}
            print(rec, flush=True)
            out.append(rec)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
