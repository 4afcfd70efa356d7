"""A whole ABPG_gain run at BASELINE config 2 as a user makes it (1000 iterations from x0, package defaults), once more
with the opt-in memoisation of the repeated value evaluation: rates, monotonicity, feasibility, and that the two runs
are the same to the bit."""
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
    f, h, L, x0 = acc.D_opt_design(2048, 32768, randseed=10)
    xd = torch.from_numpy(x0).cuda()
    out = {}
    runs = {}
    for name, memo in (("default", False), ("memoize_values", True)):
        f.memoize_values(memo)
        c0 = dict(f.calls)
        torch.cuda.synchronize()
        t = time.time()
        res = acc.ABPG_gain(f, h, L, xd, gamma=2, maxitrs=1000, verbose=False)
        torch.cuda.synchronize()
        dt = time.time() - t
        x, F = res[0], res[1]
        xs = x.cpu().numpy()
        runs[name] = res
        out[name] = {"seconds": dt, "iterations": len(F), "it_per_s": len(F) / dt, "F0": float(F[0]), "F_last": float(F[-1]),
                     "monotone": bool(np.all(np.diff(F) <= 1e-9 * np.abs(F[:-1]))), "x_min": float(xs.min()),
                     "sum_minus_1": float(xs.sum() - 1), "value_calls": f.calls["value"] - c0["value"],
                     "grad_calls": f.calls["grad"] - c0["grad"], "T_last": float(res[-1][-1])}
        print(name, out[name], flush=True)
    f.memoize_values(False)
    a, b = runs["default"], runs["memoize_values"]
    out["bit_identical"] = bool(torch.equal(a[0], b[0]) and all(np.array_equal(p, q) for p, q in zip(a[1:-1], b[1:-1])))
    print("bit-identical:", out["bit_identical"])
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
