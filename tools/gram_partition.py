"""A/B of two plans of the Gram kernel in one process (handles created under either setting of
accbpg_debug_plan_flags, interleaved rounds; value and gradient must agree to rounding).  --flags 1,0: plain contiguous
stream-K ranges against whole tiles per workgroup with XCD-compact phases where the tile list is longer than the grid
(m >= 4096).  --flags 4,0: one launch over the whole row length against one launch per column block of 32768 (rows of
65536 columns or more)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x16384,8192x32768,8192x262144")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--flags", default="1,0", help="plan flags of the baseline and of the candidate")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd import _lib
    lib = _lib.load()
    res = []
    for shp in args.shapes.split(","):
        m, n = (int(v) for v in shp.split("x"))
        gen = torch.Generator(device="cuda").manual_seed(7)
        V = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=gen)
        x = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) + 0.05
        x /= x.sum()
        objs = {}
        base, cand = (int(v) for v in args.flags.split(","))
        for flag in (base, cand):
            lib.accbpg_debug_plan_flags(flag)
            objs[flag] = acc.DOptimalObj(V)
            objs[flag].overlap_values(False)
        lib.accbpg_debug_plan_flags(0)
        vals = {flag: o.func_grad(x, 2) for flag, o in objs.items()}
        relf = abs(vals[cand][0] - vals[base][0]) / abs(vals[base][0])
        relg = float(((vals[cand][1] - vals[base][1]).abs() / vals[base][1].abs()).max())
        rec = {"shape": [m, n], "flags": [base, cand], "rel_gap_f": relf, "rel_gap_g": relg, "gram_ms": {base: [], cand: []}}
        iters = 10 if m * n <= 8192 * 32768 else 3
        for rnd in range(args.rounds):
            for flag, o in objs.items():
                o.profile(True)
                for _ in range(iters):
                    o.func_grad(x, 0)
                tot, cnt = o.profile_read()["gram"]
                o.profile(False)
                rec["gram_ms"][flag].append(tot / iters)           # (a chunked evaluation is several launches)
        flops = float(m) * m * n
        rec["baseline_ms"] = float(np.median(rec["gram_ms"][base]))
        rec["candidate_ms"] = float(np.median(rec["gram_ms"][cand]))
        rec["baseline_tflops"] = flops / rec["baseline_ms"] * 1e-9
        rec["candidate_tflops"] = flops / rec["candidate_ms"] * 1e-9
        print(rec, flush=True)
        res.append(rec)
        del objs, V
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
