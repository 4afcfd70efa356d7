"""A/B of the schedules of the direct-to-LDS Gram and gradient kernels in ONE process, interleaved rounds
(cdna_hip_programming.md 5.4 rule 24): variant 0 = reads and multiplies issued as a block at every fragment-group
boundary, 1 = dealt out between MFMA pairs (reads behind odd pairs, loads behind even pairs, barrier in the middle of the
last group; timing code 18, no longer in the table), 2 = dealt out with the reads early in the group and the barrier behind
the first fragment row, one load per M0 write through the builtin; 3 / 4 = variant 2 with the loads of the k-loop two
(production) / four to an M0 write.  The handle's switch, used for the bit-identity check and the gradient kernel's
timings: Gram 0 = production, 1 = builtin loads, 2 = four to an M0 write, 3 = block schedule; gradient kernel 0 =
production (block schedule, four loads to an M0 write), 1 = builtin loads, 2 = dealt out.
Checks first that value and gradient are bit-identical under every variant."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import accbpg_and_fw_amd as acc
    from accbpg_and_fw_amd import _lib
    from accbpg_and_fw_amd.functions import _ptr
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(3)
    V = torch.randn(args.m, args.n, dtype=torch.float64, device="cuda", generator=gen)
    x = torch.rand(args.n, dtype=torch.float64, device="cuda", generator=gen) + 0.05
    x /= x.sum()
    f = acc.DOptimalObj(V)
    f.overlap_values(False)
    base = f.func_grad(x, 2)
    same = {}
    for v in (1, 2, 0):
        lib.accbpg_debug_chol_variant(f._h, v << 30)
        r = f.func_grad(x, 2)
        same[v] = bool(r[0] == base[0] and torch.equal(r[1], base[1]))
    print("bit-identical:", same, flush=True)
    out = {"shape": [args.m, args.n], "bit_identical": same, "gram_ms": {}, "grad_ms": {}}
    ms = C.c_double(0.0)
    codes = {0: 10, 2: 19, 3: 28, 4: 29}
    for rnd in range(args.rounds):
        for v, code in codes.items():
            _lib.check(lib.accbpg_debug_gram_variant(f._h, _ptr(x), code, args.iters, C.byref(ms)), "gram variant")
            out["gram_ms"].setdefault(v, []).append(ms.value)
        for v in (0, 1, 2):
            lib.accbpg_debug_chol_variant(f._h, v << 30)             # gradient kernel: 0 production, 1 builtin loads, 2 dealt out
            f.profile(True)
            for _ in range(args.iters):
                f.func_grad(x, 1)
            tot, cnt = f.profile_read()["grad"]
            f.profile(False)
            out["grad_ms"].setdefault(v, []).append(tot / cnt)
        print(rnd, {v: round(out["gram_ms"][v][-1], 4) for v in codes}, {v: round(out["grad_ms"][v][-1], 4) for v in (0, 1, 2)},
              flush=True)
    lib.accbpg_debug_chol_variant(f._h, 0)
    for key in ("gram_ms", "grad_ms"):
        out[key + "_median"] = {v: float(np.median(t)) for v, t in out[key].items()}
        out[key + "_min"] = {v: float(np.min(t)) for v, t in out[key].items()}
    txt = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, "w") as fh:
            fh.write(txt)
    print(txt)


if __name__ == "__main__":
    main()
