"""Bit-stability soak of the two MFMA-bound kernels as the solvers drive them: the same (f, grad) evaluation repeated
many times at several shapes (one of them with value evaluations running beside it on the second stream), every result
compared bit for bit with the first.  A staging race (a load counted wrongly, a stage read before it landed) would show
as a differing run; so would a hand-off race of the one-launch Cholesky that sits between the two kernels."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import accbpg_and_fw_amd as acc
    out = {}
    for (m, n, reps, beside) in ((2048, 32768, 1500, False), (2048, 32768, 800, True), (512, 8192, 3000, False),
                                 (1024, 4096, 3000, False), (4096, 32768, 200, False)):
        gen = torch.Generator(device="cuda").manual_seed(11)
        V = torch.randn(m, n, dtype=torch.float64, device="cuda", generator=gen)
        x = torch.rand(n, dtype=torch.float64, device="cuda", generator=gen) + 0.05
        x /= x.sum()
        y = torch.flip(x, [0]).contiguous()
        f = acc.DOptimalObj(V)
        f.overlap_values(beside)
        f0, g0 = f.func_grad(x, 2)
        g0 = g0.clone()
        fy0 = f.func_grad(y, 0)
        differ = 0
        t = time.time()
        for _ in range(reps):
            if beside:
                ticket = f.value_async(y)
                fv, g = f.func_grad(x, 2)
                fy = f.value_wait(ticket)
                differ += int(fy != fy0)
            else:
                fv, g = f.func_grad(x, 2)
            differ += int(fv != f0 or not torch.equal(g, g0))
        torch.cuda.synchronize()
        key = "%dx%d%s" % (m, n, " +value beside" if beside else "")
        out[key] = {"runs": reps, "differ": differ, "seconds": round(time.time() - t, 2)}
        print(key, out[key], flush=True)
        del f, V
        torch.cuda.empty_cache()
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)
    sys.exit(1 if any(v["differ"] for v in out.values()) else 0)


if __name__ == "__main__":
    main()
