"""Repeat the sequential-vs-overlapped comparison of tests/test_gpu_parity.py many times (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
nbad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    if rep % 50 == 0: print("... rep", rep, flush=True)
    f, h, L, x0 = acc.D_opt_design(300, 3000, randseed=21)
    a = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    b = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=60, theta_eq=True, restart=True, verbose=False)
    f.overlap_values(True)
    a2 = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    b2 = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=60, theta_eq=True, restart=True, verbose=False)
    f.overlap_values(False)
    a3 = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    for name, u, v in [("gain seq/ovl", a, a2), ("abpg seq/ovl", b, b2), ("gain seq/seq", a, a3)]:
        for idx, (p, q) in enumerate(zip(u[:-1], v[:-1])):
            if not np.array_equal(p, q):
                nbad += 1
                d = np.abs(np.asarray(p) - np.asarray(q))
                first = int(np.argmax(d > 0))
                print("rep %d %s: output %d differs, max %.3e, first index %d" % (rep, name, idx, d.max(), first), flush=True)
                break
print("mismatching comparisons:", nbad)
