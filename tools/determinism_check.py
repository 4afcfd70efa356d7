"""Run-to-run bitwise reproducibility of a solver run, sequential and with the two-stream overlap."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import accbpg_and_fw_amd as acc
for (m, n) in [(300, 3000), (1024, 4096)]:
    f, h, L, x0 = acc.D_opt_design(m, n, randseed=21)
    ref = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    for mode in ("sequential", "overlap"):
        f.overlap_values(mode == "overlap")
        bad = 0
        for rep in range(12):
            r = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
            same = all(np.array_equal(p, q) for p, q in zip(ref[:-1], r[:-1]))
            if not same:
                bad += 1
                k = int(np.argmax(np.abs(r[1] - ref[1]) > 0)) if len(r[1]) == len(ref[1]) else -1
                print("   (%d,%d) %s rep %d differs: first F mismatch at k=%d, max|dx|=%.3e" % (m, n, mode, rep, k, np.max(np.abs(r[0] - ref[0]))), flush=True)
        print("(%d,%d) %-10s: %d of 12 runs differ from the first sequential run" % (m, n, mode, bad), flush=True)
    f.overlap_values(False)
