"""Deviation of the HIP path from the reference's long full-size traces (prints; the test asserts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import accbpg_and_fw_amd as acc
gd = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "large_long.npz"))
f, h, L, x0 = acc.D_opt_design(int(gd["m"]), int(gd["n"]), randseed=int(gd["seed"]))
x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=int(gd["iters"]), theta_eq=True, verbose=False)
print("ABPG %d its: l_inf x %.2e  max|F-Fref| %.2e  (F[-1] = %.10f)" % (len(F), np.max(np.abs(x - gd["abpg_x"])), np.max(np.abs(F - gd["abpg_F"])), F[-1]))
xb, Fb, Lb, Tb = acc.BPG(f, h, L, x0, maxitrs=int(gd["half"]), linesearch=True, verbose=False)
print("BPG-LS %d its: l_inf x %.2e  max|F-Fref| %.2e  max|L-Lref| %.2e" % (len(Fb), np.max(np.abs(xb - gd["bpgls_x"])), np.max(np.abs(Fb - gd["bpgls_F"])), np.max(np.abs(Lb - gd["bpgls_Ls"]))))
