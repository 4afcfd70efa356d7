"""Where does the 300-iteration ABPG_gain run at (2048, 32768) leave the reference's trace?  Prints, per iteration,
the relative gap of F, of the gain, and the call counts, around the first departures."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import accbpg_and_fw_amd as acc
from test_gpu_parity import _logged_gain_run, golden

gd = golden("large_gain_300")
f, h, L, x0 = acc.D_opt_design(2048, 32768, randseed=10)
iters = int(gd["iters"])
pos = gd["iter_call_pos"]
(x, F, Gain, Gdiv, Gavg, T), kinds, values, taps = _logged_gain_run(f, h, L, x0, iters, [pos[k] for k in gd["keep"]])
relF = np.abs(F - gd["F"]) / (1 + np.abs(gd["F"]))
relG = np.abs(Gain - gd["Gain"]) / np.abs(gd["Gain"])
print("first F gap > 1e-9 at", np.flatnonzero(relF > 1e-9)[:5], "first gain gap > 1e-12 at", np.flatnonzero(relG > 1e-12)[:5])
n = min(len(kinds), len(gd["call_kinds"]))
dk = np.flatnonzero(kinds[:n] != gd["call_kinds"][:n])
print("first call-kind difference at call", dk[:3], "of", len(kinds), len(gd["call_kinds"]))
dv = np.abs(values[:n] - gd["call_values"][:n]) / (1 + np.abs(gd["call_values"][:n]))
dv = np.where(np.isnan(dv), 0, dv)
big = np.flatnonzero(dv > 1e-9)
print("calls with value gap > 1e-9:", big[:20])
for c in big[:12]:
    k = int(np.searchsorted(pos, c, side="right") - 1)
    print("  call %d (iteration %d, kind %d): %.3e   ours %.12f ref %.12f" % (c, k, kinds[c], dv[c], values[c], gd["call_values"][c]))
for k in range(80, min(130, iters)):
    print("k=%3d  relF %.2e  relGain %.2e  gain %.6e ref %.6e  F %.10f" % (k, relF[k], relG[k], Gain[k], gd["Gain"][k], F[k]))
for k in gd["keep"]:
    k = int(k)
    if int(pos[k]) in taps:
        print("x_%d l_inf" % k, np.max(np.abs(taps[int(pos[k])].cpu().numpy() - gd["x_%d" % k])))
print("final l_inf", np.max(np.abs(x.cpu().numpy() - gd["x"])), "final relF", relF[-1], "max relF", relF.max())
