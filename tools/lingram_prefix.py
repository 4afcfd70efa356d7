"""The Gram-reuse variant (DOptimalObj.linear_gram, opt-in) against the reference's 300-iteration ABPG_gain trace at
(2048,32768) (tests/golden/large_gain_300.npz): how long does it take the same accept/reject decisions as the 8-thread
reference (its own 4-thread run leaves it at k = 86, profiles/r02_reference_self_spread.json), and how close do the
objective values stay afterwards?  Also the same for the direct evaluation, side by side."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import accbpg_and_fw_amd as acc
    gd = np.load(os.path.join(ROOT, "tests", "golden", "large_gain_300.npz"))
    iters = int(gd["iters"])
    f, h, L, x0 = acc.D_opt_design(2048, 32768, randseed=10)
    out = {}
    for name, lin in (("direct", False), ("linear_gram", True)):
        f.linear_gram(lin)
        x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, verbose=False)
        differs = np.flatnonzero(np.abs(Gain - gd["Gain"]) > 1e-12 * np.abs(gd["Gain"]))
        stable = int(differs[0]) if differs.size else iters
        relF = np.abs(F - gd["F"]) / (1 + np.abs(gd["F"]))
        out[name] = {"decision_stable_prefix": stable, "max_relF_on_prefix": float(relF[:stable].max()),
                     "max_relF": float(relF.max()), "relF_last": float(relF[-1]),
                     "first_relF_above_1e-9": int(np.flatnonzero(relF > 1e-9)[0]) if np.any(relF > 1e-9) else None,
                     "l_inf_x_final": float(np.max(np.abs(x - gd["x"])))}
        print(name, out[name], flush=True)
    f.linear_gram(False)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
