"""How far does the REFERENCE move when only the BLAS thread count changes?  Compares two runs of the real reference's
ABPG_gain(gamma=2) at D_opt_design(2048, 32768, randseed=10) made by oracle/gen_golden.py on the same machine:
tests/golden/large_gain_300.npz (8 OpenBLAS threads, 300 iterations) and a 100-iteration run with
OPENBLAS_NUM_THREADS=4 (--name large_gain_100_t4, not committed).  Writes profiles/r02_reference_self_spread.json:
the yardstick for what "parity" of the GPU path with the reference can mean on this solver at this size."""
import json
import os
import sys

import numpy as np

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
a = np.load(os.path.join(root, "tests", "golden", "large_gain_300.npz"))
b = np.load(sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "tests", "golden", "large_gain_100_t4.npz"))
n = len(b["F"])
relF = np.abs(a["F"][:n] - b["F"]) / (1 + np.abs(b["F"]))
relG = np.abs(a["Gain"][:n] - b["Gain"]) / np.abs(b["Gain"])
dg = np.flatnonzero(relG > 1e-12)
nc = min(len(a["call_kinds"]), len(b["call_kinds"]))
dk = np.flatnonzero(a["call_kinds"][:nc] != b["call_kinds"][:nc])
out = {
    "what": "real reference, ABPG_gain(gamma=2) at D_opt_design(2048,32768,randseed=10): 8 vs 4 OpenBLAS threads",
    "iterations_compared": int(n),
    "first_iteration_with_a_different_gain": int(dg[0]) if dg.size else None,
    "first_oracle_call_of_a_different_kind": int(dk[0]) if dk.size else None,
    "first_iteration_with_F_gap_above_1e-10": int(np.flatnonzero(relF > 1e-10)[0]) if np.any(relF > 1e-10) else None,
    "first_iteration_with_F_gap_above_1e-9": int(np.flatnonzero(relF > 1e-9)[0]) if np.any(relF > 1e-9) else None,
    "max_rel_gap_of_F": float(relF.max()), "at_iteration": int(relF.argmax()),
    "rel_gap_of_F_by_iteration": {str(k): float(relF[k]) for k in range(0, n, 10)},
    "l_inf_of_iterates": {},
}
for k in b["keep"]:
    key = "x_%d" % int(k)
    if key in a.files:
        out["l_inf_of_iterates"][str(int(k))] = float(np.max(np.abs(a[key] - b[key])))
if "x_%d" % n in a.files:                                     # the final iterate of the short run is x_n of the long one
    out["l_inf_of_iterates"][str(n)] = float(np.max(np.abs(a["x_%d" % n] - b["x"])))
print(json.dumps(out, indent=1))
with open(os.path.join(root, "profiles", "r02_reference_self_spread.json"), "w") as fh:
    json.dump(out, fh, indent=1)
