"""accbpg_and_fw_amd -- MI355X-native implementation of the D-optimal-design hot path of
accbpg (DredderGun/accbpg_and_fw): BPG / ABPG / ABPG_gain with the Burg-entropy simplex
kernel, and D_opt_FW / D_opt_FW_away, behind the reference's own names and signatures.

    import accbpg_and_fw_amd as accbpg
    f, h, L, x0 = accbpg.D_opt_design(80, 200)
    x, F, G, T = accbpg.BPG(f, h, L, x0, maxitrs=1000)

All arithmetic runs in hand-written gfx950 HIP kernels (accbpg_and_fw_amd/csrc) through a
ctypes C-ABI (include/accbpg_hip.h); there is no CPU fallback.
"""
from .functions import (RSmoothFunction, DOptimalObj, PoissonRegression, LegendreFunction, BurgEntropy,
                        BurgEntropyL1, BurgEntropyL2, BurgEntropySimplex)
from .algorithms import BPG, ABPG, ABPG_gain, ABPG_expo, ABDA, solve_theta
from .algorithms_fw import FW_alg_div_step
from .functions_lmo import lmo_simplex
from .D_opt_alg import D_opt_FW, D_opt_FW_away
from .applications import D_opt_design, D_opt_libsvm, D_opt_KYinit, Poisson_regrL1, Poisson_regrL2
from .utils import load_libsvm_file
from .batched import DOptimalBatch, BPG_batch, ABPG_batch, ABPG_gain_batch, solve_batch, solve_instances

__all__ = ["RSmoothFunction", "DOptimalObj", "PoissonRegression", "LegendreFunction", "BurgEntropy",
           "BurgEntropyL1", "BurgEntropyL2", "BurgEntropySimplex", "Poisson_regrL1", "Poisson_regrL2",
           "BPG", "ABPG", "ABPG_gain", "ABPG_expo", "ABDA", "solve_theta", "FW_alg_div_step", "lmo_simplex",
           "D_opt_FW", "D_opt_FW_away", "D_opt_design", "D_opt_libsvm", "D_opt_KYinit", "load_libsvm_file"]
__version__ = "0.1.0"
