import sys; sys.path.insert(0, ".")
import numpy as np, torch
import accbpg_and_fw_amd as acc
for (m, n) in [(2048, 32768), (512, 8192)]:
    f, h, L, x0 = acc.D_opt_design(m, n, randseed=1)
    f.speculate(False)
    x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, torch.from_numpy(x0).cuda(), gamma=2, maxitrs=260, verbose=False)
    r = np.round(np.log(Gain[1:] / (Gain[:-1] / 1.2)) / np.log(1.2)).astype(int)
    print(m, n, "retries per iteration k=1..:", "".join(str(min(v, 9)) for v in r))
    print("gain 200..230", np.round(Gain[200:230], 5))
