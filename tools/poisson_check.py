"""Exploratory: deviations of the HIP Poisson path from the reference goldens (prints, no asserts)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accbpg_and_fw_amd as acc

gd = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "poisson.npz"))
CFG = {"l1": (acc.Poisson_regrL1, 200, 100, 0.0001, 0), "l2": (acc.Poisson_regrL2, 100, 1000, 0.001, 0.001),
       "l1r": (acc.Poisson_regrL1, 300, 2000, 0.001, 0.01)}


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - b) / (np.abs(b) + 1e-300)))


def pre(a, b, tol=1e-12):
    n = min(len(a), len(b))
    bad = np.nonzero(np.abs(a[:n] - b[:n]) > tol * (1 + np.abs(b[:n])))[0]
    return n if bad.size == 0 else int(bad[0])


for tag, (fac, m, n, noise, lam) in CFG.items():
    f, h, L, x0 = fac(m, n, noise=noise, lamda=lam, randseed=1)
    x, y = gd[tag + "_x"], gd[tag + "_y"]
    fx, g = f.func_grad(x, 2)
    print(tag, "f rel", abs(fx - gd[tag + "_f"]) / abs(gd[tag + "_f"]), "g rel", rel(g, gd[tag + "_g"]),
          "g abs", np.max(np.abs(g - gd[tag + "_g"])), "f0", abs(f(x0) - gd[tag + "_f0"]),
          "psi", h.extra_Psi(x) - gd[tag + "_psi"])
    for idx in range(3):
        z = h.div_prox_map(y, gd[tag + "_g"], float(gd["%s_prox_L%d" % (tag, idx)]))
        print("   prox", idx, "bitwise", np.array_equal(z, gd["%s_prox_x%d" % (tag, idx)]), rel(z, gd["%s_prox_x%d" % (tag, idx)]))
    z = h.prox_map(np.abs(gd[tag + "_g"]) + 0.5, 2.0)
    print("   prox raw bitwise", np.array_equal(z, gd[tag + "_prox_raw"]))

N = 2000
f, h, L, x0 = CFG["l1"][0](200, 100, noise=0.0001, lamda=0, randseed=1)
t = time.time()
x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=N, linesearch=False, verbose=False)
print("l1 bpg", time.time() - t, "x", np.max(np.abs(x - gd["l1_bpg_x"])), "F", np.max(np.abs(F - gd["l1_bpg_F"])))
x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=N, linesearch=True, verbose=False)
p = pre(Ls, gd["l1_bpgls_Ls"])
print("l1 bpgls prefix", p, "F", np.max(np.abs(F[:p] - gd["l1_bpgls_F"][:p])), "end", F[-1] - gd["l1_bpgls_F"][-1])
for gam, key in [(1.0, "g10"), (1.5, "g15"), (2.0, "g20")]:
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=gam, maxitrs=N, theta_eq=True, verbose=False)
    print("l1 abpg", key, "x", np.max(np.abs(x - gd["l1_abpg_%s_x" % key])), "F", np.max(np.abs(F - gd["l1_abpg_%s_F" % key])),
          "G", rel(G[:500], gd["l1_abpg_%s_G" % key][:500]))
x, F, G, T = acc.ABDA(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=True, verbose=False)
print("l1 abda x", np.max(np.abs(x - gd["l1_abda_x"])), "F", np.max(np.abs(F - gd["l1_abda_F"])))
x, F, Gamma, G, T = acc.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=3, verbose=False)
p = pre(Gamma, gd["l1_expo_Gamma"])
print("l1 expo prefix", p, "F", np.max(np.abs(F[:p] - gd["l1_expo_F"][:p])), "end", F[-1] - gd["l1_expo_F"][-1])
x, F, G, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, theta_eq=False, verbose=False)
p = pre(G, gd["l1_gain_G"])
print("l1 gain prefix", p, "F", np.max(np.abs(F[:p] - gd["l1_gain_F"][:p])), "end", F[-1] - gd["l1_gain_F"][-1])

f, h, L, x0 = CFG["l2"][0](100, 1000, noise=0.001, lamda=0.001, randseed=1)
x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=N, linesearch=False, verbose=False)
print("l2 bpg x", np.max(np.abs(x - gd["l2_bpg_x"])), "F", np.max(np.abs(F - gd["l2_bpg_F"])))
x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=N, linesearch=True, ls_ratio=1.5, verbose=False)
p = pre(Ls, gd["l2_bpgls_Ls"])
print("l2 bpgls prefix", p, "F", np.max(np.abs(F[:p] - gd["l2_bpgls_F"][:p])), "end", F[-1] - gd["l2_bpgls_F"][-1])
x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=False, verbose=False)
print("l2 abpg x", np.max(np.abs(x - gd["l2_abpg_x"])), "F", np.max(np.abs(F - gd["l2_abpg_F"])))
x, F, Gamma, G, T = acc.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=1, verbose=False)
p = pre(Gamma, gd["l2_expo_Gamma"])
print("l2 expo prefix", p, "F", np.max(np.abs(F[:p] - gd["l2_expo_F"][:p])), "end", F[-1] - gd["l2_expo_F"][-1])
t = time.time()
x, F, G, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, ls_inc=1.5, ls_dec=1.5, theta_eq=True,
                                       verbose=False)
p = pre(G, gd["l2_gain_G"])
print("l2 gain", time.time() - t, "prefix", p, "F", np.max(np.abs(F[:p] - gd["l2_gain_F"][:p])), "end", F[-1] - gd["l2_gain_F"][-1])

# bandwidth at a larger size
for (m, n) in [(8192, 65536), (2048, 262144), (65536, 4096)]:
    A = torch.rand(m, n, dtype=torch.float64, device="cuda")
    b = torch.rand(m, dtype=torch.float64, device="cuda") + 0.5
    f = acc.PoissonRegression(A, b)
    x = torch.rand(n, dtype=torch.float64, device="cuda") / n
    for flag in (0, 2):
        f.func_grad(x, flag)
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            f.func_grad(x, flag)
        torch.cuda.synchronize(); dt = (time.time() - t) / 5
        byt = 8.0 * m * n * (1 if flag == 0 else 2)
        print("poisson", (m, n), "flag", flag, "ms %.3f" % (dt * 1e3), "GB/s %.0f" % (byt / dt / 1e9))
    Ah = A.cpu().numpy(); xh = x.cpu().numpy(); bh = b.cpu().numpy()
    Ax = Ah @ xh
    gref = Ah.T @ (1 - bh / Ax)
    fref = float(np.sum(bh * np.log(bh / Ax) + Ax - bh))
    fx, g = f.func_grad(x, 2)
    print("   f rel", abs(fx - fref) / abs(fref), "g rel", float(np.max(np.abs(g.cpu().numpy() - gref) / np.abs(gref))))
    del A, f
