"""Generate tests/golden/*.npz by running the REAL reference in the build container.

TEST INFRASTRUCTURE ONLY.  Runs only where /root/reference exists (never on
the GPU box).  Nothing from the reference is copied: the reference package is
imported read-only (bytecode writing disabled), called on seeded inputs, and
only inputs/outputs (numbers) are written out.  ``V`` for the Gaussian
instances is never stored; tests regenerate it with the same legacy NumPy RNG
call the factory uses (accbpg/applications.py:47-49).

Usage:
    python oracle/gen_golden.py            # small fixtures (about a minute)
    python oracle/gen_golden.py --medium   # + (256,4096) 1000-iteration traces
    python oracle/gen_golden.py --large    # + (2048,32768) per-call + short trajectory (tens of minutes)

cvxpy and jax are not installed here and are imported at module level by the
reference (functions.py:4-6) although nothing on this path uses them, so empty
stand-in modules are registered for the import to succeed.
"""
import argparse
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def load_reference():
    for name in ["cvxpy", "jax", "jax.numpy", "jax.scipy", "jax.scipy.linalg"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["jax"].numpy = sys.modules["jax.numpy"]
    sys.modules["jax"].scipy = sys.modules["jax.scipy"]
    sys.modules["jax"].jit = lambda f=None, **k: f
    sys.modules["jax.scipy"].linalg = sys.modules["jax.scipy.linalg"]
    sys.modules["jax.scipy.linalg"].cholesky = None
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import accbpg
    return accbpg


def save(name, **arrays):
    import numpy as np
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def percall(accbpg, tag, m, n, seed):
    """func_grad / prox / divergence input-output pairs at a non-trivial point."""
    import numpy as np
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    rng = np.random.RandomState(1000 + seed)
    x = rng.rand(n) + 0.05
    x /= x.sum()
    fx, g = f.func_grad(x, 2)
    f0, g0 = f.func_grad(x0, 2)
    y = rng.rand(n) + 0.05
    y /= y.sum()
    out = {"m": m, "n": n, "seed": seed, "x": x, "f": fx, "g": g, "f0": f0, "g0": g0, "y": y}
    for idx, Lc in enumerate([1.0, 0.37, 5.0]):
        out["prox_L%d" % idx] = Lc
        out["prox_x%d" % idx] = h.div_prox_map(y, g, Lc)
    out["prox_raw"] = h.prox_map(g - g.min() + 0.5, 2.0)
    out["div_xy"] = h.divergence(x, y)
    out["div_yx"] = h.divergence(y, x)
    save("percall_" + tag, **out)


def solver_traces(accbpg, tag, m, n, seed, iters):
    """Whole-run traces.  The first five calls are the ones the reference's notebook makes on
    this instance (ipynb/ex_Dopt_random.ipynb cells 1, 3, 5); the rest widen option coverage."""
    import numpy as np
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    out = {"m": m, "n": n, "seed": seed, "iters": iters}
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=iters, linesearch=False, verbose=False)
    out.update(bpg_x=x, bpg_F=F, bpg_Ls=Ls)
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=iters, linesearch=True, verbose=False)
    out.update(bpgls_x=x, bpgls_F=F, bpgls_Ls=Ls)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True, verbose=False)
    out.update(abpg_x=x, abpg_F=F, abpg_G=G)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True, restart=True, verbose=False)
    out.update(abpgrs_x=x, abpgrs_F=F, abpgrs_G=G)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, G0=0.1, theta_eq=True,
                                                  verbose=False)
    out.update(gain_x=x, gain_F=F, gain_Gain=Gain, gain_Gdiv=Gdiv, gain_Gavg=Gavg)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, G0=0.1, theta_eq=True,
                                                  restart=True, verbose=False)
    out.update(gainrs_x=x, gainrs_F=F, gainrs_Gain=Gain, gainrs_Gdiv=Gdiv, gainrs_Gavg=Gavg)
    # wider option coverage
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=1.5, maxitrs=iters, theta_eq=False, verbose=False)
    out.update(abpgk_x=x, abpgk_F=F, abpgk_G=G)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, verbose=False)
    out.update(gaindef_x=x, gaindef_F=F, gaindef_Gain=Gain, gaindef_Gdiv=Gdiv, gaindef_Gavg=Gavg)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, G0=0.1, ls_inc=1.5,
                                                  ls_dec=1.1, theta_eq=False, checkdiv=True, restart=True,
                                                  restart_rule='f', verbose=False)
    out.update(gainopt_x=x, gainopt_F=F, gainopt_Gain=Gain, gainopt_Gdiv=Gdiv, gainopt_Gavg=Gavg)
    save("traces_" + tag, **out)


def fw_traces(accbpg, tag, m, n, seed, iters, eps=1e-8):
    import numpy as np
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    V = f.H
    out = {"m": m, "n": n, "seed": seed, "iters": iters, "eps": eps}
    x, F, SP, SN, T = accbpg.D_opt_FW(V, x0, eps, iters, verbose=False)
    out.update(fw_x=x, fw_F=F, fw_SP=SP, fw_SN=SN)
    x, F, SP, SN, T = accbpg.D_opt_FW_away(V, x0, eps, iters, verbose=False)
    out.update(away_x=x, away_F=F, away_SP=SP, away_SN=SN)
    save("fw_" + tag, **out)


def housing(accbpg):
    """RNG-free instance: the LIBSVM housing file (13 x 506 after transpose),
    ipynb/ex_Dopt_LIBSVM.ipynb.  The parsed matrix is stored as fixture data."""
    import numpy as np
    path = os.path.join(REF, "parameters_free_fw", "data", "housing.txt")
    f, h, L, x0 = accbpg.D_opt_libsvm(path)
    V = np.ascontiguousarray(f.H)
    out = {"V": V}
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=1001, linesearch=False, verbose=False)
    out.update(bpg_x=x, bpg_F=F)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2, maxitrs=1001, theta_eq=False, verbose=False)
    out.update(abpg_x=x, abpg_F=F, abpg_G=G)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=1001, G0=0.1,
                                                  ls_inc=1.5, ls_dec=1.5, verbose=False)
    out.update(gain_x=x, gain_F=F, gain_Gain=Gain, gain_Gdiv=Gdiv, gain_Gavg=Gavg)
    xs, F, SP, SN, T = accbpg.D_opt_FW_away(V, x0, 1e-8, 3000, verbose=False)
    out.update(away_x=xs, away_F=F, away_SP=SP, away_SN=SN)
    save("housing", **out)


def large(accbpg, m=2048, n=32768, seed=10, iters=12):
    """Config-2 size: per-call values and a short ABPG_gain trajectory."""
    import numpy as np
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    t = time.time()
    f0, g0 = f.func_grad(x0, 2)
    print("func_grad at x0: %.1f s" % (time.time() - t), flush=True)
    rng = np.random.RandomState(77)
    x = rng.rand(n) + 0.05
    x /= x.sum()
    fx, g = f.func_grad(x, 2)
    z = h.div_prox_map(x, g, 1.0)
    out = {"m": m, "n": n, "seed": seed, "iters": iters, "f0": f0, "g0": g0,
           "x": x, "f": fx, "g": g, "prox": z, "div": h.divergence(z, x)}
    save("large_percall", **out)
    t = time.time()
    xs, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, verbose=True)
    print("ABPG_gain %d its: %.1f s" % (iters, time.time() - t), flush=True)
    save("large_gain", m=m, n=n, seed=seed, iters=iters, x=xs, F=F, Gain=Gain, Gdiv=Gdiv,
         Gavg=Gavg, ref_seconds=T[-1])
    V = f.H
    t = time.time()
    xf, F, SP, SN, T = accbpg.D_opt_FW(V, x0, 1e-8, 40, verbose=False)
    xa, Fa, SPa, SNa, Ta = accbpg.D_opt_FW_away(V, x0, 1e-8, 40, verbose=False)
    print("FW 40 its x2: %.1f s" % (time.time() - t), flush=True)
    save("large_fw", m=m, n=n, seed=seed, fw_x=xf, fw_F=F, fw_SP=SP, fw_SN=SN,
         away_x=xa, away_F=Fa, away_SP=SPa, away_SN=SNa)


def large_long(accbpg, m=2048, n=32768, seed=10, iters=120):
    """Config-2 size, longer horizon: ABPG(gamma=2, theta_eq=True) and BPG with line search for `iters`
    iterations (the two solvers whose decisions are stable to 1e-17 across BLAS thread counts; about an
    hour of CPU on 8 cores).  Stores the final iterate, a mid-run iterate and the traces."""
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    t = time.time()
    half = iters // 2
    xh, Fh, Gh, Th = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=half, theta_eq=True, verbose=False)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True, verbose=False)
    print("ABPG %d + %d its: %.1f s" % (half, iters, time.time() - t), flush=True)
    t = time.time()
    xb, Fb, Lb, Tb = accbpg.BPG(f, h, L, x0, maxitrs=half, linesearch=True, verbose=False)
    print("BPG-LS %d its: %.1f s" % (half, time.time() - t), flush=True)
    save("large_long", m=m, n=n, seed=seed, iters=iters, half=half, abpg_x=x, abpg_F=F, abpg_G=G, abpg_xh=xh,
         bpgls_x=xb, bpgls_F=Fb, bpgls_Ls=Lb)


class _CallLog:
    """Pass-through around the reference's objective that notes, for every oracle call, its kind and the value
    it returned, and keeps the argument of the value calls.  ABPG_gain asks for f at the accepted point twice
    in a row (accbpg/algorithms.py:387 and then :347 of the next iteration), which is how the iterate x_k of
    every k is recovered from one run without running the reference once per checkpoint."""

    def __init__(self, f):
        self._f = f
        self.kinds = []
        self.values = []
        self.iterates = []
        self._last = None

    def __call__(self, x):
        v = self._f(x)
        self.kinds.append(0)
        self.values.append(v)
        if self._last is None or (self._last is not None and (self._last == x).all()):
            self.iterates.append((len(self.kinds) - 1, x.copy()))
        self._last = x.copy()
        return v

    def func_grad(self, x, flag=2):
        out = self._f.func_grad(x, flag)
        self.kinds.append(flag)
        self.values.append(out[0] if flag == 2 else (out if flag == 0 else float("nan")))
        return out

    def gradient(self, x):
        self.kinds.append(1)
        self.values.append(float("nan"))
        return self._f.gradient(x)


def large_gain_long(accbpg, m=2048, n=32768, seed=10, iters=64, keep=(16, 24, 32, 40, 48, 56), name="large_gain_long"):
    """Config-2 size, the headline solver past its retry-free transient: ABPG_gain(gamma=2) for `iters` iterations
    (accbpg/algorithms.py:295-420; the inner loop :361-390 starts to retry once G has been cut below what the
    instance supports).  Stores the traces, the value every oracle call returned in call order (rejected trial
    points included), the iterates x_k at the iterations in `keep` and the final iterate."""
    import numpy as np
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    log = _CallLog(f)
    t = time.time()
    xs, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(log, h, L, x0, gamma=2, maxitrs=iters, verbose=True)
    print("ABPG_gain %d its: %.1f s" % (iters, time.time() - t), flush=True)
    xk = {k: x for k, (pos, x) in enumerate(log.iterates)}
    assert len(log.iterates) == len(F), (len(log.iterates), len(F))
    for k in range(len(F)):
        assert log.values[log.iterates[k][0]] == F[k]
    out = dict(m=m, n=n, seed=seed, iters=iters, x=xs, F=F, Gain=Gain, Gdiv=Gdiv, Gavg=Gavg,
               call_kinds=np.array(log.kinds, dtype=np.int8), call_values=np.array(log.values),
               iter_call_pos=np.array([p for p, _ in log.iterates]), ref_seconds=T[-1],
               keep=np.array([k for k in keep if k < len(F)]))
    for k in keep:
        if k < len(F):
            out["x_%d" % k] = xk[k]
    save(name, **out)


class _ValueTap:
    """Pass-through that keeps the argument of selected value calls: ABPG asks for f at the iterate x_k exactly once
    per iteration (accbpg/algorithms.py:135), so value call number k is x_k."""

    def __init__(self, f, keep):
        self._f = f
        self._keep = set(keep)
        self.count = 0
        self.kept = {}

    def __call__(self, x):
        if self.count in self._keep:
            self.kept[self.count] = x.copy()
        self.count += 1
        return self._f(x)

    def gradient(self, x):
        return self._f.gradient(x)

    def func_grad(self, x, flag=2):
        return self._f.func_grad(x, flag)


def large_abpg_1000(accbpg, m=2048, n=32768, seed=10, iters=1000, keep=(250, 500, 750)):
    """Config-2 size over the north-star's horizon: 1000 iterations of ABPG(gamma=2, theta_eq=True) -- the
    accelerated solver whose decisions are reproducible in the reference itself -- with the iterates at k = 250, 500,
    750 and the final one (about three hours of CPU on 8 cores)."""
    import numpy as np
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    tap = _ValueTap(f, keep)
    t = time.time()
    x, F, G, T = accbpg.ABPG(tap, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True, verbose=True, verbskip=50)
    print("ABPG %d its: %.1f s" % (iters, time.time() - t), flush=True)
    out = dict(m=m, n=n, seed=seed, iters=iters, x=x, F=F, G=G, keep=np.array(sorted(tap.kept)), ref_seconds=T[-1])
    for k, xk in tap.kept.items():
        out["x_%d" % k] = xk
    save("large_abpg_1000", **out)


def large_fw_long(accbpg, m=2048, n=32768, seed=10, iters=1000, name="large_fw_long"):
    """Config 3 (BASELINE.json): D_opt_FW and D_opt_FW_away at (2048, 32768) for 1000 iterations from x0 = 1/n
    (accbpg/D_opt_alg.py:9-88, 91-187): final iterates and the per-iteration traces (objective, dual gap /
    positive and negative gaps), which pin every step choice along the way."""
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    V = f.H
    t = time.time()
    xf, F, SP, SN, T = accbpg.D_opt_FW(V, x0, 1e-8, iters, verbose=True, verbskip=100)
    print("FW %d its: %.1f s" % (iters, time.time() - t), flush=True)
    t = time.time()
    xa, Fa, SPa, SNa, Ta = accbpg.D_opt_FW_away(V, x0, 1e-8, iters, verbose=True, verbskip=100)
    print("FW away %d its: %.1f s" % (iters, time.time() - t), flush=True)
    save(name, m=m, n=n, seed=seed, iters=iters, fw_x=xf, fw_F=F, fw_SP=SP, fw_SN=SN,
         away_x=xa, away_F=Fa, away_SP=SPa, away_SN=SNa)


def large_bpg_long(accbpg, m=2048, n=32768, seed=10, iters=300, name="large_bpg_long"):
    """Config-2 size: BPG with line search for 300 iterations (accbpg/algorithms.py:11-72)."""
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    t = time.time()
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=iters, linesearch=True, verbose=True, verbskip=25)
    print("BPG-LS %d its: %.1f s" % (iters, time.time() - t), flush=True)
    save(name, m=m, n=n, seed=seed, iters=iters, x=x, F=F, Ls=Ls, ref_seconds=T[-1])


def m8192(accbpg, m=8192, n=16400, seed=10, iters=3):
    """Config-5's m (two-level Cholesky, the replicated tail of the sharded evaluation) against the real
    reference: per-call values at two points and a few ABPG iterations.  n is ragged on purpose (16400 =
    128*128 + 16: the last column tile of the gradient product is partial, and 8 logical shards are unequal)."""
    import numpy as np
    import time
    f, h, L, x0 = accbpg.D_opt_design(m, n, randseed=seed)
    t = time.time()
    f0, g0 = f.func_grad(x0, 2)
    print("func_grad at x0: %.1f s" % (time.time() - t), flush=True)
    rng = np.random.RandomState(8192)
    x = rng.rand(n) + 0.05
    x /= x.sum()
    fx, g = f.func_grad(x, 2)
    out = {"m": m, "n": n, "seed": seed, "iters": iters, "f0": f0, "g0": g0, "x": x, "f": fx, "g": g}
    t = time.time()
    xs, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2, maxitrs=iters, theta_eq=True, verbose=False)
    print("ABPG %d its: %.1f s" % (iters, time.time() - t), flush=True)
    out.update(abpg_x=xs, abpg_F=F, abpg_G=G)
    save("percall_%dx%d" % (m, n), **out)


def traces_512(accbpg):
    """1000-iteration traces at the config-4 instance size (about 10 minutes of CPU)."""
    f, h, L, x0 = accbpg.D_opt_design(512, 8192, randseed=10)
    out = {"m": 512, "n": 8192, "seed": 10, "iters": 1000}
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=1000, linesearch=True, verbose=False)
    out.update(bpgls_x=x, bpgls_F=F, bpgls_Ls=Ls)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=1000, theta_eq=True, verbose=False)
    out.update(abpg_x=x, abpg_F=F, abpg_G=G)
    x, F, Gain, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=300, verbose=False)
    out.update(gain_x=x, gain_F=F, gain_Gain=Gain, gain_Gavg=Gavg)
    save("traces_512x8192", **out)


def next_rows(accbpg):
    """KYinit / libsvm instances, ABPG_expo, ABDA, FW_alg_div_step + lmo_simplex,
    with the calls of frank_wolfe_wtih_rs/ex_Dopt_design.py:12-21 on the housing instance and of
    ipynb/ex_Dopt_random.ipynb on the (80,200) instance."""
    import numpy as np
    out = {}
    for name in ["housing", "bodyfat", "mpg", "abalone"]:
        f, h, L, x0 = accbpg.D_opt_libsvm(os.path.join(REF, "parameters_free_fw", "data", name + ".txt"))
        out["libsvm_%s_shape" % name] = np.array(f.H.shape)
        out["libsvm_%s_f0" % name] = f(x0)
        out["libsvm_%s_checksum" % name] = np.array([f.H.sum(), np.abs(f.H).max(), (f.H ** 2).sum()])
    f, h, L, x0 = accbpg.D_opt_libsvm(os.path.join(REF, "parameters_free_fw", "data", "housing.txt"))
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=300, linesearch=True, ls_ratio=2, verbose=False)
    out.update(h_bpg_x=x, h_bpg_F=F, h_bpg_Ls=Ls)
    x, F, Ls, T = accbpg.FW_alg_div_step(f, h, L, x0, lmo=accbpg.lmo_simplex(), maxitrs=300, gamma=2.0,
                                         ls_ratio=2, verbose=False)
    out.update(h_fwdiv_x=x, h_fwdiv_F=F, h_fwdiv_Ls=Ls)
    x, F, Gamma, G, T = accbpg.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True, Gmargin=100,
                                         verbose=False)
    out.update(h_expo_x=x, h_expo_F=F, h_expo_Gamma=Gamma, h_expo_G=G)
    f, h, L, x0 = accbpg.D_opt_design(80, 200, randseed=10)
    x, F, Gamma, G, T = accbpg.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True, verbose=False)
    out.update(r_expo_x=x, r_expo_F=F, r_expo_Gamma=Gamma, r_expo_G=G)
    x, F, Gamma, G, T = accbpg.ABPG_expo(f, h, L, x0, gamma0=2.5, maxitrs=200, theta_eq=False, checkdiv=True,
                                         Gmargin=5, restart=True, verbose=False)
    out.update(r_expo2_x=x, r_expo2_F=F, r_expo2_Gamma=Gamma, r_expo2_G=G)
    x, F, G, T = accbpg.ABDA(f, h, L, x0, gamma=2, maxitrs=300, theta_eq=True, verbose=False)
    out.update(r_abda_x=x, r_abda_F=F, r_abda_G=G)
    x, F, Ls, T = accbpg.FW_alg_div_step(f, h, L, x0, lmo=accbpg.lmo_simplex(), maxitrs=300, gamma=2.0,
                                         ls_ratio=2, verbose=False)
    out.update(r_fwdiv_x=x, r_fwdiv_F=F, r_fwdiv_Ls=Ls)
    # Kumar-Yildirim start (ipynb/ABPGvsFW/ex_Dopt_ABPGvsFW.ipynb:171-173 uses D_opt_design(30,1000))
    f, h, L, x0 = accbpg.D_opt_design(30, 1000, randseed=4)
    np.random.seed(99)
    xky = accbpg.D_opt_KYinit(f.H)
    out.update(ky_x=xky, ky_f=f(np.maximum(xky, 0)) if xky.min() >= 0 else np.nan)
    xs, F, SP, SN, T = accbpg.D_opt_FW_away(f.H, xky, 1e-8, 2000, verbose=False)
    out.update(ky_away_x=xs, ky_away_F=F, ky_away_SP=SP, ky_away_SN=SN)
    save("next_rows", **out)


def poisson(accbpg):
    """PoissonRegression + BurgEntropyL1/L2, with the calls of ipynb/ex_Poisson_L2.ipynb
    (cells 1, 3 and 5) at a reduced iteration count.  A is not stored: the tests rebuild it with the same legacy
    NumPy RNG call sequence the factory uses (accbpg/applications.py:114-121) and compare the checksum."""
    import numpy as np
    out = {}
    N = 2000
    for tag, fac, m, n, noise, lam in [("l1", accbpg.Poisson_regrL1, 200, 100, 0.0001, 0),
                                       ("l2", accbpg.Poisson_regrL2, 100, 1000, 0.001, 0.001),
                                       ("l1r", accbpg.Poisson_regrL1, 300, 2000, 0.001, 0.01)]:
        f, h, L, x0 = fac(m, n, noise=noise, lamda=lam, randseed=1)
        out[tag + "_cfg"] = np.array([m, n, noise, lam])
        out[tag + "_A_checksum"] = np.array([f.A.sum(), np.abs(f.A).max(), (f.A ** 2).sum()])
        out[tag + "_b"] = f.b
        out[tag + "_L"] = L
        out[tag + "_x0"] = x0
        rng = np.random.RandomState(77)
        x = rng.rand(n) * (2.0 / n) + 1e-3
        y = rng.rand(n) * (2.0 / n) + 1e-3
        fx, g = f.func_grad(x, 2)
        out.update({tag + "_x": x, tag + "_y": y, tag + "_f": fx, tag + "_g": g, tag + "_f0": f(x0),
                    tag + "_g0": f.gradient(x0), tag + "_psi": h.extra_Psi(x)})
        for idx, Lc in enumerate([L, 0.37 * L, 5.0]):
            out["%s_prox_L%d" % (tag, idx)] = Lc
            out["%s_prox_x%d" % (tag, idx)] = h.div_prox_map(y, g, Lc)
        out[tag + "_prox_raw"] = h.prox_map(np.abs(g) + 0.5, 2.0)
        out[tag + "_div_xy"] = h.divergence(x, y)

    f, h, L, x0 = accbpg.Poisson_regrL1(200, 100, noise=0.0001, lamda=0, randseed=1)
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=N, linesearch=False, verbose=False)
    out.update(l1_bpg_x=x, l1_bpg_F=F)
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=N, linesearch=True, verbose=False)
    out.update(l1_bpgls_x=x, l1_bpgls_F=F, l1_bpgls_Ls=Ls)
    for gam, key in [(1.0, "g10"), (1.5, "g15"), (2.0, "g20")]:
        x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=gam, maxitrs=N, theta_eq=True, verbose=False)
        out.update({"l1_abpg_%s_x" % key: x, "l1_abpg_%s_F" % key: F, "l1_abpg_%s_G" % key: G})
    x, F, G, T = accbpg.ABDA(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=True, verbose=False)
    out.update(l1_abda_x=x, l1_abda_F=F, l1_abda_G=G)
    x, F, Gamma, G, T = accbpg.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=3, verbose=False)
    out.update(l1_expo_x=x, l1_expo_F=F, l1_expo_Gamma=Gamma, l1_expo_G=G)
    x, F, G, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, theta_eq=False, verbose=False)
    out.update(l1_gain_x=x, l1_gain_F=F, l1_gain_G=G, l1_gain_Gdiv=Gdiv, l1_gain_Gavg=Gavg)

    f, h, L, x0 = accbpg.Poisson_regrL2(100, 1000, noise=0.001, lamda=0.001, randseed=1)
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=N, linesearch=False, verbose=False)
    out.update(l2_bpg_x=x, l2_bpg_F=F)
    x, F, Ls, T = accbpg.BPG(f, h, L, x0, maxitrs=N, linesearch=True, ls_ratio=1.5, verbose=False)
    out.update(l2_bpgls_x=x, l2_bpgls_F=F, l2_bpgls_Ls=Ls)
    x, F, G, T = accbpg.ABPG(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=False, verbose=False)
    out.update(l2_abpg_x=x, l2_abpg_F=F, l2_abpg_G=G)
    x, F, Gamma, G, T = accbpg.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=1, verbose=False)
    out.update(l2_expo_x=x, l2_expo_F=F, l2_expo_Gamma=Gamma, l2_expo_G=G)
    x, F, G, Gdiv, Gavg, T = accbpg.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, ls_inc=1.5, ls_dec=1.5,
                                              theta_eq=True, verbose=False)
    out.update(l2_gain_x=x, l2_gain_F=F, l2_gain_G=G, l2_gain_Gdiv=Gdiv, l2_gain_Gavg=Gavg)
    save("poisson", **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--medium", action="store_true")
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only-large", action="store_true")
    ap.add_argument("--only-traces", action="store_true")
    ap.add_argument("--only-next", action="store_true")
    ap.add_argument("--only-poisson", action="store_true")
    ap.add_argument("--only-large-long", action="store_true")
    ap.add_argument("--only-large-gain-long", action="store_true")
    ap.add_argument("--only-large-abpg-1000", action="store_true")
    ap.add_argument("--only-large-fw-long", action="store_true")
    ap.add_argument("--only-large-bpg-long", action="store_true")
    ap.add_argument("--only-m8192", action="store_true")
    ap.add_argument("--keep", default="16,24,32,40,48,56", help="iterations whose iterate the long ABPG_gain fixture keeps")
    ap.add_argument("--name", default="large_gain_long")
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--iters", type=int, default=64)
    ap.add_argument("--out", default=None, help="write into this directory instead of tests/golden")
    args = ap.parse_args()
    accbpg = load_reference()
    if args.out:
        global OUT
        OUT = args.out
    if args.only_poisson:
        poisson(accbpg)
        return
    if args.only_m8192:
        m8192(accbpg)
        return
    if args.only_large_long:
        large_long(accbpg)
        return
    if args.only_large_gain_long:
        large_gain_long(accbpg, args.m, args.n, iters=args.iters, keep=tuple(int(k) for k in args.keep.split(",")),
                        name=args.name)
        return
    if args.only_large_abpg_1000:
        large_abpg_1000(accbpg)
        return
    if args.only_large_fw_long:
        if args.name != "large_gain_long":                     # (--iters / --name: a longer fixture beside the 1000-iteration one)
            large_fw_long(accbpg, iters=args.iters, name=args.name)
        else:
            large_fw_long(accbpg)
        return
    if args.only_large_bpg_long:
        if args.name != "large_gain_long":
            large_bpg_long(accbpg, iters=args.iters, name=args.name)
        else:
            large_bpg_long(accbpg)
        return
    if args.only_next:
        next_rows(accbpg)
        return
    if args.only_traces:
        solver_traces(accbpg, "80x200", 80, 200, 10, 1000)
        solver_traces(accbpg, "80x120", 80, 120, 10, 300)
        if args.medium:
            solver_traces(accbpg, "256x4096", 256, 4096, 10, 1000)
        return
    if not args.only_large:
        percall(accbpg, "80x200", 80, 200, 10)
        percall(accbpg, "128x1024", 128, 1024, 3)
        percall(accbpg, "200x2000", 200, 2000, 7)
        solver_traces(accbpg, "80x200", 80, 200, 10, 1000)
        solver_traces(accbpg, "80x120", 80, 120, 10, 300)
        fw_traces(accbpg, "30x1000", 30, 1000, 5, 6000)
        fw_traces(accbpg, "64x512", 64, 512, 2, 3000)
        housing(accbpg)
        next_rows(accbpg)
        poisson(accbpg)
    if args.medium:
        traces_512(accbpg)
        percall(accbpg, "512x8192", 512, 8192, 1)
        solver_traces(accbpg, "256x4096", 256, 4096, 10, 1000)
        fw_traces(accbpg, "256x4096", 256, 4096, 10, 2000)
    if args.large or args.only_large:
        large(accbpg)


if __name__ == "__main__":
    main()
