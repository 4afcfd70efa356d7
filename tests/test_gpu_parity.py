"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs, against golden vectors written by the real reference, and -- at the
BASELINE sizes -- through size-independent properties."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden, gaussian_design

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def acc():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import accbpg_and_fw_amd as a
    return a


@pytest.fixture(scope="module")
def O():
    from oracle import np_oracle
    return np_oracle


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------ MFMA engine
@pytest.mark.parametrize("config", [0, 1])
@pytest.mark.parametrize("kmajor", [0, 1])
@pytest.mark.parametrize("shape", [(64, 64, 64), (256, 128, 48), (300, 200, 77), (129, 65, 33),
                                   (512, 384, 1000), (17, 5, 3)])
def test_mfma_gemm_matches_numpy(acc, config, kmajor, shape):
    """v_mfma_f64_16x16x4_f64 tile engine: operand maps, LDS images, edge guards.
    Asymmetric random operands (a transposed result must fail)."""
    from accbpg_and_fw_amd import _lib
    lib = _lib.load()
    M, N, K = shape
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    A = rng.randn(M, K)
    B = rng.randn(K, N) if kmajor else rng.randn(N, K)
    C0 = rng.randn(M, N)
    alpha, beta = -1.25, 0.5
    ref = alpha * (A @ (B if kmajor else B.T)) + beta * C0
    Ad, Bd, Cd = dev(A), dev(B), dev(C0)
    rc = lib.accbpg_test_gemm(Ad.data_ptr(), K, Bd.data_ptr(), B.shape[1], Cd.data_ptr(), N, M, N, K,
                              kmajor, alpha, beta, config, None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = Cd.cpu().numpy()
    scale = np.abs(A) @ np.abs(B if kmajor else B.T) + np.abs(C0)
    assert np.max(np.abs(got - ref) / scale) < 1e-14       # fp64, tolerance ~ K*eps relative to |A||B|


# ------------------------------------------------------------------ objective
@pytest.mark.parametrize("tag", ["80x200", "128x1024", "200x2000", "512x8192"])
def test_func_grad_matches_reference_golden(acc, O, tag):
    gd = golden("percall_" + tag)
    m, n, seed = int(gd["m"]), int(gd["n"]), int(gd["seed"])
    f, h, L, x0 = acc.D_opt_design(m, n, randseed=seed)
    for xk, fk, gk in [("x", "f", "g"), (None, "f0", "g0")]:
        x = gd[xk] if xk else x0
        fx, g = f.func_grad(x, 2)
        # tolerance: |f| ~ 10..100 absolute 1e-11; gradient relative 1e-11 (fp64, different
        # factorisation (Cholesky vs LU) and summation order)
        assert abs(fx - float(gd[fk])) < 1e-11 * max(1.0, abs(float(gd[fk])))
        np.testing.assert_allclose(g, gd[gk], rtol=1e-11, atol=0)
        assert f(x) == fx                                   # flag 0 is the same factorisation
        np.testing.assert_array_equal(f.gradient(x), g)
    # device-tensor protocol returns device tensors
    fx2, g2 = f.func_grad(dev(gd["x"]), 2)
    assert isinstance(g2, torch.Tensor) and g2.is_cuda
    np.testing.assert_allclose(g2.cpu().numpy(), gd["g"], rtol=1e-11)


@pytest.mark.parametrize("shape", [(13, 506, 0), (80, 200, 10), (100, 1001, 4), (333, 777, 5),
                                   (768, 2048, 6), (1024, 4096, 8), (1000, 3000, 9)])
def test_func_grad_matches_oracle_ragged_sizes(acc, O, shape):
    """Sizes that are not tile multiples (odd n, odd m, big tile with edges)."""
    m, n, seed = shape
    V = gaussian_design(m, n, seed + 100)
    rng = np.random.RandomState(seed)
    x = rng.rand(n) + 0.01
    x /= x.sum()
    f = acc.DOptimalObj(V)
    fo = O.DOptOracle(V)
    fx, g = f.func_grad(x, 2)
    fr, gr = fo.func_grad(x, 2)
    assert abs(fx - fr) < 1e-11 * max(1.0, abs(fr))
    np.testing.assert_allclose(g, gr, rtol=1e-11, atol=0)


def test_func_grad_errors(acc):
    """AssertionError / ValueError behaviour of accbpg/functions.py:44-50."""
    f, h, L, x0 = acc.D_opt_design(8, 20, randseed=3)
    with pytest.raises(AssertionError):
        f.func_grad(-x0)
    with pytest.raises(AssertionError):
        f.func_grad(x0[:-1])
    z = np.zeros(20)
    z[:3] = 1.0 / 3                      # rank 3 < m: Gram matrix singular
    with pytest.raises(ValueError, match="HXHT is singular or not positive definite"):
        f.func_grad(z)
    with pytest.raises(AssertionError):
        acc.DOptimalObj(np.zeros((5, 5)))


# ------------------------------------------------------------------ Burg kernel
@pytest.mark.parametrize("tag", ["80x200", "128x1024", "200x2000", "512x8192"])
def test_prox_and_divergence_match_reference_golden(acc, tag):
    gd = golden("percall_" + tag)
    h = acc.BurgEntropySimplex()
    for idx in range(3):
        z = h.div_prox_map(gd["y"], gd["g"], float(gd["prox_L%d" % idx]))
        # same scalar algorithm; only the reduction order of the two sums differs
        np.testing.assert_allclose(z, gd["prox_x%d" % idx], rtol=1e-12, atol=0)
        assert abs(z.sum() - 1) <= 1.001e-8               # stops at |phi| <= eps, not renormalised
    gg = gd["g"] - gd["g"].min() + 0.5
    np.testing.assert_allclose(h.prox_map(gg, 2.0), gd["prox_raw"], rtol=1e-12)
    assert h.divergence(gd["x"], gd["y"]) == pytest.approx(float(gd["div_xy"]), rel=1e-12)
    assert h.divergence(gd["y"], gd["x"]) == pytest.approx(float(gd["div_yx"]), rel=1e-12)
    assert h.extra_Psi(gd["x"]) == 0


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 2048, 2049, 8193, 40000, 300000])
def test_prox_sizes_against_oracle(acc, O, n):
    """Every register-cached variant and the re-read variant, ragged n."""
    rng = np.random.RandomState(n)
    y = rng.rand(n) + 1e-3
    y /= y.sum()
    g = -rng.rand(n) * 50 - 1
    h, ho = acc.BurgEntropySimplex(), O.BurgSimplexOracle()
    for L in (1.0, 0.02):
        z = h.div_prox_map(y, g, L)
        zo = ho.div_prox_map(y, g, L)
        np.testing.assert_allclose(z, zo, rtol=1e-11, atol=0)
        assert h.last_info[1] == ho.last_newton_steps


def test_burg_errors(acc):
    h = acc.BurgEntropySimplex()
    x = np.ones(10) / 10
    with pytest.raises(AssertionError):
        h.div_prox_map(x, x, -1.0)
    with pytest.raises(AssertionError):
        h.div_prox_map(0 * x, x, 1.0)
    with pytest.raises(AssertionError):
        h.divergence(x, 0 * x)
    with pytest.raises(AssertionError):
        h.divergence(x, x[:-1])


def test_vector_helpers(acc):
    from accbpg_and_fw_amd.functions import vec_axpby, vec_dot_diff, ls_terms, vec_min_sum
    rng = np.random.RandomState(5)
    n = 5003
    x, z, g = rng.rand(n) + .1, rng.rand(n) + .1, rng.randn(n)
    th = 0.3217
    out = vec_axpby(1 - th, dev(x), th, dev(z)).cpu().numpy()
    np.testing.assert_array_equal(out, (1 - th) * x + th * z)            # bitwise: no FMA contraction
    assert vec_dot_diff(dev(g), dev(x), dev(z)) == pytest.approx(np.dot(g, x - z), rel=1e-12)
    d, dxy, dzz = ls_terms(dev(g), dev(x), dev(z), dev(z), dev(x))
    assert dxy == pytest.approx(np.sum(x / z - np.log(x / z) - 1), rel=1e-12)
    assert dzz == pytest.approx(np.sum(z / x - np.log(z / x) - 1), rel=1e-12)
    mn, sm = vec_min_sum(dev(x))
    assert mn == x.min() and sm == pytest.approx(x.sum(), rel=1e-13)


# ------------------------------------------------------------------ solver trajectories
def _close(a, b, tol):
    assert a.shape == b.shape, (a.shape, b.shape)
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def _agree_prefix(a, b, tol):
    """length of the common prefix on which two traces agree to tol"""
    n = min(len(a), len(b))
    bad = np.nonzero(np.abs(a[:n] - b[:n]) > tol * (1 + np.abs(b[:n])))[0]
    return n if bad.size == 0 else int(bad[0])


def test_bpg_abpg_trajectories_80x200(acc):
    """1000 iterations at the notebook instance D_opt_design(80,200,randseed=10) with the notebook's
    calls (ipynb/ex_Dopt_random.ipynb cells 1 and 3): iterates within the north-star tolerance
    l_inf < 1e-9, traces to 1e-9, stored stdout rows reproduced."""
    gd = golden("traces_80x200")
    f, h, L, x0 = acc.D_opt_design(80, 200, randseed=10)
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=1000, linesearch=False, verbose=False)
    assert np.max(np.abs(x - gd["bpg_x"])) < 1e-9
    _close(F, gd["bpg_F"], 1e-9)
    assert "%.3e" % F[0] == "1.910e+01" and "%.3e" % F[900] == "1.759e+01"   # ex_Dopt_random.ipynb:73,82
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=1000, linesearch=True, verbose=False)
    assert np.max(np.abs(x - gd["bpgls_x"])) < 1e-9
    _close(F, gd["bpgls_F"], 1e-9); _close(Ls, gd["bpgls_Ls"], 1e-12)
    assert "%.3e" % Ls[0] == "8.333e-01" and "%.3e" % Ls[100] == "1.938e-01"   # :243-244
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=1000, theta_eq=True, verbose=False)
    assert np.max(np.abs(x - gd["abpg_x"])) < 1e-9
    _close(F, gd["abpg_F"], 1e-9); _close(G[:500], gd["abpg_G"][:500], 1e-6)
    assert "%.3e" % G[100] == "5.529e-01"                                    # :113
    assert len(T) == len(F) and np.all(np.diff(T) >= 0)
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=1000, theta_eq=True, restart=True, verbose=False)
    assert np.max(np.abs(x - gd["abpgrs_x"])) < 1e-9
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=1.5, maxitrs=1000, theta_eq=False, verbose=False)
    assert np.max(np.abs(x - gd["abpgk_x"])) < 1e-9
    _close(F, gd["abpgk_F"], 1e-9)


_GAIN_VARIANTS = [("gain", dict(G0=0.1, theta_eq=True)),
                  ("gainrs", dict(G0=0.1, theta_eq=True, restart=True)),
                  ("gaindef", dict()),
                  ("gainopt", dict(G0=0.1, ls_inc=1.5, ls_dec=1.1, theta_eq=False, checkdiv=True,
                                   restart=True, restart_rule='f'))]


def _check_gain_runs(acc, tag, stable, tol_x):
    """ABPG_gain takes discrete line-search decisions; once converged they sit at the rounding floor
    and the reference itself is not reproducible across BLAS thread counts (first differs at k=756 on
    (80,200), k=36..79 on (256,4096); tests/test_oracle.py).  So: identical gain sequence and F on the
    decision-stable prefix (>= `stable` iterations), objective-level agreement at the end."""
    gd = golden("traces_" + tag)
    m, n, seed, iters = int(gd["m"]), int(gd["n"]), int(gd["seed"]), int(gd["iters"])
    f, h, L, x0 = acc.D_opt_design(m, n, randseed=seed)
    for key, kw in _GAIN_VARIANTS:
        x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, verbose=False, **kw)
        ref_gain = gd[key + "_Gain"]
        k = _agree_prefix(Gain, ref_gain, 1e-12)
        assert k >= min(stable, len(ref_gain) - 5), (key, k)
        ks = min(k, stable)
        _close(F[:ks], gd[key + "_F"][:ks], 1e-9)           # decision-stable prefix: tight
        _close(F[:k], gd[key + "_F"][:k], 1e-5)             # same decisions, rounding already amplified
        _close(Gavg[:k], gd[key + "_Gavg"][:k], 1e-9)
        assert len(F) == len(Gain) == len(Gdiv) == len(Gavg) == len(T)
        nf = min(len(F), len(gd[key + "_F"]))
        # after the decisions part ways only the objective level is comparable (the reference itself
        # ends 1.5e-5 apart between 1 and 8 BLAS threads on (256,4096))
        assert abs(F[nf - 1] - gd[key + "_F"][nf - 1]) < 1e-4
        if k == len(ref_gain) == len(Gain):
            assert np.max(np.abs(x - gd[key + "_x"])) < tol_x
    return f, h, L, x0


def test_abpg_gain_trajectories_80x200(acc):
    f, h, L, x0 = _check_gain_runs(acc, "80x200", stable=400, tol_x=1e-9)
    x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=200, G0=0.1, theta_eq=True,
                                               verbose=False)
    assert "%.3e" % Gain[0] == "2.488e-01" and "%.3e" % Gavg[0] == "4.988e-02"      # ex_Dopt_random.ipynb:282
    assert "%.3e" % Gain[100] == "2.986e-01" and "%.3e" % Gdiv[100] == "7.091e-01"  # :283


def test_early_stop_and_truncation_80x120(acc):
    """Stopping rules and array truncation (algorithms.py:66-71,174-179,412-419).  The stop tests
    compare rounding-level quantities with 1e-14, so the stopping iteration may move by a few
    iterations between summation orders; the traces must agree on the common prefix."""
    gd = golden("traces_80x120")
    f, h, L, x0 = acc.D_opt_design(80, 120, randseed=10)
    assert "%.3e" % f(x0) == "3.764e+01"                              # ex_Dopt_random.ipynb:398
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=300, linesearch=False, verbose=False)
    assert len(F) < 300 and len(F) == len(Ls) == len(T)
    n = min(len(F), len(gd["bpg_F"]))
    assert n >= 40
    _close(F[:n], gd["bpg_F"][:n], 1e-10)
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=300, theta_eq=True, verbose=False)
    n = min(len(F), len(gd["abpg_F"]))
    assert len(F) < 300 and n >= 40
    _close(F[:n], gd["abpg_F"][:n], 1e-10)
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=300, theta_eq=True, restart=True, verbose=False)
    n = min(len(F), len(gd["abpgrs_F"]))
    assert n >= 20
    _close(F[:n], gd["abpgrs_F"][:n], 1e-10)
    _check_gain_runs(acc, "80x120", stable=15, tol_x=1e-8)
    x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=50, G0=0.1, theta_eq=True,
                                               verbose=False)
    assert "%.3e" % Gain[0] == "5.160e-01" and "%.3e" % Gavg[0] == "7.183e-02"      # ex_Dopt_random.ipynb:398-418


def test_trajectories_256x4096(acc):
    """1000-iteration l_inf parity at a size where the CPU side is affordable."""
    gd = golden("traces_256x4096")
    f, h, L, x0 = acc.D_opt_design(256, 4096, randseed=10)
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=1000, linesearch=False, verbose=False)
    assert np.max(np.abs(x - gd["bpg_x"])) < 1e-9
    _close(F, gd["bpg_F"], 1e-9)
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=1000, linesearch=True, verbose=False)
    assert np.max(np.abs(x - gd["bpgls_x"])) < 1e-9
    _close(Ls, gd["bpgls_Ls"], 1e-12)
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=1000, theta_eq=True, verbose=False)
    assert np.max(np.abs(x - gd["abpg_x"])) < 1e-9
    _close(F, gd["abpg_F"], 1e-9)
    _check_gain_runs(acc, "256x4096", stable=30, tol_x=1e-9)


def test_reference_loop_runs_on_device_objects(acc, O):
    """Drop-in direction 2: a NumPy driver loop (here the oracle's BPG, which has the reference's
    structure) runs unchanged on this package's f / h objects."""
    gd = golden("traces_80x200")
    f, h, L, x0 = acc.D_opt_design(80, 200, randseed=10)
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=50, linesearch=False)
    _close(F, gd["bpg_F"][:50], 1e-10)


def test_housing_rng_free(acc):
    gd = golden("housing")
    V = gd["V"]
    n = V.shape[1]
    f, h = acc.DOptimalObj(V), acc.BurgEntropySimplex()
    x0 = np.ones(n) / n
    assert "%.3e" % f(x0) == "-4.137e+01"                             # ex_Dopt_LIBSVM.ipynb:191
    x, F, Ls, T = acc.BPG(f, h, 1.0, x0, maxitrs=1001, linesearch=False, verbose=False)
    _close(F, gd["bpg_F"], 1e-9)
    assert np.max(np.abs(x - gd["bpg_x"])) < 1e-9
    x, F, G, T = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=1001, verbose=False)
    _close(F, gd["abpg_F"], 1e-9)
    assert np.max(np.abs(x - gd["abpg_x"])) < 1e-9


# ------------------------------------------------------------------ Frank-Wolfe
@pytest.mark.parametrize("tag", ["30x1000", "64x512"])
def test_fw_trajectories(acc, tag):
    gd = golden("fw_" + tag)
    m, n, seed, iters = int(gd["m"]), int(gd["n"]), int(gd["seed"]), int(gd["iters"])
    V = gaussian_design(m, n, seed)
    x0 = np.ones(n) / n
    x, F, SP, SN, T = acc.D_opt_FW(V, x0, float(gd["eps"]), iters, verbose=False)
    assert len(F) == len(gd["fw_F"])
    assert np.max(np.abs(x - gd["fw_x"])) < 1e-9
    _close(F, gd["fw_F"], 1e-9); _close(SP, gd["fw_SP"], 1e-9); _close(SN, gd["fw_SN"], 1e-9)
    x, F, SP, SN, T = acc.D_opt_FW_away(V, x0, float(gd["eps"]), iters, verbose=False)
    assert abs(len(F) - len(gd["away_F"])) <= 2
    k = min(len(F), len(gd["away_F"]))
    assert np.max(np.abs(x - gd["away_x"])) < 1e-8
    _close(F[:k], gd["away_F"][:k], 1e-9); _close(SP[:k], gd["away_SP"][:k], 1e-8)
    # extension: determinant-lemma tracking of log det(H) between refactorisations
    x2, F2, SP2, SN2, T2 = acc.D_opt_FW_away(V, x0, float(gd["eps"]), iters, verbose=False, logdet_refresh=50)
    np.testing.assert_array_equal(x2, x)
    _close(F2[:k], F[:k], 1e-9)


def test_fw_housing_and_state(acc):
    gd = golden("housing")
    V = gd["V"]
    n = V.shape[1]
    x0 = np.ones(n) / n
    x, F, SP, SN, T = acc.D_opt_FW_away(V, x0, 1e-8, 3000, verbose=False)
    k = min(len(F), len(gd["away_F"]))
    _close(F[:k], gd["away_F"][:k], 1e-8)
    assert np.max(np.abs(x - gd["away_x"])) < 1e-8
    assert abs(x.sum() - 1) < 1e-9 and x.min() > -1e-15


# ------------------------------------------------------------------ BASELINE size (config 2 / 3)
@pytest.fixture(scope="module")
def large(acc):
    gd = golden("large_percall")
    f, h, L, x0 = acc.D_opt_design(int(gd["m"]), int(gd["n"]), randseed=int(gd["seed"]))
    return f, h, L, x0, gd


def test_large_percall_2048x32768(large):
    """Per-call parity at D_opt_design(2048,32768): f, g, prox, divergence against values the real
    reference produced for the same seed (oracle/gen_golden.py --large)."""
    f, h, L, x0, gd = large
    f0, g0 = f.func_grad(x0, 2)
    assert abs(f0 - float(gd["f0"])) < 1e-10 * abs(float(gd["f0"]))
    np.testing.assert_allclose(g0, gd["g0"], rtol=1e-11)
    fx, g = f.func_grad(gd["x"], 2)
    assert abs(fx - float(gd["f"])) < 1e-10 * abs(float(gd["f"]))
    np.testing.assert_allclose(g, gd["g"], rtol=1e-11)
    z = h.div_prox_map(gd["x"], gd["g"], 1.0)
    np.testing.assert_allclose(z, gd["prox"], rtol=1e-11)
    assert h.divergence(gd["prox"], gd["x"]) == pytest.approx(float(gd["div"]), rel=1e-11)


def test_large_properties_2048x32768(large):
    """Size-independent properties: sum_i x_i * (-g_i) = m (trace identity of the D-optimal
    gradient), f(c*x) = f(x) - m*log(c), gradient is (-1)-homogeneous."""
    f, h, L, x0, gd = large
    m = f.m
    x = gd["x"]
    fx, g = f.func_grad(x, 2)
    assert abs(np.dot(x, -g) - m) < 1e-9 * m
    c = 1.7
    f2, g2 = f.func_grad(c * x, 2)
    assert abs(f2 - (fx - m * np.log(c))) < 1e-9 * abs(fx)
    np.testing.assert_allclose(g2 * c, g, rtol=1e-11)
    z = h.div_prox_map(x, g, 1.0)
    assert z.min() > 0 and abs(z.sum() - 1) <= 1.001e-8
    assert h.divergence(x, x) == 0.0


def test_large_abpg_gain_trajectory_2048x32768(large, acc):
    """12 iterations of ABPG_gain(gamma=2) at config 2 against the real reference's trace."""
    f, h, L, x0, _ = large
    gd = golden("large_gain")
    iters = int(gd["iters"])
    x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, verbose=False)
    assert np.max(np.abs(x - gd["x"])) < 1e-9
    _close(F, gd["F"], 1e-9); _close(Gain, gd["Gain"], 1e-12); _close(Gdiv, gd["Gdiv"], 1e-7)


def test_large_fw_2048x32768(large, acc):
    f, h, L, x0, _ = large
    gd = golden("large_fw")
    x, F, SP, SN, T = acc.D_opt_FW(f, x0, 1e-8, 40, verbose=False)
    assert np.max(np.abs(x - gd["fw_x"])) < 1e-9
    _close(F, gd["fw_F"], 1e-9); _close(SP, gd["fw_SP"], 1e-9)
    x, F, SP, SN, T = acc.D_opt_FW_away(f, x0, 1e-8, 40, verbose=False)
    assert np.max(np.abs(x - gd["away_x"])) < 1e-9
    _close(F, gd["away_F"], 1e-8); _close(SP, gd["away_SP"], 1e-9)


# ------------------------------------------------------------------ sharding (one device, logical shards)
@pytest.mark.parametrize("shape,parts", [((96, 1000), 3), ((1024, 4096), 8), ((300, 1111), 4)])
def test_logical_shards_match_single_device(acc, shape, parts):
    """Design-point sharding with the all-reduce replaced by an in-process sum:
    same f and g as the unsharded objective, and a solver runs on it unchanged."""
    from accbpg_and_fw_amd.sharded import LogicalShards
    m, n = shape
    V = gaussian_design(m, n, 11)
    rng = np.random.RandomState(3)
    x = rng.rand(n) + 0.01
    x /= x.sum()
    f = acc.DOptimalObj(V)
    fs = LogicalShards(V, parts)
    f1, g1 = f.func_grad(x, 2)
    f2, g2 = fs.func_grad(x, 2)
    assert abs(f1 - f2) < 1e-11 * max(1.0, abs(f1))
    np.testing.assert_allclose(g2, g1, rtol=1e-11)
    assert fs(x) == f2
    h = acc.BurgEntropySimplex()
    x0 = np.ones(n) / n
    xa, Fa, Ga, Ta = acc.ABPG(f, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    xb, Fb, Gb, Tb = acc.ABPG(fs, h, 1.0, x0, gamma=2, maxitrs=15, verbose=False)
    assert np.max(np.abs(xa - xb)) < 1e-12
    np.testing.assert_allclose(Fb, Fa, rtol=1e-12, atol=1e-12)


def test_batched_instances_match_sequential(acc):
    """Config-4 style batch: independent instances solved concurrently from host threads on
    separate streams give exactly the results of solving them one after the other."""
    from accbpg_and_fw_amd.batched import solve_batch
    probs = [acc.D_opt_design(96, 640, randseed=50 + j) for j in range(6)]
    seq = [acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=40, verbose=False) for f, h, L, x0 in probs]
    par = solve_batch(probs, acc.ABPG_gain, threads=6, gamma=2, maxitrs=40, verbose=False)
    for a, b in zip(seq, par):
        np.testing.assert_array_equal(a[0], b[0])          # same kernels, same order per instance -> bitwise
        np.testing.assert_array_equal(a[1], b[1])
        np.testing.assert_array_equal(a[2], b[2])


# ------------------------------------------------------------------ SURVEY 8(f) rows 1-3
import os as _os
_DATA = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "golden", "data")


def test_libsvm_instances_gpu(acc):
    """D_opt_libsvm on the four vendored files: F(x0) as the reference computes it."""
    gd = golden("next_rows")
    for name in ["housing", "bodyfat", "mpg", "abalone"]:
        f, h, L, x0 = acc.D_opt_libsvm(_os.path.join(_DATA, name + ".txt"))
        assert f.H.shape == tuple(gd["libsvm_%s_shape" % name])
        assert f(x0) == pytest.approx(float(gd["libsvm_%s_f0" % name]), rel=1e-11)


def test_expo_abda_fwdiv_trajectories(acc):
    """ABPG_expo, ABDA and FW_alg_div_step + lmo_simplex with the calls of
    frank_wolfe_wtih_rs/ex_Dopt_design.py:12-21 (housing) and on D_opt_design(80,200,seed 10)."""
    gd = golden("next_rows")
    f, h, L, x0 = acc.D_opt_libsvm(_os.path.join(_DATA, "housing.txt"))
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=300, linesearch=True, ls_ratio=2, verbose=False)
    _close(F, gd["h_bpg_F"], 1e-9); _close(Ls, gd["h_bpg_Ls"], 1e-12)
    x, F, Ls, T = acc.FW_alg_div_step(f, h, L, x0, lmo=acc.lmo_simplex(), maxitrs=300, gamma=2.0, ls_ratio=2,
                                      verbose=False)
    k = _agree_prefix(Ls, gd["h_fwdiv_Ls"], 1e-12)
    assert k >= 150, k
    _close(F[:k], gd["h_fwdiv_F"][:k], 1e-9)
    x, F, Gamma, G, T = acc.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True, Gmargin=100, verbose=False)
    k = _agree_prefix(Gamma, gd["h_expo_Gamma"], 1e-12)
    assert k >= 100, k
    _close(F[:100], gd["h_expo_F"][:100], 1e-9)
    f, h, L, x0 = acc.D_opt_design(80, 200, randseed=10)
    x, F, Gamma, G, T = acc.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True, verbose=False)
    k = _agree_prefix(Gamma, gd["r_expo_Gamma"], 1e-12)
    assert k >= 200, k
    _close(F[:200], gd["r_expo_F"][:200], 1e-9)
    assert len(F) == len(Gamma) == len(G) == len(T)
    x, F, Gamma, G, T = acc.ABPG_expo(f, h, L, x0, gamma0=2.5, maxitrs=200, theta_eq=False, checkdiv=True,
                                      Gmargin=5, restart=True, verbose=False)
    k = _agree_prefix(Gamma, gd["r_expo2_Gamma"], 1e-12)
    assert k >= 40, k
    _close(F[:40], gd["r_expo2_F"][:40], 1e-9)
    x, F, G, T = acc.ABDA(f, h, L, x0, gamma=2, maxitrs=300, theta_eq=True, verbose=False)
    _close(F, gd["r_abda_F"], 1e-9)
    assert np.max(np.abs(x - gd["r_abda_x"])) < 1e-9
    x, F, Ls, T = acc.FW_alg_div_step(f, h, L, x0, lmo=acc.lmo_simplex(), maxitrs=300, gamma=2.0, ls_ratio=2,
                                      verbose=False)
    k = _agree_prefix(Ls, gd["r_fwdiv_Ls"], 1e-12)
    assert k >= 150, k
    _close(F[:k], gd["r_fwdiv_F"][:k], 1e-9)
    with pytest.raises(ValueError):
        acc.FW_alg_div_step(f, h, -1.0, x0, 5, 2.0, acc.lmo_simplex(), verbose=False)


def test_lmo_simplex_and_argminmax(acc):
    from accbpg_and_fw_amd.functions import vec_argminmax, vec_div_scalar
    rng = np.random.RandomState(8)
    for n in [5, 1000, 70001]:
        g = rng.randn(n)
        g[rng.randint(n)] = g.min()                 # a tie: the first index must win
        s = acc.lmo_simplex(2.0)(g)
        ref = np.zeros(n) + 1e-15
        ref[np.where(g == g.min())[0][0]] = 2.0
        np.testing.assert_array_equal(s, ref)
        imin, imax, vmin, vmax = vec_argminmax(dev(g))
        assert imin == np.argmin(g) and imax == np.argmax(g) and vmin == g.min() and vmax == g.max()
        np.testing.assert_array_equal(vec_div_scalar(dev(g), 3.7).cpu().numpy(), g / 3.7)


def test_kyinit_gpu(acc):
    """Kumar-Yildirim start: identical support and weights to the reference for the same RNG state,
    and FW-away started from it follows the reference's trajectory."""
    gd = golden("next_rows")
    f, h, L, x0 = acc.D_opt_design(30, 1000, randseed=4)
    np.random.seed(99)
    xky = acc.D_opt_KYinit(f.H)
    np.testing.assert_array_equal(xky, gd["ky_x"])
    np.random.seed(99)
    np.testing.assert_array_equal(acc.D_opt_KYinit(f), gd["ky_x"])        # also accepts the objective
    np.testing.assert_array_equal(acc.D_opt_KYinit(np.zeros((30, 60))), np.ones(60) / 60)
    xs, F, SP, SN, T = acc.D_opt_FW_away(f.H, xky, 1e-8, 2000, verbose=False)
    k = min(len(F), len(gd["ky_away_F"]))
    assert abs(len(F) - len(gd["ky_away_F"])) <= 2
    _close(F[:k], gd["ky_away_F"][:k], 1e-8)
    assert np.max(np.abs(xs - gd["ky_away_x"])) < 1e-8


# ------------------------------------------------------------------ Gram-matrix reuse through linearity
@pytest.mark.parametrize("solver", ["abpg", "abpg_gain", "abpg_restart"])
def test_linear_gram_reuse_matches_direct(acc, solver):
    """Opt-in extension: V diag(x) V^T is linear in x, so the accelerated solvers can combine resident
    Gram matrices instead of re-forming them.  Same trajectory as direct evaluation to rounding, with
    one O(m^2 n) product per pass instead of two or three."""
    gd = golden("traces_80x200")
    f, h, L, x0 = acc.D_opt_design(80, 200, randseed=10)
    if solver == "abpg":
        run = lambda ff: acc.ABPG(ff, h, L, x0, gamma=2.0, maxitrs=400, theta_eq=True, verbose=False)
        ref_x, ref_F = None, gd["abpg_F"][:400]
    elif solver == "abpg_restart":
        run = lambda ff: acc.ABPG(ff, h, L, x0, gamma=2.0, maxitrs=400, theta_eq=True, restart=True, verbose=False)
        ref_x, ref_F = None, gd["abpgrs_F"][:400]
    else:
        run = lambda ff: acc.ABPG_gain(ff, h, L, x0, gamma=2, maxitrs=400, G0=0.1, theta_eq=True, verbose=False)
        ref_x, ref_F = None, gd["gain_F"][:400]
    base = run(f)
    f2 = acc.DOptimalObj(f.H).linear_gram(True, refresh=25)
    lin = run(f2)
    assert np.max(np.abs(lin[0] - base[0])) < 1e-11
    np.testing.assert_allclose(lin[1], base[1], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(lin[1], ref_F, rtol=1e-9, atol=1e-9)          # and the reference's trace
    evals = f2.calls["value"] + f2.calls["grad"]
    assert f2.gram_launches + f2.gram_combos <= evals + 450
    assert f2.gram_launches < 0.62 * evals, (f2.gram_launches, f2.gram_combos, evals)
    # plain evaluations still work on an object with reuse enabled
    fx, g = f2.func_grad(x0, 2)
    fb, gb = f.func_grad(x0, 2)
    assert fx == fb
    np.testing.assert_array_equal(g, gb)


def test_trajectories_512x8192(acc):
    """1000-iteration l_inf parity at the BASELINE config-4 instance size D_opt_design(512,8192)
    (traces from the real reference, /tmp generation script of oracle/gen_golden.py style)."""
    gd = golden("traces_512x8192")
    f, h, L, x0 = acc.D_opt_design(512, 8192, randseed=10)
    x, F, Ls, T = acc.BPG(f, h, L, x0, maxitrs=1000, linesearch=True, verbose=False)
    assert np.max(np.abs(x - gd["bpgls_x"])) < 1e-9
    _close(F, gd["bpgls_F"], 1e-9); _close(Ls, gd["bpgls_Ls"], 1e-12)
    x, F, G, T = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=1000, theta_eq=True, verbose=False)
    assert np.max(np.abs(x - gd["abpg_x"])) < 1e-9
    _close(F, gd["abpg_F"], 1e-9)
    x, F, Gain, Gdiv, Gavg, T = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=300, verbose=False)
    k = _agree_prefix(Gain, gd["gain_Gain"], 1e-12)
    assert k >= 25, k
    _close(F[:25], gd["gain_F"][:25], 1e-9)


def test_overlapped_value_evaluation_is_identical(acc):
    """Opt-in: F[k] = f(x) on a side stream beside func_grad(y).  Same kernels on the same data, so
    the whole run is bitwise identical to the sequential one."""
    f, h, L, x0 = acc.D_opt_design(300, 3000, randseed=21)
    a = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    b = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=60, theta_eq=True, restart=True, verbose=False)
    f.overlap_values(True)
    a2 = acc.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=60, verbose=False)
    b2 = acc.ABPG(f, h, L, x0, gamma=2.0, maxitrs=60, theta_eq=True, restart=True, verbose=False)
    f.overlap_values(False)
    for u, v in [(a, a2), (b, b2)]:
        for p, q in zip(u[:-1], v[:-1]):
            np.testing.assert_array_equal(p, q)
    # an error inside the overlapped evaluation surfaces at the wait
    f.overlap_values(True)
    bad = torch.from_numpy(-x0).cuda()
    with pytest.raises(AssertionError):
        f.value_wait(f.value_async(bad))
    f.overlap_values(False)
