"""CPU tests: the NumPy oracle against golden vectors produced by the real
reference (oracle/gen_golden.py) and against the notebook rows that the
reference stores as its only regression data."""
import numpy as np
import pytest

from conftest import golden, gaussian_design
from oracle import np_oracle as O


@pytest.mark.parametrize("tag", ["80x200", "128x1024", "200x2000"])
def test_percall_matches_reference(tag):
    gd = golden("percall_" + tag)
    m, n, seed = int(gd["m"]), int(gd["n"]), int(gd["seed"])
    f, h, L, x0 = O.D_opt_design(m, n, randseed=seed)
    fx, g = f.func_grad(gd["x"], 2)
    assert fx == pytest.approx(float(gd["f"]), rel=0, abs=1e-12)
    np.testing.assert_allclose(g, gd["g"], rtol=1e-13, atol=0)
    f0, g0 = f.func_grad(x0, 2)
    assert f0 == pytest.approx(float(gd["f0"]), rel=0, abs=1e-12)
    np.testing.assert_allclose(g0, gd["g0"], rtol=1e-13)
    assert f(gd["x"]) == fx
    for idx in range(3):
        z = h.div_prox_map(gd["y"], gd["g"], float(gd["prox_L%d" % idx]))
        np.testing.assert_array_equal(z, gd["prox_x%d" % idx])   # same arithmetic -> bitwise
    gg = gd["g"] - gd["g"].min() + 0.5
    np.testing.assert_array_equal(h.prox_map(gg, 2.0), gd["prox_raw"])
    assert h.divergence(gd["x"], gd["y"]) == float(gd["div_xy"])
    assert h.divergence(gd["y"], gd["x"]) == float(gd["div_yx"])


def _check(a, b, tol):
    assert a.shape == b.shape
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def _prefix(a, b, tol):
    n = min(len(a), len(b))
    bad = np.nonzero(np.abs(a[:n] - b[:n]) > tol * (1 + np.abs(b[:n])))[0]
    return n if bad.size == 0 else int(bad[0])


# Line-search variants take discrete accept/reject decisions; once a run has converged those
# decisions sit at the rounding floor and the REFERENCE ITSELF is not reproducible across BLAS
# thread counts (measured: ABPG_gain(G0=0.1) first differs at k=756 on (80,200), at k=36..79 on
# (256,4096); BPG, BPG-LS and ABPG are stable to 1e-17).  So the line-search runs are compared on
# the decision-stable prefix and at objective level afterwards.
@pytest.mark.parametrize("tag", ["80x200", "80x120"])
def test_solver_traces_match_reference(tag):
    gd = golden("traces_" + tag)
    m, n, seed, iters = int(gd["m"]), int(gd["n"]), int(gd["seed"]), int(gd["iters"])
    f, h, L, x0 = O.D_opt_design(m, n, randseed=seed)
    tol = 1e-12
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=iters, linesearch=False)
    _check(x, gd["bpg_x"], tol); _check(F, gd["bpg_F"], tol); _check(Ls, gd["bpg_Ls"], tol)
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=iters, linesearch=True)
    _check(x, gd["bpgls_x"], 1e-10); _check(F, gd["bpgls_F"], tol); _check(Ls, gd["bpgls_Ls"], tol)
    x, F, G, T = O.ABPG(f, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True)
    _check(x, gd["abpg_x"], tol); _check(F, gd["abpg_F"], tol); _check(G, gd["abpg_G"], 1e-9)
    x, F, G, T = O.ABPG(f, h, L, x0, gamma=2.0, maxitrs=iters, theta_eq=True, restart=True)
    _check(x, gd["abpgrs_x"], tol); _check(F, gd["abpgrs_F"], tol)
    x, F, G, T = O.ABPG(f, h, L, x0, gamma=1.5, maxitrs=iters, theta_eq=False)
    _check(x, gd["abpgk_x"], tol); _check(F, gd["abpgk_F"], tol)
    stable = 500 if tag == "80x200" else 20
    for key, kw in [("gain", dict(G0=0.1, theta_eq=True)),
                    ("gainrs", dict(G0=0.1, theta_eq=True, restart=True)),
                    ("gaindef", dict()),
                    ("gainopt", dict(G0=0.1, ls_inc=1.5, ls_dec=1.1, theta_eq=False, checkdiv=True,
                                     restart=True, restart_rule='f'))]:
        x, F, Gain, Gdiv, Gavg, T = O.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=iters, **kw)
        k = _prefix(Gain, gd[key + "_Gain"], 1e-12)
        assert k >= min(stable, len(gd[key + "_Gain"])), (key, k)
        _check(F[:k], gd[key + "_F"][:k], 1e-11)
        _check(Gavg[:k], gd[key + "_Gavg"][:k], 1e-11)
        assert abs(F[-1] - gd[key + "_F"][-1]) < 1e-9


@pytest.mark.parametrize("tag", ["30x1000", "64x512", "256x4096"])
def test_fw_traces_match_reference(tag):
    gd = golden("fw_" + tag)
    m, n, seed, iters = int(gd["m"]), int(gd["n"]), int(gd["seed"]), int(gd["iters"])
    V = gaussian_design(m, n, seed)
    x0 = np.ones(n) / n
    x, F, SP, SN, T = O.D_opt_FW(V, x0, float(gd["eps"]), iters)
    _check(x, gd["fw_x"], 1e-12); _check(F, gd["fw_F"], 1e-11)
    _check(SP, gd["fw_SP"], 1e-11); _check(SN, gd["fw_SN"], 1e-11)
    x, F, SP, SN, T = O.D_opt_FW_away(V, x0, float(gd["eps"]), iters)
    _check(x, gd["away_x"], 1e-12); _check(F, gd["away_F"], 1e-11)
    _check(SP, gd["away_SP"], 1e-11); _check(SN, gd["away_SN"], 1e-11)


def test_housing_rng_free_instance():
    gd = golden("housing")
    V = gd["V"]
    assert V.shape == (13, 506)
    n = V.shape[1]
    f, h = O.DOptOracle(V), O.BurgSimplexOracle()
    x0 = np.ones(n) / n
    # F(x0) for the LIBSVM housing file, SURVEY.md 8(c) / ipynb/ex_Dopt_LIBSVM.ipynb:191
    assert "%.3e" % f(x0) == "-4.137e+01"
    x, F, Ls, T = O.BPG(f, h, 1.0, x0, maxitrs=1001, linesearch=False)
    _check(F, gd["bpg_F"], 1e-11); _check(x, gd["bpg_x"], 1e-11)
    x, F, Gain, Gdiv, Gavg, T = O.ABPG_gain(f, h, 1.0, x0, gamma=2, maxitrs=1001, G0=0.1,
                                             ls_inc=1.5, ls_dec=1.5)
    k = _prefix(Gain, gd["gain_Gain"], 1e-12)
    assert k >= 300
    _check(F[:k], gd["gain_F"][:k], 1e-10)
    x, F, SP, SN, T = O.D_opt_FW_away(V, x0, 1e-8, 3000)
    _check(x, gd["away_x"], 1e-10); _check(F, gd["away_F"], 1e-9)


def test_notebook_rows_ex_Dopt_random():
    """Rows stored in ipynb/ex_Dopt_random.ipynb for D_opt_design(80,200,randseed=10), with the
    calls that notebook makes: BPG(linesearch=False) :73,:82; BPG(linesearch=True) :243-244;
    ABPG(gamma=2, theta_eq=True) :113; ABPG_gain(gamma=2, G0=0.1, theta_eq=True) :282-283;
    second instance D_opt_design(80,120,randseed=10) :398."""
    f, h, L, x0 = O.D_opt_design(80, 200, randseed=10)
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=1000, linesearch=False)
    assert "%.3e" % F[0] == "1.910e+01" and "%.3e" % F[900] == "1.759e+01"
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=200, linesearch=True)
    assert "%.3e" % Ls[0] == "8.333e-01" and "%.3e" % Ls[100] == "1.938e-01"
    x, F, G, T = O.ABPG(f, h, L, x0, gamma=2.0, maxitrs=200, theta_eq=True)
    assert "%.3e" % G[100] == "5.529e-01"
    x, F, Gain, Gdiv, Gavg, T = O.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=200, G0=0.1, theta_eq=True)
    assert "%.3e" % Gain[0] == "2.488e-01" and "%.3e" % Gavg[0] == "4.988e-02"
    assert "%.3e" % Gain[100] == "2.986e-01" and "%.3e" % Gdiv[100] == "7.091e-01"
    f2, h2, L2, x02 = O.D_opt_design(80, 120, randseed=10)
    assert "%.3e" % f2(x02) == "3.764e+01"
    x, F, Gain, Gdiv, Gavg, T = O.ABPG_gain(f2, h2, L2, x02, gamma=2, maxitrs=50, G0=0.1, theta_eq=True)
    assert "%.3e" % Gain[0] == "5.160e-01" and "%.3e" % Gavg[0] == "7.183e-02"


def test_solve_theta_and_asserts():
    t = O.solve_theta(0.5, 2.0, 1.0)
    assert abs((1 - t) / t ** 2 - 1 / 0.25) < 1e-4
    f, h, L, x0 = O.D_opt_design(8, 20, randseed=3)
    with pytest.raises(AssertionError):
        f.func_grad(-x0)
    with pytest.raises(AssertionError):
        f.func_grad(x0[:-1])
    with pytest.raises(ValueError):
        z = np.zeros(20); z[:3] = 1.0 / 3      # rank-deficient Gram matrix: slogdet sign 0
        f.func_grad(z)
    with pytest.raises(AssertionError):
        h.div_prox_map(x0, x0, -1.0)
    with pytest.raises(AssertionError):
        h.divergence(x0, 0 * x0)


# ------------------------------------------------------------------ SURVEY 8(f) rows 1-3
DATA = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden", "data")


def test_libsvm_instances_match_reference():
    """F(x0) of the four vendored LIBSVM files (-4.136876e+01, -3.474969e+01,
    -3.416310e+01, 3.978055e+01) and the parsed matrices' checksums."""
    gd = golden("next_rows")
    expect = {"housing": "-4.136876e+01", "bodyfat": "-3.474969e+01", "mpg": "-3.416310e+01", "abalone": "3.978055e+01"}
    for name, txt in expect.items():
        f, h, L, x0 = O.D_opt_libsvm(__import__("os").path.join(DATA, name + ".txt"))
        assert tuple(gd["libsvm_%s_shape" % name]) == f.H.shape
        np.testing.assert_allclose([f.H.sum(), np.abs(f.H).max(), (f.H ** 2).sum()],
                                   gd["libsvm_%s_checksum" % name], rtol=1e-14)
        f0 = f(x0)
        assert f0 == pytest.approx(float(gd["libsvm_%s_f0" % name]), rel=1e-13)
        assert "%.6e" % f0 == txt


def test_next_row_solvers_match_reference():
    gd = golden("next_rows")
    f, h, L, x0 = O.D_opt_libsvm(__import__("os").path.join(DATA, "housing.txt"))
    x, F, Ls, T = O.FW_alg_div_step(f, h, L, x0, lmo=O.lmo_simplex(), maxitrs=300, gamma=2.0, ls_ratio=2)
    _check(F, gd["h_fwdiv_F"], 1e-11); _check(Ls, gd["h_fwdiv_Ls"], 1e-12); _check(x, gd["h_fwdiv_x"], 1e-11)
    x, F, Gamma, G, T = O.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True, Gmargin=100)
    k = _prefix(Gamma, gd["h_expo_Gamma"], 1e-12)
    assert k >= 100
    _check(F[:k], gd["h_expo_F"][:k], 1e-9)
    f, h, L, x0 = O.D_opt_design(80, 200, randseed=10)
    x, F, Gamma, G, T = O.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=300, theta_eq=True)
    k = _prefix(Gamma, gd["r_expo_Gamma"], 1e-12)
    assert k >= 200
    _check(F[:k], gd["r_expo_F"][:k], 1e-10)
    x, F, Gamma, G, T = O.ABPG_expo(f, h, L, x0, gamma0=2.5, maxitrs=200, theta_eq=False, checkdiv=True,
                                    Gmargin=5, restart=True)
    k = _prefix(Gamma, gd["r_expo2_Gamma"], 1e-12)        # decision-stable prefix (see the note above)
    assert k >= 40, k
    _check(F[:40], gd["r_expo2_F"][:40], 1e-10)
    _check(F[:k], gd["r_expo2_F"][:k], 1e-4)
    x, F, G, T = O.ABDA(f, h, L, x0, gamma=2, maxitrs=300, theta_eq=True)
    _check(F, gd["r_abda_F"], 1e-11); _check(x, gd["r_abda_x"], 1e-11)
    x, F, Ls, T = O.FW_alg_div_step(f, h, L, x0, lmo=O.lmo_simplex(), maxitrs=300, gamma=2.0, ls_ratio=2)
    _check(F, gd["r_fwdiv_F"], 1e-11); _check(Ls, gd["r_fwdiv_Ls"], 1e-12)


def test_kyinit_matches_reference():
    gd = golden("next_rows")
    f, h, L, x0 = O.D_opt_design(30, 1000, randseed=4)
    np.random.seed(99)
    xky = O.D_opt_KYinit(f.H)
    np.testing.assert_array_equal(xky, gd["ky_x"])
    assert np.count_nonzero(xky) <= 60 and abs(xky.sum() - 1) < 1e-14
    # n <= 2m falls back to the uniform point (applications.py:67-68)
    np.testing.assert_array_equal(O.D_opt_KYinit(np.zeros((30, 60))), np.ones(60) / 60)


# ---------------------------------------------------------------- Poisson + Burg L1/L2
_POISSON = {"l1": (O.Poisson_regrL1, 200, 100, 0.0001, 0), "l2": (O.Poisson_regrL2, 100, 1000, 0.001, 0.001),
            "l1r": (O.Poisson_regrL1, 300, 2000, 0.001, 0.01)}


def _poisson_instance(tag):
    """Factory instance with b and L pinned to the golden run (b = A x + noise goes through the host BLAS,
    which may differ by an ulp between machines)."""
    fac, m, n, noise, lam = _POISSON[tag]
    f, h, L, x0 = fac(m, n, noise=noise, lamda=lam, randseed=1)
    gd = golden("poisson")
    np.testing.assert_allclose(f.b, gd[tag + "_b"], rtol=1e-13)
    assert L == pytest.approx(float(gd[tag + "_L"]), rel=1e-13)
    return O.PoissonOracle(f.A, gd[tag + "_b"]), h, float(gd[tag + "_L"]), x0


@pytest.mark.parametrize("tag", ["l1", "l2", "l1r"])
def test_poisson_percall_matches_reference(tag):
    gd = golden("poisson")
    f, h, L, x0 = _poisson_instance(tag)
    A = f.A
    np.testing.assert_allclose([A.sum(), np.abs(A).max(), (A ** 2).sum()], gd[tag + "_A_checksum"], rtol=1e-14)
    np.testing.assert_array_equal(x0, gd[tag + "_x0"])
    x, y = gd[tag + "_x"], gd[tag + "_y"]
    fx, g = f.func_grad(x, 2)
    assert fx == pytest.approx(float(gd[tag + "_f"]), rel=1e-12, abs=1e-13)
    np.testing.assert_allclose(g, gd[tag + "_g"], rtol=1e-11, atol=1e-13)
    assert f(x0) == pytest.approx(float(gd[tag + "_f0"]), rel=1e-12, abs=1e-13)
    np.testing.assert_allclose(f.gradient(x0), gd[tag + "_g0"], rtol=1e-11, atol=1e-13)
    assert h.extra_Psi(x) == pytest.approx(float(gd[tag + "_psi"]), rel=1e-15)
    for idx in range(3):
        z = h.div_prox_map(y, gd[tag + "_g"], float(gd["%s_prox_L%d" % (tag, idx)]))
        np.testing.assert_array_equal(z, gd["%s_prox_x%d" % (tag, idx)])      # elementwise -> bitwise
    np.testing.assert_array_equal(h.prox_map(np.abs(gd[tag + "_g"]) + 0.5, 2.0), gd[tag + "_prox_raw"])
    assert h.divergence(x, y) == float(gd[tag + "_div_xy"])


def test_poisson_l1_solver_traces_match_reference():
    gd = golden("poisson")
    f, h, L, x0 = _poisson_instance("l1")
    N, tol = 2000, 1e-10
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=N, linesearch=False)
    _check(x, gd["l1_bpg_x"], tol); _check(F, gd["l1_bpg_F"], tol)
    for gam, key in [(1.0, "g10"), (1.5, "g15"), (2.0, "g20")]:
        x, F, G, T = O.ABPG(f, h, L, x0, gamma=gam, maxitrs=N, theta_eq=True)
        _check(x, gd["l1_abpg_%s_x" % key], 1e-8); _check(F, gd["l1_abpg_%s_F" % key], tol)
    x, F, G, T = O.ABDA(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=True)
    _check(x, gd["l1_abda_x"], 1e-8); _check(F, gd["l1_abda_F"], tol)
    # line-search runs: decision-stable prefix, then objective-level agreement (see the note above)
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=N, linesearch=True)
    p = _prefix(Ls, gd["l1_bpgls_Ls"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l1_bpgls_F"][:p], tol)
    assert abs(F[-1] - gd["l1_bpgls_F"][-1]) < 1e-6
    x, F, Gamma, G, T = O.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=3)
    p = _prefix(Gamma, gd["l1_expo_Gamma"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l1_expo_F"][:p], tol)
    x, F, G, Gdiv, Gavg, T = O.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, theta_eq=False)
    p = _prefix(G, gd["l1_gain_G"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l1_gain_F"][:p], tol)
    assert abs(F[-1] - gd["l1_gain_F"][-1]) < 1e-6


def test_poisson_l2_solver_traces_match_reference():
    gd = golden("poisson")
    f, h, L, x0 = _poisson_instance("l2")
    N, tol = 2000, 1e-10
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=N, linesearch=False)
    _check(x, gd["l2_bpg_x"], tol); _check(F, gd["l2_bpg_F"], tol)
    x, F, G, T = O.ABPG(f, h, L, x0, gamma=2.0, maxitrs=N, theta_eq=False)
    _check(x, gd["l2_abpg_x"], 1e-8); _check(F, gd["l2_abpg_F"], tol)
    x, F, Ls, T = O.BPG(f, h, L, x0, maxitrs=N, linesearch=True, ls_ratio=1.5)
    p = _prefix(Ls, gd["l2_bpgls_Ls"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l2_bpgls_F"][:p], tol)
    x, F, Gamma, G, T = O.ABPG_expo(f, h, L, x0, gamma0=3, maxitrs=N, theta_eq=False, Gmargin=1)
    p = _prefix(Gamma, gd["l2_expo_Gamma"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l2_expo_F"][:p], tol)
    x, F, G, Gdiv, Gavg, T = O.ABPG_gain(f, h, L, x0, gamma=2, maxitrs=N, G0=0.1, ls_inc=1.5, ls_dec=1.5,
                                         theta_eq=True)
    p = _prefix(G, gd["l2_gain_G"], 1e-12)
    assert p >= 100, p
    _check(F[:p], gd["l2_gain_F"][:p], tol)
