"""CPU tests of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/accbpg_hip.h declares, and the Python mirror keeps the reference's names and
signatures.  No compute call is made (no GPU here)."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "accbpg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(accbpg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from accbpg_and_fw_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), "missing export " + name
    assert sorted(_lib.EXPORTS) == names, "ctypes table and header disagree"
    assert lib.accbpg_abi_version() == 3


def test_missing_library_fails_loudly(monkeypatch):
    from accbpg_and_fw_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libaccbpg_hip.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_error_code_mapping():
    from accbpg_and_fw_amd import _lib
    _lib.load()
    with pytest.raises(AssertionError):
        _lib.check(_lib.ERR_ASSERT, "x", "DOptimalObj: x needs to be nonnegative")
    with pytest.raises(ValueError, match="HXHT is singular or not positive definite"):
        _lib.check(_lib.ERR_NOT_PD, "x")
    with pytest.raises(RuntimeError):
        _lib.check(_lib.ERR_HIP, "x")


def test_signatures_match_reference():
    """Argument names and defaults of accbpg/algorithms.py:11-12,94-95,295-297 and
    accbpg/D_opt_alg.py:9,91."""
    import accbpg_and_fw_amd as acc

    def sig(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values()]
    E = inspect.Parameter.empty
    assert sig(acc.BPG) == [("f", E), ("h", E), ("L", E), ("x0", E), ("maxitrs", E), ("epsilon", 1e-14),
                            ("linesearch", True), ("ls_ratio", 1.2), ("verbose", True), ("verbskip", 1)]
    assert sig(acc.ABPG) == [("f", E), ("h", E), ("L", E), ("x0", E), ("gamma", E), ("maxitrs", E),
                             ("epsilon", 1e-14), ("theta_eq", False), ("restart", False),
                             ("restart_rule", 'g'), ("verbose", True), ("verbskip", 1)]
    assert sig(acc.ABPG_gain) == [("f", E), ("h", E), ("L", E), ("x0", E), ("gamma", E), ("maxitrs", E),
                                  ("epsilon", 1e-14), ("G0", 1), ("ls_inc", 1.2), ("ls_dec", 1.2),
                                  ("theta_eq", True), ("checkdiv", False), ("restart", False),
                                  ("restart_rule", 'g'), ("verbose", True), ("verbskip", 1)]
    assert sig(acc.D_opt_FW) == [("V", E), ("x0", E), ("eps", E), ("maxitrs", E), ("verbose", True),
                                 ("verbskip", 1)]
    assert sig(acc.D_opt_FW_away)[:6] == sig(acc.D_opt_FW)
    assert sig(acc.D_opt_design) == [("m", E), ("n", E), ("randseed", -1)]
    assert sig(acc.solve_theta) == [("theta", E), ("gamma", E), ("gainratio", 1)]
    for name in ["func_grad", "gradient", "__call__"]:
        assert hasattr(acc.DOptimalObj, name)
    for name in ["extra_Psi", "gradient", "divergence", "prox_map", "div_prox_map"]:
        assert hasattr(acc.BurgEntropySimplex, name)


def test_solve_theta_matches_oracle():
    import accbpg_and_fw_amd as acc
    from oracle import np_oracle as O
    for th, ga, gr in [(1.0, 2.0, 1.0), (0.3, 2.0, 0.8), (0.05, 1.5, 1.2)]:
        assert acc.solve_theta(th, ga, gr) == O.solve_theta(th, ga, gr)


def test_product_never_imports_oracle():
    """The product path must not route through the CPU oracle."""
    pkg = os.path.join(ROOT, "accbpg_and_fw_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "np_oracle" not in text and "from oracle" not in text and "import oracle" not in text, fn


def test_libsvm_loader_cpu():
    """load_libsvm_file (accbpg/utils.py:22-95) is host code: CSR result, index-base detection,
    comments, error cases; matrices equal the oracle's and the reference's checksums."""
    import numpy as np
    import accbpg_and_fw_amd as acc
    from oracle import np_oracle as O
    gd = np.load(os.path.join(ROOT, "tests", "golden", "next_rows.npz"))
    for name in ["housing", "bodyfat", "mpg", "abalone"]:
        path = os.path.join(ROOT, "tests", "golden", "data", name + ".txt")
        X, y = acc.load_libsvm_file(path)
        Xo, yo = O.load_libsvm_file(path)
        np.testing.assert_array_equal(X.toarray(), Xo)
        np.testing.assert_array_equal(y, yo)
        H = X.T.toarray('C') if X.shape[0] > X.shape[1] else X.toarray('C')
        assert tuple(gd["libsvm_%s_shape" % name]) == H.shape
        np.testing.assert_allclose([H.sum(), np.abs(H).max(), (H ** 2).sum()], gd["libsvm_%s_checksum" % name],
                                   rtol=1e-14)


def test_libsvm_loader_edge_cases(tmp_path):
    import numpy as np
    import accbpg_and_fw_amd as acc
    p = tmp_path / "a.txt"
    p.write_text("1 1:0.5 3:2 # comment\n\n# whole-line comment\n-1 2:1.5\n")
    X, y = acc.load_libsvm_file(str(p))
    np.testing.assert_array_equal(X.toarray(), [[0.5, 0, 2.0], [0, 1.5, 0]])
    np.testing.assert_array_equal(y, [1.0, -1.0])
    p.write_text("1 0:0.5 2:2\n")                       # zero-based file: no shift
    X, y = acc.load_libsvm_file(str(p))
    assert X.shape == (1, 3) and X[0, 0] == 0.5
    X, y = acc.load_libsvm_file(str(p), n_features=5)
    assert X.shape == (1, 5)
    p.write_text("1 2:1 2:3\n")
    with pytest.raises(ValueError):
        acc.load_libsvm_file(str(p))
    p.write_text("1 -1:1\n")
    with pytest.raises(ValueError):
        acc.load_libsvm_file(str(p))


def test_libsvm_loader_more_edges(tmp_path, capsys):
    """Index-base switches, the n_features warning, compressed input, and which of two defects in one
    file is reported (the first in file order, as accbpg/utils.py:22-95 meets them)."""
    import gzip
    import numpy as np
    import accbpg_and_fw_amd as acc
    p = tmp_path / "b.txt"
    p.write_text("1 0:1 2:3\n")
    with pytest.raises(ValueError, match="Invalid index 0"):
        acc.load_libsvm_file(str(p), zero_based=False)
    X, _ = acc.load_libsvm_file(str(p), zero_based=True)
    assert X.shape == (1, 3)
    p.write_text("2 1:1 4:3\n")
    X, _ = acc.load_libsvm_file(str(p), n_features=2)            # one-based file, needs 4 columns
    assert X.shape == (1, 4) and "n_features increased" in capsys.readouterr().out
    X, _ = acc.load_libsvm_file(str(p), zero_based=True)         # forced zero-based: 5 columns
    assert X.shape == (1, 5)
    p.write_text("1 3:1 2:1\n1 -4:2\n")
    with pytest.raises(ValueError, match="sorted and unique"):
        acc.load_libsvm_file(str(p))
    p.write_text("1 -4:2\n1 3:1 2:1\n")
    with pytest.raises(ValueError, match="Invalid index -4"):
        acc.load_libsvm_file(str(p))
    q = tmp_path / "c.txt.gz"
    with gzip.open(q, "wt") as fh:
        fh.write("3 1:0.25 2:-1e3\n")
    X, y = acc.load_libsvm_file(str(q), dtype=np.float32)
    assert X.dtype == np.float32 and X.toarray().tolist() == [[0.25, -1000.0]] and y.tolist() == [3.0]
