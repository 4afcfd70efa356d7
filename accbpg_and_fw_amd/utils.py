"""LIBSVM / svmlight text reader with the interface of accbpg/utils.py:22-95: returns
(scipy.sparse.csr_matrix, labels).  Host-side; the matrix goes to the GPU when an objective is built on it.

The file is tokenised once and the index / value columns are converted, validated and assembled into CSR
with array operations (no per-entry Python work), which is what the datasets this package is pointed at
(hundreds of thousands of entries) want."""
from __future__ import annotations

import bz2
import gzip
import os.path

import numpy as np
import scipy.sparse as sparse

_OPENERS = {".gz": gzip.open, ".bz2": bz2.open}


def _read_text(filename):
    opener = _OPENERS.get(os.path.splitext(filename)[1], open)
    with opener(filename, "rt") as fh:
        return fh.read()


def _first_violation(ids, row_of, zero_based):
    """Position and kind of the first malformed entry in file order, or None."""
    bad_index = ids < 0
    if not zero_based:                                   # (the string "auto" counts as true here)
        bad_index |= ids == 0
    unsorted = np.zeros(ids.shape, dtype=bool)
    unsorted[1:] = (row_of[1:] == row_of[:-1]) & (ids[1:] <= ids[:-1])
    where = np.flatnonzero(bad_index | unsorted)
    if where.size == 0:
        return None
    at = int(where[0])
    return at, ("index" if bad_index[at] else "order")


def load_libsvm_file(filename, dtype=np.float64, n_features=None, zero_based="auto"):
    """Each record is ``label idx:value idx:value ...``; '#' starts a comment; blank records are skipped.
    Indices must be increasing within a record.  zero_based="auto" treats the file as one-based when its
    smallest index is positive; n_features grows (with a printed warning) when the data need more columns."""
    records = [body.split() for body in (ln.split("#", 1)[0] for ln in _read_text(filename).splitlines())]
    records = [rec for rec in records if rec]
    labels = np.array([float(rec[0]) for rec in records])
    per_row = np.fromiter((len(rec) - 1 for rec in records), dtype=np.int64, count=len(records))
    pairs = [tok.split(":", 1) for rec in records for tok in rec[1:]]
    ids = np.array([p[0] for p in pairs]).astype(np.int64) if pairs else np.zeros(0, dtype=np.int64)
    row_of = np.repeat(np.arange(len(records)), per_row)

    problem = _first_violation(ids, row_of, zero_based)
    if problem is not None:
        at, kind = problem
        if kind == "index":
            raise ValueError("Invalid index {0:d} in LibSVM data file.".format(int(ids[at])))
        raise ValueError("Feature indices in LibSVM data file"
                         "should be sorted and unique.")
    vals = np.array([dtype(p[1]) for p in pairs])

    one_based = (zero_based is False) or (zero_based == "auto" and ids.min() > 0)
    if one_based:
        ids = ids - 1
    width = int(ids.max()) + 1
    if n_features is not None and n_features < width:
        print("Warning: n_features increased to match data.")
    if n_features is None or n_features < width:
        n_features = width
    row_ptr = np.concatenate(([0], np.cumsum(per_row)))
    X = sparse.csr_matrix((vals, ids, row_ptr), (len(records), n_features))
    X.sort_indices()
    return X, labels
