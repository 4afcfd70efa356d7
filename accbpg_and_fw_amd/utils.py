"""LIBSVM text loader (accbpg/utils.py:22-95): returns (scipy.sparse.csr_matrix, labels) like the
reference.  Host-side parsing; the matrix goes to the GPU when an objective is built on it."""
from __future__ import annotations

import os.path

import numpy as np
import scipy.sparse as sparse


def _open_text(filename):
    ext = os.path.splitext(filename)[1]
    if ext == '.gz':
        import gzip
        return gzip.open(filename, 'rt')
    if ext == '.bz2':
        import bz2
        return bz2.open(filename, 'rt')
    return open(filename, 'r')


def load_libsvm_file(filename, dtype=np.float64, n_features=None, zero_based="auto"):
    """Each line: ``label idx:value idx:value ...`` with '#' comments; indices must be sorted and
    unique within a line; zero_based="auto" shifts the indices down when the smallest one is > 0;
    n_features is raised (with the reference's warning) if the data need more columns."""
    labels, values, cols, starts = [], [], [], []
    with _open_text(filename) as fh:
        for line in fh:
            hash_at = line.find('#')
            if hash_at >= 0:
                line = line[:hash_at]
            tokens = line.split()
            if not tokens:
                continue
            labels.append(float(tokens[0]))
            starts.append(len(values))
            last = -1
            for tok in tokens[1:]:
                idx_txt, val_txt = tok.split(':', 1)
                idx = int(idx_txt)
                if idx < 0 or (not zero_based and idx == 0):
                    raise ValueError("Invalid index {0:d} in LibSVM data file.".format(idx))
                if idx <= last:
                    raise ValueError("Feature indices in LibSVM data file"
                                     "should be sorted and unique.")
                cols.append(idx)
                values.append(dtype(val_txt))
                last = idx
    starts.append(len(values))
    values = np.array(values)
    starts = np.array(starts)
    cols = np.array(cols)
    if zero_based is False or (zero_based == "auto" and cols.min() > 0):
        cols -= 1
    if n_features is None:
        n_features = cols.max() + 1
    elif n_features < cols.max() + 1:
        n_features = cols.max() + 1
        print("Warning: n_features increased to match data.")
    X = sparse.csr_matrix((values, cols, starts), (starts.shape[0] - 1, n_features))
    X.sort_indices()
    return X, np.array(labels)
