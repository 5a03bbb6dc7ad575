"""Test helpers: the shared driver set-up plus error norms."""
import numpy as np

import adapters
from setup_problem import (C_M, DT, FARADAY, PSI, load_model, make_mesh)  # noqa: F401
from setup_problem import Setup as _Setup


class Setup(_Setup):
    """Driver set-up plus the oracle view of the same problem."""

    def oracle(self):
        return adapters.oracle_problem(self)

    def oracle_fields(self):
        return adapters.oracle_fields(self)


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.abs(b).max()
    return np.abs(a - b).max() / (scale if scale > 0 else 1.0)


def csr_rel_err(A, B):
    """max |A - B| relative to max |B| (patterns may differ by explicit zeros)."""
    D = (A - B).tocoo()
    scale = np.abs(B.data).max() if B.nnz else 1.0
    return (np.abs(D.data).max() if D.nnz else 0.0) / scale
