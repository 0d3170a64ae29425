"""``from pydrake.all import *`` surface needed by /root/reference/ilqr.py:
InitializeAutoDiff (ilqr.py:254) and ExtractGradient (ilqr.py:268)."""
import numpy as np

from oracle import dual as _dual

__all__ = ["InitializeAutoDiff", "ExtractGradient"]


def InitializeAutoDiff(values):
    duals = _dual.seed(values)
    out = np.empty(len(duals), dtype=object)
    for i, q in enumerate(duals):
        out[i] = q
    return out


def ExtractGradient(vec):
    return _dual.gradient(list(vec))
