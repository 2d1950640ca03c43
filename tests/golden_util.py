"""Loader for the golden vectors in tests/golden/*.npz (made by tests/golden/make_golden.py
from the REAL reference)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.cases = self.meta["cases"]

    def has(self, idx, name):
        return ("c%03d_%s_col" % (idx, name)) in self.z

    def tri(self, idx, name):
        """-> (rows, cols, col, row, val)"""
        pre = ("c%03d_%s" % (idx, name)) if idx is not None else name
        shp = self.z[pre + "_shape"]
        return int(shp[0]), int(shp[1]), self.z[pre + "_col"], self.z[pre + "_row"], self.z[pre + "_val"]

    def arr(self, idx, name):
        return self.z["c%03d_%s" % (idx, name)]


def same_pattern(t1, t2):
    return (len(t1[2]) == len(t2[2]) and np.array_equal(t1[2], t2[2]) and np.array_equal(t1[3], t2[3]))


def to_dense(t):
    rows, cols, c, r, v = t
    out = np.zeros((rows, cols), dtype=v.dtype if len(v) else np.float64)
    out[r - 1, c - 1] = v
    return out
