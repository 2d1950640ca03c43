"""Deterministic synthetic inputs shared by tests and bench.py (SURVEY 8(d) generator)."""
import numpy as np


def banded_triplets(n, h, complex_=False, shift=0.0, c0=0, c1=None):
    """H_ii = -1 + 2*((i*7919) mod 1000)/1000 (+shift), H_ij = -0.25*exp(-0.05|i-j|)/|i-j| for
    0<|i-j|<=h (1-based i); complex: times exp(i*0.1*(i-j)).  Returns NTPoly triplets
    (col,row,val), 1-based, sorted by column then row; only columns [c0, c1) (0-based) if given."""
    c1 = n if c1 is None else c1
    i = np.arange(1, n + 1, dtype=np.int64)
    cols, rows, vals = [], [], []
    diag = -1.0 + 2.0 * ((i * 7919) % 1000) / 1000.0 + shift
    offs = np.arange(-h, h + 1)
    # column-major construction: for every column j, rows j+o in range
    col = np.repeat(i[c0:c1], len(offs))
    row = col + np.tile(offs, c1 - c0)
    ok = (row >= 1) & (row <= n)
    col, row = col[ok], row[ok]
    dist = np.abs(row - col)
    with np.errstate(divide="ignore", invalid="ignore"):
        val = np.where(dist == 0, diag[col - 1], -0.25 * np.exp(-0.05 * dist) / np.maximum(dist, 1))
    if complex_:
        val = val * np.exp(1j * 0.1 * (row - col))
        val = np.where(dist == 0, diag[col - 1] + 0j, val)
    return col.astype(np.int32), row.astype(np.int32), val
