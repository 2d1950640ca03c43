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


def random_permutation(n, seed):
    """Seeded random relabelling perm[old 0-based] = new 0-based (the load-balancing permutation of
    LoadBalancerModule.F90:14-52 / PermutationModule.F90:92-107; the reference draws from the Fortran RNG, any
    permutation serves)."""
    return np.random.default_rng(seed).permutation(n).astype(np.int64)


def permuted_banded_triplets(n, h, seed, c0=0, c1=None, shift=0.0, complex_=False):
    """P^T H P of banded_triplets(n, h) under random_permutation(n, seed): entry (r, c) of H moves to
    (perm[r], perm[c]).  Only the NEW columns [c0, c1) (0-based) if given; sorted by column then row."""
    c1 = n if c1 is None else c1
    perm = random_permutation(n, seed)
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n, dtype=np.int64)
    cols, rows, vals = [], [], []
    step = 1 << 16
    offs = np.arange(-h, h + 1, dtype=np.int64)
    for b0 in range(c0, c1, step):
        b1 = min(c1, b0 + step)
        jo = inv[b0:b1]                                   # original (0-based) column of every new column
        ro = jo[:, None] + offs[None, :]                  # original rows
        ok = (ro >= 0) & (ro < n)
        rn = np.where(ok, perm[np.clip(ro, 0, n - 1)], n)  # new rows, invalid -> n (sorts last)
        dist = np.abs(ro - jo[:, None])
        i1 = jo + 1
        diag = -1.0 + 2.0 * ((i1 * 7919) % 1000) / 1000.0 + shift
        with np.errstate(divide="ignore", invalid="ignore"):
            v = np.where(dist == 0, diag[:, None], -0.25 * np.exp(-0.05 * dist) / np.maximum(dist, 1))
        if complex_:
            v = np.where(dist == 0, diag[:, None] + 0j, v * np.exp(1j * 0.1 * (ro - jo[:, None])))
        order = np.argsort(rn, axis=1, kind="stable")
        rn = np.take_along_axis(rn, order, axis=1)
        v = np.take_along_axis(v, order, axis=1)
        keep = rn < n
        cn = np.broadcast_to(np.arange(b0, b1, dtype=np.int64)[:, None], rn.shape)
        cols.append((cn[keep] + 1).astype(np.int32))
        rows.append((rn[keep] + 1).astype(np.int32))
        vals.append(v[keep])
    return np.concatenate(cols), np.concatenate(rows), np.concatenate(vals)


def lattice_stencil(r2):
    """offsets (dx, dy, dz) with dx^2 + dy^2 + dz^2 <= r2 and their Euclidean lengths (r2 = 13: 203 points)"""
    r = int(np.sqrt(r2)) + 1
    g = np.arange(-r, r + 1, dtype=np.int64)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    keep = X * X + Y * Y + Z * Z <= r2
    return X[keep], Y[keep], Z[keep], np.sqrt((X * X + Y * Y + Z * Z)[keep].astype(np.float64))


def lattice_triplets(L, r2=13, c0=0, c1=None, shift=0.0, decay=3.0):
    """A 3-D Hamiltonian WITHOUT band structure (SURVEY section 7 "hard parts": the operand the register-slab / MFMA
    tile kernels cannot take and no relabelling can turn into a narrow band): sites (x, y, z) of an L x L x L lattice
    with open boundaries, site number i = x + L y + L^2 z (0-based), couplings to every site within distance
    sqrt(r2) (r2 = 13: 203 entries per interior row; the row extent of a column is +-(3 L^2 + ...) sites),
    H_ii as banded_triplets, H_ij = -0.25 exp(-decay d) / d for the Euclidean distance d (decay = 3: the purification
    iterates hold about 1.7 times the entries of H at threshold 1e-8, as those of the banded generator do).  Returns NTPoly triplets
    (col, row, val), 1-based, sorted by column then row; only columns [c0, c1) (0-based) if given."""
    n = L * L * L
    c1 = n if c1 is None else c1
    dx, dy, dz, dist = lattice_stencil(r2)
    order = np.argsort(dx + L * dy + L * L * dz, kind="stable")   # ascending row offset: rows come out sorted
    dx, dy, dz, dist = dx[order], dy[order], dz[order], dist[order]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.where(dist == 0, 0.0, -0.25 * np.exp(-decay * dist) / np.maximum(dist, 1e-300))
    cols, rows, vals = [], [], []
    step = 1 << 15
    for b0 in range(c0, c1, step):
        b1 = min(c1, b0 + step)
        j = np.arange(b0, b1, dtype=np.int64)
        x, y, z = j % L, (j // L) % L, j // (L * L)
        xr, yr, zr = x[:, None] + dx[None, :], y[:, None] + dy[None, :], z[:, None] + dz[None, :]
        ok = (xr >= 0) & (xr < L) & (yr >= 0) & (yr < L) & (zr >= 0) & (zr < L)
        r = xr + L * yr + L * L * zr
        diag = -1.0 + 2.0 * (((j + 1) * 7919) % 1000) / 1000.0 + shift
        v = np.where(dist[None, :] == 0, diag[:, None], np.broadcast_to(w[None, :], r.shape))
        cn = np.broadcast_to(j[:, None], r.shape)
        cols.append((cn[ok] + 1).astype(np.int32))
        rows.append((r[ok] + 1).astype(np.int32))
        vals.append(v[ok])
    return np.concatenate(cols), np.concatenate(rows), np.concatenate(vals)
