"""TEST INFRASTRUCTURE ONLY: read/write the binary triplet files exchanged with
oracle/ref_driver.f90 (header int32 rows, cols, nnz, is_complex; int32 col[nnz];
int32 row[nnz]; f64 val[nnz] or interleaved re/im).  Indices are 1-based."""
import numpy as np


def write_tri(path, rows, cols, col, row, val):
    col = np.asarray(col, dtype=np.int32)
    row = np.asarray(row, dtype=np.int32)
    val = np.asarray(val)
    is_c = np.iscomplexobj(val)
    with open(path, "wb") as f:
        np.array([rows, cols, len(col), int(is_c)], dtype=np.int32).tofile(f)
        col.tofile(f)
        row.tofile(f)
        if is_c:
            val.astype(np.complex128).view(np.float64).tofile(f)
        else:
            val.astype(np.float64).tofile(f)


def read_tri(path):
    with open(path, "rb") as f:
        rows, cols, nnz, is_c = np.fromfile(f, dtype=np.int32, count=4)
        col = np.fromfile(f, dtype=np.int32, count=nnz)
        row = np.fromfile(f, dtype=np.int32, count=nnz)
        if is_c:
            val = np.fromfile(f, dtype=np.float64, count=2 * nnz).view(np.complex128)
        else:
            val = np.fromfile(f, dtype=np.float64, count=nnz)
    return int(rows), int(cols), col, row, val


def from_scipy(m):
    """scipy sparse -> (col,row,val) sorted by column then row, 1-based."""
    c = m.tocsc()
    c.sort_indices()
    c.eliminate_zeros()
    col = np.repeat(np.arange(c.shape[1], dtype=np.int32), np.diff(c.indptr)) + 1
    return col.astype(np.int32), (c.indices + 1).astype(np.int32), c.data.copy()


def to_scipy(rows, cols, col, row, val):
    import scipy.sparse as sp
    return sp.csc_matrix((val, (row - 1, col - 1)), shape=(rows, cols))
