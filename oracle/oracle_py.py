"""TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/libntpoly_oracle.so.

Matrices cross this boundary as NTPoly-style triplets: (col, row, val) arrays,
1-based, sorted by column then row (TripletModule.F90:14-25).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class _OMat(C.Structure):
    _fields_ = [("rows", C.c_int32), ("cols", C.c_int32), ("is_complex", C.c_int32),
                ("nnz", C.c_int64), ("outer", C.POINTER(C.c_int64)),
                ("inner", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double))]


class OParams(C.Structure):
    _fields_ = [("converge_diff", C.c_double), ("max_iterations", C.c_int32),
                ("threshold", C.c_double), ("monitor_convergence", C.c_int32),
                ("step_thresh", C.c_double), ("do_load_balancing", C.c_int32),
                ("perm", C.POINTER(C.c_int32))]


class _OTrace(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("cap", C.c_int32), ("value", C.POINTER(C.c_double)),
                ("energy", C.POINTER(C.c_double)), ("sigma", C.POINTER(C.c_double)),
                ("nnz", C.POINTER(C.c_int64)), ("stamp", C.POINTER(C.c_double))]


class OMonitor(C.Structure):
    _fields_ = [("win_short", C.c_double * 3), ("win_long", C.c_double * 6), ("nval", C.c_int32),
                ("loose_cutoff", C.c_double), ("tight_cutoff", C.c_double),
                ("automatic", C.c_int32)]


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "libntpoly_oracle.so")
        srcs = [os.path.join(HERE, f) for f in
                ("ntpoly_oracle.c", "ntpoly_oracle_kernels.inc", "ntpoly_oracle.h")]
        if not os.path.exists(path) or os.path.getmtime(path) < max(map(os.path.getmtime, srcs)):
            build()
        L = C.CDLL(path)
        P = C.POINTER(_OMat)
        L.omat_from_triplets.restype = P
        L.omat_from_triplets.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int32]
        L.omat_to_triplets.argtypes = [P, C.c_void_p, C.c_void_p, C.c_void_p]
        L.omat_free.argtypes = [P]
        for name in ("omat_transpose", "omat_copy", "omat_to_complex"):
            getattr(L, name).restype = P
            getattr(L, name).argtypes = [P]
        L.omat_identity.restype = P
        L.omat_identity.argtypes = [C.c_int32, C.c_int32]
        L.oracle_gemm.restype = P
        L.oracle_gemm.argtypes = [P, P, P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                  C.c_double]
        L.oracle_ps_multiply.restype = P
        L.oracle_ps_multiply.argtypes = [P, P, P, C.c_double, C.c_double, C.c_double]
        L.oracle_increment.restype = P
        L.oracle_increment.argtypes = [P, P, C.c_double, C.c_double]
        L.oracle_pairwise.restype = P
        L.oracle_pairwise.argtypes = [P, P]
        L.oracle_dot.argtypes = [P, P, C.POINTER(C.c_double)]
        L.oracle_scale.argtypes = [P, C.c_double]
        for name in ("oracle_trace", "oracle_norm", "oracle_sigma"):
            getattr(L, name).restype = C.c_double
            getattr(L, name).argtypes = [P]
        L.oracle_gershgorin.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.oracle_is_identity.argtypes = [P]
        L.oparams_default.argtypes = [C.POINTER(OParams)]
        L.otrace_new.restype = C.POINTER(_OTrace)
        L.otrace_new.argtypes = [C.c_int32]
        L.otrace_free.argtypes = [C.POINTER(_OTrace)]
        for name in ("oracle_trs2", "oracle_trs4"):
            getattr(L, name).restype = P
            getattr(L, name).argtypes = [P, P, C.c_double, C.POINTER(OParams),
                                         C.POINTER(C.c_double), C.POINTER(C.c_double),
                                         C.POINTER(_OTrace)]
        for name in ("oracle_sign", "oracle_invert", "oracle_inverse_square_root",
                     "oracle_square_root"):
            getattr(L, name).restype = P
            getattr(L, name).argtypes = [P, C.POINTER(OParams), C.POINTER(_OTrace)]
        L.omonitor_init.argtypes = [C.POINTER(OMonitor), C.c_int, C.c_double]
        L.omonitor_append.argtypes = [C.POINTER(OMonitor), C.c_double]
        L.omonitor_converged.argtypes = [C.POINTER(OMonitor)]
        L.oracle_num_threads.restype = C.c_int
        L.oracle_set_fma.argtypes = [C.c_int]
        L.oracle_get_fma.restype = C.c_int
        _LIB = L
    return _LIB


def set_fma(on):
    """arithmetic of the real multiply kernel: True = fma(a, b, acc), what the reference computes when built with FP
    contraction (oracle/build_ref.py --fma; pinned by tests/golden/ps_gemm_fma.npz); False (default) = separate multiply
    and add, its default build"""
    lib().oracle_set_fma(1 if on else 0)


def get_fma():
    return bool(lib().oracle_get_fma())


class Mat:
    """Owning handle of an oracle matrix."""

    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        if getattr(self, "ptr", None) and _LIB is not None:
            _LIB.omat_free(self.ptr)
            self.ptr = None

    @classmethod
    def from_triplets(cls, rows, cols, col, row, val):
        col = np.ascontiguousarray(col, dtype=np.int32)
        row = np.ascontiguousarray(row, dtype=np.int32)
        is_c = np.iscomplexobj(val)
        val = np.ascontiguousarray(val, dtype=np.complex128 if is_c else np.float64)
        return cls(lib().omat_from_triplets(rows, cols, len(col), col.ctypes.data, row.ctypes.data,
                                            val.ctypes.data, int(is_c)))

    @classmethod
    def from_scipy(cls, m):
        c = m.tocsc()
        c.sort_indices()
        col = np.repeat(np.arange(c.shape[1], dtype=np.int32), np.diff(c.indptr)) + 1
        return cls.from_triplets(c.shape[0], c.shape[1], col, c.indices.astype(np.int32) + 1,
                                 c.data)

    @classmethod
    def identity(cls, n, is_complex=False):
        return cls(lib().omat_identity(n, int(is_complex)))

    @property
    def rows(self):
        return self.ptr.contents.rows

    @property
    def cols(self):
        return self.ptr.contents.cols

    @property
    def nnz(self):
        return self.ptr.contents.nnz

    @property
    def is_complex(self):
        return bool(self.ptr.contents.is_complex)

    def triplets(self):
        n = self.nnz
        col = np.empty(n, dtype=np.int32)
        row = np.empty(n, dtype=np.int32)
        val = np.empty(n, dtype=np.complex128 if self.is_complex else np.float64)
        lib().omat_to_triplets(self.ptr, col.ctypes.data, row.ctypes.data, val.ctypes.data)
        return col, row, val

    def to_scipy(self):
        import scipy.sparse as sp
        col, row, val = self.triplets()
        return sp.csc_matrix((val, (row - 1, col - 1)), shape=(self.rows, self.cols))

    def copy(self):
        return Mat(lib().omat_copy(self.ptr))

    def transpose(self):
        return Mat(lib().omat_transpose(self.ptr))


def _p(m):
    return m.ptr if m is not None else None


def gemm(A, B, Cin=None, tA=False, tB=False, alpha=1.0, beta=None, threshold=0.0):
    return Mat(lib().oracle_gemm(A.ptr, B.ptr, _p(Cin), int(tA), int(tB), alpha,
                                 0.0 if beta is None else beta, int(beta is not None), threshold))


def ps_multiply(A, B, Cin=None, alpha=1.0, beta=0.0, threshold=0.0):
    return Mat(lib().oracle_ps_multiply(A.ptr, B.ptr, _p(Cin), alpha, beta, threshold))


def ps_multiply_sliced(A, B, alpha, threshold, rows, cols, slices):
    """alpha*A*B as the reference computes it on a rows x cols x slices process grid with slices > 1
    (distributed_algebra_includes/MatrixMultiply.f90:25-29, 74-80, 230-267; comm_includes/
    ReduceAndSumMatrixCleanup.f90:11-32; block multiplier 1): the inner dimension is cut into blocks of
    padded_dim / (max(rows, cols) * slices) columns dealt round-robin to the slices; slice s multiplies its share
    with threshold / (1000 * slices); the partial products are added in slice order by IncrementMatrix applied
    block by block (the same block size along the rows), the caller's threshold only in the last addition.
    Built from the oracle's own gemm / increment on sub-matrices (scipy only cuts and stacks)."""
    import scipy.sparse as sp
    n = A.rows
    lcm = slices * cols * rows
    padded = (n + lcm - 1) // lcm * lcm
    block = padded // (max(rows, cols) * slices)
    working = threshold / (slices * 1000.0)
    As, Bs = A.to_scipy().tocsc(), B
    is_c = A.is_complex
    total = None    # list of row-block matrices (scipy csc), None = empty
    nblk = (n + block - 1) // block
    acc = [None] * nblk
    for s in range(slices):
        keep = ((np.arange(n) // block) % slices) == s
        Am = As @ sp.diags(keep.astype(float))      # columns outside the slice's share removed (exact: x*1, x*0 dropped)
        Am = sp.csc_matrix(Am)
        Am.eliminate_zeros()
        Am.sort_indices()
        if is_c:
            Am = Am.astype(np.complex128)
        part = ps_multiply(Mat.from_scipy(Am) if Am.nnz else Mat.from_triplets(n, n, [], [], np.zeros(0, complex if is_c else float)),
                           Bs, None, alpha, 0.0, working).to_scipy().tocsr()
        thr = threshold if s == slices - 1 else 0.0
        for g in range(nblk):
            r0, r1 = g * block, min(n, (g + 1) * block)
            pb = sp.csc_matrix(part[r0:r1, :])
            pb.sort_indices()
            if acc[g] is None:
                acc[g] = Mat.from_triplets(r1 - r0, n, [], [], np.zeros(0, complex if is_c else float))
            acc[g] = increment(Mat.from_scipy(pb), acc[g], 1.0, thr)
    out = sp.vstack([a.to_scipy() for a in acc]).tocsc()
    out.sort_indices()
    return Mat.from_scipy(out)


def increment(A, B, alpha=1.0, threshold=0.0):
    return Mat(lib().oracle_increment(A.ptr, B.ptr, alpha, threshold))


def pairwise(A, B):
    return Mat(lib().oracle_pairwise(A.ptr, B.ptr))


def dot(A, B):
    out = (C.c_double * 2)()
    lib().oracle_dot(A.ptr, B.ptr, out)
    return complex(out[0], out[1]) if A.is_complex else out[0]


def scale(A, c):
    lib().oracle_scale(A.ptr, c)


def trace(A):
    return lib().oracle_trace(A.ptr)


def norm(A):
    return lib().oracle_norm(A.ptr)


def sigma(A):
    return lib().oracle_sigma(A.ptr)


def gershgorin(A):
    a, b = C.c_double(), C.c_double()
    lib().oracle_gershgorin(A.ptr, C.byref(a), C.byref(b))
    return a.value, b.value


def is_identity(A):
    return bool(lib().oracle_is_identity(A.ptr))


def params(converge_diff=1e-6, max_iterations=1000, threshold=0.0, monitor_convergence=True,
           perm=None):
    p = OParams()
    lib().oparams_default(C.byref(p))
    p.converge_diff = converge_diff
    p.max_iterations = max_iterations
    p.threshold = threshold
    p.monitor_convergence = int(monitor_convergence)
    if perm is not None:
        arr = np.ascontiguousarray(perm, dtype=np.int32)
        p._keep = arr
        p.perm = arr.ctypes.data_as(C.POINTER(C.c_int32))
        p.do_load_balancing = 1
    return p


def _trace_out(t):
    n = t.contents.iterations
    out = dict(iterations=n,
               value=np.array([t.contents.value[i] for i in range(n)]),
               energy=np.array([t.contents.energy[i] for i in range(n)]),
               sigma=np.array([t.contents.sigma[i] for i in range(n)]),
               nnz=np.array([t.contents.nnz[i] for i in range(n)], dtype=np.int64),
               stamp=np.array([t.contents.stamp[i] for i in range(n)]))
    lib().otrace_free(t)
    return out


def density(solver, H, ISQ, nel, p):
    """solver in {'trs2','trs4'} -> (K, energy, mu, trace dict)"""
    e, mu = C.c_double(), C.c_double()
    t = lib().otrace_new(p.max_iterations)
    K = Mat(getattr(lib(), "oracle_" + solver)(H.ptr, ISQ.ptr, nel, C.byref(p), C.byref(e),
                                               C.byref(mu), t))
    return K, e.value, mu.value, _trace_out(t)


def matrix_function(solver, A, p):
    """solver in {'sign','invert','inverse_square_root','square_root'} -> (Out, trace dict)"""
    t = lib().otrace_new(p.max_iterations)
    out = Mat(getattr(lib(), "oracle_" + solver)(A.ptr, C.byref(p), t))
    return out, _trace_out(t)
