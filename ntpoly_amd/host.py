"""Python mirror of the reference's C++/SWIG classes (Source/CPlusPlus/*.h, the objects its own
unit tests use: ProcessGrid, TripletList_r/_c, Matrix_ps, Matrix_lsr/_lsc, MatrixMemoryPool,
Permutation, SolverParameters, DensityMatrixSolvers, SignSolvers, InverseSolvers,
SquareRootSolvers, LoadBalancer, EigenBounds) on top of the C ABI of libntpoly_amd.so.

Same method names and argument meaning as the reference so that the parity tests read like
the reference's tests.  numpy arrays are only used at the boundary (triplets in / out); all
matrix data lives in HBM.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import b, d, handle, i, lib, ll, s


# ------------------------------------------------------------------ process grid
def ConstructGlobalProcessGrid(process_rows=None, process_columns=None, process_slices=None, world_comm=0):
    """ProcessGrid.cc:12-48 (ConstructGlobalProcessGrid overloads)."""
    if process_rows is None and process_slices is None:
        lib.ConstructGlobalProcessGrid_default_wrp(i(world_comm))
    elif process_rows is None:
        lib.ConstructGlobalProcessGrid_onlyslice_wrp(i(world_comm), i(process_slices))
    else:
        lib.ConstructGlobalProcessGrid_wrp(i(world_comm), i(process_rows), i(process_columns), i(process_slices))


def WriteGridInfo():
    lib.WriteGlobalProcessGridInfo_wrp()


def DestructGlobalProcessGrid():
    lib.DestructGlobalProcessGrid_wrp()


def GetGlobalIsRoot():
    return bool(lib.GetGlobalIsRoot_wrp())


def init_comm(unique_id=None, rank=0, nranks=1):
    """RCCL bootstrap (replaces MPI_Init + communicator handles of the reference)."""
    if nranks <= 1:
        lib.ntpoly_amd_init_comm(b"\0" * 128, i(0), i(1))
    else:
        lib.ntpoly_amd_init_comm(unique_id, i(rank), i(nranks))


def get_unique_id():
    buf = C.create_string_buffer(128)
    lib.ntpoly_amd_get_unique_id(buf)
    return buf.raw


def init_comm_from_torch():
    """One process per GPU under torch.distributed.run: the 128-byte RCCL id is broadcast over a
    torch.distributed *gloo* group (control plane only) and the engine builds its own RCCL
    communicator (data plane, xGMI).  torch's GPU runtime is never initialised: the PyTorch wheel
    bundles a second HIP/RCCL runtime and two initialised runtimes in one process corrupt the heap
    at exit.  The engine picks its GPU from LOCAL_RANK.  Returns (rank, world_size)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        init_comm()
        return 0, 1
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo")
    box = [get_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    init_comm(box[0], rank, world)
    return rank, world


def init_comm_from_env(timeout=300.0):
    """One process per GPU under `python -m torch.distributed.run` (or any launcher that sets RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_PORT) on ONE node, without importing torch: rank 0 creates the 128-byte RCCL id and hands
    it to the other ranks through a file in /tmp named after the launcher's pid (all ranks are children of the
    same launcher) and the master port; the engine then builds its own RCCL communicator and picks its GPU from
    LOCAL_RANK.  Returns (rank, world_size)."""
    import os
    import time
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world <= 1:
        init_comm()
        return 0, 1
    if os.environ.get("NTPOLY_AMD_COMM", "").startswith("shm:"):
        # shared-memory test transport: the segment name is in the environment, there is no id to hand over
        init_comm(get_unique_id(), rank, world)
        barrier()
        return rank, world
    # NTPOLY_AMD_RDV names the file explicitly (launchers whose ranks are not children of one process)
    path = os.environ.get("NTPOLY_AMD_RDV") or "/tmp/ntpoly_amd_rdv_%d_%s" % (os.getppid(), os.environ.get("MASTER_PORT", "0"))
    # a per-launch nonce keeps a stale file of an earlier (crashed) launch from being taken for this one's id
    # (an explicitly named file may be shared by ranks of different parents: NTPOLY_AMD_RDV_NONCE stands in for the pid)
    who = os.environ.get("NTPOLY_AMD_RDV_NONCE", "") if os.environ.get("NTPOLY_AMD_RDV") else str(os.getppid())
    # (an elastic restart of a crashed worker group keeps run id, port and parent: the restart count tells the new
    # group's file from the dead group's)
    # (hashed, not truncated: a long user-supplied run id must not push the restart count, port and parent out of the field)
    import hashlib
    nonce = hashlib.sha256((os.environ.get("TORCHELASTIC_RUN_ID", "") + ":" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0") + ":" +
                            os.environ.get("MASTER_PORT", "0") + ":" + who).encode()).hexdigest().encode()[:64].ljust(64, b"\0")
    if rank == 0:
        uid = get_unique_id()
        try:
            os.unlink(path)          # left behind by a launch that died between writing and joining
        except FileNotFoundError:
            pass
        tmp = "%s.%d.tmp" % (path, os.getpid())
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(nonce + uid)
        os.rename(tmp, path)
    else:
        t0 = time.time()
        uid = None
        while uid is None:
            try:
                with open(path, "rb") as f:
                    blob = f.read()
                if len(blob) == 64 + 128 and blob[:64] == nonce:
                    uid = blob[64:]
            except FileNotFoundError:
                pass
            if uid is None:
                if time.time() - t0 > timeout:
                    raise RuntimeError("rendezvous file %s did not appear (or belongs to another launch)" % path)
                time.sleep(0.01)
    init_comm(uid, rank, world)
    barrier()
    if rank == 0:
        os.unlink(path)
    return rank, world


def barrier():
    lib.ntpoly_amd_barrier()


def allreduce_max(x):
    v = C.c_double(float(x))
    lib.ntpoly_amd_allreduce_max(C.byref(v), i(1))
    return v.value


def set_option(name, value):
    lib.ntpoly_amd_set_option(name.encode(), i(value))


def get_option(name):
    lib.ntpoly_amd_get_option.restype = C.c_int
    return int(lib.ntpoly_amd_get_option(name.encode()))


def synchronize():
    lib.ntpoly_amd_synchronize()


# ------------------------------------------------------------------ triplets
class Triplet_r:
    """Triplet.h: (index_column, index_row, point_value), 1-based indices"""
    def __init__(self, index_column=0, index_row=0, point_value=0.0):
        self.index_column, self.index_row, self.point_value = index_column, index_row, point_value

    def __iter__(self):
        return iter((self.index_column, self.index_row, self.point_value))


class Triplet_c(Triplet_r):
    pass


class _TripletList:
    _c = False

    def __init__(self, size=0):
        self.ih = handle()
        getattr(lib, "ConstructTripletList_%s_wrp" % self._sfx)(self.ih, i(size))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            getattr(lib, "DestructTripletList_%s_wrp" % self._sfx)(self.ih)
            self.ih = None

    def GetSize(self):
        return int(getattr(lib, "GetTripletListSize_%s_wrp" % self._sfx)(self.ih))

    def Append(self, index_column, index_row=None, point_value=None):
        """Append(triplet) as the reference's SWIG classes take it (a Triplet_r / Triplet_c object), or the three
        fields directly"""
        if index_row is None:
            t = index_column
            index_column, index_row, point_value = t.index_column, t.index_row, t.point_value
        if self._c:
            lib.AppendToTripletList_c_wrp(self.ih, i(index_column), i(index_row), d(point_value.real), d(point_value.imag))
        else:
            lib.AppendToTripletList_r_wrp(self.ih, i(index_column), i(index_row), d(point_value))

    def GetTripletAt(self, index):
        """0-based index, as the reference's C++ / SWIG layer; returns a Triplet object that also unpacks as
        (index_column, index_row, point_value)"""
        col, row = C.c_int(), C.c_int()
        if self._c:
            re, im = C.c_double(), C.c_double()
            lib.GetTripletAt_c_wrp(self.ih, i(index + 1), C.byref(col), C.byref(row), C.byref(re), C.byref(im))
            return Triplet_c(col.value, row.value, complex(re.value, im.value))
        val = C.c_double()
        lib.GetTripletAt_r_wrp(self.ih, i(index + 1), C.byref(col), C.byref(row), C.byref(val))
        return Triplet_r(col.value, row.value, val.value)

    # bulk transfer (extension; the reference ABI moves one triplet per call)
    def set_arrays(self, col, row, val):
        col = np.ascontiguousarray(col, dtype=np.int32)
        row = np.ascontiguousarray(row, dtype=np.int32)
        if self._c:
            val = np.ascontiguousarray(val, dtype=np.complex128)
            lib.ntpoly_amd_triplets_set_c(self.ih, ll(len(col)), col.ctypes.data_as(C.c_void_p),
                                          row.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p))
        else:
            val = np.ascontiguousarray(val, dtype=np.float64)
            lib.ntpoly_amd_triplets_set_r(self.ih, ll(len(col)), col.ctypes.data_as(C.c_void_p),
                                          row.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p))

    def arrays(self):
        n = self.GetSize()
        col = np.empty(n, dtype=np.int32)
        row = np.empty(n, dtype=np.int32)
        val = np.empty(n, dtype=np.complex128 if self._c else np.float64)
        lib.ntpoly_amd_triplets_get(self.ih, col.ctypes.data_as(C.c_void_p), row.ctypes.data_as(C.c_void_p),
                                    val.ctypes.data_as(C.c_void_p))
        return col, row, val


class TripletList_r(_TripletList):
    _sfx, _c = "r", False


class TripletList_c(_TripletList):
    _sfx, _c = "c", True


def _tlist_from(col, row, val):
    t = TripletList_c() if np.iscomplexobj(val) else TripletList_r()
    t.set_arrays(col, row, val)
    return t


# ------------------------------------------------------------------ permutation / parameters / pools
class Permutation:
    def __init__(self, matrix_dimension):
        self.ih = handle()
        self.dim = matrix_dimension
        lib.ConstructDefaultPermutation_wrp(self.ih, i(matrix_dimension))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructPermutation_wrp(self.ih)
            self.ih = None

    def _rebuild(self, fn):
        lib.DestructPermutation_wrp(self.ih)
        getattr(lib, fn)(self.ih, i(self.dim))

    def SetDefaultPermutation(self):
        self._rebuild("ConstructDefaultPermutation_wrp")

    def SetReversePermutation(self):
        self._rebuild("ConstructReversePermutation_wrp")

    def SetRandomPermutation(self):
        self._rebuild("ConstructRandomPermutation_wrp")

    def set_lookup(self, index_lookup):
        arr = np.ascontiguousarray(index_lookup, dtype=np.int32)
        lib.ntpoly_amd_permutation_set(self.ih, i(len(arr)), arr.ctypes.data_as(C.c_void_p))


class SolverParameters:
    def __init__(self):
        self.ih = handle()
        lib.ConstructSolverParameters_wrp(self.ih)
        self._perm = None

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructSolverParameters_wrp(self.ih)
            self.ih = None

    def SetConvergeDiff(self, v):
        lib.SetParametersConvergeDiff_wrp(self.ih, d(v))

    def SetMaxIterations(self, v):
        lib.SetParametersMaxIterations_wrp(self.ih, i(v))

    def SetVerbosity(self, v):
        lib.SetParametersBeVerbose_wrp(self.ih, b(v))

    def SetThreshold(self, v):
        lib.SetParametersThreshold_wrp(self.ih, d(v))

    def SetLoadBalance(self, permutation):
        self._perm = permutation
        lib.SetParametersLoadBalance_wrp(self.ih, permutation.ih)

    def SetStepThreshold(self, v):
        lib.SetParametersStepThreshold_wrp(self.ih, d(v))

    def SetMonitorConvergence(self, v):
        lib.SetParametersMonitorConvergence_wrp(self.ih, b(v))


class PMatrixMemoryPool:
    def __init__(self, matrix):
        self.ih = handle()
        lib.ConstructMatrixMemoryPool_p_wrp(self.ih, matrix.ih)

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructMatrixMemoryPool_p_wrp(self.ih)
            self.ih = None


class MatrixMemoryPool_r:
    def __init__(self, columns, rows):
        self.ih = handle()
        lib.ConstructMatrixMemoryPool_lr_wrp(self.ih, i(columns), i(rows))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructMatrixMemoryPool_lr_wrp(self.ih)
            self.ih = None


class MatrixMemoryPool_c(MatrixMemoryPool_r):
    def __init__(self, columns, rows):
        self.ih = handle()
        lib.ConstructMatrixMemoryPool_lc_wrp(self.ih, i(columns), i(rows))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructMatrixMemoryPool_lc_wrp(self.ih)
            self.ih = None


# ------------------------------------------------------------------ distributed matrix
class Matrix_ps:
    """PSMatrix.h:20-197."""

    def __init__(self, arg=None):
        self.ih = handle()
        if isinstance(arg, Matrix_ps):
            lib.ConstructEmptyMatrix_ps_wrp(self.ih, i(arg.GetActualDimension()))
            lib.CopyMatrix_ps_wrp(arg.ih, self.ih)
        elif isinstance(arg, str):
            raw, n = s(arg)
            if arg.endswith(".mtx"):
                lib.ConstructMatrixFromMatrixMarket_ps_wrp(self.ih, raw, n)
            else:
                lib.ConstructMatrixFromBinary_ps_wrp(self.ih, raw, n)
        else:
            lib.ConstructEmptyMatrix_ps_wrp(self.ih, i(arg))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            lib.DestructMatrix_ps_wrp(self.ih)
            self.ih = None

    # -- construction helpers used by the tests
    @classmethod
    def from_triplets(cls, dim, col, row, val):
        m = cls(dim)
        m.FillFromTripletList(_tlist_from(col, row, val))
        return m

    @classmethod
    def from_scipy(cls, mat):
        c = mat.tocsc()
        c.sort_indices()
        col = np.repeat(np.arange(c.shape[1], dtype=np.int32), np.diff(c.indptr)) + 1
        return cls.from_triplets(c.shape[0], col, c.indices.astype(np.int32) + 1, c.data)

    def triplets(self):
        """local triplets (col,row,val), 1-based global indices, sorted by column then row"""
        t = TripletList_c() if self.IsComplex() else TripletList_r()
        self.GetTripletList(t)
        return t.arrays()

    def to_scipy(self):
        import scipy.sparse as sp
        col, row, val = self.triplets()
        n = self.GetActualDimension()
        return sp.csc_matrix((val, (row - 1, col - 1)), shape=(n, n))

    # -- reference API
    def WriteToBinary(self, path):
        raw, n = s(path)
        lib.WriteMatrixToBinary_ps_wrp(self.ih, raw, n)

    def WriteToMatrixMarket(self, path):
        raw, n = s(path)
        lib.WriteMatrixToMatrixMarket_ps_wrp(self.ih, raw, n)

    def FillFromTripletList(self, tlist, prepartitioned=False):
        if prepartitioned:  # Fortran API: prepartitioned_in=.TRUE. (each rank passes its own columns)
            lib.ntpoly_amd_fill_prepartitioned(self.ih, tlist.ih)
        elif tlist._c:
            lib.FillMatrixFromTripletList_psc_wrp(self.ih, tlist.ih)
        else:
            lib.FillMatrixFromTripletList_psr_wrp(self.ih, tlist.ih)

    def FillPermutation(self, permutation, permuterows=True):
        lib.FillMatrixPermutation_ps_wrp(self.ih, permutation.ih, b(permuterows))

    def FillIdentity(self):
        lib.FillMatrixIdentity_ps_wrp(self.ih)

    def GetActualDimension(self):
        v = C.c_int()
        lib.GetMatrixActualDimension_ps_wrp(self.ih, C.byref(v))
        return v.value

    def GetLogicalDimension(self):
        v = C.c_int()
        lib.GetMatrixLogicalDimension_ps_wrp(self.ih, C.byref(v))
        return v.value

    def GetSize(self):
        v = C.c_long()
        lib.GetMatrixSize_ps_wrp(self.ih, C.byref(v))
        return v.value

    def IsComplex(self):
        return bool(lib.ntpoly_amd_matrix_is_complex(self.ih))

    def local_columns(self):
        a, c = C.c_int(), C.c_int()
        lib.ntpoly_amd_matrix_local_columns(self.ih, C.byref(a), C.byref(c))
        return a.value, c.value

    def GetTripletList(self, tlist):
        if tlist._c:
            lib.GetMatrixTripletList_psc_wrp(self.ih, tlist.ih)
        else:
            lib.GetMatrixTripletList_psr_wrp(self.ih, tlist.ih)

    def Transpose(self, matA):
        lib.TransposeMatrix_ps_wrp(matA.ih, self.ih)

    def Conjugate(self):
        lib.ConjugateMatrix_ps_wrp(self.ih)

    def Dot(self, matB):
        if self.IsComplex() or matB.IsComplex():
            re, im = C.c_double(), C.c_double()
            lib.DotMatrix_psc_wrp(self.ih, matB.ih, C.byref(re), C.byref(im))
            return complex(re.value, im.value)
        v = C.c_double()
        lib.DotMatrix_psr_wrp(self.ih, matB.ih, C.byref(v))
        return v.value

    def Increment(self, matB, alpha=1.0, threshold=0.0):
        """this <- alpha*matB + this (PSMatrix.cc Increment)"""
        lib.IncrementMatrix_ps_wrp(matB.ih, self.ih, d(alpha), d(threshold))

    def PairwiseMultiply(self, matA, matB):
        lib.MatrixPairwiseMultiply_ps_wrp(matA.ih, matB.ih, self.ih)

    def Gemm(self, matA, matB, memory_pool=None, alpha=1.0, beta=0.0, threshold=0.0):
        """this = alpha*matA*matB + beta*this (PSMatrix.cc:208-213)"""
        pool = memory_pool if memory_pool is not None else PMatrixMemoryPool(matA)
        lib.MatrixMultiply_ps_wrp(matA.ih, matB.ih, self.ih, d(alpha), d(beta), d(threshold), pool.ih)

    def Scale(self, constant):
        lib.ScaleMatrix_ps_wrp(self.ih, d(constant))

    def Norm(self):
        return float(lib.MatrixNorm_ps_wrp(self.ih))

    def MeasureAsymmetry(self):
        return float(lib.MeasureAsymmetry_ps_wrp(self.ih))

    def Trace(self):
        v = C.c_double()
        lib.MatrixTrace_ps_wrp(self.ih, C.byref(v))
        return v.value

    def IsIdentity(self):
        return bool(lib.IsIdentity_ps_wrp(self.ih))

    def Symmetrize(self):
        lib.SymmetrizeMatrix_ps_wrp(self.ih)

    def CommSplit(self):
        """(copy of the WHOLE matrix on this process's half of the grid, colour 0 / 1, True when the grid was split along its
        slices) -- CommSplitMatrix, PSMatrixModule.F90:1489-1541; the copy lives on a sub-communicator: everything done with
        it afterwards (products, reductions, solvers) stays inside the half"""
        out = Matrix_ps.__new__(Matrix_ps)
        out.ih = handle()
        color = C.c_int(0)
        split_slice = C.c_bool(False)
        lib.ntpoly_amd_comm_split_matrix(self.ih, out.ih, C.byref(color), C.byref(split_slice))
        return out, int(color.value), bool(split_slice.value)

    def grid_comm_info(self):
        """(rank on the communicator of this matrix's grid, its size, True when that is a sub-communicator)"""
        g = handle()
        lib.GetMatrixProcessGrid_ps_wrp(self.ih, g)
        out = (C.c_int * 3)()
        lib.ntpoly_amd_grid_comm_info(g, out)
        return int(out[0]), int(out[1]), bool(out[2])

    def GatherMatrixToProcess(self, within_slice_id=None):
        """the whole matrix as a LOCAL matrix on every process (None) or on the process with this rank inside its slice (the
        others get None) -- PSMatrixModule.F90:1704-1808"""
        cls = Matrix_lsc if self.IsComplex() else Matrix_lsr
        ih = handle()
        lib.ntpoly_amd_gather_matrix_to_process(self.ih, ih, i(-1 if within_slice_id is None else within_slice_id))
        if not any(ih):
            return None
        out = cls.__new__(cls)
        out.ih = ih
        return out


# ------------------------------------------------------------------ local matrices (config 2)
class _Matrix_ls:
    def __init__(self, rows=None, columns=None, tlist=None, path=None):
        self.ih = handle()
        sfx = self._sfx
        if path is not None:
            raw, n = s(path)
            getattr(lib, "ConstructMatrixFromFile_%s_wrp" % sfx)(self.ih, raw, n)
        elif tlist is not None:
            getattr(lib, "ConstructMatrixFromTripletList_%s_wrp" % sfx)(self.ih, tlist.ih, i(rows), i(columns))
        else:
            getattr(lib, "ConstructZeroMatrix_%s_wrp" % sfx)(self.ih, i(rows), i(columns))

    def __del__(self):
        if getattr(self, "ih", None) is not None and lib is not None:
            getattr(lib, "DestructMatrix_%s_wrp" % self._sfx)(self.ih)
            self.ih = None

    @classmethod
    def from_triplets(cls, rows, cols, col, row, val):
        return cls(rows, cols, tlist=_tlist_from(col, row, val))

    def GetRows(self):
        v = C.c_int()
        getattr(lib, "GetMatrixRows_%s_wrp" % self._sfx)(self.ih, C.byref(v))
        return v.value

    def GetColumns(self):
        v = C.c_int()
        getattr(lib, "GetMatrixColumns_%s_wrp" % self._sfx)(self.ih, C.byref(v))
        return v.value

    def Scale(self, c):
        getattr(lib, "ScaleMatrix_%s_wrp" % self._sfx)(self.ih, d(c))

    def Increment(self, matB, alpha=1.0, threshold=0.0):
        getattr(lib, "IncrementMatrix_%s_wrp" % self._sfx)(matB.ih, self.ih, d(alpha), d(threshold))

    def PairwiseMultiply(self, matA, matB):
        getattr(lib, "PairwiseMultiplyMatrix_%s_wrp" % self._sfx)(matA.ih, matB.ih, self.ih)

    def Gemm(self, matA, matB, isATransposed=False, isBTransposed=False, alpha=1.0, beta=0.0, threshold=0.0,
             memory_pool=None):
        """this = alpha*op(matA)*op(matB) + beta*this (SMatrix.cc Gemm)"""
        pool = memory_pool if memory_pool is not None else self._pool(matB.GetColumns(), matA.GetRows())
        getattr(lib, "MatrixMultiply_%s_wrp" % self._sfx)(matA.ih, matB.ih, self.ih, b(isATransposed), b(isBTransposed),
                                                          d(alpha), d(beta), d(threshold), pool.ih)

    def Transpose(self, matA):
        getattr(lib, "TransposeMatrix_%s_wrp" % self._sfx)(matA.ih, self.ih)

    def triplets(self):
        t = TripletList_c() if self._sfx == "lsc" else TripletList_r()
        getattr(lib, "MatrixToTripletList_%s_wrp" % self._sfx)(self.ih, t.ih)
        return t.arrays()


class Matrix_lsr(_Matrix_ls):
    _sfx, _pool = "lsr", MatrixMemoryPool_r

    def Dot(self, matB):
        v = C.c_double()
        lib.DotMatrix_lsr_wrp(self.ih, matB.ih, C.byref(v))
        return v.value


class Matrix_lsc(_Matrix_ls):
    _sfx, _pool = "lsc", MatrixMemoryPool_c

    def Dot(self, matB):
        re, im = C.c_double(), C.c_double()
        lib.DotMatrix_lsc_wrp(self.ih, matB.ih, C.byref(re), C.byref(im))
        return complex(re.value, im.value)

    def Conjugate(self):
        lib.ConjugateMatrix_lsc_wrp(self.ih)


# ------------------------------------------------------------------ solvers (static-method classes as in Source/CPlusPlus)
class DensityMatrixSolvers:
    @staticmethod
    def _run(fn, Hamiltonian, InverseSquareRoot, trace, Density, solver_parameters):
        e, mu = C.c_double(), C.c_double()
        fn(Hamiltonian.ih, InverseSquareRoot.ih, d(trace), Density.ih, C.byref(e), C.byref(mu), solver_parameters.ih)
        return e.value, mu.value

    @staticmethod
    def PM(H, ISQ, trace, Density, solver_parameters):
        return DensityMatrixSolvers._run(lib.PM_wrp, H, ISQ, trace, Density, solver_parameters)

    @staticmethod
    def TRS2(H, ISQ, trace, Density, solver_parameters):
        return DensityMatrixSolvers._run(lib.TRS2_wrp, H, ISQ, trace, Density, solver_parameters)

    @staticmethod
    def TRS4(H, ISQ, trace, Density, solver_parameters):
        return DensityMatrixSolvers._run(lib.TRS4_wrp, H, ISQ, trace, Density, solver_parameters)

    @staticmethod
    def HPCP(H, ISQ, trace, Density, solver_parameters):
        return DensityMatrixSolvers._run(lib.HPCP_wrp, H, ISQ, trace, Density, solver_parameters)


def trs2_step(X, X2, WH, trace, threshold, trace_x=None):
    """one TRS2 iteration (what TRS2_wrp runs inside its loop) -> (sigma, energy, trace of the new X);
    trace_x = trace of X from the previous step's return value (None: computed).  X2 is SCRATCH: on the fused path the
    product X*X never exists as a matrix (it is merged into X inside the SpGEMM kernel's epilogue) and X2 is left
    untouched; only the unfused path leaves X*X in it."""
    e, sg = C.c_double(), C.c_double()
    tr = C.c_double(float("nan") if trace_x is None else trace_x)
    lib.ntpoly_amd_trs2_step(X.ih, X2.ih, WH.ih, d(trace), d(threshold), C.byref(e), C.byref(sg), C.byref(tr))
    return sg.value, e.value, tr.value


class SignSolvers:
    @staticmethod
    def ComputeSign(mat, signmat, solver_parameters):
        lib.SignFunction_wrp(mat.ih, signmat.ih, solver_parameters.ih)

    @staticmethod
    def ComputePolarDecomposition(mat, umat, hmat, solver_parameters):
        lib.PolarDecomposition_wrp(mat.ih, umat.ih, hmat.ih, solver_parameters.ih)


class InverseSolvers:
    @staticmethod
    def Invert(mat, inverse, solver_parameters):
        lib.Invert_wrp(mat.ih, inverse.ih, solver_parameters.ih)

    @staticmethod
    def PseudoInverse(mat, inverse, solver_parameters):
        lib.PseudoInverse_wrp(mat.ih, inverse.ih, solver_parameters.ih)


class SquareRootSolvers:
    @staticmethod
    def SquareRoot(mat, out, solver_parameters):
        lib.SquareRoot_wrp(mat.ih, out.ih, solver_parameters.ih)

    @staticmethod
    def InverseSquareRoot(mat, out, solver_parameters):
        lib.InverseSquareRoot_wrp(mat.ih, out.ih, solver_parameters.ih)

    @staticmethod
    def with_order(mat, out, solver_parameters, inverse, order):
        lib.ntpoly_amd_square_root_order(mat.ih, out.ih, solver_parameters.ih, i(int(inverse)), i(order))


class _Poly:
    """Polynomial.h / ChebyshevSolvers.h / HermiteSolvers.h: coefficient `degree` (0-based) multiplies
    x^degree, T_degree(x) or H_degree(x)"""
    _construct = _destruct = _set = None

    def __init__(self, degree):
        self.ih = handle()
        getattr(lib, self._construct)(self.ih, i(degree))

    def __del__(self):
        try:
            getattr(lib, self._destruct)(self.ih)
        except Exception:
            pass

    def SetCoefficient(self, degree, coefficient):
        getattr(lib, self._set)(self.ih, i(degree + 1), d(coefficient))


class Polynomial(_Poly):
    _construct, _destruct, _set = "ConstructPolynomial_wrp", "DestructPolynomial_wrp", "SetCoefficient_wrp"

    def HornerCompute(self, InputMat, OutputMat, solver_parameters):
        lib.HornerCompute_wrp(InputMat.ih, OutputMat.ih, self.ih, solver_parameters.ih)

    def PatersonStockmeyerCompute(self, InputMat, OutputMat, solver_parameters):
        lib.PatersonStockmeyerCompute_wrp(InputMat.ih, OutputMat.ih, self.ih, solver_parameters.ih)


class ChebyshevPolynomial(_Poly):
    _construct, _destruct, _set = ("ConstructChebyshevPolynomial_wrp", "DestructChebyshevPolynomial_wrp",
                                   "SetChebyshevCoefficient_wrp")

    def Compute(self, InputMat, OutputMat, solver_parameters):
        lib.ChebyshevCompute_wrp(InputMat.ih, OutputMat.ih, self.ih, solver_parameters.ih)

    def ComputeFactorized(self, InputMat, OutputMat, solver_parameters):
        lib.FactorizedChebyshevCompute_wrp(InputMat.ih, OutputMat.ih, self.ih, solver_parameters.ih)


class HermitePolynomial(_Poly):
    _construct, _destruct, _set = ("ConstructHermitePolynomial_wrp", "DestructHermitePolynomial_wrp",
                                   "SetHermiteCoefficient_wrp")

    def Compute(self, InputMat, OutputMat, solver_parameters):
        lib.HermiteCompute_wrp(InputMat.ih, OutputMat.ih, self.ih, solver_parameters.ih)


class ExponentialSolvers:
    @staticmethod
    def ComputeExponential(InputMat, OutputMat, solver_parameters):
        lib.ComputeExponential_wrp(InputMat.ih, OutputMat.ih, solver_parameters.ih)

    @staticmethod
    def ComputeLogarithm(InputMat, OutputMat, solver_parameters):
        lib.ComputeLogarithm_wrp(InputMat.ih, OutputMat.ih, solver_parameters.ih)

    @staticmethod
    def ComputeExponentialPade(InputMat, OutputMat, solver_parameters):
        lib.ComputeExponentialPade_wrp(InputMat.ih, OutputMat.ih, solver_parameters.ih)


class TrigonometrySolvers:
    @staticmethod
    def Sine(InputMat, OutputMat, solver_parameters):
        lib.Sine_wrp(InputMat.ih, OutputMat.ih, solver_parameters.ih)

    @staticmethod
    def Cosine(InputMat, OutputMat, solver_parameters):
        lib.Cosine_wrp(InputMat.ih, OutputMat.ih, solver_parameters.ih)


class RootSolvers:
    @staticmethod
    def ComputeRoot(InputMat, OutputMat, root, solver_parameters):
        lib.ComputeRoot_wrp(InputMat.ih, OutputMat.ih, i(root), solver_parameters.ih)

    @staticmethod
    def ComputeInverseRoot(InputMat, OutputMat, root, solver_parameters):
        lib.ComputeInverseRoot_wrp(InputMat.ih, OutputMat.ih, i(root), solver_parameters.ih)


class LinearSolvers:
    """Source/CPlusPlus/LinearSolvers.h"""
    @staticmethod
    def CGSolver(AMat, XMat, BMat, solver_parameters):
        lib.CGSolver_wrp(AMat.ih, XMat.ih, BMat.ih, solver_parameters.ih)

    @staticmethod
    def CholeskyDecomposition(AMat, LMat, solver_parameters):
        lib.CholeskyDecomposition_wrp(AMat.ih, LMat.ih, solver_parameters.ih)


class Analysis:
    """Source/CPlusPlus/Analysis.h"""
    @staticmethod
    def PivotedCholeskyDecomposition(AMat, LMat, rank, solver_parameters):
        lib.PivotedCholeskyDecomposition_wrp(AMat.ih, LMat.ih, i(rank), solver_parameters.ih)

    @staticmethod
    def ReduceDimension(AMat, dim, RMat, solver_parameters):
        lib.ReduceDimension_wrp(AMat.ih, i(dim), RMat.ih, solver_parameters.ih)


class GeometryOptimization:
    """Source/CPlusPlus/GeometryOptimization.h"""
    @staticmethod
    def PurificationExtrapolate(PreviousDensity, Overlap, trace, NewDensity, solver_parameters):
        lib.PurificationExtrapolate_wrp(PreviousDensity.ih, Overlap.ih, d(trace), NewDensity.ih, solver_parameters.ih)

    @staticmethod
    def LowdinExtrapolate(PreviousDensity, OldOverlap, NewOverlap, NewDensity, solver_parameters):
        lib.LowdinExtrapolate_wrp(PreviousDensity.ih, OldOverlap.ih, NewOverlap.ih, NewDensity.ih, solver_parameters.ih)


class MatrixConversion:
    """Source/CPlusPlus/MatrixConversion.h"""
    @staticmethod
    def SnapMatrixToSparsityPattern(mat, pattern):
        lib.SnapMatrixToSparsityPattern_wrp(mat.ih, pattern.ih)


class EigenSolvers:
    """Source/CPlusPlus/EigenSolvers.h (the dense eigensolver is the engine's own Jacobi method on the GPU)"""
    @staticmethod
    def EigenDecomposition(matrix, eigenvalues, nvals, eigenvectors, solver_parameters):
        lib.EigenDecomposition_wrp(matrix.ih, eigenvalues.ih, i(nvals), eigenvectors.ih, solver_parameters.ih)

    @staticmethod
    def EigenValues(matrix, eigenvalues, nvals, solver_parameters):
        lib.EigenDecomposition_novec_wrp(matrix.ih, eigenvalues.ih, i(nvals), solver_parameters.ih)

    @staticmethod
    def SingularValueDecomposition(matrix, leftvectors, rightvectors, singularvalues, solver_parameters):
        lib.SingularValueDecompostion_wrp(matrix.ih, leftvectors.ih, rightvectors.ih, singularvalues.ih,
                                          solver_parameters.ih)

    @staticmethod
    def EstimateGap(H, K, chemical_potential, solver_parameters):
        gap = C.c_double()
        lib.EstimateGap_wrp(H.ih, K.ih, d(chemical_potential), C.byref(gap), solver_parameters.ih)
        return gap.value


class FermiOperator:
    """Source/CPlusPlus/FermiOperator.h; each returns (energy, chemical potential or None)"""
    @staticmethod
    def ComputeDenseFOE(H, ISQ, trace, Density, inv_temp, solver_parameters):
        e, mu = C.c_double(), C.c_double()
        lib.ComputeDenseFOE_wrp(H.ih, ISQ.ih, d(trace), Density.ih, d(inv_temp), C.byref(e), C.byref(mu), solver_parameters.ih)
        return e.value, mu.value

    @staticmethod
    def WOM_GC(H, ISQ, Density, chemical_potential, inv_temp, solver_parameters):
        e = C.c_double()
        lib.WOM_GC_wrp(H.ih, ISQ.ih, Density.ih, d(chemical_potential), d(inv_temp), C.byref(e), solver_parameters.ih)
        return e.value

    @staticmethod
    def WOM_C(H, ISQ, Density, trace, inv_temp, solver_parameters):
        e = C.c_double()
        lib.WOM_C_wrp(H.ih, ISQ.ih, Density.ih, d(trace), d(inv_temp), C.byref(e), solver_parameters.ih)
        return e.value


class DenseSolvers:
    """the reference's Dense* entry points (f(A) = V f(L) V^H through the eigendecomposition)"""
    @staticmethod
    def DenseDensity(H, ISQ, trace, Density, solver_parameters):
        return DensityMatrixSolvers._run(lib.DenseDensity_wrp, H, ISQ, trace, Density, solver_parameters)

    @staticmethod
    def _f(name, In, Out, solver_parameters):
        getattr(lib, name)(In.ih, Out.ih, solver_parameters.ih)

    SquareRoot = staticmethod(lambda a, o, p: DenseSolvers._f("DenseSquareRoot_wrp", a, o, p))
    InverseSquareRoot = staticmethod(lambda a, o, p: DenseSolvers._f("DenseInverseSquareRoot_wrp", a, o, p))
    Exponential = staticmethod(lambda a, o, p: DenseSolvers._f("ComputeDenseExponential_wrp", a, o, p))
    Logarithm = staticmethod(lambda a, o, p: DenseSolvers._f("ComputeDenseLogarithm_wrp", a, o, p))
    Sine = staticmethod(lambda a, o, p: DenseSolvers._f("DenseSine_wrp", a, o, p))
    Cosine = staticmethod(lambda a, o, p: DenseSolvers._f("DenseCosine_wrp", a, o, p))
    Invert = staticmethod(lambda a, o, p: DenseSolvers._f("DenseInvert_wrp", a, o, p))
    SignFunction = staticmethod(lambda a, o, p: DenseSolvers._f("DenseSignFunction_wrp", a, o, p))


class LoadBalancer:
    @staticmethod
    def PermuteMatrix(mat_in, mat_out, permutation, memorypool=None):
        pool = memorypool if memorypool is not None else PMatrixMemoryPool(mat_in)
        lib.PermuteMatrix_wrp(mat_in.ih, mat_out.ih, permutation.ih, pool.ih)

    @staticmethod
    def UndoPermuteMatrix(mat_in, mat_out, permutation, memorypool=None):
        pool = memorypool if memorypool is not None else PMatrixMemoryPool(mat_in)
        lib.UndoPermuteMatrix_wrp(mat_in.ih, mat_out.ih, permutation.ih, pool.ih)


class EigenBounds:
    @staticmethod
    def PowerBounds(matrix, solver_parameters):
        v = C.c_double()
        lib.PowerBounds_wrp(matrix.ih, C.byref(v), solver_parameters.ih)
        return v.value

    @staticmethod
    def GershgorinBounds(matrix):
        mn, mx = C.c_double(), C.c_double()
        lib.GershgorinBounds_wrp(matrix.ih, C.byref(mn), C.byref(mx))
        return mn.value, mx.value


def ActivateLogger(start_document=False, file_name=None):
    if file_name:
        raw, n = s(file_name)
        lib.ActivateLoggerFile_wrp(b(start_document), raw, n)
    else:
        lib.ActivateLogger_wrp(b(start_document))


def DeactivateLogger():
    lib.DeactivateLogger_wrp()


# ------------------------------------------------------------------ statistics (extensions)
def last_spgemm_stats():
    out = (C.c_longlong * 13)()
    a, t = C.c_float(), C.c_float()
    lib.ntpoly_amd_last_spgemm_stats(out, C.byref(a), C.byref(t))
    return dict(nnz_a=out[0], nnz_b=out[1], nnz_c=out[2], products=out[3], tmp_entries=out[4],
                bins=[out[5 + k] for k in range(6)], overflow=out[11], slab=out[12], ms_numeric=a.value, ms_total=t.value)


def band_order(M):
    """(new position of every index, bandwidth) of the bandwidth-reducing order of M's pattern (csrc/relabel.hip), or None"""
    pos = np.zeros(M.GetActualDimension(), dtype=np.int32)
    bw = C.c_longlong()
    lib.ntpoly_amd_band_order.restype = C.c_int
    ok = lib.ntpoly_amd_band_order(M.ih, pos.ctypes.data_as(C.c_void_p), C.byref(bw))
    return (pos, int(bw.value)) if ok else None


def fusion_counts():
    """(steps X*X, steps 2X - X*X) computed inside the SpGEMM kernel's epilogue and fused steps repeated unfused"""
    out = (C.c_longlong * 3)()
    lib.ntpoly_amd_fusion_counts(out)
    return dict(square=out[0], update=out[1], repeated=out[2])


def tile2_counts():
    """(multiplies computed in the two-block geometry of the MFMA kernel, launches of it repeated on k_spgemm_tile)"""
    out = (C.c_longlong * 2)()
    lib.ntpoly_amd_tile2_counts(out)
    return dict(done=out[0], repeated=out[1])


def band_scope_counts():
    """(solves run in a recovered band order across ranks, operands searched for a hidden band)"""
    out = (C.c_longlong * 2)()
    lib.ntpoly_amd_band_scope_counts(out)
    return dict(solves=out[0], searched=out[1])


def block_scope_counts():
    """(solves run in a block order across ranks, panel products of such solves on the block path)"""
    out = (C.c_longlong * 2)()
    lib.ntpoly_amd_block_scope_counts(out)
    return dict(solves=out[0], products=out[1])


def band_searches():
    """searches for a bandwidth-reducing order since start (one per sparsity pattern)"""
    out = C.c_longlong()
    lib.ntpoly_amd_band_searches(C.byref(out))
    return int(out.value)


def slab_algebra_counts():
    """operations of the solver loops done on matrices in slab form since start, and those that went back to compressed columns"""
    out = (C.c_longlong * 4)()
    lib.ntpoly_amd_slab_algebra_counts(out)
    return dict(products=out[0], merges=out[1], others=out[2], refusals=out[3])


def panel_product_counts():
    """products of slab sessions on more than one rank since start: done in slab form on every rank, declined"""
    out = (C.c_longlong * 3)()
    lib.ntpoly_amd_panel_product_counts(out)
    return dict(slab=out[0], declined=out[1], host_syncs=out[2])


def last_grouped_stats():
    """grouped LDS-hash path of the last SpGEMM (csrc/spgemm_grouped.hip)"""
    out = (C.c_longlong * 6)()
    r = C.c_double()
    lib.ntpoly_amd_last_grouped_stats(out, C.byref(r))
    return dict(used=int(out[0]), failed_cols=int(out[1]), groups=int(out[2]), level=int(out[3]), minhash=int(out[4]),
                tile_rows=int(out[5]), union_ratio=r.value)


def last_block_stats():
    """block path of the last SpGEMM (csrc/spgemm_block.hip: 16 x 16 tiles of a clustered index order on the FP64 matrix cores)"""
    out = (C.c_longlong * 3)()
    r = C.c_double()
    lib.ntpoly_amd_last_block_stats(out, C.byref(r))
    return dict(used=int(out[0]), tile_products=int(out[1]), candidates=int(out[2]), fill=r.value)


def last_spgemm_thin():
    """1: the last SpGEMM ran on the thin-left kernel (csrc/spgemm_thin.hip)"""
    lib.ntpoly_amd_last_spgemm_thin.restype = C.c_int
    return int(lib.ntpoly_amd_last_spgemm_thin())


def block_order(M):
    """position of every index in the block order the engine multiplies matrices of M's dimension in (made from M if none exists)"""
    pos = np.zeros(M.GetActualDimension(), dtype=np.int32)
    lib.ntpoly_amd_block_order.restype = C.c_int
    ok = lib.ntpoly_amd_block_order(M.ih, pos.ctypes.data_as(C.c_void_p))
    return pos if ok else None


def increment_identity(Identity, B, alpha):
    """IncrementMatrix(Identity, B, alpha) as the solver loops call it (in place where every column stores its diagonal)"""
    lib.ntpoly_amd_increment_identity(Identity.ih, B.ih, d(alpha))


def norm_axpby(A, B, alpha, beta):
    """MatrixNorm(alpha A + beta B) without forming it; None when the engine would form the difference"""
    r = C.c_double()
    lib.ntpoly_amd_norm_axpby.restype = C.c_int
    ok = lib.ntpoly_amd_norm_axpby(A.ih, B.ih, d(alpha), d(beta), C.byref(r))
    return r.value if ok else None


def column_fused_counts():
    """operations on compressed columns done without the merge pass (csrc/column_fused.hip) since start"""
    out = (C.c_longlong * 2)()
    lib.ntpoly_amd_column_fused_counts(out)
    return dict(identity_in_place=int(out[0]), norms_of_differences=int(out[1]))


def block_algebra_counts():
    """(operations of solver loops / C-ABI loops done on matrices in block form, fallbacks to compressed columns)"""
    out = (C.c_longlong * 2)()
    lib.ntpoly_amd_block_algebra_counts(out)
    return dict(operations=int(out[0]), fallbacks=int(out[1]))


def drop_block_caches():
    lib.ntpoly_amd_drop_block_caches()


def exchange_stats():
    """(halo exchanges of distributed multiplies so far, host synchronisations inside them, ALL host synchronisations of
    the process so far) -- the synchronisations are counted where the host waits (sync_stream in csrc/common.cpp)"""
    out = (C.c_longlong * 3)()
    lib.ntpoly_amd_exchange_stats(out)
    return int(out[0]), int(out[1]), int(out[2])


def reset_spgemm_accum():
    lib.ntpoly_amd_reset_spgemm_accum()


def spgemm_accum():
    out = (C.c_longlong * 3)()
    dd = (C.c_double * 3)()
    lib.ntpoly_amd_get_spgemm_accum(out, dd)
    return dict(calls=out[0], products=out[1], nnz_c=out[2], alg_bytes=dd[0], ms_numeric=dd[1], ms_total=dd[2])


def solver_trace():
    n = int(lib.ntpoly_amd_trace_iterations())
    v = np.zeros(n)
    e = np.zeros(n)
    sg = np.zeros(n)
    nz = np.zeros(n, dtype=np.int64)
    if n:
        lib.ntpoly_amd_trace_get(v.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p),
                                 sg.ctypes.data_as(C.c_void_p), nz.ctypes.data_as(C.c_void_p))
    a, c = C.c_double(), C.c_double()
    lib.ntpoly_amd_trace_times(C.byref(a), C.byref(c))
    return dict(iterations=n, value=v, energy=e, sigma=sg, nnz=nz, setup_ms=a.value, loop_ms=c.value)


def malloc_stats():
    n, ms = C.c_longlong(), C.c_double()
    lib.ntpoly_amd_malloc_stats(C.byref(n), C.byref(ms))
    return n.value, ms.value


def release_cache():
    """frees what the engine keeps on the device between solves (operand caches, kept transposes) -- ntpoly_amd_release_cache"""
    lib.ntpoly_amd_release_cache()


def memory():
    a, c = C.c_longlong(), C.c_longlong()
    lib.ntpoly_amd_memory(C.byref(a), C.byref(c))
    return a.value, c.value
