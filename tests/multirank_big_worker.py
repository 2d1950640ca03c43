#!/usr/bin/env python3
"""One rank of the multi-process engine test at sizes that matter (tests/test_gpu_multirank_big.py): BASELINE
configs[1] (N = 65 536, halfband 50) in natural order and under the load balancer's random relabelling
(LoadBalancerModule.F90:14-52), and a 24^3 lattice Hamiltonian -- one distributed product
(distributed_algebra_includes/MatrixMultiply.f90:92-267) and a few TRS2 iterations each.  RANK / WORLD_SIZE /
NTPOLY_AMD_COMM come from the environment; the ranks share ONE GPU and exchange through the shared-memory test transport.

Results of this size are compared through ADDITIVE digests: every entry (column, row, value bits) is mixed into a 64-bit
word and the words are summed modulo 2^64, so the digests of the panels add up to the digest of the whole matrix
whatever the number of ranks -- equal sums = the same entries with the same bits.

    python tests/multirank_big_worker.py <out-prefix>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

M1, M2, M3 = np.uint64(0x9E3779B97F4A7C15), np.uint64(0xC2B2AE3D27D4EB4F), np.uint64(0x165667B19E3779F9)


def digest(col, row, val):
    """(entries, additive digest of (column, row, value bits), additive digest of the pattern alone)"""
    with np.errstate(over="ignore"):
        c = col.astype(np.uint64) * M1
        r = row.astype(np.uint64) * M2
        p = (c ^ (r + (c >> np.uint64(29)))) * M3
        p ^= p >> np.uint64(31)
        v = np.ascontiguousarray(val, dtype=np.float64).view(np.uint64)
        e = (p + v * M1) * M2
        e ^= e >> np.uint64(33)
        return np.array([len(col), int(e.sum(dtype=np.uint64)), int(p.sum(dtype=np.uint64))], dtype=np.uint64)


def main():
    out = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import ntpoly_amd as nt
    from gen import banded_triplets, permuted_banded_triplets, lattice_triplets
    nt.init_comm(nt.get_unique_id(), rank, world)
    nt.ConstructGlobalProcessGrid(1, world, 1)
    nt.set_option("time_kernels", 1)
    res = {}
    cases = (("band", 65536, lambda c0, c1: banded_triplets(65536, 50, c0=c0, c1=c1)),
             ("perm", 65536, lambda c0, c1: permuted_banded_triplets(65536, 50, 42, c0=c0, c1=c1)),
             ("latt", 24 ** 3, lambda c0, c1: lattice_triplets(24, c0=c0, c1=c1)))
    only = os.environ.get("NTPOLY_AMD_BIG_ONLY", "")
    for tag, n, gen in cases:
        if only and tag not in only.split(","):
            continue
        A = nt.Matrix_ps(n)
        c0, c1 = A.local_columns()
        t = nt.TripletList_r()
        t.set_arrays(*gen(c0, c1))
        A.FillFromTripletList(t, prepartitioned=True)
        del t
        # ---- one distributed product, threshold 1e-8 (bit for bit the one-rank product: a column's arithmetic does not
        # depend on who owns it -- in the unfused mode always, in FMA mode as long as the k order is the label order)
        C = nt.Matrix_ps(n)
        C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
        st, gs, bs = nt.last_spgemm_stats(), nt.last_grouped_stats(), nt.last_block_stats()
        res[tag + "_kernel"] = np.array([int(bool(st.get("slab"))), int(bool(bs.get("used"))), int(bool(gs.get("used")))])
        res[tag + "_AA"] = digest(*C.triplets())
        res[tag + "_AA_scal"] = np.array([C.Trace(), C.Norm(), float(np.real(C.Dot(A)))])
        del C
        # ---- six TRS2 iterations (fixed count, monitor off): the energies of every iteration and the density
        Ident = nt.Matrix_ps(n)
        Ident.FillIdentity()
        p = nt.SolverParameters()
        p.SetThreshold(1e-8)
        p.SetConvergeDiff(1e-30)
        p.SetMaxIterations(6)
        p.SetMonitorConvergence(False)
        K = nt.Matrix_ps(n)
        f0, e0, b0, k0 = nt.fusion_counts(), nt.exchange_stats(), nt.band_scope_counts(), nt.block_scope_counts()
        energy, mu = nt.DensityMatrixSolvers.TRS2(A, Ident, n / 2.0, K, p)
        f1, e1, b1, k1 = nt.fusion_counts(), nt.exchange_stats(), nt.band_scope_counts(), nt.block_scope_counts()
        res[tag + "_trs2_block_scope"] = np.array([k1["solves"] - k0["solves"], k1["products"] - k0["products"]])
        bs2 = nt.last_block_stats()
        tr = nt.solver_trace()
        res[tag + "_trs2_log"] = np.array(tr["energy"])
        res[tag + "_trs2_sigma"] = np.array(tr["sigma"])
        res[tag + "_trs2_nnz"] = np.array(tr["nnz"])
        res[tag + "_trs2_scal"] = np.array([energy, mu])
        res[tag + "_trs2_fused"] = np.array([f1[k] - f0[k] for k in ("square", "update", "repeated")])
        res[tag + "_trs2_block"] = np.array([int(bool(bs2.get("used")))])
        res[tag + "_trs2_band_scope"] = np.array([b1["solves"] - b0["solves"], b1["searched"] - b0["searched"]])
        res[tag + "_trs2_syncs"] = np.array([e1[0] - e0[0], e1[1] - e0[1], e1[2] - e0[2]])
        kc, kr, kv = K.triplets()
        res[tag + "_K"] = digest(kc, kr, kv)
        res[tag + "_K_sums"] = np.array([float(np.sum(kv)), float(np.sum(kv * kv)), float(np.sum(np.abs(kv)))])
        del K, Ident, A
    np.savez(out + ".%d.npz" % rank, **res)
    nt.DestructGlobalProcessGrid()


if __name__ == "__main__":
    main()
