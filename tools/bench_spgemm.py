#!/usr/bin/env python3
"""A/B timing of the SpGEMM numeric-kernel variants in ONE process on the same operand
(cdna_hip_programming.md 5.4 rule 24): X after `--iters` TRS2 iterations at BASELINE configs[2]."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=262144)
    ap.add_argument("--halfband", type=int, default=100)
    ap.add_argument("--threshold", type=float, default=1e-8)
    ap.add_argument("--iters", type=int, default=6)
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--variants", type=str, default="400,351")
    ap.add_argument("--fma", type=int, default=0)
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--waves", type=int, default=0)
    ap.add_argument("--complex", type=int, default=0, help="Hermitian complex operand (BASELINE configs[4] family)")
    ap.add_argument("--permute", type=int, default=None, metavar="SEED", help="operand under a seeded random relabelling")
    args = ap.parse_args()
    import ntpoly_amd as nt
    from gen import banded_triplets, permuted_banded_triplets
    from bench import trs2_step
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("time_kernels", 1)
    nt.set_option("spgemm_fma", args.fma)
    nt.set_option("tile_rows", args.rows)
    nt.set_option("tile_waves", args.waves)
    n, h, thr = args.n, args.halfband, args.threshold
    if args.permute is None:
        col, row, val = banded_triplets(n, h, complex_=bool(args.complex))
    else:
        col, row, val = permuted_banded_triplets(n, h, args.permute, complex_=bool(args.complex))
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    del col, row, val
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    X = nt.Matrix_ps(H)
    X.Scale(-1.0)
    X.Increment(Ident, e_max, 0.0)
    X.Scale(1.0 / (e_max - e_min))
    X2 = nt.Matrix_ps(n)
    pool = nt.PMatrixMemoryPool(H)
    for _ in range(args.iters):
        if args.complex:   # the same recurrence through the public calls (the fused step is a real-arithmetic entry)
            X2.Gemm(X, X, pool, 1.0, 0.0, thr)
            if np.real(X.Trace()) > n / 2.0:
                X, X2 = X2, X
            else:
                X.Scale(2.0)
                X.Increment(X2, -1.0, thr)
        else:
            trs2_step(nt, X, X2, H, pool, n / 2.0, thr)
    variants = [int(v) for v in args.variants.split(",")]
    ref = None
    res = {v: [] for v in variants}
    for rep in range(args.reps):
        for v in variants:
            nt.set_option("spgemm_variant", v)
            X2.Gemm(X, X, pool, 1.0, 0.0, thr)
            st = nt.last_spgemm_stats()
            sig = (st["nnz_c"], X2.Dot(H), X2.Trace())
            if v not in (401, 402, 403, 404) and not (511 <= v <= 541 and v != 518):  # ablation variants compute garbage on purpose
                if ref is None:
                    ref = sig
                assert sig == ref or args.fma, ("variant %d differs" % v, sig, ref)
            res[v].append((st["ms_numeric"], st["ms_total"]))
    st = nt.last_spgemm_stats()
    print("grouped path:", nt.last_grouped_stats())
    print("max span per bin:", nt.last_spgemm_stats())
    print("operand nnz %d, products %.3e, nnz_c %d, bins %s" % (st["nnz_a"], st["products"], st["nnz_c"], st["bins"]))
    for v in variants:
        a = np.array(res[v])
        print("variant %d: numeric ms min %.3f med %.3f | total ms min %.3f med %.3f | %.3e products/s" % (
            v, a[:, 0].min(), np.median(a[:, 0]), a[:, 1].min(), np.median(a[:, 1]), st["products"] / (a[:, 0].min() * 1e-3)))


if __name__ == "__main__":
    main()
