#!/usr/bin/env python3
"""Block path (csrc/spgemm_block.hip) on the lattice iterate: X after `--iters` TRS2 iterations on the L^3 lattice
Hamiltonian, then `--reps` products X * X through MatrixMultiply_ps_wrp; prints the numeric kernel's time, the tile
statistics and the matrix-core efficiency (useful products / issued).  NTPOLY_AMD_BS_ABLATE=1/2/3 with a library built
with -DNTP_ABLATIONS (NTPOLY_AMD_LIB) runs the experiments without A loads / B loads / matrix instructions."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lattice", type=int, default=64)
    ap.add_argument("--threshold", type=float, default=1e-8)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ablate", type=int, default=0, help="experiment build only: applied to the timed products, not to the iterations before them")
    args = ap.parse_args()
    import ntpoly_amd as nt
    from gen import lattice_triplets
    from bench import trs2_step
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("time_kernels", 1)
    nt.set_option("spgemm_fma", 1)
    nt.set_option("slab_algebra", 0)
    L, thr = args.lattice, args.threshold
    n = L ** 3
    H = nt.Matrix_ps.from_triplets(n, *lattice_triplets(L))
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    X = nt.Matrix_ps(H)
    X.Scale(-1.0)
    X.Increment(Ident, e_max, 0.0)
    X.Scale(1.0 / (e_max - e_min))
    X2 = nt.Matrix_ps(n)
    pool = nt.PMatrixMemoryPool(H)
    tr = None
    for _ in range(args.iters):
        _, e, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr, tr)
    out = []
    if args.ablate:
        os.environ["NTPOLY_AMD_BS_ABLATE"] = str(args.ablate)
    for r in range(args.reps):
        C = nt.Matrix_ps(n)
        C.Gemm(X, X, pool, 1.0, 0.0, thr)
        nt.synchronize()
        st = nt.last_spgemm_stats()
        bs = nt.last_block_stats()
        issued = bs["tile_products"] * 4096.0
        out.append(dict(ms_numeric=st["ms_numeric"], ms_total=st["ms_total"], nnz_a=st["nnz_a"], nnz_c=st["nnz_c"], products=st["products"],
                        tile_products=bs["tile_products"], fill=bs["fill"], candidates=bs["candidates"],
                        useful_fraction=st["products"] / issued if issued else None,
                        mfma_tflops_issued=2 * issued / (st["ms_numeric"] * 1e-3) / 1e12 if st["ms_numeric"] else None,
                        useful_tflops=2 * st["products"] / (st["ms_numeric"] * 1e-3) / 1e12 if st["ms_numeric"] else None))
        del C
    print(json.dumps(dict(lattice=L, n=n, ablate=args.ablate, reps=out)))


if __name__ == "__main__":
    main()
