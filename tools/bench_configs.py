#!/usr/bin/env python3
"""Timings of the other BASELINE configs on one MI355X (documentation numbers for DESIGN.md section 7; the driver's
bench.py measures configs[2]).  Prints one JSON object.

  configs[1]  local A*A, banded N = 65 536, h = 50 (101 nnz/row), threshold 0 and 1e-8
  configs[3]  the N = 1 048 576, h = 100 operand as ONE A*A on one GPU (the 8-GPU config's whole problem)
  configs[4]  Hermitian complex N = 131 072, h = 50: SignFunction of the indefinite H, InverseSquareRoot of H + 2I
  configs[2] operand under the other solvers of the path: TRS4, real SignFunction / InverseSquareRoot (N = 262 144)

  --arithmetic fma|unfused   option spgemm_fma (real operands; complex ones always run unfused)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arithmetic", choices=("fma", "unfused"), default="fma")
    ap.add_argument("--only", default="", help="comma-separated subset: products,config3,lattice,solvers,complex")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    import ntpoly_amd as nt
    from gen import banded_triplets, permuted_banded_triplets
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("time_kernels", 1)
    nt.set_option("spgemm_fma", 1 if args.arithmetic == "fma" else 0)
    out = {"arithmetic": args.arithmetic}

    def want(tag):
        return not only or tag in only

    def product(n, h, thr, reps=5, complex_=False, permute=None, lattice=None):
        if lattice is not None:
            from gen import lattice_triplets
            n = lattice ** 3
            col, row, val = lattice_triplets(lattice)
        elif permute is None:
            col, row, val = banded_triplets(n, h, complex_=complex_)
        else:
            col, row, val = permuted_banded_triplets(n, h, permute, complex_=complex_)
        A = nt.Matrix_ps.from_triplets(n, col, row, val)
        del col, row, val
        C = nt.Matrix_ps(n)
        C.Gemm(A, A, None, 1.0, 0.0, thr)   # warm-up (allocator, first-touch)
        # wall time of MatrixMultiply_ps_wrp with the statistics off (they cost a pass and a read-back of their own) ...
        nt.set_option("time_kernels", 0)
        C.Gemm(A, A, None, 1.0, 0.0, thr)
        dt = None
        for _ in range(reps):
            nt.synchronize()
            t0 = time.perf_counter()
            C.Gemm(A, A, None, 1.0, 0.0, thr)
            nt.synchronize()
            d1 = time.perf_counter() - t0
            dt = d1 if dt is None else min(dt, d1)
        # ... and the kernel time and the product count from a call with them on
        nt.set_option("time_kernels", 1)
        C.Gemm(A, A, None, 1.0, 0.0, thr)
        st = nt.last_spgemm_stats()
        per = 20 if complex_ else 12
        alg = per * (st["nnz_a"] + st["nnz_b"] + st["nnz_c"]) + 4 * (3 * n + 3)
        gs = nt.last_grouped_stats()
        bs = nt.last_block_stats()
        return dict(n=n, halfband=h, threshold=thr, permute_seed=permute, grouped_hash=int(gs.get("used", 0)), block_path=bs["used"],
                    block_fill=bs["fill"], block_tile_products=bs["tile_products"], wall_ms=1e3 * dt, kernel_ms=st["ms_numeric"], nnz_out=st["nnz_c"],
                    products=st["products"], nnz_out_per_s=st["nnz_c"] / dt, products_per_s=st["products"] / (st["ms_numeric"] * 1e-3),
                    alg_GBps_kernel=alg / (st["ms_numeric"] * 1e-3) / 1e9, slab=st["slab"])

    if want("products"):
        out["config1_thr0"] = product(65536, 50, 0.0)
        out["config1_thr1e-8"] = product(65536, 50, 1e-8)
        out["config2_one_product"] = product(262144, 100, 1e-8)
    if want("config3"):
        out["config3_one_product_1gpu"] = product(1048576, 100, 1e-8, reps=3)
    # the same operand under the seeded relabelling ("with and without random permutation"): one product on
    # the grouped LDS-hash kernel (a single multiply has no loop to amortise a recovered band order over)
        out["config3_one_product_1gpu_relabelled"] = product(1048576, 100, 1e-8, reps=3, permute=42)
    if want("lattice"):
        # one product H * H of the 64^3 lattice Hamiltonian (no band: the block path, csrc/spgemm_block.hip)
        out["lattice64_one_product"] = product(0, 0, 1e-8, reps=3, lattice=64)
    # TRS2 on the configs[3] operand, natural order and relabelled (label-ordered slab steps)
    nt.set_option("time_kernels", 0)   # (whole solves below: wall time, no per-product statistics)
    for tag, perm in (("config3_trs2_1gpu", None), ("config3_trs2_1gpu_relabelled", 42)) if want("config3") else ():
        n3, h3 = 1048576, 100
        col, row, val = banded_triplets(n3, h3) if perm is None else permuted_banded_triplets(n3, h3, perm)
        H3 = nt.Matrix_ps.from_triplets(n3, col, row, val)
        del col, row, val
        I3 = nt.Matrix_ps(n3)
        I3.FillIdentity()
        res = []
        for iters in (4, 24, 4, 24):   # (the first two solves warm the allocator up and, relabelled, find the band order)
            K3 = nt.Matrix_ps(n3)
            p3 = nt.SolverParameters()
            p3.SetThreshold(1e-8)
            p3.SetConvergeDiff(1e-30)
            p3.SetMaxIterations(iters)
            p3.SetMonitorConvergence(False)
            nt.synchronize()
            t0 = time.perf_counter()
            e3, _ = nt.DensityMatrixSolvers.TRS2(H3, I3, n3 / 2.0, K3, p3)
            nt.synchronize()
            res.append((time.perf_counter() - t0, e3, K3.GetSize()))
            del K3
        out[tag] = dict(n=n3, halfband=h3, threshold=1e-8, permute_seed=perm, ms_per_iteration=1e3 * (res[3][0] - res[2][0]) / 20,
                        wall_s_first_solve_4_iterations=res[0][0], wall_s_24_iterations=res[3][0], energy_24=res[3][1],
                        nnz_K_24=res[3][2])
        del H3, I3

    def timed_solver(fn, n):
        O = nt.Matrix_ps(n)
        fn(O)   # warm-up
        dt = None
        for _ in range(3):   # (best of three)
            nt.synchronize()
            t0 = time.perf_counter()
            fn(O)
            nt.synchronize()
            d1 = time.perf_counter() - t0
            dt = d1 if dt is None else min(dt, d1)
        tr = nt.solver_trace()
        return dict(n=n, wall_s=dt, iterations=tr["iterations"], s_per_iteration=dt / max(1, tr["iterations"]),
                    nnz_result=O.GetSize())

    if want("solvers"):
        # the headline operand under the other solvers the path serves (real arithmetic): TRS4 by differencing two
        # iteration caps, SignFunction on the indefinite H, InverseSquareRoot on H + 2I
        n, h, thr = 262144, 100, 1e-8
        col, row, val = banded_triplets(n, h)
        H = nt.Matrix_ps.from_triplets(n, col, row, val)
        I = nt.Matrix_ps(n)
        I.FillIdentity()
        res = []
        for iters in (4, 14, 4, 14):
            K = nt.Matrix_ps(n)
            p = nt.SolverParameters()
            p.SetThreshold(thr)
            p.SetConvergeDiff(1e-30)
            p.SetMaxIterations(iters)
            p.SetMonitorConvergence(False)
            nt.synchronize()
            t0 = time.perf_counter()
            e, _ = nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, K, p)
            nt.synchronize()
            res.append((time.perf_counter() - t0, e, K.GetSize()))
            del K
        out["config2_trs4"] = dict(n=n, halfband=h, threshold=thr, ms_per_iteration=1e3 * (res[3][0] - res[2][0]) / 10,
                                   energy_14=res[3][1], nnz_K_14=res[3][2])
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        p.SetConvergeDiff(1e-8)
        out["config2_sign_real_indefinite"] = timed_solver(lambda o: nt.SignSolvers.ComputeSign(H, o, p), n)
        H.Increment(I, 2.0, 0.0)
        out["config2_inverse_square_root_real"] = timed_solver(lambda o: nt.SquareRootSolvers.InverseSquareRoot(H, o, p), n)
        del H, I

    if want("complex"):
        n, h, thr = 131072, 50, 1e-8
        p = nt.SolverParameters()
        p.SetThreshold(thr)
        p.SetConvergeDiff(1e-8)
        # SignFunction of the INDEFINITE Hermitian H (the operand of tests/test_gpu_scale.py::
        # test_config4_sign_of_the_indefinite_operand: eigenvalues of both signs, sign(H) far from the identity)
        col, row, val = banded_triplets(n, h, complex_=True)
        H = nt.Matrix_ps.from_triplets(n, col, row, val)
        del col, row, val
        r = timed_solver(lambda o: nt.SignSolvers.ComputeSign(H, o, p), n)
        r.update(halfband=h, threshold=thr, operand="H (indefinite)")
        out["config4_sign"] = r
        del H
        col, row, val = banded_triplets(n, h, complex_=True, shift=2.0)
        H = nt.Matrix_ps.from_triplets(n, col, row, val)
        del col, row, val
        p.SetConvergeDiff(1e-10)
        r = timed_solver(lambda o: nt.SquareRootSolvers.InverseSquareRoot(H, o, p), n)
        r.update(halfband=h, threshold=thr, operand="H + 2 I")
        out["config4_inverse_square_root"] = r
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
