#!/usr/bin/env python3
"""A roofline line per solver loop of the path: the products of iterations 5..14 of TRS4,
SignFunction and InverseSquareRoot on the headline operand (real, N = 262 144, h = 100, threshold 1e-8) and of
SignFunction / InverseSquareRoot on the configs[4] operand (complex, N = 131 072, h = 50) -- algorithmic bytes of the
SpGEMMs (12 (20 complex) x (nnzA + nnzB + nnzC) + 4 x (columns)) over the time of their numeric kernels,
against 8 TB/s, plus the wall time per iteration with everything else in it.  The counters are differences between a
solve capped at 14 and one capped at 4 iterations (kernel timers on).
    python3 tools/solver_roofline.py [fma|unfused] > profiles/r03_solver_roofline.json"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import ntpoly_amd as nt
from gen import banded_triplets
arith = sys.argv[1] if len(sys.argv) > 1 else "fma"
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("spgemm_fma", 1 if arith == "fma" else 0)
out = {"arithmetic": arith, "peak_GBps": 8000.0, "solvers": {}}


def run(tag, solver, n, h, cplx, shift):
    col, row, val = banded_triplets(n, h, complex_=cplx, shift=shift)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    I = nt.Matrix_ps(n); I.FillIdentity()
    res = {}
    for timed in (0, 1):          # wall time with the statistics off, counters with them on
        nt.set_option("time_kernels", timed)
        for rep in range(2):
            for iters in (4, 14):
                K = nt.Matrix_ps(n)
                p = nt.SolverParameters(); p.SetThreshold(1e-8); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
                nt.reset_spgemm_accum(); nt.synchronize(); t0 = time.perf_counter()
                if solver == "trs4":
                    nt.DensityMatrixSolvers.TRS4(H, I, n / 2.0, K, p)
                elif solver == "sign":
                    nt.SignSolvers.ComputeSign(H, K, p)
                else:
                    nt.SquareRootSolvers.InverseSquareRoot(H, K, p)
                nt.synchronize()
                res[(timed, iters)] = (time.perf_counter() - t0, nt.spgemm_accum() if timed else None)
                del K
    wall = (res[(0, 14)][0] - res[(0, 4)][0]) / 10
    a14, a4 = res[(1, 14)][1], res[(1, 4)][1]
    calls = (a14["calls"] - a4["calls"]) / 10
    alg = (a14["alg_bytes"] - a4["alg_bytes"]) / 10
    ms = (a14["ms_numeric"] - a4["ms_numeric"]) / 10
    out["solvers"][tag] = dict(n=n, halfband=h, complex=bool(cplx), operand="H" if shift == 0 else "H + %g I" % shift,
                               ms_per_iteration_wall=1e3 * wall, products_per_iteration=calls,
                               spgemm_alg_bytes_per_iteration=alg, spgemm_kernel_ms_per_iteration=ms,
                               roofline=dict(bound="hbm", achieved=alg / (ms * 1e-3) / 1e9 if ms > 0 else None, peak=8000.0, unit="GB/s",
                                             frac=(alg / (ms * 1e-3) / 1e9 / 8000.0) if ms > 0 else None),
                               share_of_wall_in_spgemm_kernels=ms / (1e3 * wall) if wall > 0 else None)
    del H, I


run("trs4_real_headline", "trs4", 262144, 100, False, 0.0)
run("sign_real_headline_indefinite", "sign", 262144, 100, False, 0.0)
run("inverse_square_root_real_headline", "isr", 262144, 100, False, 2.0)
run("sign_complex_config4_indefinite", "sign", 131072, 50, True, 0.0)
run("inverse_square_root_complex_config4", "isr", 131072, 50, True, 2.0)
print(json.dumps(out, indent=1))
