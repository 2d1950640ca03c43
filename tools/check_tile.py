#!/usr/bin/env python3
"""MFMA tile kernel (option spgemm_fma = 1) against the v_fma_f64 loop of the register-slab kernel (spgemm_fma = 3):
the same FMA chain in ascending k, so products must agree BIT FOR BIT; TRS2 steps through the fused tile epilogues
against the separate passes of the FMA loop: same sigma, entry counts and iterate, energies to 1e-12.  Prints timings."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def setup(nt, n, h):
    from gen import banded_triplets
    col, row, val = banded_triplets(n, h)
    H = nt.Matrix_ps.from_triplets(n, col, row, val)
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    Ident = nt.Matrix_ps(n)
    Ident.FillIdentity()
    X = nt.Matrix_ps(H)
    X.Scale(-1.0)
    X.Increment(Ident, e_max, 0.0)
    X.Scale(1.0 / (e_max - e_min))
    return H, X


def trip(M):
    c, r, v = M.triplets()
    o = np.lexsort((r, c))
    return c[o], r[o], v[o]


def same(a, b):
    return all(x.shape == y.shape and np.array_equal(x, y) for x, y in zip(a, b))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=65536)
    ap.add_argument("--halfband", type=int, default=100)
    ap.add_argument("--threshold", type=float, default=1e-8)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--rows", type=int, default=2)
    ap.add_argument("--waves", type=int, default=0)
    args = ap.parse_args()
    import ntpoly_amd as nt
    from bench import trs2_step
    nt.init_comm()
    nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("time_kernels", 1)
    nt.set_option("tile_rows", args.rows)
    nt.set_option("tile_waves", args.waves)
    n, h, thr = args.n, args.halfband, args.threshold
    ok = True
    # ---- single products on an iterate
    H, X = setup(nt, n, h)
    X2 = nt.Matrix_ps(n)
    pool = nt.PMatrixMemoryPool(H)
    for _ in range(3):
        trs2_step(nt, X, X2, H, pool, n / 2.0, thr)
    out = {}
    for fma in (3, 1, 0):
        nt.set_option("spgemm_fma", fma)
        best = (1e9, 1e9)
        for _ in range(args.reps):
            X2.Gemm(X, X, pool, 1.0, 0.0, thr)
            st = nt.last_spgemm_stats()
            best = min(best, (st["ms_numeric"], st["ms_total"]))
        out[fma] = trip(X2)
        print("product fma=%d: nnz_c %d numeric %.3f ms total %.3f ms" % (fma, st["nnz_c"], best[0], best[1]), flush=True)
    eq = same(out[1], out[3])
    print("product: tile (fma=1) == fma loop (fma=3) bit for bit:", eq)
    ok &= eq
    d = np.max(np.abs(out[1][2] - out[0][2])) if out[1][2].shape == out[0][2].shape else float("nan")
    print("product: tile vs unfused default: same pattern %s, max |diff| %.3e" % (out[1][2].shape == out[0][2].shape, d))
    # threshold 0 and alpha != 1
    for alpha, t in ((1.0, 0.0), (-0.75, 1e-6)):
        res = {}
        for fma in (3, 1):
            nt.set_option("spgemm_fma", fma)
            X2.Gemm(X, X, pool, alpha, 0.0, t)
            res[fma] = trip(X2)
        eq = same(res[1], res[3])
        print("product alpha=%g thr=%g: bitwise equal %s (nnz %d)" % (alpha, t, eq, res[1][2].size))
        ok &= eq
    # ---- TRS2 steps
    logs = {}
    for fma in (3, 1):
        nt.set_option("spgemm_fma", fma)
        H, X = setup(nt, n, h)
        X2 = nt.Matrix_ps(n)
        log = []
        nt.synchronize()
        t0 = time.perf_counter()
        for it in range(args.iters):
            s, e, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr)
            log.append((s, e, tr, X.GetSize()))
        nt.synchronize()
        dt = (time.perf_counter() - t0) / args.iters
        logs[fma] = (log, trip(X))
        print("trs2 fma=%d: %.3f ms/iter; fusion counts %s" % (fma, dt * 1e3, nt.fusion_counts() if hasattr(nt, "fusion_counts") else "?"), flush=True)
    la, lb = logs[3][0], logs[1][0]
    for it, (a, b) in enumerate(zip(la, lb)):
        good = a[0] == b[0] and a[3] == b[3] and abs(a[1] - b[1]) <= 1e-12 * abs(a[1]) and abs(a[2] - b[2]) <= 1e-12 * abs(a[2])
        if not good:
            print("  iteration %d differs: %r vs %r" % (it, a, b))
        ok &= good
    eq = same(logs[1][1], logs[3][1])
    print("trs2: final iterate bit for bit:", eq, "entries", logs[1][1][2].size)
    ok &= eq
    print("ALL OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
