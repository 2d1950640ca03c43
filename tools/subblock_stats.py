#!/usr/bin/env python3
"""How empty are the tiles of the block path at 4 x 4 granularity?  (profiles/README.md 97: the 4x4x4 f64 matrix instruction
has the 16x16x4's rate at a quarter of its granularity.)  The L^3 lattice iterate after `--iters` TRS2 steps, in the engine's
block order; tile pairs (A tile, B tile) sharing their k block are sampled and three costs compared, in units of one 16x16x4
instruction: what the kernel issues now (slices of four k positions with entries on both sides), what four-block instructions in
the diagonal assignment would issue (an instruction only when one of its four sub-block pairs has entries on both sides, 1/4
each), and the bound of one 4x4x4 product per sub-block pair with entries on both sides (1/16 each).
    python3 tools/subblock_stats.py --lattice 32 --iters 8 > profiles/r05_subblock_stats.json"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lattice", type=int, default=32)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--samples", type=int, default=2000000)
    args = ap.parse_args()
    import ntpoly_amd as nt
    from gen import lattice_triplets
    from bench import trs2_step
    nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
    nt.set_option("spgemm_fma", 1); nt.set_option("block_path", 2)
    L = args.lattice; n = L ** 3; thr = 1e-8
    H = nt.Matrix_ps.from_triplets(n, *lattice_triplets(L))
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    I = nt.Matrix_ps(n); I.FillIdentity()
    X = nt.Matrix_ps(H); X.Scale(-1.0); X.Increment(I, e_max, 0.0); X.Scale(1.0 / (e_max - e_min))
    X2 = nt.Matrix_ps(n); pool = nt.PMatrixMemoryPool(H); tr = None
    for _ in range(args.iters):
        _, e, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr, tr)
    pos = np.asarray(nt.block_order(X), dtype=np.int64)
    col, row, val = X.triplets()
    pr, pc = pos[row - 1], pos[col - 1]
    tr_, tc_ = pr // 16, pc // 16
    nb = int(max(tr_.max(), tc_.max())) + 1
    tile = tr_ * nb + tc_
    bit = ((pr % 16) // 4) * 4 + ((pc % 16) // 4)          # (row group, column group) of the entry inside its tile
    ut, inv = np.unique(tile, return_inverse=True)
    mask = np.zeros(len(ut), dtype=np.int64)
    np.bitwise_or.at(mask, inv, np.int64(1) << bit)
    cnt = np.bincount(inv)
    t_r, t_c = ut // nb, ut % nb
    # masks as 4 x 4 boolean arrays: M[tile][rg][cg]
    M = ((mask[:, None] >> np.arange(16)[None, :]) & 1).reshape(-1, 4, 4).astype(bool)
    # A role: tile (rb, kb): rows = rg (a), k slice = cg (q).  B role: tile (kb, cb): k slice = rg (q), column group = cg (c)
    order_c = np.argsort(t_c, kind="stable"); order_r = np.argsort(t_r, kind="stable")
    startc = np.searchsorted(t_c[order_c], np.arange(nb + 1)); startr = np.searchsorted(t_r[order_r], np.arange(nb + 1))
    na, nbt = np.diff(startc), np.diff(startr)                 # A tiles in block column kb, B tiles in block row kb
    w = (na * nbt).astype(np.float64)
    rng = np.random.default_rng(7)
    kb = rng.choice(nb, size=args.samples, p=w / w.sum())
    ia = order_c[startc[kb] + (rng.random(args.samples) * na[kb]).astype(np.int64)]
    ib = order_r[startr[kb] + (rng.random(args.samples) * nbt[kb]).astype(np.int64)]
    A, B = M[ia], M[ib]                                        # A[s][a][q], B[s][q][c]
    colA = A.any(axis=1)                                       # [s][q]
    rowB = B.any(axis=2)                                       # [s][q]
    now = (colA & rowB).sum(axis=1).astype(np.float64)         # 16x16x4 instructions issued
    pair = A[:, :, :, None] & B.transpose(0, 1, 2)[:, None, :, :]   # [s][a][q][c]
    bound = pair.sum(axis=(1, 2, 3)) / 16.0
    diag = np.zeros(args.samples)
    blk = np.arange(4)
    for t in range(4):
        need = pair[:, blk, :, (blk + t) % 4]                 # [blk][s][q] (advanced indexing moves the index axis first)
        diag += need.any(axis=0).sum(axis=1) / 4.0
    print(json.dumps(dict(
        what="tile pairs sharing their k block, sampled; costs in units of one 16x16x4 matrix instruction per pair",
        lattice=L, n=n, iters=args.iters, nnz=int(len(val)), tiles=int(len(ut)), fill=float(cnt.mean() / 256.0),
        subblocks_nonempty_per_tile=float(M.sum(axis=(1, 2)).mean()), slices_nonempty_per_tile=float(M.any(axis=1).sum(axis=1).mean()),
        issued_now=float(now.mean()), four_block_diagonal=float(diag.mean()), per_subblock_bound=float(bound.mean()),
        ratio_diagonal=float(diag.mean() / now.mean()), ratio_bound=float(bound.mean() / now.mean()), samples=args.samples)))


if __name__ == "__main__":
    main()
