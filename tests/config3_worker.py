#!/usr/bin/env python3
"""One rank of tests/test_gpu_config3.py: BASELINE configs[3]'s operand (N = 1 048 576, halfband 100) as column panels on
WORLD_SIZE ranks sharing the box's GPU over the shared-memory test transport, ONE distributed A * A in the library's default
arithmetic (distributed_algebra_includes/MatrixMultiply.f90:92-267), the panel of the result as an additive digest
(tests/multirank_big_worker.py: the digests of the panels add up to the digest of the whole product).

    python tests/config3_worker.py <out-prefix>
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out = sys.argv[1]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import ntpoly_amd as nt
    from gen import banded_triplets
    from multirank_big_worker import digest
    n, h = 1048576, 100
    nt.init_comm(nt.get_unique_id(), rank, world)
    nt.ConstructGlobalProcessGrid(1, world, 1)
    A = nt.Matrix_ps(n)
    c0, c1 = A.local_columns()
    t = nt.TripletList_r()
    t.set_arrays(*banded_triplets(n, h, c0=c0, c1=c1))
    A.FillFromTripletList(t, prepartitioned=True)
    del t
    C = nt.Matrix_ps(n)
    C.Gemm(A, A, None, 1.0, 0.0, 1e-8)
    st, gs, bs = nt.last_spgemm_stats(), nt.last_grouped_stats(), nt.last_block_stats()
    res = {"kernel": np.array([int(bool(st.get("slab"))), int(bool(bs.get("used"))), int(bool(gs.get("used")))]),
           "AA": digest(*C.triplets())}
    np.savez(out + ".%d.npz" % rank, **res)
    del C, A
    nt.DestructGlobalProcessGrid()


if __name__ == "__main__":
    main()
