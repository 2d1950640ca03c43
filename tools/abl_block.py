#!/usr/bin/env python3
"""Ablation builds of the block kernel (compile-time variants, -DNTP_BS_ABL=k: a run-time switch inside k_bs_numeric costs it
30-100 %): make the 64^3 iterate with the plain build and write it out (MODE=make), then time products X * X of the file
with the library NTPOLY_AMD_LIB names (MODE=time).  The ablated products are WRONG by construction; only their time counts."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("ABL_ROOT", ROOT)); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntpoly_amd as nt
from gen import lattice_triplets
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("time_kernels", 1); nt.set_option("spgemm_fma", 1); nt.set_option("slab_algebra", 0)
L = int(os.environ.get("LATTICE", "64")); n = L ** 3; thr = 1e-8
path = os.environ.get("XFILE", "/tmp/x_iterate.bin")
if os.environ.get("MODE", "time") == "make":
    sys.path.insert(0, os.environ.get("ABL_ROOT", ROOT))
    from bench import trs2_step
    H = nt.Matrix_ps.from_triplets(n, *lattice_triplets(L))
    e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
    I = nt.Matrix_ps(n); I.FillIdentity()
    X = nt.Matrix_ps(H); X.Scale(-1.0); X.Increment(I, e_max, 0.0); X.Scale(1.0 / (e_max - e_min))
    X2 = nt.Matrix_ps(n); pool = nt.PMatrixMemoryPool(H); tr = None
    for _ in range(8):
        _, e, tr = trs2_step(nt, X, X2, H, pool, n / 2.0, thr, tr)
    X.WriteToBinary(path)
    print("wrote", path, X.GetSize())
else:
    X = nt.Matrix_ps(path)
    pool = nt.PMatrixMemoryPool(X)
    ms = []
    for r in range(4):
        C = nt.Matrix_ps(n)
        C.Gemm(X, X, pool, 1.0, 0.0, thr)
        nt.synchronize()
        ms.append(nt.last_spgemm_stats()["ms_numeric"])
        del C
    print(json.dumps(dict(lib=os.path.basename(os.environ.get("NTPOLY_AMD_LIB", "default")), ms_numeric=ms, block=nt.last_block_stats().get("used"))))
