#!/usr/bin/env python3
"""Timing experiments on the headline iterate: one TRS2 step on a COPY of the iterate after 10 normal steps, per setting
(kernel time from the engine's HIP-event timers).  Ablated launches (NTPOLY_AMD_T2_ABLATE) give wrong results by design."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ntpoly_amd as nt
from gen import banded_triplets
nt.init_comm(); nt.ConstructGlobalProcessGrid(1, 1, 1)
nt.set_option("time_kernels", 1)
n, h, thr = 262144, 100, 1e-8
H = nt.Matrix_ps.from_triplets(n, *banded_triplets(n, h))
e_min, e_max = nt.EigenBounds.GershgorinBounds(H)
I = nt.Matrix_ps(n); I.FillIdentity()
X = nt.Matrix_ps(H); X.Scale(-1.0); X.Increment(I, e_max, 0.0); X.Scale(1.0 / (e_max - e_min))
X2 = nt.Matrix_ps(n)
tr = None
for _ in range(int(os.environ.get("T2_WARM", "10"))):
    _, e, tr = nt.trs2_step(X, X2, H, n / 2.0, thr, tr)
settings = [("tile2", {"tile2": 1}, "0"), ("tile", {"tile2": 0}, "0"), ("tile2 no epilogue", {"tile2": 1}, "1"), ("tile2 again", {"tile2": 1}, "0")]
if os.environ.get('T2_ONLY'):
    settings = [s_ for s_ in settings if s_[0] in (('tile2',) if os.environ['T2_ONLY'] == '2' else ('tile2', 'tile2 no epilogue'))]
for name, opts, abl in settings:
    for k, v in opts.items():
        nt.set_option(k, v)
    os.environ["NTPOLY_AMD_T2_ABLATE"] = abl
    ms = []
    for rep in range(3):
        Y = nt.Matrix_ps(X)
        Y2 = nt.Matrix_ps(n)
        # bring the copy into the step's form (slab form) with one normal... the copy of a slab-form matrix is packed: the
        # first step on it runs from compressed columns; time the SECOND step
        os.environ["NTPOLY_AMD_T2_ABLATE"] = "0"
        _, e, t2 = nt.trs2_step(Y, Y2, H, n / 2.0, thr, tr)
        _, e, t2 = nt.trs2_step(Y, Y2, H, n / 2.0, thr, t2)     # (the step after the first carries runs only)
        os.environ["NTPOLY_AMD_T2_ABLATE"] = abl
        c0 = nt.tile2_counts()
        nt.synchronize()
        nt.reset_spgemm_accum()
        _, e, t3 = nt.trs2_step(Y, Y2, H, n / 2.0, thr, t2)
        nt.synchronize()
        acc = nt.spgemm_accum()
        ms.append(acc["ms_numeric"])
        used = nt.tile2_counts()["done"] - c0["done"]
        del Y, Y2
    print("%-20s kernel ms %s (two-block geometry used: %d)" % (name, " ".join("%.3f" % m for m in ms), used), flush=True)
os.environ["NTPOLY_AMD_T2_ABLATE"] = "0"
