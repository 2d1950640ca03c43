#!/usr/bin/env python3
"""What a TRS2 solve of the RELABELLED headline operand does on two ranks sharing one GPU (shared-memory test transport):
times of solves capped at 5 / 15 / 25 iterations, band-scope and panel-step counters.  Without RANK in the environment it
starts the ranks itself:  NTPOLY_AMD_SHM_MB=1024 python tools/scope_diag.py [ranks] [n] [natural]"""
import os
import subprocess
import sys
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if "RANK" not in os.environ:
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    name = "d%s" % uuid.uuid4().hex[:12]
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", NTPOLY_AMD_COMM="shm:" + name,
                                       NTPOLY_AMD_SHM_MB=os.environ.get("NTPOLY_AMD_SHM_MB", "1024")))
             for r in range(world)]
    rc = 0
    for p in procs:
        rc |= p.wait()
    try:
        os.unlink("/dev/shm/ntpoly_amd_" + name)
    except OSError:
        pass
    sys.exit(rc)

import ntpoly_amd as nt
from gen import permuted_banded_triplets, banded_triplets
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
h, thr = 100, 1e-8
nt.init_comm(nt.get_unique_id(), rank, world)
nt.ConstructGlobalProcessGrid(1, world, 1)
H = nt.Matrix_ps(n)
c0, c1 = H.local_columns()
t = nt.TripletList_r()
natural = len(sys.argv) > 3 and sys.argv[3] == "natural"
t.set_arrays(*(banded_triplets(n, h, c0=c0, c1=c1) if natural else permuted_banded_triplets(n, h, 42, c0=c0, c1=c1)))
H.FillFromTripletList(t, prepartitioned=True)
I = nt.Matrix_ps(n); I.FillIdentity()
if len(sys.argv) > 3 and sys.argv[3] == "gather":
    # the natural operand: a solve, then ONE large transfer through the transport (the whole matrix gathered), then the solve again
    H0 = nt.Matrix_ps(n)
    t0_ = nt.TripletList_r()
    t0_.set_arrays(*banded_triplets(n, h, c0=c0, c1=c1))
    H0.FillFromTripletList(t0_, prepartitioned=True)
    def solve(tag):
        for iters in (5, 15):
            K = nt.Matrix_ps(n)
            p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
            nt.synchronize(); t0 = time.perf_counter()
            nt.DensityMatrixSolvers.TRS2(H0, I, n / 2.0, K, p)
            nt.synchronize()
            if rank == 0:
                print("%s, iterations %2d: %.3f s" % (tag, iters, time.perf_counter() - t0), flush=True)
    solve("natural operand, fresh processes")
    G = H0.GatherMatrixToProcess()
    del G
    solve("natural operand after one gather of the whole matrix through the transport")
    if os.environ.get("SCOPE_DIAG_LONG_KERNEL"):
        import torch
        a = torch.randn(24576, 24576, device="cuda")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        b = a @ a
        torch.cuda.synchronize()
        if rank == 0:
            print("one long kernel in each process (a dense product through torch): %.3f s" % (time.perf_counter() - t0), flush=True)
        del a, b
        solve("natural operand after one long-running kernel in each process")
    perm = nt.Permutation(n)
    perm.SetRandomPermutation()
    Hp = nt.Matrix_ps(n)
    nt.LoadBalancer.PermuteMatrix(H0, Hp, perm)
    del Hp
    solve("natural operand after a random permutation of the matrix (redistribution through sends and receives)")
    nt.lib.ntpoly_amd_finalize_comm()
    sys.exit(0)
both = len(sys.argv) > 3 and sys.argv[3] == "both"   # (relabelled solves first, then the natural operand in the same processes)
for iters in ((5, 5, 15, 25, 5, 25) if not both else (5, 10)):
    K = nt.Matrix_ps(n)
    p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
    nt.synchronize(); nt.barrier() if hasattr(nt, "barrier") else None
    b0 = nt.band_scope_counts(); t0 = time.perf_counter()
    e, _ = nt.DensityMatrixSolvers.TRS2(H, I, n / 2.0, K, p)
    nt.synchronize(); t1 = time.perf_counter()
    if rank == 0:
        print("iterations %2d: %.3f s, energy %.6f, band scope counts %s -> %s" % (iters, t1 - t0, e, list(b0), list(nt.band_scope_counts())), flush=True)

if both:
    if os.environ.get("SCOPE_DIAG_RELEASE"):
        nt.release_cache()
        nt.synchronize()
        if rank == 0:
            print("caches released: in use / cached", nt.memory(), flush=True)
    elif rank == 0:
        print("in use / cached", nt.memory(), flush=True)
    H2 = nt.Matrix_ps(n)
    t2 = nt.TripletList_r()
    t2.set_arrays(*banded_triplets(n, h, c0=c0, c1=c1))
    H2.FillFromTripletList(t2, prepartitioned=True)
    for iters in (5, 10):
        K = nt.Matrix_ps(n)
        p = nt.SolverParameters(); p.SetThreshold(thr); p.SetConvergeDiff(1e-30); p.SetMaxIterations(iters); p.SetMonitorConvergence(False)
        nt.synchronize(); t0 = time.perf_counter()
        e, _ = nt.DensityMatrixSolvers.TRS2(H2, I, n / 2.0, K, p)
        nt.synchronize(); t1 = time.perf_counter()
        if rank == 0:
            print("natural operand after the relabelled solves, iterations %2d: %.3f s" % (iters, t1 - t0), flush=True)
del H, I
nt.synchronize()
try:
    del K
except NameError:
    pass
nt.lib.ntpoly_amd_finalize_comm()
