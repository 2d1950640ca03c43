#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz by running the REAL NTPoly
reference (oracle/_ref/ref_driver, built by oracle/build_ref.py from /root/reference)
on seeded inputs.  Run in the build container only (needs /root/reference, flang,
MPICH); the .npz files it writes are committed and are all the GPU box needs.

    python tests/golden/make_golden.py            # regenerate everything

Every .npz holds inputs and reference outputs as NTPoly triplets (1-based col,row,val
sorted by column then row) plus scalars; `meta` is a JSON string with the parameters.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.trifile import from_scipy, read_tri, write_tri  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
DRV = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
ENV = dict(os.environ, OMP_NUM_THREADS="1", LD_LIBRARY_PATH="/opt/conda/lib")
MPIEXEC = "/opt/conda/bin/mpiexec"


def run(args, nranks=1):
    cmd = [DRV] + [str(a) for a in args]
    if nranks > 1:
        cmd = [MPIEXEC, "-n", str(nranks)] + cmd
    r = subprocess.run(cmd, env=ENV, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_driver failed: %s\n%s\n%s" % (cmd, r.stdout, r.stderr))
    return r.stdout


def tri(m):
    """scipy -> dict of triplet arrays"""
    c, r, v = from_scipy(sp.csc_matrix(m))
    return c, r, v


def put(d, name, shape, t):
    d[name + "_shape"] = np.array(shape, dtype=np.int32)
    d[name + "_col"], d[name + "_row"], d[name + "_val"] = t


def rnd(rng, rows, cols, density, complex_=False):
    m = sp.random(rows, cols, density, random_state=rng, format="csc",
                  data_rvs=lambda n: rng.uniform(-1, 1, n))
    if complex_:
        im = sp.random(rows, cols, density, random_state=rng, format="csc",
                       data_rvs=lambda n: rng.uniform(-1, 1, n))
        m = (m + 1j * im).tocsc()
    m.sort_indices()
    return m


def banded(n, h, complex_=False):
    """SURVEY 8(d) generator: H_ii = -1 + 2*((i*7919) mod 1000)/1000,
    H_ij = -0.25*exp(-0.05|i-j|)/|i-j| (1-based i); complex: times exp(i*0.1*(i-j))."""
    i = np.arange(1, n + 1)
    diags, offs = [(-1.0 + 2.0 * ((i * 7919) % 1000) / 1000.0)], [0]
    for d in range(1, h + 1):
        v = np.full(n - d, -0.25 * np.exp(-0.05 * d) / d)
        if complex_:
            diags += [v * np.exp(-1j * 0.1 * d), v * np.exp(1j * 0.1 * d)]
        else:
            diags += [v, v]
        offs += [d, -d]
    m = sp.diags(diags, offs, shape=(n, n), format="csc", dtype=complex if complex_ else float)
    m.sort_indices()
    return m


def save(name, d, meta):
    d["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print("wrote", name, {k: v for k, v in meta.items() if k != "cases"})


def local_gemm_case(tmp, A, B, Cin, tA, tB, alpha, beta, thr):
    write_tri(tmp + "/A.tri", A.shape[0], A.shape[1], *tri(A))
    write_tri(tmp + "/B.tri", B.shape[0], B.shape[1], *tri(B))
    cin = "none"
    if Cin is not None:
        write_tri(tmp + "/Cin.tri", Cin.shape[0], Cin.shape[1], *tri(Cin))
        cin = tmp + "/Cin.tri"
    run(["lgemm", tmp + "/A.tri", tmp + "/B.tri", cin, int(tA), int(tB), repr(alpha), repr(beta),
         repr(thr), tmp + "/C.tri"])
    return read_tri(tmp + "/C.tri")


def gen_local_gemm(tmp):
    """test_matrix.py:224-362 (multiply nn/nt/tn/tt/zero) with seeded inputs, plus sparse-branch
    cases (density < 10 %), alpha/beta/threshold, real + complex."""
    rng = np.random.default_rng(20240601)
    cases = []
    d = {}
    # the reference suite's own shapes (test_matrix.py:80-88): (rows, cols, density)
    ref_shapes = [(2, 4, 0.0), (8, 8, 0.0), (2, 2, 1.0), (4, 4, 1.0), (19, 19, 1.0), (4, 2, 1.0),
                  (2, 4, 1.0), (4, 4, 0.2), (8, 8, 1.0)]
    specs = []
    for (r, c, dens) in ref_shapes:
        for (tA, tB) in [(0, 0), (0, 1), (1, 0), (1, 1)]:
            specs.append(dict(m=r, k=c, n=r, da=dens, db=dens, tA=tA, tB=tB,
                              alpha=float(rng.uniform(1, 2)), beta=None, thr=0.0))
    # genuinely sparse x sparse: the branch the HIP kernel replaces
    for (m, k, n, da, db) in [(120, 120, 120, 0.04, 0.04), (90, 140, 70, 0.05, 0.06),
                              (48, 200, 48, 0.03, 0.08), (161, 161, 161, 0.015, 0.09)]:
        for (tA, tB) in [(0, 0), (0, 1), (1, 0), (1, 1)]:
            specs.append(dict(m=m, k=k, n=n, da=da, db=db, tA=tA, tB=tB,
                              alpha=float(rng.uniform(-2, 2)), beta=None, thr=0.0))
        specs.append(dict(m=m, k=k, n=n, da=da, db=db, tA=0, tB=0, alpha=1.0, beta=None, thr=1e-2))
        specs.append(dict(m=m, k=k, n=n, da=da, db=db, tA=0, tB=0, alpha=-0.7, beta=0.5, thr=1e-3))
        specs.append(dict(m=m, k=k, n=n, da=da, db=db, tA=1, tB=1, alpha=2.5, beta=-1.25, thr=0.0))
    # one sparse operand, one dense (min sparsity decides the branch: GemmMatrix.f90:59)
    specs.append(dict(m=40, k=40, n=40, da=0.05, db=0.9, tA=0, tB=0, alpha=1.0, beta=None, thr=1e-3))
    specs.append(dict(m=40, k=40, n=40, da=0.5, db=0.5, tA=0, tB=0, alpha=3.0, beta=None, thr=0.3))
    for is_c in (False, True):
        for s in specs:
            if is_c and (s["m"] > 150 or s["thr"] == 0.3):
                continue
            idx = len(cases)
            sa = (s["k"], s["m"]) if s["tA"] else (s["m"], s["k"])
            sb = (s["n"], s["k"]) if s["tB"] else (s["k"], s["n"])
            A = rnd(rng, sa[0], sa[1], s["da"], is_c)
            B = rnd(rng, sb[0], sb[1], s["db"], is_c)
            Cin = rnd(rng, s["m"], s["n"], 0.05, is_c) if s["beta"] is not None else None
            rows, cols, c, r, v = local_gemm_case(tmp, A, B, Cin, s["tA"], s["tB"], s["alpha"],
                                                  0.0 if s["beta"] is None else s["beta"], s["thr"])
            pre = "c%03d_" % idx
            put(d, pre + "A", A.shape, tri(A))
            put(d, pre + "B", B.shape, tri(B))
            if Cin is not None:
                put(d, pre + "Cin", Cin.shape, tri(Cin))
            put(d, pre + "C", (rows, cols), (c, r, v))
            cases.append(dict(s, complex=is_c, nnzA=int(A.nnz), nnzB=int(B.nnz), nnzC=int(len(c))))
    save("local_gemm", d, dict(kind="local_gemm", cases=cases, ncases=len(cases)))


def gen_local_increment(tmp):
    rng = np.random.default_rng(7)
    d, cases = {}, []
    for is_c in (False, True):
        for (m, n, da, db, alpha, thr) in [(50, 50, 0.1, 0.1, 1.0, 0.0), (50, 70, 0.2, 0.05, -2.0, 0.0),
                                           (120, 120, 0.05, 0.3, 0.37, 0.25), (30, 30, 0.0, 0.2, 2.0, 0.1),
                                           (30, 30, 0.2, 0.0, 2.0, 0.1), (64, 64, 1.0, 1.0, -1.0, 0.5)]:
            A = rnd(rng, m, n, da, is_c)
            B = rnd(rng, m, n, db, is_c)
            if alpha == -1.0:  # force exact cancellations on the shared pattern
                B = A.copy()
            write_tri(tmp + "/A.tri", m, n, *tri(A))
            write_tri(tmp + "/B.tri", m, n, *tri(B))
            run(["lincr", tmp + "/A.tri", tmp + "/B.tri", repr(alpha), repr(thr), tmp + "/C.tri"])
            rows, cols, c, r, v = read_tri(tmp + "/C.tri")
            pre = "c%03d_" % len(cases)
            put(d, pre + "A", A.shape, tri(A))
            put(d, pre + "B", B.shape, tri(B))
            put(d, pre + "C", (rows, cols), (c, r, v))
            cases.append(dict(alpha=alpha, thr=thr, complex=is_c))
    save("local_increment", d, dict(kind="local_increment", cases=cases))


def gen_ps(tmp):
    """test_psmatrixalgebra.py:193-218 (multiply, N=33, densities :103-107) on a 1x1x1 grid with
    seeded inputs + banded/sparse cases with threshold, alpha, beta; plus increment and scalars."""
    rng = np.random.default_rng(33)
    d, cases = {}, []

    def one(A, B, Cin, alpha, beta, thr, tag):
        n = A.shape[0]
        write_tri(tmp + "/A.tri", n, n, *tri(A))
        write_tri(tmp + "/B.tri", n, n, *tri(B))
        cin = "none"
        if Cin is not None:
            write_tri(tmp + "/Cin.tri", n, n, *tri(Cin))
            cin = tmp + "/Cin.tri"
        run(["pgemm", 1, 1, 1, tmp + "/A.tri", tmp + "/B.tri", cin, repr(alpha), repr(beta), repr(thr),
             tmp + "/C.tri"])
        rows, cols, c, r, v = read_tri(tmp + "/C.tri")
        pre = "c%03d_" % len(cases)
        put(d, pre + "A", A.shape, tri(A))
        put(d, pre + "B", B.shape, tri(B))
        if Cin is not None:
            put(d, pre + "Cin", Cin.shape, tri(Cin))
        put(d, pre + "C", (rows, cols), (c, r, v))
        dens = min(A.nnz, B.nnz) / float(n * n)
        cases.append(dict(tag=tag, alpha=alpha, beta=beta, thr=thr, n=n, dense_branch=bool(dens > 0.1),
                          complexA=bool(np.iscomplexobj(A.data)), complexB=bool(np.iscomplexobj(B.data))))

    for (da, db) in [(1.0, 1.0), (0.2, 0.2), (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)]:
        for (ca, cb) in [(False, False), (True, True), (True, False), (False, True)]:
            one(rnd(rng, 33, 33, da, ca), rnd(rng, 33, 33, db, cb), None, 1.0, 0.0, 0.0, "ref_n33")
    for (ca, cb) in [(False, False), (True, True), (True, False), (False, True)]:
        A, B = rnd(rng, 129, 129, 0.04, ca), rnd(rng, 129, 129, 0.05, cb)
        one(A, B, None, 1.0, 0.0, 0.0, "sparse_n129")
        one(A, B, None, -1.5, 0.0, 1e-2, "sparse_n129_thr")
        one(A, B, rnd(rng, 129, 129, 0.03, ca or cb), 0.75, -2.0, 1e-3, "sparse_n129_beta")
    Hb = banded(512, 20)
    one(Hb, Hb, None, 1.0, 0.0, 0.0, "banded_n512_h20")
    one(Hb, Hb, None, 1.0, 0.0, 1e-8, "banded_n512_h20_thr1e-8")
    one(Hb, Hb, None, 1.0, 0.0, 1e-4, "banded_n512_h20_thr1e-4")
    Hc = banded(384, 12, True)
    one(Hc, Hc, None, 1.0, 0.0, 1e-6, "cbanded_n384_h12")
    perm = rng.permutation(512)
    Hp = sp.csc_matrix(Hb[perm][:, perm])
    one(Hp, Hp, None, 1.0, 0.0, 1e-8, "banded_n512_h20_permuted")
    save("ps_gemm", d, dict(kind="ps_gemm", cases=cases))

    # increment + scalars
    d, cases = {}, []
    for (ca, cb) in [(False, False), (True, True), (True, False), (False, True)]:
        for (alpha, thr) in [(1.0, 0.0), (-2.5, 0.2)]:
            A, B = rnd(rng, 65, 65, 0.2, ca), rnd(rng, 65, 65, 0.15, cb)
            write_tri(tmp + "/A.tri", 65, 65, *tri(A))
            write_tri(tmp + "/B.tri", 65, 65, *tri(B))
            run(["pincr", 1, 1, 1, tmp + "/A.tri", tmp + "/B.tri", repr(alpha), repr(thr), tmp + "/C.tri"])
            rows, cols, c, r, v = read_tri(tmp + "/C.tri")
            pre = "c%03d_" % len(cases)
            put(d, pre + "A", A.shape, tri(A))
            put(d, pre + "B", B.shape, tri(B))
            put(d, pre + "C", (rows, cols), (c, r, v))
            cases.append(dict(alpha=alpha, thr=thr))
    save("ps_increment", d, dict(kind="ps_increment", cases=cases))

    d, cases = {}, []
    mats = [(rnd(rng, 65, 65, 0.2), rnd(rng, 65, 65, 0.3)), (banded(512, 20), banded(512, 7)),
            (rnd(rng, 48, 48, 0.3, True), rnd(rng, 48, 48, 0.3, True)),
            (banded(200, 9, True), banded(200, 14, True))]
    for (A, B) in mats:
        n = A.shape[0]
        write_tri(tmp + "/A.tri", n, n, *tri(A))
        write_tri(tmp + "/B.tri", n, n, *tri(B))
        run(["pscalars", 1, 1, 1, tmp + "/A.tri", tmp + "/B.tri", tmp + "/s.txt"])
        sc = {k: float(v) for k, v in (ln.split() for ln in open(tmp + "/s.txt"))}
        pre = "c%03d_" % len(cases)
        put(d, pre + "A", A.shape, tri(A))
        put(d, pre + "B", B.shape, tri(B))
        cases.append(sc)
    save("ps_scalars", d, dict(kind="ps_scalars", cases=cases))


def parse_log(path):
    """pull the per-iteration values out of the reference's YAML-ish log"""
    conv, energy = [], []
    total = None
    for ln in open(path):
        m = re.match(r"\s*- Convergence:\s*(\S+)", ln)
        if m:
            conv.append(float(m.group(1)))
        m = re.match(r"\s*Energy Value:\s*(\S+)", ln)
        if m:
            energy.append(float(m.group(1)))
        m = re.match(r"\s*Total Iterations:\s*(\S+)", ln)
        if m:
            total = int(m.group(1))
    return conv, energy, total


def spd_from(rng, n, density):
    """symmetric, diagonally dominant (so Gershgorin-scaled iterations converge), sparse"""
    m = rnd(rng, n, n, density)
    m = (m + m.T) * 0.5
    m = m + sp.diags(np.asarray(abs(m).sum(axis=1)).ravel() + 1.0)
    return sp.csc_matrix(m)


def gen_solvers(tmp):
    rng = np.random.default_rng(4242)
    d, cases = {}, []

    def solve(solver, H, ISQ, nel, thr, conv, maxit, monitor, tag):
        n = H.shape[0]
        write_tri(tmp + "/H.tri", n, n, *tri(H))
        isq = "none"
        if isinstance(ISQ, str):
            isq = ISQ
        elif ISQ is not None:
            write_tri(tmp + "/ISQ.tri", n, n, *tri(ISQ))
            isq = tmp + "/ISQ.tri"
        run(["solve", 1, 1, 1, solver, tmp + "/H.tri", isq, repr(nel), repr(thr), repr(conv), maxit,
             int(monitor), tmp + "/K.tri", tmp + "/log.yaml", tmp + "/s.txt"])
        rows, cols, c, r, v = read_tri(tmp + "/K.tri")
        sc = {k: float(x) for k, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        lc, le, total = parse_log(tmp + "/log.yaml")
        pre = "c%03d_" % len(cases)
        put(d, pre + "H", H.shape, tri(H))
        if ISQ is not None and not isinstance(ISQ, str):
            put(d, pre + "ISQ", ISQ.shape, tri(ISQ))
        put(d, pre + "K", (rows, cols), (c, r, v))
        d[pre + "log_convergence"] = np.array(lc)
        d[pre + "log_energy"] = np.array(le)
        cases.append(dict(tag=tag, solver=solver, nel=nel, thr=thr, conv=conv, maxit=maxit,
                          monitor=bool(monitor), isq=("file" if pre + "ISQ_col" in d else isq),
                          energy=sc["energy"], mu=sc["mu"], nnz=int(sc["nnz"]),
                          total_iterations_logged=total))
        return sp.csc_matrix((v, (r - 1, c - 1)), shape=(rows, cols))

    # --- the reference's own shipped fixture: Examples/PremadeMatrix -------------
    from scipy.io import mmread
    ex = "/root/reference/Examples/PremadeMatrix/"
    Hm, Sm = sp.csc_matrix(mmread(ex + "Hamiltonian.mtx")), sp.csc_matrix(mmread(ex + "Overlap.mtx"))
    Dref = sp.csc_matrix(mmread(ex + "Density-Reference.mtx"))
    ISQ = solve("isq", Sm, None, 0.0, 1e-6, 1e-3, 1000, 1, "premade_isq")
    solve("trs2", Hm, ISQ, 5.0, 1e-6, 1e-5, 1000, 1, "premade_trs2_nel5")
    put(d, "premade_density_reference", Dref.shape, tri(Dref))

    # --- banded synthetic (BASELINE.md golden scalars use the same generator) ----------------
    Hb = banded(512, 16)
    solve("trs2", Hb, "identity", 256.0, 1e-8, 1e-30, 8, 0, "banded512_trs2_8it")
    solve("trs2", Hb, "identity", 256.0, 1e-8, 1e-6, 1000, 1, "banded512_trs2_conv")
    solve("trs2", banded(96, 6), "identity", 40.0, 0.0, 1e-8, 1000, 1, "banded96_trs2_thr0")
    solve("trs4", Hb, "identity", 256.0, 1e-8, 1e-6, 1000, 1, "banded512_trs4_conv")
    Hc = banded(256, 10, True)
    solve("trs2", Hc, "identity", 128.0, 1e-8, 1e-6, 1000, 1, "cbanded256_trs2_conv")

    # --- random symmetric, non-trivial overlap (test_chemistry.py:193-233 style) --------------
    n = 64
    Hr = rnd(rng, n, n, 0.08)
    Hr = sp.csc_matrix((Hr + Hr.T) * 0.5 + sp.diags(np.linspace(-1, 1, n)))
    Sr = spd_from(rng, n, 0.06)
    ISQr = solve("isq", Sr, None, 0.0, 1e-9, 1e-8, 1000, 1, "rand64_isq")
    solve("trs2", Hr, ISQr, 20.0, 1e-9, 1e-8, 1000, 1, "rand64_trs2")
    solve("trs4", Hr, ISQr, 20.0, 1e-9, 1e-8, 1000, 1, "rand64_trs4")

    # --- matrix functions (test_solvers.py:139-162,211-234,259-281,364-386 style) -------------
    Sp = spd_from(rng, 96, 0.05)
    solve("invert", Sp, None, 0.0, 1e-10, 1e-8, 1000, 1, "spd96_invert")
    solve("isq", Sp, None, 0.0, 1e-10, 1e-8, 1000, 1, "spd96_isq")
    solve("sqrt", Sp, None, 0.0, 1e-10, 1e-8, 1000, 1, "spd96_sqrt")
    Sg = rnd(rng, 96, 96, 0.05)
    Sg = sp.csc_matrix((Sg + Sg.T) * 0.5 + sp.diags(np.where(np.arange(96) % 2 == 0, 2.0, -2.0)))
    solve("sign", Sg, None, 0.0, 1e-10, 1e-8, 1000, 1, "sym96_sign")
    solve("sign", Sg, None, 0.0, 1e-10, 1e-8, 1000, 0, "sym96_sign_nomonitor")
    Hs = sp.csc_matrix(banded(256, 10) + 2.0 * sp.identity(256))
    solve("isq", Hs, None, 0.0, 1e-8, 1e-6, 1000, 1, "banded256_shift_isq")
    solve("invert", Hs, None, 0.0, 1e-8, 1e-6, 1000, 1, "banded256_shift_invert")
    Hcs = sp.csc_matrix(banded(128, 8, True) + 2.0 * sp.identity(128))
    solve("isq", Hcs, None, 0.0, 1e-8, 1e-6, 1000, 1, "cbanded128_shift_isq")
    solve("sign", sp.csc_matrix(banded(128, 8, True)), None, 0.0, 1e-8, 1e-6, 1000, 1,
          "cbanded128_sign")
    save("solvers", d, dict(kind="solvers", cases=cases))


def gen_solvers_extra(tmp):
    """PM / HPCP (test_chemistry.py:266-280), PseudoInverse, PolarDecomposition
    (test_solvers.py:164-188,388-407 style) -- the cheap extras next to the four north-star solvers."""
    rng = np.random.default_rng(777)
    d, cases = {}, []

    def solve(solver, H, ISQ, nel, thr, conv, maxit, monitor, tag):
        n = H.shape[0]
        write_tri(tmp + "/H.tri", n, n, *tri(H))
        isq = ISQ if isinstance(ISQ, str) else "none"
        run(["solve", 1, 1, 1, solver, tmp + "/H.tri", isq, repr(nel), repr(thr), repr(conv), maxit,
             int(monitor), tmp + "/K.tri", tmp + "/log.yaml", tmp + "/s.txt"])
        rows, cols, c, r, v = read_tri(tmp + "/K.tri")
        sc = {k: float(x) for k, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        lc, le, total = parse_log(tmp + "/log.yaml")
        pre = "c%03d_" % len(cases)
        put(d, pre + "H", H.shape, tri(H))
        put(d, pre + "K", (rows, cols), (c, r, v))
        d[pre + "log_convergence"] = np.array(lc)
        d[pre + "log_energy"] = np.array(le)
        cases.append(dict(tag=tag, solver=solver, nel=nel, thr=thr, conv=conv, maxit=maxit, monitor=bool(monitor),
                          isq=isq, energy=sc["energy"], mu=sc["mu"], nnz=int(sc["nnz"]), total_iterations_logged=total))

    Hb = banded(256, 12)
    solve("pm", Hb, "identity", 128.0, 1e-8, 1e-6, 1000, 1, "banded256_pm")
    solve("hpcp", Hb, "identity", 128.0, 1e-8, 1e-6, 1000, 1, "banded256_hpcp")
    solve("pm", banded(128, 6, True), "identity", 50.0, 1e-9, 1e-7, 1000, 1, "cbanded128_pm")
    Sp = spd_from(rng, 80, 0.06)
    solve("pinv", Sp, None, 0.0, 1e-10, 1e-8, 1000, 1, "spd80_pinv")
    G = rnd(rng, 80, 80, 0.06)
    G = sp.csc_matrix(G + sp.diags(np.full(80, 3.0)))
    solve("polar", G, None, 0.0, 1e-10, 1e-8, 1000, 1, "gen80_polar")
    save("solvers_extra", d, dict(kind="solvers_extra", cases=cases))


def gen_scalefold(tmp):
    """ScaleAndFold (DensityMatrixSolversModule.F90:953-1117): needs estimates of homo and lumo; taken from the
    dense spectrum with a margin inside the gap, handed to the driver through the environment."""
    d, cases = {}, []
    for (n, h, nel, thr, tag) in ((256, 12, 128.0, 1e-8, "banded256_scalefold"), (192, 8, 60.0, 0.0, "banded192_scalefold_thr0")):
        H = banded(n, h)
        ev = np.linalg.eigvalsh(H.toarray())
        k = int(nel)
        gap = ev[k] - ev[k - 1]
        homo, lumo = ev[k - 1] + 0.05 * gap, ev[k] - 0.05 * gap
        write_tri(tmp + "/H.tri", n, n, *tri(H))
        ENV["REF_HOMO"], ENV["REF_LUMO"] = repr(float(homo)), repr(float(lumo))
        run(["solve", 1, 1, 1, "scalefold", tmp + "/H.tri", "identity", repr(nel), repr(thr), repr(1e-6), 1000, 1,
             tmp + "/K.tri", tmp + "/log.yaml", tmp + "/s.txt"])
        rows, cols, c, r, v = read_tri(tmp + "/K.tri")
        sc = {kk: float(x) for kk, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        lc, le, total = parse_log(tmp + "/log.yaml")
        pre = "c%03d_" % len(cases)
        put(d, pre + "H", H.shape, tri(H))
        put(d, pre + "K", (rows, cols), (c, r, v))
        d[pre + "log_convergence"] = np.array(lc)
        d[pre + "log_energy"] = np.array(le)
        cases.append(dict(tag=tag, solver="scalefold", nel=nel, thr=thr, conv=1e-6, maxit=1000, monitor=True,
                          isq="identity", homo=float(homo), lumo=float(lumo), energy=sc["energy"], nnz=int(sc["nnz"]),
                          total_iterations_logged=total))
    save("solvers_scalefold", d, dict(kind="solvers_scalefold", cases=cases))


def gen_polynomials(tmp):
    """Horner / Paterson-Stockmeyer / Chebyshev (standard + recursive) / Hermite matrix polynomials
    (test_solvers.py:409-560 style) on a banded symmetric matrix scaled into [-1, 1], real and complex."""
    d, cases = {}, []
    rng = np.random.default_rng(4242)
    for cplx in (False, True):
        H = banded(160, 6, cplx)
        H = sp.csc_matrix(H * (1.0 / (1.05 * abs(H).sum(axis=0).max())))
        write_tri(tmp + "/A.tri", 160, 160, *tri(H))
        put(d, "A%d" % int(cplx), H.shape, tri(H))
        for kind, ncoef, thr in (("horner", 1, 0.0), ("horner", 2, 0.0), ("horner", 7, 1e-9), ("ps", 2, 0.0), ("ps", 6, 0.0),
                                 ("ps", 11, 1e-9), ("cheby", 1, 0.0), ("cheby", 2, 0.0), ("cheby", 3, 0.0), ("cheby", 9, 1e-9),
                                 ("chebyfact", 1, 0.0), ("chebyfact", 2, 0.0), ("chebyfact", 5, 0.0), ("chebyfact", 9, 1e-9),
                                 ("chebyfact", 16, 1e-9), ("hermite", 1, 0.0), ("hermite", 2, 0.0), ("hermite", 6, 1e-9)):
            coef = [float(x) for x in np.round(rng.uniform(-1.0, 1.0, ncoef), 6)]
            run(["poly", 1, 1, 1, kind, tmp + "/A.tri", repr(thr), tmp + "/K.tri", ncoef] + [repr(c) for c in coef])
            rows, cols, c, r, v = read_tri(tmp + "/K.tri")
            pre = "c%03d_" % len(cases)
            put(d, pre + "K", (rows, cols), (c, r, v))
            cases.append(dict(kind=kind, complex=cplx, coef=coef, thr=thr, nnz=int(len(c))))
    save("polynomials", d, dict(kind="polynomials", cases=cases))


def gen_functions(tmp):
    """exp / log / sin / cos / roots / inverse roots / PowerBounds (test_solvers.py:586-800 style) on small banded
    matrices: symmetric indefinite for exp/sin/cos, shifted positive definite for log and the roots."""
    d, cases = {}, []
    Hs = banded(128, 5)
    Hp = sp.csc_matrix(banded(128, 5) + 2.5 * sp.identity(128))
    Hc = sp.csc_matrix(banded(96, 4, True))
    for name, M in (("sym", Hs), ("spd", Hp), ("csym", Hc)):
        put(d, "M_" + name, M.shape, tri(M))
    jobs = [("exp", "sym", 1e-9, 0), ("exp", "csym", 1e-9, 0), ("sin", "sym", 1e-9, 0), ("cos", "sym", 1e-9, 0),
            ("cos", "csym", 1e-9, 0), ("log", "spd", 1e-10, 0), ("power", "spd", 0.0, 0), ("power", "sym", 0.0, 0)]
    for r in (1, 2, 3, 4, 5, 6, 7, 8):
        jobs.append(("root", "spd", 1e-10, r))
        jobs.append(("invroot", "spd", 1e-10, r))
    for kind, name, thr, root in jobs:
        M = dict(sym=Hs, spd=Hp, csym=Hc)[name]
        n = M.shape[0]
        write_tri(tmp + "/A.tri", n, n, *tri(M))
        run(["func", 1, 1, 1, kind, tmp + "/A.tri", repr(thr), repr(1e-8), tmp + "/K.tri", tmp + "/s.txt", root])
        rows, cols, c, r, v = read_tri(tmp + "/K.tri")
        sc = {kk: float(x) for kk, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        pre = "c%03d_" % len(cases)
        if kind != "power":
            put(d, pre + "K", (rows, cols), (c, r, v))
        cases.append(dict(kind=kind, matrix=name, thr=thr, conv=1e-8, root=root, bound=sc["bound"], nnz=int(len(c))))
    save("functions", d, dict(kind="functions", cases=cases))


def gen_extras(tmp):
    """The remaining solver families run by the real reference (ref_driver extra ...): CG, Pade exponential, geometry
    extrapolation, sparsity-pattern snap, dense eigendecomposition / SVD / matrix functions, gap estimate, Fermi
    operator (dense FOE, dense step function, WOM_GC, WOM_C), Cholesky, pivoted Cholesky, ReduceDimension."""
    rng = np.random.default_rng(77)
    d, cases = {}, []
    n, nel = 96, 30
    Hs = banded(n, 5)
    Hp = sp.csc_matrix(Hs + 2.5 * sp.identity(n))
    Hc = sp.csc_matrix(banded(64, 4, True))
    Hcp = sp.csc_matrix(Hc + 2.5 * sp.identity(64))
    B1 = banded(n, 3)
    S_old = sp.csc_matrix(sp.identity(n) + 0.05 * B1)
    S_new = sp.csc_matrix(sp.identity(n) + 0.06 * B1)
    w, v = np.linalg.eigh(Hs.toarray())
    Dd = v[:, :nel] @ v[:, :nel].T
    Dd[np.abs(Dd) < 1e-12] = 0.0
    D = sp.csc_matrix(Dd)
    mu_mid = 0.5 * (w[nel - 1] + w[nel])
    ws, vs = np.linalg.eigh(S_old.toarray())
    ISQ = sp.csc_matrix((vs / np.sqrt(ws)) @ vs.T)
    Ident = sp.identity(n, format="csc")
    R = sp.csc_matrix(rnd(rng, n, n, 0.08) + 2.0 * sp.identity(n))
    Ra, Rb = rnd(rng, n, n, 0.10), rnd(rng, n, n, 0.12)
    Rc = rnd(rng, n, n, 0.10, True)
    mats = dict(Hs=Hs, Hp=Hp, Hc=Hc, Hcp=Hcp, S_old=S_old, S_new=S_new, D=D, ISQ=ISQ, I=Ident, R=R, Ra=Ra, Rb=Rb, Rc=Rc)
    for name, M in mats.items():
        M = sp.csc_matrix(M)
        M.sort_indices()
        mats[name] = M
        put(d, "M_" + name, M.shape, tri(M))
    jobs = [  # kind, A, B, C, thr, conv, p1, p2
        ("cg", "Hp", "Hs", None, 0.0, 1e-9, 0, 0),      # (with a threshold the residual becomes exactly zero and the
        ("cg", "Hcp", "Hc", None, 0.0, 1e-9, 0, 0),     #  reference's loop divides 0 / 0 before its monitor stops it)
        ("pade", "Hs", None, None, 1e-9, 1e-8, 0, 0),
        ("purify", "D", "S_old", None, 1e-9, 1e-8, nel, 0),
        ("lowdin", "D", "S_old", "S_new", 1e-9, 1e-8, 0, 0),
        ("snap", "Ra", "Rb", None, 0.0, 1e-8, 0, 0),
        ("snap", "Rc", "Rb", None, 0.0, 1e-8, 0, 0),
        ("eig", "Hs", None, None, 1e-12, 1e-8, n, 0),
        ("eig", "Hs", None, None, 1e-12, 1e-8, 10, 0),
        ("eig", "Hc", None, None, 1e-12, 1e-8, 64, 0),
        ("svd", "R", None, None, 1e-12, 1e-10, 0, 0),
        ("gap", "Hs", "D", None, 1e-10, 1e-8, mu_mid, 0),
        ("foe", "Hs", "I", None, 1e-10, 1e-8, nel, 20.0),
        ("foe", "Hs", "ISQ", None, 1e-10, 1e-8, nel, 50.0),
        ("foe", "Hc", "none_identity", None, 1e-10, 1e-8, 20, 30.0),
        ("density", "Hs", "I", None, 1e-10, 1e-8, nel, 0),
        ("density", "Hs", "ISQ", None, 1e-10, 1e-8, nel + 0.5, 0),
        ("womgc", "Hs", "I", None, 1e-8, 1e-8, mu_mid, 4.0),
        ("womc", "Hs", "ISQ", None, 1e-8, 1e-8, nel, 4.0),
        ("chol", "Hp", None, None, 0.0, 1e-8, 0, 0),
        ("chol", "Hp", None, None, 1e-6, 1e-8, 0, 0),
        ("pchol", "D", None, None, 1e-10, 1e-8, nel, 0),
        ("pchol", "Hp", None, None, 1e-8, 1e-8, 40, 0),
        ("reduce", "Hs", None, None, 1e-9, 1e-8, nel, 0),
        ("dsqrt", "Hp", None, None, 1e-10, 1e-8, 0, 0),
        ("dsqrt", "Hcp", None, None, 1e-10, 1e-8, 0, 0),
        ("disqrt", "Hp", None, None, 1e-10, 1e-8, 0, 0),
        ("dexp", "Hs", None, None, 1e-10, 1e-8, 0, 0),
        ("dlog", "Hp", None, None, 1e-10, 1e-8, 0, 0),
        ("dsin", "Hs", None, None, 1e-10, 1e-8, 0, 0),
        ("dcos", "Hc", None, None, 1e-10, 1e-8, 0, 0),
        ("dinv", "Hp", None, None, 1e-10, 1e-8, 0, 0),
        ("dsign", "Hs", None, None, 1e-10, 1e-8, 0, 0),
    ]
    IdentC = sp.identity(64, format="csc")
    for kind, a, b, c, thr, conv, p1, p2 in jobs:
        files = []
        for tag, nm in (("A", a), ("B", b), ("C", c)):
            if nm is None:
                files.append("none")
                continue
            M = IdentC if nm == "none_identity" else mats[nm]
            write_tri("%s/%s.tri" % (tmp, tag), M.shape[0], M.shape[1], *tri(M))
            files.append("%s/%s.tri" % (tmp, tag))
        run(["extra", 1, 1, 1, kind] + files + [repr(thr), repr(conv), tmp + "/K.tri", tmp + "/s.txt", repr(float(p1)), repr(float(p2))])
        rows, cols, cc, rr, vv = read_tri(tmp + "/K.tri")
        sc = {kk: float(x) for kk, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        pre = "c%03d_" % len(cases)
        put(d, pre + "K", (rows, cols), (cc, rr, vv))
        if kind in ("eig", "svd"):
            rows2, cols2, c2, r2, v2 = read_tri(tmp + "/K.tri.2")
            put(d, pre + "K2", (rows2, cols2), (c2, r2, v2))
        cases.append(dict(kind=kind, A=a, B=b, C=c, thr=thr, conv=conv, p1=float(p1), p2=float(p2), s1=sc["s1"], s2=sc["s2"],
                          nnz=int(len(cc))))
        print("  ", kind, a, b, "nnz", len(cc), "s1", sc["s1"], "s2", sc["s2"])
    save("extras", d, dict(kind="extras", cases=cases))


def gen_example_complexmatrix(tmp):
    """Examples/ComplexMatrix: the Hermitian "Guo" matrix built from the reference's input graph (tests/golden/
    reference_data/complexmatrix_input.mtx, the example's own construction restated in numpy) and the reference's
    ComputeExponential of it at the ReadMe's threshold 1e-6.  (With eigenvalues up to 25.8 the reference's Chebyshev
    scheme is far from scipy's expm here; what is pinned is what the reference computes.)"""
    import scipy.io
    A = scipy.io.mmread(os.path.join(OUT, "reference_data", "complexmatrix_input.mtx")).toarray()
    n = len(A)
    S = A + A.T - np.diag(np.diag(A))
    S = np.where((A != 0) & (A.T != 0) & ~np.eye(n, dtype=bool), A + A.T, S)
    Cm = np.where((S - A) != 0, 1j, 0.0)
    G = sp.csc_matrix(0.5 * (Cm.conj().T + Cm + S))
    G.sort_indices()
    write_tri(tmp + "/A.tri", n, n, *tri(G))
    run(["func", 1, 1, 1, "exp", tmp + "/A.tri", repr(1e-6), repr(1e-6), tmp + "/K.tri", tmp + "/s.txt", 0])
    rows, cols, c, r, v = read_tri(tmp + "/K.tri")
    d = {}
    put(d, "G", (n, n), tri(G))
    put(d, "K", (rows, cols), (c, r, v))
    save("example_complexmatrix", d, dict(kind="example_complexmatrix", thr=1e-6, cases=[]))


def gen_ps_gemm_fma(tmp):
    """Distributed multiply computed by the reference built WITH floating-point contraction (oracle/build_ref.py
    --fma: -ffp-contract=fast -march=haswell, `acc + a*b` is one FMA) -- what the reference produces on targets
    with baseline FMA.  Pins the engine's spgemm_fma option (v_fma_f64 in the register-slab kernel)."""
    global DRV
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    if not build_ref.build(fma=True):
        raise SystemExit("reference not buildable here")
    drv0, DRV = DRV, os.path.join(ROOT, "oracle", "_ref", "fma", "ref_driver")
    try:
        d, cases = {}, []
        for (n, h, h2, alpha, thr, tag) in ((512, 16, 16, 1.0, 0.0, "b512_sq_thr0"), (1024, 40, 40, 1.0, 1e-8, "b1024_sq"),
                                            (1500, 70, 90, 0.5, 1e-7, "b1500_ab"), (1800, 80, 80, 1.0, 1e-9, "b1800_wide")):   # all < 10 % dense: sparse branch
            A = banded(n, h)
            B = A if h2 == h and alpha == 1.0 else sp.csc_matrix(banded(n, h2) * 1.01)
            write_tri(tmp + "/A.tri", n, n, *tri(A))
            write_tri(tmp + "/B.tri", n, n, *tri(B))
            run(["pgemm", 1, 1, 1, tmp + "/A.tri", tmp + "/B.tri", "none", repr(alpha), "0.0", repr(thr), tmp + "/C.tri"])
            rows, cols, c, r, v = read_tri(tmp + "/C.tri")
            pre = "c%03d_" % len(cases)
            put(d, pre + "A", A.shape, tri(A))
            put(d, pre + "B", B.shape, tri(B))
            put(d, pre + "C", (rows, cols), (c, r, v))
            cases.append(dict(tag=tag, alpha=alpha, thr=thr, n=n, same=bool(B is A)))
        save("ps_gemm_fma", d, dict(kind="ps_gemm_fma", cases=cases))
    finally:
        DRV = drv0


def gen_multirank(tmp):
    """Same product on 1, 4 (2x2x1) and 8 (2x2x2) reference ranks: pins that values do not
    depend on the grid when slices == 1 and records the slices>1 behaviour."""
    d, cases = {}, []
    Hb = banded(256, 12)
    n = 256
    write_tri(tmp + "/A.tri", n, n, *tri(Hb))
    put(d, "A", Hb.shape, tri(Hb))
    for (pr, pc, ps) in [(1, 1, 1), (2, 2, 1), (1, 4, 1), (2, 2, 2)]:
        nr = pr * pc * ps
        run(["pgemm", pr, pc, ps, tmp + "/A.tri", tmp + "/A.tri", "none", "1.0", "0.0", "1e-6",
             tmp + "/C.tri"], nranks=nr)
        parts = []
        for rk in range(nr):
            fn = tmp + "/C.tri" + ("" if nr == 1 else ".%d" % rk)
            parts.append(read_tri(fn))
        c = np.concatenate([p[2] for p in parts])
        r = np.concatenate([p[3] for p in parts])
        v = np.concatenate([p[4] for p in parts])
        # slices replicate the matrix: keep unique (col,row)
        key = c.astype(np.int64) * (n + 1) + r
        _, first = np.unique(key, return_index=True)
        c, r, v = c[first], r[first], v[first]
        put(d, "C_%d%d%d" % (pr, pc, ps), (n, n), (c, r, v))
        cases.append(dict(grid=[pr, pc, ps], nnz=int(len(c))))
    save("ps_gemm_grids", d, dict(kind="ps_gemm_grids", thr=1e-6, cases=cases))


def gen_scale_logs(tmp):
    """Parity at scale: the REAL reference on 8 ranks (grid 8x1x1) at sizes well beyond the small
    goldens -- TRS2 and TRS4 at N = 16 384, h = 100 (the headline band), complex InverseSquareRoot (H + 2I) and
    SignFunction (the indefinite H itself) at N = 8 192, h = 50.  Only the per-iteration logs and a few scalars of the
    result are kept (the inputs are the closed-form generator; the results have millions of entries)."""
    d, cases = {}, []

    def solve(solver, n, h, cplx, shift, nel, thr, conv, maxit, monitor, tag, nranks=8):
        H = banded(n, h, cplx)
        if shift:
            H = sp.csc_matrix(H + shift * sp.identity(n))
        write_tri(tmp + "/H.tri", n, n, *tri(H))
        isq = "identity" if solver in ("trs2", "trs4") else "none"
        run(["solve", nranks, 1, 1, solver, tmp + "/H.tri", isq, repr(nel), repr(thr), repr(conv), maxit,
             int(monitor), tmp + "/K.tri", tmp + "/log.yaml", tmp + "/s.txt"], nranks=nranks)
        parts = [read_tri(tmp + "/K.tri.%d" % q) for q in range(nranks)] if nranks > 1 else [read_tri(tmp + "/K.tri")]
        c = np.concatenate([q[2] for q in parts])
        r = np.concatenate([q[3] for q in parts])
        v = np.concatenate([q[4] for q in parts])
        sc = {k: float(x) for k, x in (ln.split() for ln in open(tmp + "/s.txt"))}
        assert len(v) == int(sc["nnz"])
        lc, le, total = parse_log(tmp + "/log.yaml")
        pre = "c%03d_" % len(cases)
        d[pre + "log_convergence"] = np.array(lc)
        d[pre + "log_energy"] = np.array(le)
        diag = v[c == r]
        cases.append(dict(tag=tag, solver=solver, n=n, h=h, cplx=bool(cplx), shift=shift, nel=nel, thr=thr, conv=conv,
                          maxit=maxit, monitor=bool(monitor), energy=sc["energy"], mu=sc["mu"], nnz=int(sc["nnz"]),
                          total_iterations_logged=total, grid=[nranks, 1, 1],
                          trace_re=float(np.real(diag).sum()), frob2=float((np.abs(v) ** 2).sum()),
                          sum_re=float(np.real(v).sum()), sum_im=float(np.imag(v).sum())))
        print(tag, "iterations", total, "nnz", int(sc["nnz"]), flush=True)

    solve("trs2", 16384, 100, False, 0.0, 8192.0, 1e-8, 1e-6, 1000, 1, "banded16384_h100_trs2_conv")
    solve("trs4", 16384, 100, False, 0.0, 8192.0, 1e-8, 1e-6, 1000, 1, "banded16384_h100_trs4_conv")
    solve("isq", 8192, 50, True, 2.0, 0.0, 1e-8, 1e-6, 1000, 1, "cbanded8192_shift_isq")
    solve("sign", 8192, 50, True, 0.0, 0.0, 1e-8, 1e-6, 1000, 1, "cbanded8192_sign")
    save("scale_logs", d, dict(kind="scale_logs", cases=cases))


def gen_scale_logs_lb(tmp):
    """a LOAD-BALANCED converged TRS2 solve of the REAL reference on 8 ranks at N = 16 384, h = 100
    (DensityMatrixSolversModule.F90 with do_load_balancing: LoadBalancerModule.F90:14-52 permutes H and the result with
    the solver parameters' permutation).  The permutation is given to the reference explicitly (ref_driver REF_PERM;
    its own would come from the Fortran RNG) and stored with the case, so the engine is driven with the very same one.
    APPENDS the case to tests/golden/scale_logs.npz (the other cases are kept as they are)."""
    path = os.path.join(OUT, "scale_logs.npz")
    old = np.load(path, allow_pickle=False)
    d = {k: old[k] for k in old.files if k != "meta"}
    meta = json.loads(str(old["meta"]))
    cases = [c for c in meta["cases"] if c["tag"] != "banded16384_h100_trs2_lb"]
    n, h, nranks, thr, conv = 16384, 100, 8, 1e-8, 1e-6
    H = banded(n, h, False)
    write_tri(tmp + "/H.tri", n, n, *tri(H))
    perm = (np.random.default_rng(20260716).permutation(n) + 1).astype(np.int32)   # index_lookup, 1-based
    np.savetxt(tmp + "/perm.txt", perm, fmt="%d")
    ENV["REF_PERM"] = tmp + "/perm.txt"
    try:
        run(["solve", nranks, 1, 1, "trs2", tmp + "/H.tri", "identity", repr(n / 2.0), repr(thr), repr(conv), 1000, 1,
             tmp + "/K.tri", tmp + "/log.yaml", tmp + "/s.txt"], nranks=nranks)
    finally:
        del ENV["REF_PERM"]
    parts = [read_tri(tmp + "/K.tri.%d" % q) for q in range(nranks)]
    c = np.concatenate([q[2] for q in parts])
    r = np.concatenate([q[3] for q in parts])
    v = np.concatenate([q[4] for q in parts])
    sc = {k: float(x) for k, x in (ln.split() for ln in open(tmp + "/s.txt"))}
    assert len(v) == int(sc["nnz"])
    lc, le, total = parse_log(tmp + "/log.yaml")
    pre = "c%03d_" % len(cases)
    d[pre + "log_convergence"] = np.array(lc)
    d[pre + "log_energy"] = np.array(le)
    d[pre + "perm"] = perm
    diag = v[c == r]
    cases.append(dict(tag="banded16384_h100_trs2_lb", solver="trs2", n=n, h=h, cplx=False, shift=0.0, nel=n / 2.0, thr=thr,
                      conv=conv, maxit=1000, monitor=True, energy=sc["energy"], mu=sc["mu"], nnz=int(sc["nnz"]),
                      total_iterations_logged=total, grid=[nranks, 1, 1], load_balanced=True,
                      trace_re=float(np.real(diag).sum()), frob2=float((np.abs(v) ** 2).sum()),
                      sum_re=float(np.real(v).sum()), sum_im=float(np.imag(v).sum())))
    print("banded16384_h100_trs2_lb iterations", total, "nnz", int(sc["nnz"]), flush=True)
    meta["cases"] = cases
    save("scale_logs", d, meta)


def gen_slices(tmp):
    """K-split (process slices > 1) semantics of the reference's distributed multiply (MatrixMultiply.f90:25-29,
    230-267, comm_includes/ReduceAndSumMatrixCleanup.f90): every slice multiplies its share of the inner dimension
    with threshold / (1000 slices), the partial products are summed in slice order, the last addition applies the
    caller's threshold by the IncrementMatrix rules.  Results therefore depend on the grid; recorded here for several
    grids, two thresholds, A != B, alpha != 1, a dimension that needs padding, real and complex."""
    rng = np.random.default_rng(777)
    d, cases = {}, []
    for ci, (n, dens, cplx) in enumerate([(250, 0.05, False), (197, 0.08, True)]):
        A = rnd(rng, n, n, dens, cplx)
        B = rnd(rng, n, n, dens, cplx)
        write_tri(tmp + "/A.tri", n, n, *tri(A))
        write_tri(tmp + "/B.tri", n, n, *tri(B))
        put(d, "m%d_A" % ci, A.shape, tri(A))
        put(d, "m%d_B" % ci, B.shape, tri(B))
        for (pr, pc, ps) in [(1, 1, 2), (1, 1, 4), (2, 1, 2), (1, 2, 2), (2, 2, 2), (1, 1, 1)]:
            for thr in (1e-6, 2e-2):
                nr = pr * pc * ps
                run(["pgemm", pr, pc, ps, tmp + "/A.tri", tmp + "/B.tri", "none", "-0.7", "0.0", repr(thr),
                     tmp + "/C.tri"], nranks=nr)
                parts = [read_tri(tmp + "/C.tri" + ("" if nr == 1 else ".%d" % rk)) for rk in range(nr)]
                c = np.concatenate([q[2] for q in parts])
                r = np.concatenate([q[3] for q in parts])
                v = np.concatenate([q[4] for q in parts])
                key = c.astype(np.int64) * (n + 1) + r      # slices replicate the matrix: keep one copy
                _, first = np.unique(key, return_index=True)
                c, r, v = c[first], r[first], v[first]
                tag = "m%d_C_%d%d%d_%g" % (ci, pr, pc, ps, thr)
                put(d, tag, (n, n), (c, r, v))
                cases.append(dict(matrix=ci, n=n, cplx=cplx, grid=[pr, pc, ps], thr=thr, alpha=-0.7, key=tag, nnz=int(len(c))))
                print(tag, len(c), flush=True)
    save("ps_gemm_slices", d, dict(kind="ps_gemm_slices", cases=cases))


def main():
    if not os.path.exists(DRV):
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import build_ref
        if not build_ref.build():
            raise SystemExit("reference not buildable here")
    only = sys.argv[1:] 
    with tempfile.TemporaryDirectory() as tmp:
        if only:
            for name in only:
                globals()["gen_" + name](tmp)
            return
        gen_solvers_extra(tmp)
        gen_scalefold(tmp)
        gen_polynomials(tmp)
        gen_functions(tmp)
        gen_local_gemm(tmp)
        gen_local_increment(tmp)
        gen_ps(tmp)
        gen_solvers(tmp)
        gen_multirank(tmp)


if __name__ == "__main__":
    main()
