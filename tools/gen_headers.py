#!/usr/bin/env python3
"""Generate include/*.h from the engine's own C ABI definitions (ntpoly_amd/csrc/wrp.cpp).

For every exported *_wrp symbol the header records which reference declaration it replaces
(Source/C/<header>:line) and which Fortran wrapper implements it there (Source/Wrapper/<file>:line).
Run in the build container (needs /root/reference for the line numbers); the generated headers
are committed.
"""
import glob
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Source"
SRC = open(os.path.join(ROOT, "ntpoly_amd/csrc/wrp.cpp")).read()


def ref_index():
    cdecl, fbind = {}, {}
    for h in sorted(glob.glob(REF + "/C/*.h")):
        for i, ln in enumerate(open(h), 1):
            m = re.search(r"\b(\w+_wrp)\s*\(", ln)
            if m and m.group(1) not in cdecl:
                cdecl[m.group(1)] = ("Source/C/" + os.path.basename(h), i)
    for f in sorted(glob.glob(REF + "/Wrapper/*.F90")):
        for i, ln in enumerate(open(f), 1):
            m = re.search(r'name\s*=\s*"(\w+)"', ln, re.I)
            if m:
                fbind[m.group(1)] = ("Source/Wrapper/" + os.path.basename(f), i)
    return cdecl, fbind


def expand_macros(src):
    """expand the two function-generating macros of wrp.cpp into plain prototypes"""
    out = src
    dens = re.findall(r"^DENSITY_SOLVER\((\w+), \w+\)", src, re.M)
    protos = []
    for n in dens:
        protos.append("void %s(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, const double* trace, "
                      "int* ih_Density, const double* energy_value_out, const double* chemical_potential_out, "
                      "const int* ih_solver_parameters) {" % n)
    m = re.search(r"#define LOCAL_API\(SUF, CPLX\)(.*?)\n\nLOCAL_API", src, re.S)
    body = m.group(1).replace("\\\n", "\n")
    for suf in ("lsr", "lsc"):
        protos.append(body.replace("##SUF##", suf).replace("_##SUF", "_" + suf))
    return out + "\n" + "\n".join(protos)


def prototypes(src):
    src = expand_macros(src)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"^\s*((?:void|int|double|bool)\s+(\w+)\s*\(([^)]*)\))\s*\{", src, re.M):
        name = m.group(2)
        if name.endswith("_wrp") or name.startswith("ntpoly_amd_"):
            protos.setdefault(name, re.sub(r"\s+", " ", m.group(1)).strip())
    return protos


GROUPS = [
    ("ntpoly_amd_process_grid.h", "ProcessGrid_c.h", "process grid (ProcessGridModule.F90:15-56)"),
    ("ntpoly_amd_psmatrix.h", "PSMatrix_c.h", "distributed matrix + algebra (PSMatrixModule.F90, PSMatrixAlgebraModule.F90)"),
    ("ntpoly_amd_smatrix.h", "SMatrix_c.h", "local matrix + algebra (SMatrixModule.F90, SMatrixAlgebraModule.F90)"),
    ("ntpoly_amd_triplet_list.h", "TripletList_c.h", "triplet lists (TripletListModule.F90)"),
    ("ntpoly_amd_permutation.h", "Permutation_c.h", "permutations (PermutationModule.F90)"),
    ("ntpoly_amd_memory_pool.h", "MatrixMemoryPool_c.h", "local memory pools (MatrixMemoryPoolModule.F90)"),
    ("ntpoly_amd_memory_pool.h", "PMatrixMemoryPool_c.h", "distributed memory pools (PMatrixMemoryPoolModule.F90)"),
    ("ntpoly_amd_solver_parameters.h", "SolverParameters_c.h", "solver parameters (SolverParametersModule.F90:14-113)"),
    ("ntpoly_amd_density_matrix_solvers.h", "DensityMatrixSolvers_c.h", "density matrix solvers (DensityMatrixSolversModule.F90)"),
    ("ntpoly_amd_sign_solvers.h", "SignSolvers_c.h", "sign function / polar decomposition (SignSolversModule.F90)"),
    ("ntpoly_amd_inverse_solvers.h", "InverseSolvers_c.h", "inverse solvers (InverseSolversModule.F90)"),
    ("ntpoly_amd_square_root_solvers.h", "SquareRootSolvers_c.h", "square root solvers (SquareRootSolversModule.F90)"),
    ("ntpoly_amd_load_balancer.h", "LoadBalancer_c.h", "load balancer (LoadBalancerModule.F90)"),
    ("ntpoly_amd_eigen_bounds.h", "EigenBounds_c.h", "eigenvalue bounds (EigenBoundsModule.F90:29-56)"),
    ("ntpoly_amd_logging.h", "Logging_c.h", "logger (LoggingModule.F90)"),
    ("ntpoly_amd_polynomial_solvers.h", "Polynomial_c.h", "matrix polynomials, Horner / Paterson-Stockmeyer (PolynomialSolversModule.F90)"),
    ("ntpoly_amd_polynomial_solvers.h", "ChebyshevSolvers_c.h", "Chebyshev polynomials (ChebyshevSolversModule.F90)"),
    ("ntpoly_amd_polynomial_solvers.h", "HermiteSolvers_c.h", "Hermite polynomials (HermiteSolversModule.F90)"),
    ("ntpoly_amd_function_solvers.h", "ExponentialSolvers_c.h", "exponential / logarithm (ExponentialSolversModule.F90)"),
    ("ntpoly_amd_function_solvers.h", "TrigonometrySolvers_c.h", "sine / cosine (TrigonometrySolversModule.F90)"),
    ("ntpoly_amd_function_solvers.h", "RootSolvers_c.h", "roots / inverse roots (RootSolversModule.F90)"),
    ("ntpoly_amd_linear_solvers.h", "LinearSolvers_c.h", "CG, Cholesky (LinearSolversModule.F90)"),
    ("ntpoly_amd_linear_solvers.h", "Analysis_c.h", "pivoted Cholesky, ReduceDimension (AnalysisModule.F90)"),
    ("ntpoly_amd_eigen_solvers.h", "EigenSolvers_c.h", "eigendecomposition, SVD, gap estimate (EigenSolversModule.F90, SingularValueSolversModule.F90)"),
    ("ntpoly_amd_eigen_solvers.h", "FermiOperator_c.h", "dense FOE, wave-operator minimisation (FermiOperatorModule.F90)"),
    ("ntpoly_amd_geometry.h", "GeometryOptimization_c.h", "density matrix extrapolation (GeometryOptimizationModule.F90)"),
    ("ntpoly_amd_geometry.h", "MatrixConversion_c.h", "sparsity-pattern snap (MatrixConversionModule.F90)"),
]


def main():
    cdecl, fbind = ref_index()
    protos = prototypes(SRC)
    inc = os.path.join(ROOT, "include")
    os.makedirs(inc, exist_ok=True)
    files = {}
    used = set()
    for fname, refh, title in GROUPS:
        names = [n for n, (h, _) in cdecl.items() if h.endswith("/" + refh) and n in protos]
        names.sort(key=lambda n: cdecl[n][1])
        lines = files.setdefault(fname, [])
        lines.append("/* ---- %s: drop-in for %s ---- */" % (title, "Source/C/" + refh))
        for n in names:
            used.add(n)
            c = cdecl[n]
            w = fbind.get(n)
            lines.append("/* replaces %s:%d%s */" % (c[0], c[1], (" (wrapper %s:%d)" % w) if w else ""))
            lines.append(protos[n] + ";")
        missing = [n for n, (h, _) in cdecl.items() if h.endswith("/" + refh) and n not in protos]
        if missing:
            lines.append("/* not on the hot path, not exported (SURVEY 2a/8b): %s */" % ", ".join(sorted(missing)))
        lines.append("")
    ext = [n for n in protos if n.startswith("ntpoly_amd_")]
    extra_wrp = [n for n in protos if n.endswith("_wrp") and n not in used]
    for fname, lines in files.items():
        guard = fname.upper().replace(".", "_")
        with open(os.path.join(inc, fname), "w") as f:
            f.write("/* C ABI of the MI355X engine (libntpoly_amd.so).  GENERATED by tools/gen_headers.py from\n"
                    " * ntpoly_amd/csrc/wrp.cpp; same symbol names, argument order and by-reference calling\n"
                    " * convention as the reference's BIND(C) wrapper layer.  Handles are caller-owned\n"
                    " * int[NTPOLY_AMD_SIZE_WRP] buffers (Source/C/Wrapper.h:4). */\n")
            f.write("#ifndef %s\n#define %s\n#include <stdbool.h>\n#ifdef __cplusplus\nextern \"C\" {\n#endif\n" % (guard, guard))
            f.write("#ifndef NTPOLY_AMD_SIZE_WRP\n#define NTPOLY_AMD_SIZE_WRP 12\n#endif\n\n")
            f.write("\n".join(lines))
            f.write("\n#ifdef __cplusplus\n}\n#endif\n#endif\n")
    with open(os.path.join(inc, "ntpoly_amd.h"), "w") as f:
        f.write("/* Umbrella header + extension entry points of libntpoly_amd.so that have no counterpart in the\n"
                " * reference ABI (RCCL bootstrap replacing MPI_Init/communicators, bulk triplet transfer,\n"
                " * statistics for bench.py, test knobs).  GENERATED by tools/gen_headers.py. */\n")
        f.write("#ifndef NTPOLY_AMD_H\n#define NTPOLY_AMD_H\n")
        for fname in files:
            f.write('#include "%s"\n' % fname)
        f.write("#ifdef __cplusplus\nextern \"C\" {\n#endif\n\n")
        for n in sorted(ext):
            f.write(protos[n] + ";\n")
        if extra_wrp:
            f.write("\n/* *_wrp symbols without a declaration in Source/C (none expected) */\n")
            for n in sorted(extra_wrp):
                f.write(protos[n] + ";\n")
        f.write("\n#ifdef __cplusplus\n}\n#endif\n#endif\n")
    print("headers:", sorted(files) + ["ntpoly_amd.h"], "symbols:", len(used), "+", len(ext), "ext", extra_wrp)


if __name__ == "__main__":
    main()
