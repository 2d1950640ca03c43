// Remaining solver families of the reference, all of them callers of the distributed algebra (SpGEMM, increment,
// dot, norm) plus, for the "dense" family, one Hermitian eigendecomposition on the GPU (dense.hip, Jacobi):
//   LinearSolversModule.F90 (CG, Cholesky), AnalysisModule.F90 (pivoted Cholesky, ReduceDimension),
//   ExponentialSolversModule.F90:152-271 (Pade), GeometryOptimizationModule.F90, EigenSolversModule.F90,
//   SingularValueSolversModule.F90, FermiOperatorModule.F90, MatrixConversionModule.F90.
// Control flow, scalar formulas and stopping rules follow the cited lines; the arithmetic is the engine's.
#include <algorithm>
#include <cmath>
#include <functional>

#include "engine.hpp"
#include "kernels.hpp"

namespace ntp {

namespace {
SolverParameters with_monitor(const SolverParameters& p, Monitor& mon) {
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  return p;
}
void conj_transpose(const PSMatrix& A, PSMatrix& AT) {
  ps_transpose(A, AT);
  if (AT.cplx) ps_conjugate(AT);
}
}  // namespace

// FilterMatrix (PSMatrixModule.F90:1318-1357): entries with |v| > threshold stay
void ps_filter(PSMatrix& m, double threshold) {
  use_grid_comm(m.grid);
  m.loc = filter(m.loc, threshold);
}

// GatherMatrixTripletList: every rank receives every entry (ordered by column, then row)
void ps_gather_triplets(const PSMatrix& m, HostTriplets& t) {
  use_grid_comm(m.grid);
  DevMat full = ps_gather_full(m);
  to_triplets(full, 0, t);
}

// ------------------------------------------------------------------ CG (LinearSolversModule.F90:31-171)
void solver_cg(const PSMatrix& AMat, PSMatrix& XMat, const PSMatrix& BMat, const SolverParameters& p_in) {
  use_grid_comm(AMat.grid);
  Monitor mon;
  const SolverParameters p = with_monitor(p_in, mon);
  if (p.be_verbose) {
    log_header("Linear Solver");
    log_enter();
    log_element("Method", "CG");
    print_parameters(p);
  }
  PSMatrix Identity, ABalanced, BBalanced, RMat, PMat, QMat, RMatT, PMatT, TempMat, X;
  ps_construct_like(Identity, AMat);
  ps_fill_identity(Identity);
  if (p.do_load_balancing) {
    ps_permute(Identity, Identity, p.balance_permutation, false);
    ps_permute(AMat, ABalanced, p.balance_permutation, false);
    ps_permute(BMat, BBalanced, p.balance_permutation, false);
  } else {
    ps_copy(AMat, ABalanced);
    ps_copy(BMat, BBalanced);
  }
  ps_copy(Identity, X);                                                        // initial guess X = I (:89)
  ps_multiply(ABalanced, X, TempMat, 1.0, 0.0, p.threshold);
  ps_copy(BBalanced, RMat);
  ps_increment(TempMat, RMat, -1.0, 0.0);
  ps_copy(RMat, PMat);
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  for (II = 1; II <= p.max_iterations; ++II) {                                 // :101-139
    ps_multiply(ABalanced, PMat, QMat, 1.0, 0.0, p.threshold);
    conj_transpose(RMat, RMatT);
    ps_multiply(RMatT, RMat, TempMat, 1.0, 0.0, p.threshold);
    const double top = ps_trace(TempMat);
    conj_transpose(PMat, PMatT);
    ps_multiply(PMatT, QMat, TempMat, 1.0, 0.0, p.threshold);
    const double bottom = ps_trace(TempMat);
    double step_size = top / bottom;
    ps_increment(PMat, X, step_size, 0.0);
    const double norm_value = std::fabs(step_size * ps_norm(PMat));
    ps_increment(QMat, RMat, -1.0 * step_size, 0.0);
    conj_transpose(RMat, RMatT);
    ps_multiply(RMatT, RMat, TempMat, 1.0, 0.0, p.threshold);
    const double new_top = ps_trace(TempMat);
    step_size = new_top / top;
    ps_scale(PMat, step_size);
    ps_increment(RMat, PMat, 1.0, 0.0);
    monitor_append(mon, norm_value);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II - 1);
    print_matrix_information(X);
  }
  if (p.do_load_balancing) ps_permute(X, X, p.balance_permutation, true);
  if (p.be_verbose) log_exit();
  XMat = std::move(X);
}

// ------------------------------------------------------------------ Pade exponential (ExponentialSolversModule.F90:152-271)
void compute_exponential_pade(const PSMatrix& In, PSMatrix& OutMat, const SolverParameters& p) {
  use_grid_comm(In.grid);
  if (p.be_verbose) {
    log_header("Exponential Solver");
    log_enter();
    log_element("Method", "Pade");
    print_parameters(p);
  }
  PSMatrix IdentityMat, ScaledMat, TempMat, B1, B2, B3, P1, P2, LeftMat, RightMat, Out;
  ps_construct_like(IdentityMat, In);
  ps_fill_identity(IdentityMat);
  const double spectral_radius = ps_norm(In);
  double sigma_val = 1.0;
  int sigma_counter = 1;
  while (spectral_radius / sigma_val > 1.0) {
    sigma_val *= 2;
    ++sigma_counter;
  }
  ps_copy(In, ScaledMat);
  // the reference divides in default REAL kind: 1.0 / sigma_val with sigma_val a power of two is exact either way
  ps_scale(ScaledMat, 1.0 / sigma_val);
  if (p.be_verbose) {
    log_element("Sigma", sigma_val);
    log_element("Scaling Steps", sigma_counter);
  }
  SolverParameters sub = p;
  sub.threshold = sub.threshold / sigma_val;
  ps_multiply(ScaledMat, ScaledMat, B1, 1.0, 0.0, sub.threshold);
  ps_multiply(B1, B1, B2, 1.0, 0.0, sub.threshold);
  ps_multiply(B2, B2, B3, 1.0, 0.0, sub.threshold);
  ps_copy(IdentityMat, P1);                                                    // :222-226
  ps_scale(P1, 17297280.0);
  ps_increment(B1, P1, 1995840.0, 0.0);
  ps_increment(B2, P1, 25200.0, 0.0);
  ps_increment(B3, P1, 56.0, 0.0);
  ps_copy(IdentityMat, TempMat);                                               // :228-234
  ps_scale(TempMat, 8648640.0);
  ps_increment(B1, TempMat, 277200.0, 0.0);
  ps_increment(B2, TempMat, 1512.0, 0.0);
  ps_increment(B3, TempMat, 1.0, 0.0);
  ps_multiply(ScaledMat, TempMat, P2, 1.0, 0.0, sub.threshold);
  ps_copy(P1, LeftMat);
  ps_increment(P2, LeftMat, -1.0, 0.0);
  ps_copy(P1, RightMat);
  ps_increment(P2, RightMat, 1.0, 0.0);
  solver_cg(LeftMat, Out, RightMat, sub);
  for (int II = 1; II <= sigma_counter - 1; ++II) {                            // undo the scaling by squaring
    ps_multiply(Out, Out, TempMat, 1.0, 0.0, p.threshold);
    ps_copy(TempMat, Out);
  }
  if (p.be_verbose) {
    print_matrix_information(Out);
    log_exit();
  }
  OutMat = std::move(Out);
}

// ------------------------------------------------------------------ GeometryOptimizationModule.F90
void purification_extrapolate(const PSMatrix& PreviousDensity, const PSMatrix& Overlap, double trace, PSMatrix& NewDensityOut,
                              const SolverParameters& p_in) {
  use_grid_comm(PreviousDensity.grid);                  // :24-136
  Monitor mon;
  const SolverParameters p = with_monitor(p_in, mon);
  if (p.be_verbose) {
    log_header("Density Matrix Extrapolator");
    log_enter();
    log_element("Method", "Purification");
    log_header("Citations");
    log_enter();
    log_list_element("niklasson2010trace");
    log_exit();
    print_parameters(p);
  }
  PSMatrix NewDensity, WorkingDensity, WorkingOverlap, TempMat;
  ps_construct_like(NewDensity, PreviousDensity);
  ps_copy(PreviousDensity, WorkingDensity);
  ps_copy(Overlap, WorkingOverlap);
  if (p.do_load_balancing) {
    ps_permute(WorkingDensity, WorkingDensity, p.balance_permutation, false);
    ps_permute(WorkingOverlap, WorkingOverlap, p.balance_permutation, false);
  }
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  for (II = 1; II <= p.max_iterations; ++II) {
    ps_multiply(WorkingDensity, WorkingOverlap, TempMat, 1.0, 0.0, p.threshold);
    ps_multiply(TempMat, WorkingDensity, NewDensity, 1.0, 0.0, p.threshold);
    double d[2];
    ps_dot(WorkingDensity, WorkingOverlap, d);
    const double trace_value = d[0];
    if (trace > trace_value) {
      ps_scale(NewDensity, -1.0);
      ps_increment(WorkingDensity, NewDensity, 2.0, 0.0);
    }
    ps_increment(NewDensity, WorkingDensity, -1.0, 0.0);
    const double norm_value = ps_norm(WorkingDensity);
    ps_copy(NewDensity, WorkingDensity);
    monitor_append(mon, norm_value);
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Trace", trace_value);
      log_exit();
    }
  }
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(NewDensity);
  }
  if (p.do_load_balancing) ps_permute(NewDensity, NewDensity, p.balance_permutation, true);
  if (p.be_verbose) log_exit();
  NewDensityOut = std::move(NewDensity);
}

void lowdin_extrapolate(const PSMatrix& PreviousDensity, const PSMatrix& OldOverlap, const PSMatrix& NewOverlap,
                        PSMatrix& NewDensity, const SolverParameters& p) {
  use_grid_comm(PreviousDensity.grid);     // :137-214
  if (p.be_verbose) {
    log_header("Density Matrix Extrapolator");
    log_enter();
    log_element("Method", "Lowdin");
    log_header("Citations");
    log_enter();
    log_list_element("exner2002comparison");
    log_exit();
    print_parameters(p);
  }
  PSMatrix SQRMat, ISQMat, TempMat, Out;
  solver_square_root(OldOverlap, SQRMat, p, false, 5);
  solver_square_root(NewOverlap, ISQMat, p, true, 5);
  ps_similarity(PreviousDensity, SQRMat, SQRMat, TempMat, p.threshold);
  ps_similarity(TempMat, ISQMat, ISQMat, Out, p.threshold);
  if (p.be_verbose) log_exit();
  NewDensity = std::move(Out);
}

// ------------------------------------------------------------------ MatrixConversionModule.F90:12-43
void snap_to_sparsity_pattern(PSMatrix& mat, const PSMatrix& pattern) {
  use_grid_comm(pattern.grid);
  PSMatrix ones, zeros, filtered;
  if (pattern.cplx) ps_to_real(pattern, ones);
  else ps_copy(pattern, ones);
  {  // every stored value of the pattern becomes 1
    const int32_t width = ones.c1 - ones.c0;
    ones.loc = to_real(ones.loc);
    if (ones.loc.nnz) {
      std::vector<double> one((size_t)ones.loc.nnz, 1.0);
      ones.loc.val.upload(one.data(), one.size());
      sync_stream();
    }
    (void)width;
  }
  ps_copy(ones, zeros);
  ps_scale(zeros, 0.0);
  if (mat.cplx) {
    PSMatrix zc, oc;
    ps_to_complex(zeros, zc);
    ps_to_complex(ones, oc);
    zeros = std::move(zc);
    ones = std::move(oc);
  }
  ps_increment(zeros, mat, 1.0, -1.0);   // union pattern, explicit zeros kept (threshold -1)
  ps_copy(mat, filtered);
  ps_pairwise(ones, filtered, mat);
}

// ------------------------------------------------------------------ eigendecomposition (EigenSolversModule.F90:33-71,
// eigenexa_includes/EigenSerial.f90): gather, dense Hermitian eigensolver (dense.hip), drop the pairs past nvals, sparsify with
// the threshold, hand every rank its panel.  Every rank factors the gathered matrix itself (same input, same
// code, same device type), so no broadcast of the vectors is needed.
void ps_eigendecomposition(const PSMatrix& A, PSMatrix& eigenvalues, PSMatrix* eigenvectors, int nvals,
                           const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (p.be_verbose) {
    log_header("Eigen Solver");
    log_enter();
    log_element("Method", "Jacobi");
    log_element("NVALS", nvals);
    log_exit();
    print_parameters(p);
  }
  const int32_t n = A.dim;
  const size_t w = A.cplx ? 2 : 1;
  PSMatrix vals, vecs;
  ps_construct_like(vals, A);
  ps_construct_like(vecs, A);
  std::vector<double> hw((size_t)n);
  {
    DevMat full = ps_gather_full(A);
    DevBuf<double> dense((size_t)n * (size_t)n * w), W((size_t)std::max(n, 1));
    to_dense(full, dense.p, n);
    dense_eigh(dense.p, n, A.cplx, W.p);
    if (nvals < n) dense_zero_columns(dense.p, n, n, std::max(nvals, 0), n, A.cplx);
    if (n) {
      HIP_CHECK(hipMemcpyAsync(hw.data(), W.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream()));
      sync_stream();
    }
    if (eigenvectors) vecs.loc = from_dense(dense.p, n, n, vecs.c0, vecs.c1 - vecs.c0, A.cplx, p.threshold);
    sync_stream();
  }
  HostTriplets t;
  t.cplx = false;
  for (int32_t c = vals.c0; c < vals.c1; ++c) {
    if (c >= nvals || !(std::fabs(hw[(size_t)c]) > p.threshold)) continue;
    t.col.push_back(c + 1);
    t.row.push_back(c + 1);
    t.val.push_back(hw[(size_t)c]);
  }
  vals.cplx = false;  // eigenvalues come back as a real matrix also for complex input (EigenSolversModule.F90:316-318)
  vals.loc = from_triplets(t, n, vals.c1 - vals.c0, vals.c0);
  eigenvalues = std::move(vals);
  if (eigenvectors) *eigenvectors = std::move(vecs);
}

// DenseMatrixFunction (EigenSolversModule.F90:74-131): f(A) = V f(L) V^H
void dense_matrix_function(const PSMatrix& A, PSMatrix& Result, const std::function<double(double)>& func,
                           const SolverParameters& p) {
  use_grid_comm(A.grid);
  PSMatrix vecs, vecsT, vals, Out;
  ps_eigendecomposition(A, vals, &vecs, A.dim, p);
  HostTriplets t;
  ps_gather_triplets(vals, t);
  for (size_t i = 0; i < t.size(); ++i) t.val[i] = func(t.val[i]);
  conj_transpose(vecs, vecsT);   // taken BEFORE the scaling (:132-133)
  if (A.cplx) {
    HostTriplets tc;
    tc.cplx = true;
    tc.col = t.col;
    tc.row = t.row;
    tc.val.resize(2 * t.size());
    for (size_t i = 0; i < t.size(); ++i) { tc.val[2 * i] = t.val[i]; tc.val[2 * i + 1] = 0.0; }
    ps_diagonal_scale(vecs, tc);
  } else {
    ps_diagonal_scale(vecs, t);
  }
  ps_multiply(vecs, vecsT, Out, 1.0, 0.0, p.threshold);
  Result = std::move(Out);
}

// SingularValueDecomposition (SingularValueSolversModule.F90:14-52): polar decomposition, then the eigenpairs of H
void ps_svd(const PSMatrix& A, PSMatrix& left, PSMatrix& right, PSMatrix& singular, const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (p.be_verbose) {
    log_header("Singular Value Solver");
    log_enter();
    log_element("Method", "Polar");
    print_parameters(p);
  }
  PSMatrix UMat, HMat, L, R, S;
  solver_polar(A, UMat, &HMat, p);
  ps_eigendecomposition(HMat, S, &R, HMat.dim, p);
  ps_multiply(UMat, R, L, 1.0, 0.0, p.threshold);
  if (p.be_verbose) log_exit();
  left = std::move(L);
  right = std::move(R);
  singular = std::move(S);
}

// EstimateGap (EigenSolversModule.F90:153-226)
void estimate_gap(const PSMatrix& H, const PSMatrix& K, double chemical_potential, double* gap, const SolverParameters& p) {
  use_grid_comm(H.grid);
  if (p.be_verbose) {
    log_header("Estimate Gap");
    log_enter();
    print_parameters(p);
  }
  PSMatrix KH, ShiftH;
  double e_min, e_max;
  ps_multiply(K, H, KH, 1.0, 0.0, p.threshold);
  if (p.be_verbose) {
    log_header("Estimate Minimum");
    log_enter();
  }
  power_bounds(KH, &e_min, p, false);
  if (p.be_verbose) log_exit();
  if (e_min > 0.0) ps_gershgorin(H, &e_min, &e_max);
  if (p.be_verbose) log_element("Estimated e_min", e_min);
  ps_construct_like(ShiftH, H);
  ps_fill_identity(ShiftH);
  ps_scale(ShiftH, -e_min);
  ps_increment(H, ShiftH, 1.0, 0.0);
  ps_multiply(K, ShiftH, KH, 1.0, 0.0, p.threshold);
  power_bounds(KH, &e_max, p, false);
  e_max = e_max + e_min;
  *gap = 2.0 * (chemical_potential - e_max);
  if (p.be_verbose) {
    log_element("HOMO Estimate", e_max);
    log_element("Gap Estimate", *gap);
    log_exit();
  }
}

// ------------------------------------------------------------------ FermiOperatorModule.F90
namespace {
double foe_erf(double x) {  // :531-546 (its own rational approximation, kept for identical occupations)
  const double a1 = 0.254829592, a2 = -0.284496736, a3 = 1.421413741, a4 = -1.453152027, a5 = 1.061405429, pp = 0.3275911;
  const double z = std::fabs(x);
  const double t = 1.0 / (1.0 + pp * z);
  const double tau = t * (a1 + t * (a2 + t * (a3 + t * (a4 + t * a5))));
  return std::copysign(1.0, x) * (1.0 - tau * std::exp(-z * z));
}
}  // namespace

void compute_dense_foe(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, const double* inv_temp_in,
                       double* energy_out, double* mu_out, const SolverParameters& p) {
  use_grid_comm(H.grid);  // :31-248
  const bool do_smearing = inv_temp_in != nullptr;
  const double inv_temp = do_smearing ? *inv_temp_in : 0.0;
  if (p.be_verbose) {
    log_header("Density Matrix Solver");
    log_enter();
    if (do_smearing) {
      log_element("Method", "Dense FOE");
      log_element("Inverse Temperature", inv_temp);
    } else {
      log_element("Method", "Dense Step Function");
    }
    print_parameters(p);
  }
  PSMatrix ISQT, WH, WD, vecs, vecsT, vals, Temp, Out;
  ps_transpose(ISQ, ISQT);
  ps_multiply(ISQ, H, Temp, 1.0, 0.0, p.threshold);
  ps_multiply(Temp, ISQT, WH, 1.0, 0.0, p.threshold);
  ps_eigendecomposition(WH, vals, &vecs, WH.dim, p);
  HostTriplets t;
  ps_gather_triplets(vals, t);
  const int num_eigs = H.dim;
  std::vector<double> eigs((size_t)num_eigs, 0.0), occ;
  for (size_t i = 0; i < t.size() && i < (size_t)num_eigs; ++i) eigs[i] = t.val[i];
  double chemical_potential = 0.0;
  int JJ = 1;
  if (do_smearing) {                                                           // bisection on mu (:115-136)
    occ.resize((size_t)num_eigs);
    double left = *std::min_element(eigs.begin(), eigs.end());
    double right = *std::max_element(eigs.begin(), eigs.end());
    for (JJ = 1; JJ <= 10 * p.max_iterations; ++JJ) {
      chemical_potential = left + (right - left) / 2;
      double sv = 0.0;
      for (int i = 0; i < num_eigs; ++i) {
        const double sval = eigs[(size_t)i] - chemical_potential;
        if (inv_temp * sval > 30) occ[(size_t)i] = 0.5 * (1.0 - foe_erf(inv_temp * sval));
        else occ[(size_t)i] = 1.0 / (1.0 + std::exp(inv_temp * sval));
        sv += occ[(size_t)i];
      }
      if (std::fabs(trace - sv) < 1e-8) break;
      else if (sv > trace) right = chemical_potential;
      else left = chemical_potential;
    }
  } else {                                                                     // :137-143
    const int fl = (int)std::floor(trace);
    const double homo = eigs[(size_t)std::max(fl - 1, 0)];
    const double lumo = eigs[(size_t)std::min(fl, num_eigs - 1)];
    const double occ_temp = fl + 1 - trace;
    chemical_potential = homo + occ_temp * 0.5 * (lumo - homo);
  }
  if (p.be_verbose) {
    log_header("Chemical Potential Search");
    log_enter();
    log_element("Potential", chemical_potential);
    log_element("Iterations", JJ);
    log_exit();
  }
  double energy_value = 0.0;                                                   // occupations -> sqrt factors (:153-180)
  for (size_t i = 0; i < t.size(); ++i) {
    if (!do_smearing) {
      if (t.col[i] <= (int)std::floor(trace)) {
        energy_value += t.val[i];
        t.val[i] = 1.0;
      } else if (t.col[i] == (int)std::ceil(trace)) {
        const double occ_temp = trace - std::floor(trace);
        energy_value += occ_temp * t.val[i];
        t.val[i] = std::sqrt(occ_temp);
      } else {
        t.val[i] = 0.0;
      }
    } else {
      const double sval = t.val[i] - chemical_potential;
      const double occ_temp = 1.0 / (1.0 + std::exp(inv_temp * sval));
      energy_value += occ_temp * t.val[i];
      t.val[i] = occ_temp < 0 ? 0.0 : std::sqrt(occ_temp);
    }
  }
  if (vecs.cplx) {
    HostTriplets tc;
    tc.cplx = true;
    tc.col = t.col;
    tc.row = t.row;
    tc.val.resize(2 * t.size());
    for (size_t i = 0; i < t.size(); ++i) { tc.val[2 * i] = t.val[i]; tc.val[2 * i + 1] = 0.0; }
    ps_diagonal_scale(vecs, tc);
  } else {
    ps_diagonal_scale(vecs, t);
  }
  ps_filter(vecs, p.threshold);
  conj_transpose(vecs, vecsT);
  ps_multiply(vecs, vecsT, WD, 1.0, 0.0, p.threshold);
  ps_multiply(ISQT, WD, Temp, 1.0, 0.0, p.threshold);
  ps_multiply(Temp, ISQ, Out, 1.0, 0.0, p.threshold);
  if (energy_out) *energy_out = energy_value;
  if (mu_out) *mu_out = chemical_potential;
  if (p.be_verbose) log_exit();
  K = std::move(Out);
}

namespace {
// ComputeX (:449-473): X = W (I - W^2), optionally hands W^2 back
void wom_compute_x(const PSMatrix& W, const PSMatrix& I, double threshold, PSMatrix& Out, PSMatrix* W2_out) {
  PSMatrix W2, Temp;
  ps_multiply(W, W, W2, 1.0, 0.0, threshold);
  ps_copy(W2, Temp);
  ps_scale(Temp, -1.0);
  ps_increment(I, Temp, 1.0, threshold);
  ps_multiply(W, Temp, Out, 1.0, 0.0, threshold);
  if (W2_out) *W2_out = std::move(W2);
}
void wom_gc_step(const PSMatrix& X, const PSMatrix& A, double threshold, PSMatrix& Out) {  // :474-483
  ps_multiply(X, A, Out, -0.5, 0.0, threshold);
}
void wom_c_step(const PSMatrix& X, const PSMatrix& A, const PSMatrix& W, double threshold, PSMatrix& Out) {  // :484-507
  PSMatrix XA;
  ps_multiply(X, A, XA, 1.0, 0.0, threshold);
  double d[2];
  ps_dot(X, W, d);
  const double denom = d[0];
  ps_dot(W, XA, d);
  const double num = d[0];
  ps_copy(X, Out);
  ps_scale(Out, -1.0 * num / denom);
  ps_increment(XA, Out, 1.0, 0.0);
  ps_scale(Out, -0.5);
}
}  // namespace

// WOM_Implementation (:317-447): adaptive Heun integration of the wave-operator flow up to inv_temp
void solver_wom(const PSMatrix& H, const PSMatrix& ISQ, PSMatrix& K, double inv_temp, const double* trace_in,
                const double* mu_in, double* energy_out, const SolverParameters& p) {
  use_grid_comm(H.grid);
  const bool GC = mu_in != nullptr;
  if (p.be_verbose) {
    log_header("Density Matrix Solver");
    log_enter();
    log_element("Method", GC ? "WOM_GC" : "WOM_C");
    log_element("Inverse Temperature", inv_temp);
    if (GC) log_element("Chemical Potential", *mu_in);
    else log_element("Target Trace", *trace_in);
    print_parameters(p);
  }
  PSMatrix ISQT, WH, IMat, RK1, RK2, K0, K1, Temp, W, A, X, KOrth, Out;
  ps_construct_like(IMat, H);
  ps_fill_identity(IMat);
  ps_transpose(ISQ, ISQT);
  ps_similarity(H, ISQ, ISQT, WH, p.threshold);
  if (p.do_load_balancing) {
    ps_permute(WH, WH, p.balance_permutation, false);
    ps_permute(IMat, IMat, p.balance_permutation, false);
  }
  ps_copy(WH, A);
  if (GC) ps_increment(IMat, A, -1.0 * (*mu_in), 0.0);
  ps_copy(IMat, W);
  if (GC) ps_scale(W, 1.0 / std::sqrt(2.0));
  else ps_scale(W, std::sqrt(*trace_in / (double)WH.dim));
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  auto gradient = [&](const PSMatrix& Xm, const PSMatrix& Wm, PSMatrix& Kout) {
    if (GC) wom_gc_step(Xm, A, p.threshold, Kout);
    else wom_c_step(Xm, A, Wm, p.threshold, Kout);
  };
  int II = 0;
  double B_I = 0.0, step = 1.0, energy = 0.0;
  while (B_I < inv_temp) {
    step = std::fmin(step, inv_temp - B_I);
    wom_compute_x(W, IMat, p.threshold, X, &KOrth);
    double d[2];
    ps_dot(WH, KOrth, d);
    energy = d[0];
    gradient(X, W, K0);
    ++II;
    double err = 0.0;
    auto trial = [&]() {                                                       // :358-377 / :380-398
      ps_copy(K0, RK1);
      ps_scale(RK1, step);
      ps_increment(W, RK1, 1.0, p.threshold);
      wom_compute_x(RK1, IMat, p.threshold, X, nullptr);
      gradient(X, RK1, K1);
      ++II;
      ps_copy(W, RK2);
      ps_increment(K0, RK2, step * 0.5, p.threshold);
      ps_increment(K1, RK2, step * 0.5, p.threshold);
      ps_copy(RK1, Temp);
      ps_increment(RK2, Temp, -1.0, p.threshold);
      err = ps_norm(Temp);
    };
    trial();
    while (err > 1.1 * p.step_thresh) {
      step = step * std::pow(p.step_thresh / err, 0.5);
      trial();
    }
    ps_copy(RK2, Temp);
    ps_increment(W, Temp, -1.0, p.threshold);
    const double err2 = ps_norm(Temp);
    if (err2 < p.converge_diff) break;                                         // "Early Exit Triggered"
    ps_copy(RK2, W);
    const double B_I_old = B_I;
    B_I = B_I + step;
    step = step * std::pow(p.step_thresh / err, 0.5);
    if (p.be_verbose) {
      const double sparsity = (double)ps_size(W) / ((double)W.dim * (double)W.dim);
      log_list_element("Gradient Evaluations", (double)II);
      log_enter();
      log_element("Beta", B_I_old);
      log_element("Sparsity", sparsity);
      log_element("Energy", energy);
      log_element("Norm of Change", err2);
      log_exit();
    }
  }
  ps_multiply(W, W, KOrth, 1.0, 0.0, p.threshold);
  if (p.be_verbose) {
    log_exit();
    log_element("Total_Iterations", II);
    print_matrix_information(W);
  }
  if (energy_out) {
    double d[2];
    ps_dot(WH, KOrth, d);
    *energy_out = d[0];
  }
  if (p.do_load_balancing) ps_permute(KOrth, KOrth, p.balance_permutation, true);
  ps_similarity(KOrth, ISQT, ISQ, Out, p.threshold);
  if (p.be_verbose) log_exit();
  K = std::move(Out);
}

// ------------------------------------------------------------------ Cholesky (LinearSolversModule.F90:174-300,
// AnalysisModule.F90:26-196).  The reference keeps a dense copy of every rank's block and walks the columns one
// at a time with a broadcast per column; here every rank factors the gathered dense matrix on its GPU and keeps
// its own column panel of L.  rank < 0: plain Cholesky.
void ps_cholesky(const PSMatrix& A, PSMatrix& L, int rank, const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (A.cplx) NTP_FATAL("CholeskyDecomposition: real matrices only (as the reference, LinearSolversModule.F90:186-187)");
  if (p.be_verbose) {
    log_header("Linear Solver");
    log_enter();
    if (rank < 0) {
      log_element("Method", "Cholesky Decomposition");
    } else {
      log_element("Method", "Pivoted Cholesky Decomposition");
      log_element("Target_Rank", rank);
      log_header("Citations");
      log_enter();
      log_list_element("aquilante2006fast");
      log_exit();
    }
    print_parameters(p);
  }
  const int32_t n = A.dim;
  PSMatrix Out;
  ps_construct_like(Out, A);
  {
    DevMat full = ps_gather_full(A);
    DevBuf<double> dA((size_t)n * (size_t)n), dL((size_t)n * (size_t)n);
    to_dense(full, dA.p, n);
    dense_cholesky(dA.p, dL.p, n, p.threshold, rank);
    Out.loc = from_dense(dL.p, n, n, Out.c0, Out.c1 - Out.c0, false, 0.0);
    sync_stream();
  }
  if (p.be_verbose) {
    print_matrix_information(Out);
    log_exit();
  }
  L = std::move(Out);
}

// ReduceDimension (AnalysisModule.F90:199-245)
void reduce_dimension(const PSMatrix& A, int dim, PSMatrix& Reduced, const SolverParameters& p) {
  use_grid_comm(A.grid);
  PSMatrix Identity, PMat, PVec, PVecT, VAV;
  ps_construct_like(Identity, A);
  ps_fill_identity(Identity);
  double energy, mu;
  solver_trs4(A, Identity, (double)dim, PMat, &energy, &mu, p);
  ps_cholesky(PMat, PVec, dim, p);
  conj_transpose(PVec, PVecT);
  ps_similarity(A, PVecT, PVec, VAV, p.threshold);
  ps_get_slice(VAV, Reduced, 1, dim, 1, dim);
}

}  // namespace ntp
