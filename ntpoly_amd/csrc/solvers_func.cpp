// Matrix functions built on the polynomial solvers and on SquareRoot / Invert:
//   EigenBoundsModule.F90:60-189        PowerBounds (power iteration + Aitken extrapolation)
//   ExponentialSolversModule.F90:32-143 ComputeExponential (scaling and squaring around a degree-15 Chebyshev fit)
//   ExponentialSolversModule.F90       ComputeLogarithm (repeated square roots + degree-31 Chebyshev fit of log(1+x))
//   TrigonometrySolversModule.F90:30-121, 263-411  Sine / Cosine (Chebyshev fit of cos + double-angle recurrence)
//   RootSolversModule.F90:31-337        ComputeRoot / ComputeInverseRoot (coupled Newton iteration, Higham)
// Same control flow and constants as the reference (the Chebyshev coefficient tables are its data).
#include <cmath>
#include <vector>

#include "engine.hpp"

namespace ntp {

// ------------------------------------------------------------------ PowerBounds
void power_bounds(const PSMatrix& A, double* max_value_out, const SolverParameters& p_in, bool defaults) {
  use_grid_comm(A.grid);
  SolverParameters p = p_in;
  if (defaults) p.max_iterations = 10;  // :81-86
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  if (p.be_verbose) {
    log_header("Power Bounds Solver");
    log_enter();
    print_parameters(p);
  }
  // the reference's "vector" is a matrix whose first row is 1/N in every column (:98-113): every column carries the
  // same power iteration started from e_1
  PSMatrix vec, vec2;
  ps_construct_like(vec, A);
  {
    HostTriplets t;
    t.cplx = false;
    for (int32_t c = vec.c0; c < vec.c1; ++c) {
      t.col.push_back(c + 1);
      t.row.push_back(1);
      t.val.push_back(1.0 / (double)vec.dim);
    }
    PSMatrix tmp;
    ps_construct_empty(tmp, A.dim, A.grid, false);
    tmp.loc = from_triplets(t, tmp.dim, tmp.c1 - tmp.c0, tmp.c0);
    if (A.cplx) ps_to_complex(tmp, vec);
    else vec = std::move(tmp);
  }
  double ritz[3] = {0, 0, 0}, aitken[3] = {0, 0, 0}, max_value = 0.0;
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  for (II = 1; II <= p.max_iterations; ++II) {                     // :121-165
    ps_multiply(A, vec, vec2, 1.0, 0.0, p.threshold);
    double d1[2], d2[2];
    ps_dot(vec, vec, d1);
    ps_dot(vec, vec2, d2);
    max_value = d2[0] / d1[0];
    const double scale_value = 1.0 / ps_norm(vec2);
    ps_scale(vec2, scale_value);
    std::swap(vec.loc, vec2.loc);
    ritz[0] = ritz[1]; ritz[1] = ritz[2]; ritz[2] = max_value;
    aitken[0] = aitken[1]; aitken[1] = aitken[2];
    if (II >= 3) {
      const double num = ritz[2] * ritz[0] - ritz[1] * ritz[1];
      const double den = ritz[2] - 2 * ritz[1] + ritz[0];
      aitken[2] = std::fabs(den) > 1e-14 ? num / den : ritz[2];
    } else {
      aitken[2] = ritz[2];
    }
    monitor_append(mon, -(aitken[2] - aitken[1]));
    if (monitor_converged(mon, p.be_verbose)) {
      if (std::fabs(aitken[2] - ritz[2]) < mon.loose_cutoff) break;
    }
    if (p.be_verbose) {
      log_enter();
      log_element("Estimate", ritz[2]);
      log_element("Aitken Estimate", aitken[2]);
      log_exit();
    }
  }
  *max_value_out = aitken[2];
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II - 1);
    log_element("Max Eigen Value", aitken[2]);
    log_exit();
  }
}

// ------------------------------------------------------------------ exponential
void compute_exponential(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(In.grid);
  SolverParameters sub = p, psub = p;
  psub.max_iterations = 10;
  if (p.be_verbose) {
    log_header("Exponential Solver");
    log_enter();
    log_element("Method", "Chebyshev");
    print_parameters(p);
  }
  double spectral_radius;
  power_bounds(In, &spectral_radius, psub, false);
  double sigma_val = 1.0;
  int sigma_counter = 1;
  while (spectral_radius / sigma_val > 1.0) {
    sigma_val *= 2;
    sigma_counter += 1;
  }
  PSMatrix Scaled, R, Temp;
  ps_copy(In, Scaled);
  ps_scale(Scaled, 1.0 / sigma_val);
  sub.threshold = sub.threshold / sigma_val;
  if (p.be_verbose) log_element("Sigma", sigma_val);
  // Chebyshev coefficients of exp on [-1, 1] (ExponentialSolversModule.F90:84-100)
  static const double c[16] = {1.266065877752007e+00, 1.130318207984970e+00, 2.714953395340771e-01, 4.433684984866504e-02,
                               5.474240442092110e-03, 5.429263119148932e-04, 4.497732295351912e-05, 3.198436462630565e-06,
                               1.992124801999838e-07, 1.103677287249654e-08, 5.505891628277851e-10, 2.498021534339559e-11,
                               1.038827668772902e-12, 4.032447357431817e-14, 2.127980007794583e-15, -1.629151584468762e-16};
  chebyshev_compute(Scaled, R, std::vector<double>(c, c + 16), sub);
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(R, t, p.balance_permutation, false);
    R = std::move(t);
  }
  for (int k = 1; k <= sigma_counter - 1; ++k) {
    ps_multiply(R, R, Temp, 1.0, 0.0, p.threshold);
    std::swap(R.loc, Temp.loc);
  }
  if (p.be_verbose) print_matrix_information(R);
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(R, t, p.balance_permutation, true);
    R = std::move(t);
  }
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ trigonometry
namespace {
void scale_square_trig(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p) {  // :263-411
  if (p.be_verbose) {
    log_header("Trigonometry Solver");
    log_enter();
    log_element("Method", "Chebyshev");
    log_header("Citations");
    log_enter();
    log_list_element("serbin1980algorithm");
    log_list_element("higham2003computing");
    log_list_element("yau1993reducing");
    log_exit();
    print_parameters(p);
  }
  double e_min, e_max;
  ps_gershgorin(In, &e_min, &e_max);
  const double spectral_radius = std::max(std::fabs(e_min), std::fabs(e_max));
  double sigma_val = 1.0;
  int sigma_counter = 1;
  while (spectral_radius / sigma_val > 1.0) {
    sigma_val *= 2;
    sigma_counter += 1;
  }
  PSMatrix Scaled, Ident, T2, T4, T6, T8, R, Temp;
  ps_copy(In, Scaled);
  ps_scale(Scaled, 1.0 / sigma_val);
  ps_construct_like(Ident, In);
  ps_fill_identity(Ident);
  if (p.do_load_balancing) {
    PSMatrix t, u;
    ps_permute(Scaled, t, p.balance_permutation, false);
    Scaled = std::move(t);
    ps_permute(Ident, u, p.balance_permutation, false);
    Ident = std::move(u);
  }
  // Chebyshev coefficients of cos on [-1, 1], even terms only (:309-325), 1-based as in the reference
  double c[18] = {0};
  c[1] = 7.651976865579664e-01;  c[3] = -2.298069698638004e-01; c[5] = 4.953277928219409e-03;
  c[7] = -4.187667600472235e-05; c[9] = 1.884468822397086e-07;  c[11] = -5.261224549346905e-10;
  c[13] = 9.999906645345580e-13; c[15] = -2.083597362700025e-15; c[17] = 9.181480886537484e-17;
  SlabSession slab(!Scaled.cplx);   // (engine.hpp: the Chebyshev evaluation and the squarings on matrices kept in slab form)
  ps_multiply(Scaled, Scaled, T2, 2.0, 0.0, p.threshold);
  ps_increment(Ident, T2, -1.0, 0.0);
  ps_multiply(T2, T2, T4, 2.0, 0.0, p.threshold);
  ps_increment(Ident, T4, -1.0, 0.0);
  ps_multiply(T4, T2, T6, 2.0, 0.0, p.threshold);
  ps_increment(T2, T6, -1.0, 0.0);
  ps_multiply(T6, T2, T8, 2.0, 0.0, p.threshold);
  ps_increment(T4, T8, -1.0, 0.0);

  ps_copy(T8, R);
  ps_scale(R, 0.5 * c[17]);
  ps_increment(T6, R, 0.5 * c[15], 0.0);
  ps_increment(T4, R, 0.5 * c[13], 0.0);
  ps_increment(T2, R, 0.5 * c[11], 0.0);
  ps_multiply(T8, R, Temp, 1.0, 0.0, p.threshold);

  ps_copy(T8, R);
  ps_scale(R, c[9]);
  ps_increment(T6, R, c[7] + 0.5 * c[11], 0.0);
  ps_increment(T4, R, c[5] + 0.5 * c[13], 0.0);
  ps_increment(T2, R, c[3] + 0.5 * c[15], 0.0);
  ps_increment(Ident, R, c[1] + 0.5 * c[17], 0.0);
  ps_increment(Temp, R, 1.0, 0.0);

  for (int II = 1; II <= sigma_counter - 1; ++II) {   // cos(2x) = 2 cos^2(x) - 1
    ps_multiply(R, R, Temp, 1.0, 0.0, p.threshold);
    std::swap(R.loc, Temp.loc);
    ps_scale(R, 2.0);
    ps_increment(Ident, R, -1.0, 0.0);
  }
  slab.close();
  ps_slab_leave(R);
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(R, t, p.balance_permutation, true);
    R = std::move(t);
  }
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}
}  // namespace

void compute_cosine(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(In.grid); scale_square_trig(In, Out, p); }

void compute_sine(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(In.grid);  // sin(x) = cos(x - pi/2), :30-64
  const double PI = 4 * std::atan(1.0);
  PSMatrix Shifted, Ident;
  ps_copy(In, Shifted);
  ps_construct_like(Ident, In);
  ps_fill_identity(Ident);
  ps_increment(Ident, Shifted, -1.0 * PI / 2.0, 0.0);
  scale_square_trig(Shifted, Out, p);
}

// ------------------------------------------------------------------ roots
void compute_inverse_root(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p);

namespace {
void inverse_root_impl(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p) {  // :177-337
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  if (p.be_verbose) {
    log_header("Root Solver");
    log_enter();
    log_header("Citations");
    log_enter();
    log_list_element("nicholas2008functions");
    log_exit();
    print_parameters(p);
  }
  double e_min, e_max;
  ps_gershgorin(In, &e_min, &e_max);
  const double scaling_factor = e_max / std::pow(std::sqrt(2.0), 1.0 / root);
  int target_root;
  if (root % 4 == 0) target_root = root / 4;
  else if (root % 4 == 1 || root % 4 == 3) target_root = root;
  else target_root = (root - 2) / 2 + 1;
  PSMatrix SqrtMat, Fthrt, Ident, Mk, Inter, InterP, Temp, R;
  solver_square_root(In, SqrtMat, p, false, 5);
  solver_square_root(SqrtMat, Fthrt, p, false, 5);
  ps_construct_like(Ident, In);
  ps_fill_identity(Ident);
  if (p.do_load_balancing) {
    PSMatrix t, u;
    ps_permute(Fthrt, t, p.balance_permutation, false);
    Fthrt = std::move(t);
    ps_permute(Ident, u, p.balance_permutation, false);
    Ident = std::move(u);
  }
  ps_copy(Ident, R);
  ps_scale(R, 1.0 / scaling_factor);
  ps_copy(Fthrt, Mk);
  ps_scale(Mk, 1.0 / std::pow(scaling_factor, target_root));
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  double norm_value = p.converge_diff + 1.0;
  int II;
  SlabSession slab(!R.cplx && !Mk.cplx);   // (engine.hpp: the loop's matrices stay in slab form between its operations)
  for (II = 1; II <= p.max_iterations; ++II) {
    if (p.be_verbose && II > 1) log_list_element("Convergence", norm_value);
    ps_copy(Ident, Inter);
    ps_scale(Inter, (double)(target_root + 1));
    ps_increment(Mk, Inter, -1.0, 0.0);
    ps_scale(Inter, 1.0 / target_root);
    ps_multiply(R, Inter, Temp, 1.0, 0.0, p.threshold);
    std::swap(R.loc, Temp.loc);
    ps_copy(Inter, InterP);
    for (int JJ = 1; JJ <= target_root - 1; ++JJ) {
      ps_multiply(Inter, InterP, Temp, 1.0, 0.0, p.threshold);
      std::swap(InterP.loc, Temp.loc);
    }
    ps_multiply(InterP, Mk, Temp, 1.0, 0.0, p.threshold);
    ps_copy(Temp, Mk);
    if (!ps_norm_axpby(Ident, Temp, -1.0, 1.0, &norm_value)) {       // (Temp - I is only there for its norm)
      ps_increment(Ident, Temp, -1.0, 0.0);
      norm_value = ps_norm(Temp);
    }
    monitor_append(mon, norm_value);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II - 1);
    print_matrix_information(R);
  }
  if (root % 4 == 1 || root % 4 == 3) {
    ps_multiply(R, R, Temp, 1.0, 0.0, p.threshold);
    PSMatrix R2;
    ps_multiply(Temp, Temp, R2, 1.0, 0.0, p.threshold);
    R = std::move(R2);
  } else if (root % 4 != 0) {
    ps_multiply(R, R, Temp, 1.0, 0.0, p.threshold);
    std::swap(R.loc, Temp.loc);
  }
  slab.close();
  ps_slab_leave(R);
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(R, t, p.balance_permutation, true);
    R = std::move(t);
  }
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

void root_impl(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p) {  // :86-121
  std::vector<double> c((size_t)root, 0.0);
  c[(size_t)root - 1] = 1.0;
  PSMatrix Raised, Temp, R;
  polynomial_paterson_stockmeyer(In, Raised, c, p);   // InputMat^(root-1)
  compute_inverse_root(Raised, Temp, root, p);
  ps_multiply(In, Temp, R, 1.0, 0.0, p.threshold);
  Out = std::move(R);
}
}  // namespace

void compute_root(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p) {
  use_grid_comm(In.grid);  // :31-83
  if (p.be_verbose) {
    log_header("Root Solver");
    log_enter();
    log_element("Root", root);
    print_parameters(p);
  }
  if (root == 1) {
    PSMatrix R;
    ps_copy(In, R);
    Out = std::move(R);
  } else if (root == 2) {
    PSMatrix R;
    solver_square_root(In, R, p, false, 5);
    Out = std::move(R);
  } else if (root == 3) {
    PSMatrix Temp;
    ps_multiply(In, In, Temp, 1.0, 0.0, p.threshold);
    root_impl(Temp, Out, 6, p);
  } else if (root == 4) {
    PSMatrix Temp, R;
    solver_square_root(In, Temp, p, false, 5);
    solver_square_root(Temp, R, p, false, 5);
    Out = std::move(R);
  } else {
    root_impl(In, Out, root, p);
  }
  if (p.be_verbose) log_exit();
}

void compute_inverse_root(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p) {
  use_grid_comm(In.grid);  // :124-174
  if (p.be_verbose) {
    log_header("Inverse Root Solver");
    log_enter();
    log_element("Root", root);
    print_parameters(p);
  }
  if (root == 1) {
    PSMatrix R;
    solver_invert(In, R, p);
    Out = std::move(R);
  } else if (root == 2) {
    PSMatrix R;
    solver_square_root(In, R, p, true, 5);
    Out = std::move(R);
  } else if (root == 3) {
    PSMatrix Temp, R;
    compute_root(In, Temp, 3, p);
    solver_invert(Temp, R, p);
    Out = std::move(R);
  } else if (root == 4) {
    PSMatrix Temp, R;
    solver_square_root(In, Temp, p, false, 5);
    solver_square_root(Temp, R, p, true, 5);
    Out = std::move(R);
  } else {
    inverse_root_impl(In, Out, root, p);
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ logarithm
void compute_logarithm(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(In.grid);
  SolverParameters isub = p, psub = p, fsub = p;
  psub.max_iterations = 16;
  if (p.be_verbose) {
    log_header("Logarithm Solver");
    log_enter();
    log_element("Method", "Chebyshev");
    print_parameters(p);
  }
  PSMatrix Ident, Scaled, R;
  ps_construct_like(Ident, In);
  ps_fill_identity(Ident);
  int sigma_val = 1, sigma_counter = 1;
  double spectral_radius;
  power_bounds(In, &spectral_radius, psub, false);
  while (spectral_radius > std::sqrt(2.0)) {
    spectral_radius = std::sqrt(spectral_radius);
    sigma_val *= 2;
    sigma_counter += 1;
  }
  if (p.be_verbose) log_element("Sigma", sigma_val);
  fsub.threshold = fsub.threshold / (double)(1 << (sigma_counter - 1));
  compute_root(In, Scaled, sigma_val, isub);
  ps_increment(Ident, Scaled, -1.0, 0.0);
  // Chebyshev coefficients of log(1 + x) (ExponentialSolversModule.F90, ComputeLogarithm)
  static const double c[32] = {
      -0.485101351704, 1.58828112379, -0.600947731795, 0.287304748177, -0.145496447103, 0.0734013668818,
      -0.0356277942958, 0.0161605505166, -0.0066133591188, 0.00229833505456, -0.000577804103964, 2.2849332964e-05,
      8.37426826403e-05, -6.10822859027e-05, 2.58132364523e-05, -5.87577322647e-06, -8.56711062722e-07,
      1.52066488969e-06, -7.12760496253e-07, 1.23102245249e-07, 6.03168259043e-08, -5.1865499826e-08,
      1.43185107512e-08, 2.58449717089e-09, -3.73189861771e-09, 1.18469334815e-09, 1.51569931066e-10,
      -2.89595999673e-10, 1.26720668874e-10, -3.00079067694e-11, 3.91175568865e-12, -2.21155654398e-13};
  chebyshev_factorized(Scaled, R, std::vector<double>(c, c + 32), fsub);
  ps_scale(R, (double)(1 << (sigma_counter - 1)));
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

}  // namespace ntp
