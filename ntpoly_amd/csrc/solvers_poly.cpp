// Matrix polynomials: host-side callers of the distributed algebra.
//   PolynomialSolversModule.F90  : Horner (Compute_stand :74-161), Paterson-Stockmeyer (FactorizedCompute_stand :165-282)
//   ChebyshevSolversModule.F90   : three-term recurrence (Compute_cheby :69-160), divide and conquer
//                                  (FactorizedCompute_cheby :163-247 + ComputeRecursive :250-365)
//   HermiteSolversModule.F90     : physicists' Hermite recurrence (Compute_horner :52-184)
// Same control flow, same order of operations, same use (or, for Paterson-Stockmeyer, non-use) of the threshold.
#include <cmath>
#include <vector>

#include "engine.hpp"

namespace ntp {

namespace {
struct Balanced {  // identity + input, permuted when load balancing is on (the common preamble of these solvers)
  PSMatrix Identity, Input;
};
void balanced_setup(const PSMatrix& In, const SolverParameters& p, Balanced& b) {
  ps_construct_like(b.Identity, In);
  ps_fill_identity(b.Identity);
  ps_copy(In, b.Input);
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(b.Identity, t, p.balance_permutation, false);
    b.Identity = std::move(t);
    PSMatrix u;
    ps_permute(b.Input, u, p.balance_permutation, false);
    b.Input = std::move(u);
  }
}
void balanced_finish(PSMatrix& Out, const SolverParameters& p) {
  if (p.do_load_balancing) {
    PSMatrix t;
    ps_permute(Out, t, p.balance_permutation, true);
    Out = std::move(t);
  }
}
void poly_header(const char* solver, const char* method, const char* citation, int degree, const SolverParameters& p,
                 bool degree_first) {
  if (!p.be_verbose) return;
  log_header(solver);
  log_enter();
  log_element("Method", method);
  if (citation) {
    log_header("Citations");
    log_enter();
    log_list_element(citation);
    log_exit();
  }
  if (degree_first) log_element("Degree", degree - 1);
  print_parameters(p);
  if (!degree_first) log_element("Degree", degree - 1);
}
}  // namespace

// ------------------------------------------------------------------ Horner
void polynomial_horner(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p) {
  use_grid_comm(In.grid);
  const int degree = (int)c.size();
  if (degree < 1) NTP_FATAL("polynomial without coefficients");
  poly_header("Polynomial Solver", "Horner", nullptr, degree, p, false);
  Balanced b;
  balanced_setup(In, p, b);
  SlabSession slab(!In.cplx);   // (engine.hpp: products, merges and scalings of the evaluation on matrices kept in slab form)
  PSMatrix R, Temporary;
  ps_copy(b.Identity, R);
  if (degree == 1) {
    ps_scale(R, c[(size_t)degree - 1]);
  } else {
    ps_scale(R, c[(size_t)degree - 2]);
    ps_increment(b.Input, R, c[(size_t)degree - 1], 0.0);
    for (int II = degree - 2; II >= 1; --II) {
      ps_multiply(b.Input, R, Temporary, 1.0, 0.0, p.threshold);
      std::swap(R.loc, Temporary.loc);
      ps_increment(b.Identity, R, c[(size_t)II - 1], 0.0);
    }
  }
  slab.close();
  ps_slab_leave(R);
  balanced_finish(R, p);
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Paterson-Stockmeyer
void polynomial_paterson_stockmeyer(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c,
                                    const SolverParameters& p) {
  use_grid_comm(In.grid);
  const int degree = (int)c.size();
  if (degree < 2) NTP_FATAL("Paterson-Stockmeyer needs a polynomial of degree >= 1");
  const int m_value = degree - 1;
  const int s_value = (int)std::sqrt((float)m_value);
  const int r_value = m_value / s_value;
  poly_header("Polynomial Solver", "Paterson Stockmeyer", "paterson1973number", degree, p, false);
  PSMatrix Identity;
  ps_construct_like(Identity, In);
  ps_fill_identity(Identity);
  SlabSession slab(!In.cplx);
  std::vector<PSMatrix> x_powers((size_t)s_value + 1);
  ps_construct_like(x_powers[0], In);
  ps_fill_identity(x_powers[0]);
  for (int II = 1; II <= s_value; ++II)  // no threshold here (the reference passes none, :229-232)
    ps_multiply(In, x_powers[(size_t)II - 1], x_powers[(size_t)II], 1.0, 0.0, 0.0);
  PSMatrix Xs, Bk, R, Temp;
  ps_copy(x_powers[(size_t)s_value], Xs);
  auto coef = [&](int one_based) { return c[(size_t)one_based - 1]; };

  ps_copy(Identity, Bk);
  ps_scale(Bk, coef(s_value * r_value + 1));
  for (int II = 1; II <= m_value - s_value * r_value; ++II) {
    const int c_index = s_value * r_value + II;
    ps_increment(x_powers[(size_t)II], Bk, coef(c_index + 1), 0.0);
  }
  ps_multiply(Bk, Xs, R, 1.0, 0.0, 0.0);

  int k_value = r_value - 1;
  ps_copy(Identity, Bk);
  ps_scale(Bk, coef(s_value * k_value + 1));
  for (int II = 1; II <= s_value - 1; ++II) {
    const int c_index = s_value * k_value + II;
    ps_increment(x_powers[(size_t)II], Bk, coef(c_index + 1), 0.0);
  }
  ps_increment(Bk, R, 1.0, 0.0);

  for (k_value = r_value - 2; k_value >= 0; --k_value) {
    ps_copy(Identity, Bk);
    ps_scale(Bk, coef(s_value * k_value + 1));
    for (int II = 1; II <= s_value - 1; ++II) {
      const int c_index = s_value * k_value + II;
      ps_increment(x_powers[(size_t)II], Bk, coef(c_index + 1), 0.0);
    }
    ps_multiply(Xs, R, Temp, 1.0, 0.0, 0.0);
    std::swap(R.loc, Temp.loc);
    ps_increment(Bk, R, 1.0, 0.0);
  }
  slab.close();
  ps_slab_leave(R);
  ps_slab_leave(const_cast<PSMatrix&>(In));   // (the session may have turned the caller's operand into slab form where it was)
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Chebyshev, three-term recurrence
void chebyshev_compute(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p) {
  use_grid_comm(In.grid);
  const int degree = (int)c.size();
  if (degree < 1) NTP_FATAL("polynomial without coefficients");
  poly_header("Chebyshev Solver", "Standard", nullptr, degree, p, true);
  Balanced b;
  balanced_setup(In, p, b);
  SlabSession slab(!In.cplx);
  PSMatrix Tk, Tkminus1, Tkminus2, R;
  ps_copy(b.Identity, Tkminus2);
  if (degree == 1) {
    ps_copy(Tkminus2, R);
    ps_scale(R, c[0]);
  } else {
    ps_copy(b.Input, Tkminus1);
    ps_copy(Tkminus2, R);
    ps_scale(R, c[0]);
    ps_increment(Tkminus1, R, c[1], 0.0);
    if (degree > 2) {
      ps_multiply(b.Input, Tkminus1, Tk, 2.0, 0.0, p.threshold);
      ps_increment(Tkminus2, Tk, -1.0, 0.0);
      ps_increment(Tk, R, c[2], 0.0);
      for (int II = 4; II <= degree; ++II) {
        std::swap(Tkminus2.loc, Tkminus1.loc);  // Tkminus2 <- Tkminus1
        std::swap(Tkminus1.loc, Tk.loc);        // Tkminus1 <- Tk (Tk now holds scratch)
        ps_multiply(b.Input, Tkminus1, Tk, 2.0, 0.0, p.threshold);
        ps_increment(Tkminus2, Tk, -1.0, 0.0);
        ps_increment(Tk, R, c[(size_t)II - 1], 0.0);
      }
    }
  }
  slab.close();
  ps_slab_leave(R);
  if (p.be_verbose) print_matrix_information(R);
  balanced_finish(R, p);
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Chebyshev, divide and conquer
namespace {
void cheby_recursive(const std::vector<PSMatrix>& T, const std::vector<double>& c, PSMatrix& Out, int depth,
                     const SolverParameters& p) {
  const int n = (int)c.size();
  if (n == 1) {
    ps_copy(T[0], Out);
    ps_scale(Out, c[0]);
  } else if (n == 2) {
    ps_copy(T[0], Out);
    ps_scale(Out, c[0]);
    ps_increment(T[1], Out, c[1], 0.0);
  } else {
    const int mid = n / 2;
    std::vector<double> left(c.begin(), c.begin() + mid), right(c.begin() + mid, c.end());
    for (int II = 2; II <= (int)left.size(); ++II) left[(size_t)II - 1] -= c[(size_t)(n - II + 2) - 1];
    PSMatrix LeftMat, RightMat, R;
    cheby_recursive(T, left, LeftMat, depth + 1, p);
    const int full_midpoint = (int)T.size() - depth + 1;  // 1-based index into T_Powers
    cheby_recursive(T, right, RightMat, depth + 1, p);
    ps_multiply(T[(size_t)full_midpoint - 1], RightMat, R, 2.0, 0.0, p.threshold);
    ps_increment(LeftMat, R, 1.0, 0.0);
    ps_increment(T[(size_t)full_midpoint - 1], R, -1.0 * right[0], 0.0);
    Out = std::move(R);
  }
}
}  // namespace

void chebyshev_factorized(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p) {
  use_grid_comm(In.grid);
  const int degree = (int)c.size();
  if (degree < 1) NTP_FATAL("polynomial without coefficients");
  poly_header("Chebyshev Solver", "Recursive", nullptr, degree, p, true);
  Balanced b;
  balanced_setup(In, p, b);
  SlabSession slab(!In.cplx);
  int log2degree = 1;
  while ((1 << log2degree) <= degree) ++log2degree;
  std::vector<PSMatrix> T((size_t)log2degree);
  PSMatrix R;
  ps_copy(b.Identity, T[0]);
  if (degree == 1) {
    ps_copy(T[0], R);
  } else {
    ps_copy(b.Input, T[1]);
    for (int II = 3; II <= log2degree; ++II) {
      ps_multiply(T[(size_t)II - 2], T[(size_t)II - 2], T[(size_t)II - 1], 2.0, 0.0, p.threshold);
      ps_increment(b.Identity, T[(size_t)II - 1], -1.0, 0.0);
    }
    cheby_recursive(T, c, R, 1, p);
  }
  slab.close();
  ps_slab_leave(R);
  if (p.be_verbose) print_matrix_information(R);
  balanced_finish(R, p);
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Hermite
void hermite_compute(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p) {
  use_grid_comm(In.grid);
  const int degree = (int)c.size();
  if (degree < 1) NTP_FATAL("polynomial without coefficients");
  poly_header("Hermite Solver", "Standard", nullptr, degree, p, true);
  Balanced b;
  balanced_setup(In, p, b);
  SlabSession slab(!In.cplx);
  PSMatrix Hk, Hkminus1, Hkplus1, Hkprime, R;
  ps_copy(b.Identity, Hkminus1);
  ps_copy(Hkminus1, R);
  ps_scale(R, c[0]);
  if (degree > 1) {
    ps_copy(b.Input, Hk);
    ps_scale(Hk, 2.0);
    ps_increment(Hk, R, c[1], 0.0);
    if (degree > 2) {
      ps_copy(Hkminus1, Hkprime);
      ps_scale(Hkprime, 2.0);
      for (int II = 3; II <= degree; ++II) {
        ps_multiply(b.Input, Hk, Hkplus1, 2.0, 0.0, p.threshold);
        ps_increment(Hkprime, Hkplus1, -1.0, 0.0);
        ps_copy(Hk, Hkprime);
        ps_scale(Hkprime, (double)(2 * (II - 1)));
        std::swap(Hkminus1.loc, Hk.loc);   // Hkminus1 <- Hk
        std::swap(Hk.loc, Hkplus1.loc);    // Hk <- Hkplus1
        ps_increment(Hk, R, c[(size_t)II - 1], 0.0);
      }
    }
  }
  slab.close();
  ps_slab_leave(R);
  if (p.be_verbose) print_matrix_information(R);
  balanced_finish(R, p);
  Out = std::move(R);
  if (p.be_verbose) log_exit();
}

}  // namespace ntp
