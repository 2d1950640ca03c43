// Solver loops of the hot path, driving the device-resident algebra.  Control flow, scalar
// arithmetic and stopping rules follow the reference line by line so that iteration counts match:
//   DensityMatrixSolversModule.F90 (PM :37-281, TRS2 :285-481, TRS4 :485-716, HPCP :720-929)
//   SignSolversModule.F90:150-258, InverseSolversModule.F90:29-149,187-298,
//   SquareRootSolversModule.F90:164-531, ConvergenceMonitorModule.F90, SolverParametersModule.F90,
//   LoggingModule.F90.
// Matrices never leave HBM inside a loop; only scalars (trace, dot, norm, nnz) cross PCIe.
#include <chrono>
#include <cmath>
#include <cstring>

#include "engine.hpp"

namespace ntp {

// ------------------------------------------------------------------ logger
Logger& logger() {
  static Logger* l = new Logger();
  return *l;
}
void log_activate(bool start_document, const char* file_name) {
  Logger& l = logger();
  if (l.owns_file && l.out) std::fclose(l.out);
  l.owns_file = false;
  l.out = stdout;
  if (file_name && file_name[0]) {
    FILE* f = std::fopen(file_name, "w");
    if (!f) NTP_FATAL(std::string("cannot open log file ") + file_name);
    l.out = f;
    l.owns_file = true;
  }
  l.active = true;
  if (start_document) std::fprintf(l.out, "---\n");
}
void log_deactivate() {
  Logger& l = logger();
  if (l.owns_file && l.out) std::fclose(l.out);
  l.owns_file = false;
  l.out = stdout;
  l.active = false;
  l.level = 0;
}
void log_enter() { logger().level += 1; }
void log_exit() { logger().level -= 1; }
static void indent() {
  Logger& l = logger();
  for (int i = 0; i < l.level * 2; ++i) std::fputc(' ', l.out);
}
void log_header(const char* h) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "%s:\n", h);
}
void log_element(const char* key, double v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "%s: %22.14E\n", key, v);
}
void log_element(const char* key, int v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "%s: %20d\n", key, v);
}
void log_element(const char* key, const char* v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "%s: %s\n", key, v);
}
void log_element(const char* key, bool v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "%s: %s\n", key, v ? "True" : "False");
}
void log_list_element(const char* key, double v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "- %s: %22.14E\n", key, v);
}
void log_list_element(const char* key) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "- %s\n", key);
}
static void log_list_int(const char* key, long long v) {
  if (!logger().active) return;
  indent();
  std::fprintf(logger().out, "- %s: %10lld\n", key, v);
}

// ------------------------------------------------------------------ monitor
void monitor_construct(Monitor& m, bool automatic, double tight_cutoff) {
  m = Monitor();
  m.automatic = automatic;
  m.tight_cutoff = tight_cutoff;
}
void monitor_append(Monitor& m, double v) {  // ConvergenceMonitorModule.F90:101-119
  for (int i = 0; i < 2; ++i) m.win_short[i] = m.win_short[i + 1];
  for (int i = 0; i < 5; ++i) m.win_long[i] = m.win_long[i + 1];
  m.win_short[2] = v;
  m.win_long[5] = v;
  m.nval += 1;
}
bool monitor_converged(const Monitor& m, bool be_verbose) {  // :122-191
  const double last = m.win_short[2], last2 = m.win_short[1];
  if (be_verbose) log_list_element("Convergence", last);
  bool conv;
  if (std::fabs(last) > m.tight_cutoff) {
    conv = false;
  } else {
    conv = true;
    log_enter();
    log_element("Trigger", "Tight Criteria");
    log_exit();
  }
  if (!m.automatic || conv) return conv;
  conv = true;
  if (m.nval < 6) conv = false;
  double s = 0;
  for (int i = 0; i < 3; ++i) s = s + m.win_short[i];
  const double avg_short = s / 3;
  s = 0;
  for (int i = 0; i < 6; ++i) s = s + m.win_long[i];
  const double avg_long = s / 6;
  if (be_verbose) {
    log_enter();
    log_element("Avg Short", avg_short);
    log_element("Avg Long", avg_long);
    log_exit();
  }
  if (!(10 * avg_short > avg_long && avg_short / 10 < avg_long)) conv = false;
  if (!(10 * last > avg_long && last / 10 < avg_long)) conv = false;
  if (last < 0) conv = false;
  if (std::fabs(last) < std::fabs(last2)) conv = false;
  if (avg_long > m.loose_cutoff) conv = false;
  if (conv) {
    log_enter();
    log_element("Trigger", "Automatic");
    log_exit();
  }
  return conv;
}

void print_parameters(const SolverParameters& p) {  // SolverParametersModule.F90:197-218
  log_header("Solver Parameters");
  log_enter();
  log_element("Verbosity", p.be_verbose);
  log_element("Load Balancing", p.do_load_balancing);
  log_element("Convergence Difference", p.converge_diff);
  log_element("Threshold", p.threshold);
  log_element("Maximum Iterations", p.max_iterations);
  log_element("Step Threshold", p.step_thresh);
  log_element("Monitor Convergence", p.monitor_convergence);
  // (not in the reference's log: which of its two builds the products of this run reproduce -- DESIGN.md section 4.
  // "fma" = the FP-contracted build, every product entry one chain of fma(); "unfused" = the default x86-64 build,
  // separate multiply and add; complex run-like operands in fma mode: two FMA chains per part, a tolerance mode)
  log_element("Arithmetic", options().spgemm_fma == 0 ? "unfused" : "fma");
  log_exit();
}

void print_matrix_information(const PSMatrix& m) {
  use_grid_comm(m.grid);  // PSMatrixModule.F90:1248-1266
  double mn = (double)m.loc.nnz, mx = (double)m.loc.nnz;
  comm_allreduce_min(&mn, 1);
  comm_allreduce_max(&mx, 1);
  const double sparsity = (double)ps_size(m) / ((double)m.dim * (double)m.dim);
  log_header("Load_Balance");
  log_enter();
  log_list_int("min_size", (long long)mn);
  log_list_int("max_size", (long long)mx);
  log_exit();
  log_element("Dimension", (int)m.dim);
  log_element("Sparsity", sparsity);
}

SolverTrace& last_trace() {
  static SolverTrace* t = new SolverTrace();
  return *t;
}

namespace {
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) {
  sync_stream();
  return std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
}
void trace_reset() {   // (every solver's first statement: whatever a panel step of an earlier solve prepared for a successor goes here)
  last_trace() = SolverTrace();
  drop_pending_exchange();
}
void trace_rec(double value, double energy, double sigma, const PSMatrix& X) {
  SolverTrace& t = last_trace();
  t.value.push_back(value);
  t.energy.push_back(energy);
  t.sigma.push_back(sigma);
  t.nnz.push_back(X.loc.nnz);
  t.iterations += 1;
}
double real_dot(const PSMatrix& A, const PSMatrix& B) {
  double out[2];
  ps_dot(A, B, out);
  return out[0];
}
void density_header(const char* method, const char* citation, const SolverParameters& p) {
  if (!p.be_verbose) return;
  log_header("Density Matrix Solver");
  log_enter();
  log_element("Method", method);
  log_header("Citations");
  log_enter();
  log_list_element(citation);
  log_exit();
  print_parameters(p);
}
// common prologue of the density solvers: WH = ISQ H ISQ^T, optional permutation, Gershgorin
void density_setup(const PSMatrix& H, const PSMatrix& ISQ, const SolverParameters& p, PSMatrix& IMat, PSMatrix& ISQT,
                   PSMatrix& WH, double* e_min, double* e_max) {
  ps_construct_like(IMat, H);
  ps_fill_identity(IMat);
  ps_transpose(ISQ, ISQT);
  ps_similarity(H, ISQ, ISQT, WH, p.threshold);
  if (p.do_load_balancing) {
    ps_permute(WH, WH, p.balance_permutation, false);
    ps_permute(IMat, IMat, p.balance_permutation, false);
  }
  ps_gershgorin(WH, e_min, e_max);
}
void density_finish(PSMatrix& X, const PSMatrix& ISQT, const PSMatrix& ISQ, PSMatrix& K, const SolverParameters& p) {
  if (p.do_load_balancing) ps_permute(X, X, p.balance_permutation, true);
  ps_similarity(X, ISQT, ISQ, K, p.threshold);
  // what the fused / relabelled steps keep for the NEXT solve on the same operand (the expanded WH, WH in the recovered
  // band order: about 8 (nnz + 32 columns) + 12 nnz + 8 n bytes, 1.3 GB at N = 262 144) goes now when the caller said so
  if (!options().operand_cache) drop_operand_caches();
}
}  // namespace

// ------------------------------------------------------------------ TRS2
// One iteration of the TRS2 loop (DensityMatrixSolversModule.F90:380-404).  The update
// "ScaleMatrix(X,2); IncrementMatrix(X2,X,-1,threshold); DotMatrix(X,WH)" runs as ONE pass over X, X2
// and WH (same arithmetic: 2*x is exact, then the AddSparseVectors rules, then the energy).
double trs2_step(PSMatrix& X, PSMatrix& X2, const PSMatrix& WH, double trace_target, double threshold, double* sigma,
                 double* trace_io) {
  use_grid_comm(WH.grid);
  // trace_io (optional): in = trace(X) if the caller already has it (NaN: compute it), out = trace of the new X,
  // accumulated in the pass that produces the energy -> one reduction + read-back less per iteration
  const double trace_value = (trace_io && *trace_io == *trace_io) ? *trace_io : ps_trace(X);
  *sigma = (trace_target - trace_value < 0.0) ? -1.0 : 1.0;
  double out[4] = {0, 0, 0, 0};
  if (*sigma > 0.0) {
    ps_square_update_dot(X, X2, threshold, WH, out, trace_io != nullptr);  // X = 2X - X*X; energy; trace (X2: scratch -- holds X*X only on the unfused path)
  } else {
    ps_square_dot(X, X2, threshold, WH, out, trace_io != nullptr);         // X = X*X; energy; trace
  }
  if (trace_io) *trace_io = out[2];
  return out[0];
}

void solver_trs2(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy_out, double* mu_out,
                 const SolverParameters& p) {
  use_grid_comm(H.grid);
  // (several ranks, an operand without runs: the whole solve in a recovered band order -- band_scope.cpp)
  if (band_scope_try({&H, &ISQ}, {&K}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) {
        solver_trs2(*in[0], *in[1], trace, *out[0], energy_out, mu_out, p);
      }))
    return;
  trace_reset();
  auto t0 = Clock::now();
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  density_header("TRS2", "niklasson2002expansion", p);
  std::vector<double> sigma_array((size_t)p.max_iterations + 1, 0.0);
  PSMatrix WH, IMat, ISQT, X, X2;
  double e_min, e_max;
  density_setup(H, ISQ, p, IMat, ISQT, WH, &e_min, &e_max);       // :344-365
  ps_copy(WH, X);                                                  // :368-371
  ps_scale(X, -1.0);
  ps_increment(IMat, X, e_max, 0.0);
  ps_scale(X, 1.0 / (e_max - e_min));
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  last_trace().setup_ms = ms_since(t0);
  auto t1 = Clock::now();
  double energy_value = 0.0, energy_old;
  double trace_x = std::nan("");  // trace of the current iterate, handed from step to step
  int II;
  for (II = 1; II <= p.max_iterations; ++II) {                     // :380-413
    energy_old = energy_value;
    static const bool step_times = std::getenv("NTPOLY_AMD_DEBUG_STEPTIME") != nullptr;   // (host clock per iteration, rank 0)
    const auto ts0 = Clock::now();
    energy_value = trs2_step(X, X2, WH, trace, p.threshold, &sigma_array[(size_t)II], &trace_x);
    const double ts_step = step_times ? ms_since(ts0) : 0.0;   // (ms_since waits for the stream: only when asked for)
    monitor_append(mon, energy_value - energy_old);
    trace_rec(energy_value - energy_old, energy_value, sigma_array[(size_t)II], X);
    if (step_times && world().rank == 0)
      std::fprintf(stderr, "[trs2] iteration %d: step %.3f ms, with the trace record %.3f ms\n", II, ts_step, ms_since(ts0));
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Energy Value", energy_value);
      log_exit();
    }
  }
  const int total_iterations = II - 1;
  pack(X.loc);  // (the steps leave the iterate loose, kernels.hpp)
  last_trace().loop_ms = ms_since(t1);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(X);
  }
  if (energy_out) *energy_out = energy_value;
  density_finish(X, ISQT, ISQ, K, p);                              // :427-434
  if (mu_out) {                                                    // :444-472
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p.max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double zero_value = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ) {
        if (sigma_array[(size_t)JJ] < 0.0) zero_value = zero_value * zero_value;
        else zero_value = 2.0 * zero_value - zero_value * zero_value;
      }
      if (zero_value < 0.5) interval_a = midpoint;
      else interval_b = midpoint;
      if (std::fabs(zero_value - 0.5) < p.converge_diff) break;
    }
    *mu_out = e_max + (e_min - e_max) * midpoint;
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ TRS4
// ------------------------------------------------------------------ ScaleAndFold
// DensityMatrixSolversModule.F90:953-1117 (rubensson2011nonmonotonic): like TRS2 with the polynomials scaled by the
// running estimates Beta / BetaBar of where lumo and homo have moved to
void solver_scale_and_fold(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double homo, double lumo,
                           double* energy_out, const SolverParameters& p) {
  use_grid_comm(H.grid);
  trace_reset();
  auto t0 = Clock::now();
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  density_header("Scale and Fold", "rubensson2011nonmonotonic", p);
  PSMatrix WH, IMat, ISQT, X, X2;
  double e_min, e_max;
  density_setup(H, ISQ, p, IMat, ISQT, WH, &e_min, &e_max);       // :1022-1034
  ps_copy(WH, X);                                                  // :1036-1039
  ps_scale(X, -1.0);
  ps_increment(IMat, X, e_max, 0.0);
  ps_scale(X, 1.0 / (e_max - e_min));
  double Beta = (e_max - lumo) / (e_max - e_min);
  double BetaBar = (e_max - homo) / (e_max - e_min);
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  last_trace().setup_ms = ms_since(t0);
  auto t1 = Clock::now();
  double energy_value = 0.0, energy_old;
  int II;
  SlabSession slab(!X.cplx && !WH.cplx);
  for (II = 1; II <= p.max_iterations; ++II) {                     // :1049-1081
    const double trace_value = ps_trace(X);
    double alpha;
    if (trace_value > trace) {
      alpha = 2.0 / (2.0 - Beta);
      ps_axpby(IMat, X, 1.0 - alpha, alpha, 0.0);                   // ScaleMatrix(X, alpha); IncrementMatrix(I, X, 1 - alpha)
      ps_multiply(X, X, X2, 1.0, 0.0, p.threshold);
      std::swap(X.loc, X2.loc);  // CopyMatrix(X_k2, X_k); X2 is scratch
      Beta = (alpha * Beta + 1 - alpha) * (alpha * Beta + 1 - alpha);
      BetaBar = (alpha * BetaBar + 1 - alpha) * (alpha * BetaBar + 1 - alpha);
    } else {
      alpha = 2.0 / (1.0 + BetaBar);
      ps_multiply(X, X, X2, 1.0, 0.0, p.threshold);
      ps_axpby(X2, X, -1.0 * alpha * alpha, 2 * alpha, 0.0);        // ScaleMatrix(X, 2 alpha); IncrementMatrix(X2, X, -alpha^2)
      Beta = 2.0 * alpha * Beta - alpha * alpha * Beta * Beta;
      BetaBar = 2.0 * alpha * BetaBar - alpha * alpha * BetaBar * BetaBar;
    }
    energy_old = energy_value;
    energy_value = 2.0 * real_dot(X, WH);
    monitor_append(mon, energy_value - energy_old);
    trace_rec(energy_value - energy_old, energy_value, trace_value > trace ? -1.0 : 1.0, X);
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Energy Value", energy_value);
      log_exit();
    }
  }
  slab.close();
  ps_slab_leave(X);
  last_trace().loop_ms = ms_since(t1);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(X);
  }
  if (energy_out) *energy_out = energy_value;
  density_finish(X, ISQT, ISQ, K, p);                              // :1095-1101
  if (p.be_verbose) log_exit();
}

// EnergyDensityMatrix (:1165-1187): ED = D H D
void energy_density_matrix(const PSMatrix& H, const PSMatrix& D, PSMatrix& ED, double threshold) {
  use_grid_comm(H.grid);
  ps_similarity(H, D, D, ED, threshold);
}

// McWeenyStep (:1190-1231): DOut = 3 DSD - 2 DSDSD
void mcweeny_step(const PSMatrix& D, PSMatrix& DOut, const PSMatrix* S, double threshold) {
  use_grid_comm(D.grid);
  PSMatrix DS, DSD;
  if (S) ps_multiply(D, *S, DS, 1.0, 0.0, threshold);
  else ps_copy(D, DS);
  ps_multiply(DS, D, DSD, 1.0, 0.0, threshold);
  PSMatrix Out;
  ps_multiply(DS, DSD, Out, -2.0, 0.0, threshold);
  ps_increment(DSD, Out, 3.0, 0.0);
  DOut.grid = Out.grid; DOut.dim = Out.dim; DOut.cplx = Out.cplx; DOut.c0 = Out.c0; DOut.c1 = Out.c1;
  DOut.loc = std::move(Out.loc);
}

void solver_trs4(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy_out, double* mu_out,
                 const SolverParameters& p) {
  use_grid_comm(H.grid);
  // (several ranks, an operand without runs: the whole solve in a recovered band order -- band_scope.cpp)
  if (band_scope_try({&H, &ISQ}, {&K}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) {
        solver_trs4(*in[0], *in[1], trace, *out[0], energy_out, mu_out, p);
      }))
    return;
  trace_reset();
  auto t0 = Clock::now();
  const double sigma_min = 0.0, sigma_max = 6.0;
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  density_header("TRS4", "niklasson2002expansion", p);
  std::vector<double> sigma_array((size_t)p.max_iterations + 1, 0.0);
  PSMatrix WH, IMat, ISQT, X, X2, Fx, Gx, Temp;
  double e_min, e_max;
  density_setup(H, ISQ, p, IMat, ISQT, WH, &e_min, &e_max);
  ps_copy(WH, X);
  ps_scale(X, -1.0);
  ps_increment(IMat, X, e_max, 0.0);
  ps_scale(X, 1.0 / (e_max - e_min));
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  last_trace().setup_ms = ms_since(t0);
  auto t1 = Clock::now();
  double energy_value = 0.0, energy_old;
  int II;
  SlabSession slab(!X.cplx && !WH.cplx);   // (the loop's matrices stay in slab form between its operations where they can)
  for (II = 1; II <= p.max_iterations; ++II) {                     // :586-638
    ps_multiply(X, X, X2, 1.0, 0.0, p.threshold);
    // Fx = 4 X - 3 X2, Gx = I - 2 X + X2 and their traces against X2: in a slab session one pass over X and X2 leaves the
    // traces without building Fx and Gx (the same element arithmetic; IMat is the identity also under load balancing)
    double trace_fx = 0.0, trace_gx = 0.0;
    const bool chain_fused = ps_trs4_traces(X, X2, &trace_fx, &trace_gx);
    if (!chain_fused) {
      ps_copy_axpby(X2, X, Fx, 4.0, -3.0, 0.0);                     // CopyMatrix(X2, Fx); ScaleMatrix(Fx, -3); IncrementMatrix(X, Fx, 4)
      ps_copy_axpby(IMat, X, Gx, -2.0, 1.0, 0.0);                   // CopyMatrix(Identity, Gx); IncrementMatrix(X, Gx, -2)
      ps_increment(X2, Gx, 1.0, 0.0);
      trace_fx = real_dot(X2, Fx);
      trace_gx = real_dot(X2, Gx);
    }
    if (std::fabs(trace_gx) < 1.0e-14) sigma_array[(size_t)II] = 0.5 * (sigma_max - sigma_min);
    else sigma_array[(size_t)II] = (trace - trace_fx) / trace_gx;
    if (sigma_array[(size_t)II] > sigma_max) {
      ps_copy_axpby(X, X2, Temp, -1.0, 2.0, 0.0);                   // CopyMatrix(X, Temp); ScaleMatrix(Temp, 2); IncrementMatrix(X2, Temp, -1)
    } else if (sigma_array[(size_t)II] < sigma_min) {
      ps_copy(X2, Temp);
    } else {
      if (!(chain_fused && ps_trs4_operand(X, X2, sigma_array[(size_t)II], Gx))) {
        if (chain_fused) {   // (the traces came from the fused pass but the operand cannot: build the intermediates now)
          ps_copy_axpby(X2, X, Fx, 4.0, -3.0, 0.0);
          ps_copy_axpby(IMat, X, Gx, -2.0, 1.0, 0.0);
          ps_increment(X2, Gx, 1.0, 0.0);
        }
        ps_axpby(Fx, Gx, 1.0, sigma_array[(size_t)II], 0.0);        // ScaleMatrix(Gx, sigma); IncrementMatrix(Fx, Gx)
      }
      ps_multiply(X2, Gx, Temp, 1.0, 0.0, p.threshold);
    }
    // :630-631 IncrementMatrix(TempMat, X_k, -1) is overwritten by the copy that follows it; the copy itself is a
    // hand-over here (Temp is rebuilt in every iteration)
    std::swap(X.loc, Temp.loc);
    energy_old = energy_value;
    energy_value = real_dot(X, WH);
    monitor_append(mon, energy_value - energy_old);
    trace_rec(energy_value - energy_old, energy_value, sigma_array[(size_t)II], X);
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Energy Value", energy_value);
      log_exit();
    }
  }
  const int total_iterations = II - 1;
  slab.close();
  ps_slab_leave(X);
  last_trace().loop_ms = ms_since(t1);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(X);
  }
  if (energy_out) *energy_out = energy_value;
  density_finish(X, ISQT, ISQ, K, p);
  if (mu_out) {                                                    // :669-704
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p.max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double z = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ) {
        const double sg = sigma_array[(size_t)JJ];
        if (sg > sigma_max) z = 2.0 * z - z * z;
        else if (sg < sigma_min) z = z * z;
        else {
          const double tempfx = (z * z) * (4.0 * z - 3.0 * z * z);
          const double tempgx = (z * z) * (1.0 - z) * (1.0 - z);
          z = tempfx + sg * tempgx;
        }
      }
      if (z < 0.5) interval_a = midpoint;
      else interval_b = midpoint;
      if (std::fabs(z - 0.5) < p.converge_diff) break;
    }
    *mu_out = e_max + (e_min - e_max) * midpoint;
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ PM
void solver_pm(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy_out, double* mu_out,
               const SolverParameters& p) {
  use_grid_comm(H.grid);
  // (several ranks, an operand without runs: the whole solve in a recovered band order -- band_scope.cpp)
  if (band_scope_try({&H, &ISQ}, {&K}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) {
        solver_pm(*in[0], *in[1], trace, *out[0], energy_out, mu_out, p);
      }))
    return;
  trace_reset();
  auto t0 = Clock::now();
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  density_header("PM", "palser1998canonical", p);
  std::vector<double> sigma_array((size_t)p.max_iterations + 1, 0.0);
  PSMatrix WH, IMat, ISQT, X, X2, X3, Temp;
  double e_min, e_max;
  density_setup(H, ISQ, p, IMat, ISQT, WH, &e_min, &e_max);       // :96-118
  ps_copy(WH, X);
  const double dim = (double)H.dim;
  double trace_value = ps_trace(X);                                // :124-125
  const double lambda = trace_value / dim;
  const double alpha1 = trace / (e_max - lambda);                  // :128-130
  const double alpha2 = (dim - trace) / (lambda - e_min);
  const double alpha = std::fmin(alpha1, alpha2);
  double factor = -alpha / dim;
  ps_scale(X, factor);
  factor = (alpha * lambda + trace) / dim;
  ps_increment(IMat, X, factor, 0.0);
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  last_trace().setup_ms = ms_since(t0);
  auto t1 = Clock::now();
  double energy_value = 0.0, energy_old;
  int II;
  // (no slab session here: whenever sigma > 1/2 the update scales X by a1 = 0 -- stored zeros whose tails steer the merges
  // that follow, which the slab form cannot hold -- so half the iterations would fall back and convert to and fro)
  for (II = 1; II <= p.max_iterations; ++II) {                     // :145-200
    ps_multiply(X, X, X2, 1.0, 0.0, p.threshold);
    ps_multiply(X, X2, X3, 1.0, 0.0, p.threshold);
    ps_copy(X, Temp);
    ps_increment(X2, Temp, -1.0, p.threshold);
    trace_value = ps_trace(Temp);
    const double trace_value2 = real_dot(Temp, X);
    if (trace_value <= 2.2250738585072014e-308) sigma_array[(size_t)II] = 1.0;
    else sigma_array[(size_t)II] = trace_value2 / trace_value;
    const double sg = sigma_array[(size_t)II];
    double a1, a2, a3;
    if (sg > 0.5) {
      a1 = 0.0;
      a2 = 1.0 + 1.0 / sg;
      a3 = -1.0 / sg;
    } else {
      a1 = (1.0 - 2.0 * sg) / (1.0 - sg);
      a2 = (1.0 + sg) / (1.0 - sg);
      a3 = -1.0 / (1.0 - sg);
    }
    ps_axpby(X2, X, a2, a1, p.threshold);                           // ScaleMatrix(X, a1); IncrementMatrix(X2, X, a2)
    ps_increment(X3, X, a3, p.threshold);
    energy_old = energy_value;
    energy_value = real_dot(X, WH);
    monitor_append(mon, energy_value - energy_old);
    trace_rec(energy_value - energy_old, energy_value, sg, X);
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Energy Value", energy_value);
      log_exit();
    }
  }
  const int total_iterations = II - 1;
  last_trace().loop_ms = ms_since(t1);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(X);
  }
  if (energy_out) *energy_out = energy_value;
  density_finish(X, ISQT, ISQ, K, p);
  if (mu_out) {                                                    // :234-268
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p.max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double z = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ) {
        const double sg = sigma_array[(size_t)JJ];
        if (sg > 0.5) {
          z = ((1.0 + sg) * (z * z)) - (z * z * z);
          z = z / sg;
        } else {
          z = ((1.0 - 2.0 * sg) * z) + ((1.0 + sg) * (z * z)) - (z * z * z);
          z = z / (1.0 - sg);
        }
      }
      if (z < 0.5) interval_a = midpoint;
      else interval_b = midpoint;
      if (std::fabs(z - 0.5) < p.converge_diff) break;
    }
    *mu_out = lambda - (dim * midpoint - trace) / alpha;
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ HPCP
void solver_hpcp(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy_out, double* mu_out,
                 const SolverParameters& p) {
  use_grid_comm(H.grid);
  // (several ranks, an operand without runs: the whole solve in a recovered band order -- band_scope.cpp)
  if (band_scope_try({&H, &ISQ}, {&K}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) {
        solver_hpcp(*in[0], *in[1], trace, *out[0], energy_out, mu_out, p);
      }))
    return;
  trace_reset();
  auto t0 = Clock::now();
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  density_header("HPCP", "truflandier2016communication", p);
  std::vector<double> sigma_array((size_t)p.max_iterations + 1, 0.0);
  PSMatrix WH, IMat, ISQT, TempMat, D1, DH, DDH, D2DH;
  double e_min, e_max;
  density_setup(H, ISQ, p, IMat, ISQT, WH, &e_min, &e_max);       // :789-808
  const double dim = (double)H.dim;
  double mu = ps_trace(WH) / dim;                                  // :809-817
  const double sigma_bar = (dim - trace) / dim;
  const double sigma = 1.0 - sigma_bar;
  const double beta = sigma / (e_max - mu);
  const double beta_bar = sigma_bar / (mu - e_min);
  const double beta_1 = sigma;
  const double beta_2 = std::fmin(beta, beta_bar);
  ps_copy(IMat, D1);                                               // :820-826
  ps_scale(D1, beta_1);
  ps_copy(IMat, TempMat);
  ps_scale(TempMat, mu);
  ps_increment(WH, TempMat, -1.0, 0.0);
  ps_scale(TempMat, beta_2);
  ps_increment(TempMat, D1, 1.0, 0.0);
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  last_trace().setup_ms = ms_since(t0);
  auto t1 = Clock::now();
  double energy_value = 0.0, energy_old, trace_value = 0.0;
  int II;
  SlabSession slab(!D1.cplx && !WH.cplx);
  for (II = 1; II <= p.max_iterations; ++II) {                     // :836-872
    ps_copy(D1, DH);
    ps_increment_identity(IMat, DH, -1.0);                         // IncrementMatrix(Identity, DH, -1)
    ps_scale(DH, -1.0);
    ps_multiply(D1, DH, DDH, 1.0, 0.0, p.threshold);
    trace_value = ps_trace(DDH);
    ps_multiply(D1, DDH, D2DH, 1.0, 0.0, p.threshold);
    sigma_array[(size_t)II] = ps_trace(D2DH) / trace_value;
    ps_increment(D2DH, D1, 2.0, 0.0);
    ps_increment(DDH, D1, -1.0 * 2.0 * sigma_array[(size_t)II], 0.0);
    energy_old = energy_value;
    energy_value = real_dot(D1, WH);
    monitor_append(mon, energy_value - energy_old);
    trace_rec(energy_value - energy_old, energy_value, sigma_array[(size_t)II], D1);
    if (monitor_converged(mon, p.be_verbose)) break;
    if (p.be_verbose) {
      log_enter();
      log_element("Energy Value", energy_value);
      log_exit();
    }
  }
  const int total_iterations = II - 1;
  slab.close();
  ps_slab_leave(D1);
  last_trace().loop_ms = ms_since(t1);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(D1);
  }
  if (energy_out) *energy_out = energy_value;
  density_finish(D1, ISQT, ISQ, K, p);
  if (mu_out) {                                                    // :896-921
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p.max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double z = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ)
        z = z + 2.0 * (((z * z)) * (1.0 - z) - sigma_array[(size_t)JJ] * z * (1.0 - z));
      if (z < 0.5) interval_a = midpoint;
      else interval_b = midpoint;
      if (std::fabs(z - 0.5) < p.converge_diff) break;
    }
    *mu_out = mu + (beta_1 - midpoint) / beta_2;
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Sign / Polar
namespace {
// CoreComputation (SignSolversModule.F90:150-258)
void sign_core(const PSMatrix& InMat, PSMatrix& OutMat, const SolverParameters& p, bool needs_transpose) {
  const double alpha = 1.69770248526;
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  PSMatrix Identity, Temp1, Temp2, OutMatT, Out;
  ps_construct_like(Identity, InMat);
  ps_fill_identity(Identity);
  if (p.do_load_balancing) {
    ps_permute(Identity, Identity, p.balance_permutation, false);
    ps_permute(InMat, Out, p.balance_permutation, false);
  } else {
    ps_copy(InMat, Out);
  }
  double e_min, e_max;
  ps_gershgorin(InMat, &e_min, &e_max);
  double xk = std::fabs(e_min / e_max);
  ps_scale(Out, 1.0 / std::fabs(e_max));
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  // (complex operands under FMA arithmetic: the loop's products, identity increment and norm take them in slab form as well)
  const bool complex_session = Out.cplx && options().complex_sessions != 0 && options().spgemm_fma == 1 && options().complex_tile != 0;
  SlabSession slab(!needs_transpose && (!Out.cplx || complex_session), false, complex_session);
  for (II = 1; II <= p.max_iterations; ++II) {
    const double alpha_k = std::fmin(std::sqrt(3.0 / (1.0 + xk + xk * xk)), alpha);
    xk = 0.5 * alpha_k * xk * (3.0 - (alpha_k * alpha_k) * (xk * xk));
    if (needs_transpose) {
      ps_transpose(Out, OutMatT);
      if (OutMatT.cplx) ps_conjugate(OutMatT);
      ps_multiply(OutMatT, Out, Temp1, -1.0 * (alpha_k * alpha_k), 0.0, p.threshold);
    } else {
      ps_multiply(Out, Out, Temp1, -1.0 * (alpha_k * alpha_k), 0.0, p.threshold);
    }
    ps_increment_identity(Identity, Temp1, 3.0);                   // IncrementMatrix(Identity, Temp1, 3)
    ps_multiply(Out, Temp1, Temp2, 0.5 * alpha_k, 0.0, p.threshold);
    double norm_value = 0.0;   // (the difference Out - Temp2 is only there for its norm)
    if (!ps_norm_axpby(Temp2, Out, -1.0, 1.0, &norm_value)) {
      ps_increment(Temp2, Out, -1.0, 0.0);
      norm_value = ps_norm(Out);
    }
    std::swap(Out.loc, Temp2.loc);   // CopyMatrix(Temp2, Out): Temp2 is rebuilt by the next multiply
    monitor_append(mon, norm_value);
    trace_rec(norm_value, 0.0, alpha_k, Out);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  slab.close();
  ps_slab_leave(Out);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II - 1);
    print_matrix_information(Out);
  }
  if (p.do_load_balancing) ps_permute(Out, Out, p.balance_permutation, true);
  OutMat = std::move(Out);
}
}  // namespace

void solver_sign(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (band_scope_try({&A}, {&Out}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) { solver_sign(*in[0], *out[0], p); }))
    return;
  trace_reset();
  if (p.be_verbose) {
    log_header("Sign Function Solver");
    log_enter();
    log_header("Citations");
    log_enter();
    log_list_element("nicholas2008functions");
    log_exit();
    print_parameters(p);
  }
  sign_core(A, Out, p, false);
  if (p.be_verbose) log_exit();
}

void solver_polar(const PSMatrix& A, PSMatrix& U, PSMatrix* Hm, const SolverParameters& p) {
  use_grid_comm(A.grid);
  trace_reset();
  if (p.be_verbose) {
    log_header("Polar Decomposition Solver");
    log_enter();
    log_header("Citations");
    log_enter();
    log_list_element("nicholas2008functions");
    log_exit();
    print_parameters(p);
  }
  sign_core(A, U, p, true);
  if (Hm) {  // SignSolversModule.F90:129-137
    PSMatrix UT;
    ps_transpose(U, UT);
    if (UT.cplx) ps_conjugate(UT);
    ps_multiply(UT, A, *Hm, 1.0, 0.0, p.threshold);
  }
  if (p.be_verbose) log_exit();
}

// ------------------------------------------------------------------ Invert / PseudoInverse
namespace {
void invert_core(const PSMatrix& InputMat, PSMatrix& OutputMat, const SolverParameters& p, bool log_top) {
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  if (p.be_verbose) {
    log_header("Inverse Solver");
    log_enter();
    log_header("Citations");
    log_enter();
    log_list_element("palser1998canonical");
    log_exit();
    print_parameters(p);
  }
  PSMatrix Temp1, Temp2, Identity, Balanced, Out;
  ps_construct_like(Identity, InputMat);
  ps_fill_identity(Identity);
  if (p.do_load_balancing) {
    ps_permute(Identity, Identity, p.balance_permutation, false);
    ps_permute(InputMat, Balanced, p.balance_permutation, false);
  } else {
    ps_copy(InputMat, Balanced);
  }
  const double sigma = ps_sigma(Balanced);
  ps_copy(Balanced, Out);
  ps_scale(Out, sigma);
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  double norm_value = p.converge_diff + 1.0;
  int II;
  // (complex operands under FMA arithmetic: products, merges, scalings, copies and norms take them in slab form as well)
  const bool complex_session = Out.cplx && options().complex_sessions != 0 && options().spgemm_fma == 1 && options().complex_tile != 0;
  SlabSession slab(!Out.cplx || complex_session, false, complex_session);
  for (II = 1; II <= p.max_iterations; ++II) {
    if (log_top && p.be_verbose && II > 1) log_list_element("Convergence", norm_value);
    ps_multiply(Out, Balanced, Temp1, 1.0, 0.0, p.threshold);
    if (!ps_norm_axpby(Temp1, Identity, -1.0, 1.0, &norm_value)) {   // (I - Temp1 is only there for its norm)
      ps_copy_axpby(Identity, Temp1, Temp2, -1.0, 1.0, 0.0);       // CopyMatrix(Identity, Temp2); IncrementMatrix(Temp1, Temp2, -1)
      norm_value = ps_norm(Temp2);
    }
    PSMatrix T2;
    ps_multiply(Temp1, Out, T2, -1.0, 0.0, p.threshold);
    ps_axpby(T2, Out, 1.0, 2.0, p.threshold);                       // ScaleMatrix(Out, 2); IncrementMatrix(T2, Out)
    monitor_append(mon, norm_value);
    trace_rec(norm_value, 0.0, sigma, Out);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  slab.close();
  ps_slab_leave(Out);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II - 1);
    print_matrix_information(Out);
  }
  if (p.do_load_balancing) ps_permute(Out, Out, p.balance_permutation, true);
  if (p.be_verbose) log_exit();
  OutputMat = std::move(Out);
}
}  // namespace

void solver_invert(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (band_scope_try({&A}, {&Out}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) { solver_invert(*in[0], *out[0], p); }))
    return;
  trace_reset();
  invert_core(A, Out, p, true);
}
void solver_pseudoinverse(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p) {
  use_grid_comm(A.grid);
  if (band_scope_try({&A}, {&Out}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) { solver_pseudoinverse(*in[0], *out[0], p); }))
    return;
  trace_reset();
  invert_core(A, Out, p, false);
}

// ------------------------------------------------------------------ (inverse) square root
namespace {
void isr_header(const SolverParameters& p) {
  if (!p.be_verbose) return;
  log_header("Newton Schultz Inverse Square Root");
  log_enter();
  log_header("Citations");
  log_enter();
  log_list_element("jansik2007linear");
  log_exit();
  print_parameters(p);
}

// NewtonSchultzISROrder2 (SquareRootSolversModule.F90:198-338)
void isr_order2(const PSMatrix& InMat, PSMatrix& OutMat, const SolverParameters& p, bool compute_inverse) {
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  isr_header(p);
  PSMatrix X, T, Temp, Identity, SR, ISR;
  ps_construct_like(Identity, InMat);
  ps_fill_identity(Identity);
  double e_min, e_max;
  ps_gershgorin(InMat, &e_min, &e_max);
  double max_between = std::fmax(std::fabs(e_min), std::fabs(e_max));
  double lambda = 1.0 / max_between;
  ps_construct_like(ISR, InMat);
  ps_fill_identity(ISR);
  ps_copy(InMat, SR);
  if (p.do_load_balancing) {
    ps_permute(SR, SR, p.balance_permutation, false);
    ps_permute(Identity, Identity, p.balance_permutation, false);
    ps_permute(ISR, ISR, p.balance_permutation, false);
  }
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  SlabSession slab(!SR.cplx);
  for (II = 1; II <= p.max_iterations; ++II) {
    ps_multiply(SR, ISR, X, 1.0, 0.0, p.threshold);
    ps_gershgorin(X, &e_min, &e_max);
    max_between = std::fmax(std::fabs(e_min), std::fabs(e_max));
    lambda = 1.0 / max_between;
    ps_scale(X, lambda);
    double norm_value = 0.0;   // (I - X is only there for its norm: Temp is handed over as storage below)
    if (!ps_norm_axpby(X, Identity, -1.0, 1.0, &norm_value)) {
      ps_copy_axpby(Identity, X, Temp, -1.0, 1.0, 0.0);             // CopyMatrix(Identity, Temp); IncrementMatrix(X, Temp, -1)
      norm_value = ps_norm(Temp);
    }
    ps_copy_axpby(Identity, X, T, -1.0, 3.0, 0.0);                  // CopyMatrix(Identity, T); ScaleMatrix(T, 3); IncrementMatrix(X, T, -1)
    ps_scale(T, 0.5);
    std::swap(ISR, Temp);                                           // CopyMatrix(ISR, Temp): ISR is rebuilt by the multiply
    ps_multiply(Temp, T, ISR, 1.0, 0.0, p.threshold);
    ps_scale(ISR, std::sqrt(lambda));
    std::swap(SR, Temp);
    ps_multiply(T, Temp, SR, 1.0, 0.0, p.threshold);
    ps_scale(SR, std::sqrt(lambda));
    monitor_append(mon, norm_value);
    trace_rec(norm_value, 0.0, lambda, ISR);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  slab.close();
  ps_slab_leave(ISR);
  ps_slab_leave(SR);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(ISR);
  }
  PSMatrix Out;
  ps_copy(compute_inverse ? ISR : SR, Out);
  if (p.do_load_balancing) ps_permute(Out, Out, p.balance_permutation, true);
  if (p.be_verbose) log_exit();
  OutMat = std::move(Out);
}

// NewtonSchultzISRTaylor (SquareRootSolversModule.F90:342-531)
void isr_taylor(const PSMatrix& InMat, PSMatrix& OutMat, const SolverParameters& p, int order, bool compute_inverse) {
  Monitor mon;
  monitor_construct(mon, p.monitor_convergence, p.converge_diff);
  isr_header(p);
  PSMatrix X, Temp, Temp2, Identity, SR, ISR;
  ps_construct_like(Identity, InMat);
  ps_fill_identity(Identity);
  double e_min, e_max;
  ps_gershgorin(InMat, &e_min, &e_max);                            // :389-391
  const double max_between = std::fmax(std::fabs(e_min), std::fabs(e_max));
  const double lambda = 1.0 / max_between;
  ps_construct_like(ISR, InMat);                                   // :394-396
  ps_fill_identity(ISR);
  ps_copy(InMat, SR);
  ps_scale(SR, lambda);
  if (p.do_load_balancing) {                                       // :399-406
    ps_permute(SR, SR, p.balance_permutation, false);
    ps_permute(Identity, Identity, p.balance_permutation, false);
    ps_permute(ISR, ISR, p.balance_permutation, false);
  }
  if (p.be_verbose) {
    log_header("Iterations");
    log_enter();
  }
  int II;
  // (complex operands under FMA arithmetic: the loop's whole vocabulary takes them in slab form -- SquareRootSolversModule.F90:415-497)
  const bool complex_session = SR.cplx && options().complex_sessions != 0 && options().spgemm_fma == 1 && options().complex_tile != 0;
  SlabSession slab(!SR.cplx || complex_session, false, complex_session);
  for (II = 1; II <= p.max_iterations; ++II) {                     // :415-497
    ps_multiply(ISR, SR, X, 1.0, 0.0, p.threshold);
    ps_increment_identity(Identity, X, -1.0);
    const double norm_value = ps_norm(X);
    if (order == 3) {                                              // :425-433
      ps_multiply(X, X, Temp, 1.0, 0.0, p.threshold);
      ps_axpby(Identity, X, 1.0, -0.5, 0.0);                        // ScaleMatrix(X, -1/2); IncrementMatrix(I, X)
      ps_increment(Temp, X, 0.375, 0.0);
    } else {                                                       // :434-479 (Knuth's 2-multiply quartic)
      const double aa = -40.0 / 35.0, bb = 48.0 / 35.0, cc = -64.0 / 35.0, dd = 128.0 / 35.0;
      const double a = (aa - 1.0) / 2.0;
      const double b = bb * (a + 1.0) - cc - a * ((a + 1.0) * (a + 1.0));
      const double c = bb - b - a * (a + 1.0);
      const double d = dd - b * c;
      ps_multiply(X, X, Temp, 1.0, 0.0, p.threshold);
      ps_increment(X, Temp, a, 0.0);
      ps_copy_axpby(Identity, X, Temp2, 1.0, b, 0.0);               // CopyMatrix(Identity, Temp2); ScaleMatrix(Temp2, b); IncrementMatrix(X, Temp2)
      ps_increment(Temp, Temp2, 1.0, 0.0);
      ps_increment_identity(Identity, Temp, c);
      ps_multiply(Temp2, Temp, X, 1.0, 0.0, p.threshold);
      ps_increment_identity(Identity, X, d);
      ps_scale(X, 35.0 / 128.0);
    }
    std::swap(ISR, Temp);                                          // :483-485 (the copy is a hand-over: ISR is rebuilt)
    ps_multiply(X, Temp, ISR, 1.0, 0.0, p.threshold);
    std::swap(SR, Temp);                                           // :488-490
    ps_multiply(Temp, X, SR, 1.0, 0.0, p.threshold);
    monitor_append(mon, norm_value);
    trace_rec(norm_value, 0.0, lambda, ISR);
    if (monitor_converged(mon, p.be_verbose)) break;
  }
  slab.close();
  ps_slab_leave(ISR);
  ps_slab_leave(SR);
  if (p.be_verbose) {
    log_exit();
    log_element("Total Iterations", II);
    print_matrix_information(ISR);
  }
  PSMatrix Out;
  if (compute_inverse) {                                           // :505-511
    ps_scale(ISR, std::sqrt(lambda));
    ps_copy(ISR, Out);
  } else {
    ps_scale(SR, 1.0 / std::sqrt(lambda));
    ps_copy(SR, Out);
  }
  if (p.do_load_balancing) ps_permute(Out, Out, p.balance_permutation, true);
  if (p.be_verbose) log_exit();
  OutMat = std::move(Out);
}
}  // namespace

void solver_square_root(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p, bool inverse, int order) {
  use_grid_comm(A.grid);
  if (band_scope_try({&A}, {&Out}, [&](const std::vector<const PSMatrix*>& in, const std::vector<PSMatrix*>& out) { solver_square_root(*in[0], *out[0], p, inverse, order); }))
    return;
  trace_reset();
  // SquareRootSelector (SquareRootSolversModule.F90:164-194): default order 5
  if (order == 2) isr_order2(A, Out, p, inverse);
  else isr_taylor(A, Out, p, order == 3 ? 3 : 5, inverse);
}

}  // namespace ntp
