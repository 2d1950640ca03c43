/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of NTPoly's SpGEMM-driven hot path, used as the
 * parity oracle for the HIP engine and as the "port" CPU baseline in bench.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (ntpoly_amd/) never links, imports or calls it.
 *
 * Pinning: every function here is checked against the REAL reference (built by
 * oracle/build_ref.py from /root/reference, driven by oracle/ref_driver.f90)
 * through the golden vectors in tests/golden/ (tests/test_oracle_golden.py),
 * and against the reference's own shipped fixture
 * Examples/PremadeMatrix/{Hamiltonian,Overlap,Density-Reference}.mtx.
 *
 * Layout follows the reference's local matrix type (SMatrixModule.F90:15-30):
 * column-compressed; outer[cols+1] 0-based offsets, inner[nnz] row ids (kept
 * 0-based here, the reference keeps them 1-based), values (re,im interleaved
 * when is_complex).
 */
#ifndef NTPOLY_ORACLE_H
#define NTPOLY_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct omat {
  int32_t rows, cols, is_complex;
  int64_t nnz;
  int64_t *outer; /* cols+1 */
  int32_t *inner; /* nnz, 0-based row index, ascending within a column */
  double *val;    /* nnz (real) or 2*nnz (complex, interleaved) */
} omat;

/* containers */
omat *omat_new(int32_t rows, int32_t cols, int32_t is_complex, int64_t nnz);
void omat_free(omat *m);
omat *omat_copy(const omat *m);
/* triplets are 1-based (index_column, index_row, value) like TripletModule.F90:14-25 */
omat *omat_from_triplets(int32_t rows, int32_t cols, int64_t n, const int32_t *col,
                         const int32_t *row, const double *val, int32_t is_complex);
void omat_to_triplets(const omat *m, int32_t *col, int32_t *row, double *val);
omat *omat_transpose(const omat *m);
omat *omat_identity(int32_t n, int32_t is_complex);
omat *omat_to_complex(const omat *m);
void omat_conjugate(omat *m);

/* local algebra (SMatrixAlgebraModule) */
omat *oracle_gemm(const omat *A, const omat *B, const omat *Cin, int tA, int tB, double alpha,
                  double beta, int has_beta, double threshold);
omat *oracle_increment(const omat *A, const omat *B, double alpha, double threshold);
omat *oracle_pairwise(const omat *A, const omat *B);
void oracle_dot(const omat *A, const omat *B, double out[2]);
void oracle_scale(omat *A, double c);
double oracle_trace(const omat *A);
double oracle_norm(const omat *A);
void oracle_gershgorin(const omat *A, double *emin, double *emax);
double oracle_sigma(const omat *A);
int oracle_is_identity(const omat *A);

/* distributed-level algebra on a 1x1x1 grid (PSMatrixAlgebraModule) */
omat *oracle_ps_multiply(const omat *A, const omat *B, const omat *Cin, double alpha, double beta,
                         double threshold);

/* solver parameters (SolverParametersModule.F90:14-33) */
typedef struct oparams {
  double converge_diff;
  int32_t max_iterations;
  double threshold;
  int32_t monitor_convergence;
  double step_thresh;
  int32_t do_load_balancing;
  const int32_t *perm; /* index_lookup, 1-based values, length = dim (may be NULL) */
} oparams;
void oparams_default(oparams *p);

/* per-iteration trace recorded by the solvers (mirrors the be_verbose YAML log) */
typedef struct otrace {
  int32_t iterations;  /* number of loop bodies executed */
  int32_t cap;
  double *value;       /* "Convergence" value appended to the monitor each iteration */
  double *energy;      /* TRS2/TRS4 energy per iteration (else 0) */
  double *sigma;       /* TRS2 sigma per iteration */
  int64_t *nnz;        /* nnz of the iterate after the update */
  double *stamp;       /* monotonic wall clock (seconds) when the iteration was recorded: differences time single iterations */
} otrace;
otrace *otrace_new(int32_t cap);
void otrace_free(otrace *t);

omat *oracle_trs2(const omat *H, const omat *ISQ, double trace, const oparams *p, double *energy,
                  double *mu, otrace *tr);
omat *oracle_trs4(const omat *H, const omat *ISQ, double trace, const oparams *p, double *energy,
                  double *mu, otrace *tr);
omat *oracle_sign(const omat *A, const oparams *p, otrace *tr);
omat *oracle_invert(const omat *A, const oparams *p, otrace *tr);
omat *oracle_inverse_square_root(const omat *A, const oparams *p, otrace *tr);
omat *oracle_square_root(const omat *A, const oparams *p, otrace *tr);

/* convergence monitor (ConvergenceMonitorModule.F90:14-191), exposed for unit tests */
typedef struct omonitor {
  double win_short[3], win_long[6];
  int32_t nval;
  double loose_cutoff, tight_cutoff;
  int32_t automatic;
} omonitor;
void omonitor_init(omonitor *m, int automatic, double tight_cutoff);
void omonitor_append(omonitor *m, double v);
int omonitor_converged(const omonitor *m);

int oracle_num_threads(void);
/* arithmetic of the real multiply kernel: 1 = fma(a, b, acc), the reference built with FP contraction; 0 (default) unfused */
void oracle_set_fma(int on);
int oracle_get_fma(void);

#ifdef __cplusplus
}
#endif
#endif
