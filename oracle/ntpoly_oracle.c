/* TEST INFRASTRUCTURE ONLY -- see ntpoly_oracle.h.
 *
 * Plain-C restatement of the reference's hot path.  Every routine cites the
 * reference file:line it follows (paths relative to /root/reference/Source/Fortran).
 * Build: gcc -O2 -std=gnu11 -fopenmp -ffp-contract=off -fPIC -shared (oracle/Makefile).
 * -ffp-contract=off matters: the reference build targets baseline x86-64 (no FMA),
 * so every a*b+c is two roundings; the oracle and the HIP engine do the same.
 */
#include "ntpoly_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ containers */
omat *omat_new(int32_t rows, int32_t cols, int32_t is_complex, int64_t nnz) {
  omat *m = (omat *)calloc(1, sizeof(omat));
  m->rows = rows;
  m->cols = cols;
  m->is_complex = is_complex;
  m->nnz = nnz;
  m->outer = (int64_t *)calloc((size_t)cols + 1, sizeof(int64_t));
  m->inner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz > 0 ? nnz : 1));
  m->val = (double *)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1) * (is_complex ? 2 : 1));
  return m;
}

void omat_free(omat *m) {
  if (!m) return;
  free(m->outer);
  free(m->inner);
  free(m->val);
  free(m);
}

omat *omat_copy(const omat *m) {
  omat *c = omat_new(m->rows, m->cols, m->is_complex, m->nnz);
  memcpy(c->outer, m->outer, sizeof(int64_t) * ((size_t)m->cols + 1));
  memcpy(c->inner, m->inner, sizeof(int32_t) * (size_t)m->nnz);
  memcpy(c->val, m->val, sizeof(double) * (size_t)m->nnz * (m->is_complex ? 2 : 1));
  return c;
}

static int cmp_i32(const void *a, const void *b) {
  const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
  return (x > y) - (x < y);
}

typedef struct {
  int64_t key;
  int64_t pos;
} keypos;
static int cmp_keypos(const void *a, const void *b) {
  const keypos *x = (const keypos *)a, *y = (const keypos *)b;
  if (x->key != y->key) return (x->key > y->key) - (x->key < y->key);
  return (x->pos > y->pos) - (x->pos < y->pos);
}

/* SortTripletList (triplet_includes/SortTripletList.f90:20-67) followed by
 * ConstructMatrixFromTripletList (sparse_includes/ConstructMatrixFromTripletList.f90:17-27):
 * order by column then row, duplicates are kept as separate entries. */
omat *omat_from_triplets(int32_t rows, int32_t cols, int64_t n, const int32_t *col,
                         const int32_t *row, const double *val, int32_t is_complex) {
  keypos *kp = (keypos *)malloc(sizeof(keypos) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; ++i) {
    kp[i].key = (int64_t)(col[i] - 1) * ((int64_t)rows + 1) + (row[i] - 1);
    kp[i].pos = i;
  }
  qsort(kp, (size_t)n, sizeof(keypos), cmp_keypos);
  omat *m = omat_new(rows, cols, is_complex, n);
  const int w = is_complex ? 2 : 1;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t p = kp[i].pos;
    m->inner[i] = row[p] - 1;
    for (int k = 0; k < w; ++k) m->val[i * w + k] = val[p * w + k];
    m->outer[col[p]] += 1;
  }
  for (int32_t j = 0; j < cols; ++j) m->outer[j + 1] += m->outer[j];
  free(kp);
  return m;
}

void omat_to_triplets(const omat *m, int32_t *col, int32_t *row, double *val) {
  const int w = m->is_complex ? 2 : 1;
  for (int32_t j = 0; j < m->cols; ++j)
    for (int64_t p = m->outer[j]; p < m->outer[j + 1]; ++p) {
      col[p] = j + 1;
      row[p] = m->inner[p] + 1;
    }
  memcpy(val, m->val, sizeof(double) * (size_t)m->nnz * w);
}

/* TransposeMatrix (sparse_includes/TransposeMatrix.f90:19-44): histogram, offsets, scatter. */
omat *omat_transpose(const omat *m) {
  omat *t = omat_new(m->cols, m->rows, m->is_complex, m->nnz);
  const int w = m->is_complex ? 2 : 1;
  int64_t *off = (int64_t *)calloc((size_t)m->rows + 1, sizeof(int64_t));
  for (int64_t p = 0; p < m->nnz; ++p) off[m->inner[p] + 1] += 1;
  for (int32_t i = 0; i < m->rows; ++i) off[i + 1] += off[i];
  memcpy(t->outer, off, sizeof(int64_t) * ((size_t)m->rows + 1));
  for (int32_t j = 0; j < m->cols; ++j)
    for (int64_t p = m->outer[j]; p < m->outer[j + 1]; ++p) {
      const int64_t q = off[m->inner[p]]++;
      t->inner[q] = j;
      for (int k = 0; k < w; ++k) t->val[q * w + k] = m->val[p * w + k];
    }
  free(off);
  return t;
}

/* FillMatrixIdentity (distributed_includes/FillMatrixIdentity.f90:9-22). */
omat *omat_identity(int32_t n, int32_t is_complex) {
  omat *m = omat_new(n, n, is_complex, n);
  const int w = is_complex ? 2 : 1;
  for (int32_t j = 0; j < n; ++j) {
    m->outer[j + 1] = j + 1;
    m->inner[j] = j;
    m->val[(int64_t)j * w] = 1.0;
    if (is_complex) m->val[(int64_t)j * w + 1] = 0.0;
  }
  return m;
}

/* ConvertMatrixToComplex (PSMatrixModule.F90:1687-1698). */
omat *omat_to_complex(const omat *m) {
  if (m->is_complex) return omat_copy(m);
  omat *c = omat_new(m->rows, m->cols, 1, m->nnz);
  memcpy(c->outer, m->outer, sizeof(int64_t) * ((size_t)m->cols + 1));
  memcpy(c->inner, m->inner, sizeof(int32_t) * (size_t)m->nnz);
  for (int64_t p = 0; p < m->nnz; ++p) {
    c->val[2 * p] = m->val[p];
    c->val[2 * p + 1] = 0.0;
  }
  return c;
}

void omat_conjugate(omat *m) {
  if (!m->is_complex) return;
  for (int64_t p = 0; p < m->nnz; ++p) m->val[2 * p + 1] = -m->val[2 * p + 1];
}

/* ------------------------------------------------------------ generic kernels */
/* Arithmetic of the real multiply kernel: 0 = separate multiply and add (the reference's default x86-64 build),
 * 1 = one rounding per product, fma(a, b, acc) (the reference built with FP contraction, build_ref.py --fma). */
static int oracle_fma_mode = 0;
void oracle_set_fma(int on) { oracle_fma_mode = on ? 1 : 0; }
int oracle_get_fma(void) { return oracle_fma_mode; }
static inline double muladd_r(double a, double b, double acc) {
  if (oracle_fma_mode) return fma(a, b, acc);
  const double prod = a * b;
  return acc + prod;
}

#define T double
#define FN(name) name##_r
#define MULADD_T(a, b, acc) muladd_r((a), (b), (acc))
#define ABS_T(x) fabs(x)
#define CONJ_T(x) (x)
#define REAL_T(x) (x)
#define IS_COMPLEX_T 0
#include "ntpoly_oracle_kernels.inc"
#undef T
#undef FN
#undef MULADD_T
#undef ABS_T
#undef CONJ_T
#undef REAL_T
#undef IS_COMPLEX_T

#define T double _Complex
#define FN(name) name##_c
#define MULADD_T(a, b, acc) ((acc) + (a) * (b))
#define ABS_T(x) cabs(x)
#define CONJ_T(x) conj(x)
#define REAL_T(x) creal(x)
#define IS_COMPLEX_T 1
#include "ntpoly_oracle_kernels.inc"
#undef T
#undef FN
#undef MULADD_T
#undef ABS_T
#undef CONJ_T
#undef REAL_T
#undef IS_COMPLEX_T

/* --------------------------------------------------------------- local algebra */
void oracle_scale(omat *A, double c) { /* sparse_includes/ScaleMatrix.f90:1 */
  const int64_t n = A->nnz * (A->is_complex ? 2 : 1);
  for (int64_t p = 0; p < n; ++p) A->val[p] = c * A->val[p];
}

omat *oracle_increment(const omat *A, const omat *B, double alpha, double threshold) {
  /* mixed real/complex is up-cast as IncrementMatrix_ps does (PSMatrixAlgebraModule.F90:441-447) */
  if (A->is_complex != B->is_complex) {
    omat *Ac = omat_to_complex(A), *Bc = omat_to_complex(B);
    omat *C = increment_c(Ac, Bc, alpha, threshold);
    omat_free(Ac);
    omat_free(Bc);
    return C;
  }
  return A->is_complex ? increment_c(A, B, alpha, threshold) : increment_r(A, B, alpha, threshold);
}

omat *oracle_pairwise(const omat *A, const omat *B) {
  return A->is_complex ? pairwise_c(A, B, 0) : pairwise_r(A, B, 0);
}

/* DotMatrix_lsr / DotMatrix_lsc (SMatrixAlgebraModule.F90:178-211): materialise the
 * Hadamard product (conj(A) for complex) then MatrixGrandSum. */
void oracle_dot(const omat *A, const omat *B, double out[2]) {
  if (A->is_complex) {
    omat *C = pairwise_c(A, B, 1);
    double _Complex s = grand_sum_c(C);
    out[0] = creal(s);
    out[1] = cimag(s);
    omat_free(C);
  } else {
    omat *C = pairwise_r(A, B, 0);
    out[0] = grand_sum_r(C);
    out[1] = 0;
    omat_free(C);
  }
}

double oracle_trace(const omat *A) { return A->is_complex ? trace_c(A) : trace_r(A); }

/* MatrixNorm (sparse_includes/MatrixNorm.f90:1-2, distributed_algebra_includes/MatrixNorm.f90:1-16) */
double oracle_norm(const omat *A) {
  double *cn = (double *)calloc((size_t)A->cols + 1, sizeof(double));
  if (A->is_complex)
    column_norm_c(A, cn);
  else
    column_norm_r(A, cn);
  double mx = A->cols > 0 ? cn[0] : 0;
  for (int32_t j = 1; j < A->cols; ++j)
    if (cn[j] > mx) mx = cn[j];
  free(cn);
  return mx;
}

void oracle_gershgorin(const omat *A, double *emin, double *emax) {
  if (A->is_complex)
    gershgorin_c(A, emin, emax);
  else
    gershgorin_r(A, emin, emax);
}

/* MatrixSigma (distributed_algebra_includes/MatrixSigma.f90:1-20) */
double oracle_sigma(const omat *A) {
  const double n = oracle_norm(A);
  return 1.0 / (n * n);
}

int oracle_is_identity(const omat *A) { return A->is_complex ? is_identity_c(A) : is_identity_r(A); }

/* GemmMatrix (sparse_includes/GemmMatrix.f90:1-101) + SparseBranch (SparseBranch.f90:1-21).
 * The dense branch (GemmMatrix.f90:59-61 -> DenseBranch.f90:1-18 -> DGEMM/ZGEMM,
 * DMatrixModule.F90:307,590) is restated with the same per-element arithmetic as the
 * sparse branch but DenseBranch's order of threshold and alpha; BLAS's summation order
 * is not reproduced (stated tolerance in tests: 1e-13 relative). */
omat *oracle_gemm(const omat *A, const omat *B, const omat *Cin, int tA, int tB, double alpha,
                  double beta, int has_beta, double threshold) {
  const double sparsity_a = (double)A->nnz / ((double)A->rows * (double)A->cols);
  const double sparsity_b = (double)B->nnz / ((double)B->rows * (double)B->cols);
  const int dense_rule = (sparsity_a < sparsity_b ? sparsity_a : sparsity_b) > 0.1;
  /* SparseBranch.f90:2-7: transpose whichever operand is not pre-transposed */
  omat *AT = tA ? (omat *)A : omat_transpose(A);
  omat *BT = tB ? (omat *)B : omat_transpose(B);
  omat *CT = A->is_complex ? multiply_block_c(AT, BT, alpha, threshold, dense_rule)
                           : multiply_block_r(AT, BT, alpha, threshold, dense_rule);
  if (!tA) omat_free(AT);
  if (!tB) omat_free(BT);
  omat *AB = omat_transpose(CT); /* PruneList.f90:35-38: sorted triplets -> CSC */
  omat_free(CT);
  /* GemmMatrix.f90:88-98 */
  if (has_beta && fabs(beta) > 0 && Cin) {
    omat *Cs = omat_copy(Cin);
    oracle_scale(Cs, beta);
    omat *R = oracle_increment(AB, Cs, 1.0, 0.0);
    omat_free(Cs);
    omat_free(AB);
    return R;
  }
  return AB;
}

/* ------------------------------------------------- distributed level, 1x1x1 grid */
/* MatrixMultiply_ps (PSMatrixAlgebraModule.F90:108-211) with the body of
 * distributed_algebra_includes/MatrixMultiply.f90 on one process, one block:
 * working_threshold = threshold (:25-29, slices == 1), one GemmAB task with both
 * operands pre-transposed (:216-224), then beta handling (:324-329). */
omat *oracle_ps_multiply(const omat *A, const omat *B, const omat *Cin, double alpha, double beta,
                         double threshold) {
  omat *Ac = NULL, *Bc = NULL;
  if (A->is_complex != B->is_complex) { /* up-casting, PSMatrixAlgebraModule.F90:171-188 */
    Ac = omat_to_complex(A);
    Bc = omat_to_complex(B);
    A = Ac;
    B = Bc;
  }
  omat *AB = oracle_gemm(A, B, NULL, 0, 0, alpha, 0.0, 0, threshold);
  omat_free(Ac);
  omat_free(Bc);
  if (fabs(beta) < DBL_MIN || !Cin) return AB;
  omat *Cs = omat_copy(Cin);
  oracle_scale(Cs, beta);
  omat *R = oracle_increment(AB, Cs, 1.0, 0.0);
  omat_free(Cs);
  omat_free(AB);
  return R;
}

/* ---------------------------------------------------------- convergence monitor */
/* ConstructMonitor (ConvergenceMonitorModule.F90:35-89) */
void omonitor_init(omonitor *m, int automatic, double tight_cutoff) {
  memset(m, 0, sizeof(*m));
  m->loose_cutoff = 1e-2;
  m->tight_cutoff = tight_cutoff;
  m->automatic = automatic;
}
/* AppendValue (:101-119) */
void omonitor_append(omonitor *m, double v) {
  for (int i = 0; i < 2; ++i) m->win_short[i] = m->win_short[i + 1];
  for (int i = 0; i < 5; ++i) m->win_long[i] = m->win_long[i + 1];
  m->win_short[2] = v;
  m->win_long[5] = v;
  m->nval += 1;
}
/* CheckConverged (:122-191) */
int omonitor_converged(const omonitor *m) {
  const double last = m->win_short[2], last2 = m->win_short[1];
  int conv = !(fabs(last) > m->tight_cutoff);
  if (!m->automatic || conv) return conv;
  conv = 1;
  if (m->nval < 6) conv = 0;
  double s = 0;
  for (int i = 0; i < 3; ++i) s = s + m->win_short[i];
  const double avg_short = s / 3;
  s = 0;
  for (int i = 0; i < 6; ++i) s = s + m->win_long[i];
  const double avg_long = s / 6;
  if (!(10 * avg_short > avg_long && avg_short / 10 < avg_long)) conv = 0;
  if (!(10 * last > avg_long && last / 10 < avg_long)) conv = 0;
  if (last < 0) conv = 0;
  if (fabs(last) < fabs(last2)) conv = 0;
  if (avg_long > m->loose_cutoff) conv = 0;
  return conv;
}

/* ------------------------------------------------------------------ parameters */
void oparams_default(oparams *p) { /* SolverParametersModule.F90:48-50,77-112 */
  p->converge_diff = 1e-6;
  p->max_iterations = 1000;
  p->threshold = 0.0;
  p->monitor_convergence = 1;
  p->step_thresh = 1e-2;
  p->do_load_balancing = 0;
  p->perm = NULL;
}

otrace *otrace_new(int32_t cap) {
  otrace *t = (otrace *)calloc(1, sizeof(otrace));
  t->cap = cap;
  t->value = (double *)calloc((size_t)cap + 1, sizeof(double));
  t->energy = (double *)calloc((size_t)cap + 1, sizeof(double));
  t->sigma = (double *)calloc((size_t)cap + 1, sizeof(double));
  t->nnz = (int64_t *)calloc((size_t)cap + 1, sizeof(int64_t));
  t->stamp = (double *)calloc((size_t)cap + 1, sizeof(double));
  return t;
}
void otrace_free(otrace *t) {
  if (!t) return;
  free(t->value);
  free(t->energy);
  free(t->sigma);
  free(t->nnz);
  free(t->stamp);
  free(t);
}
static void trace_rec(otrace *t, int it, double value, double energy, double sigma, int64_t nnz) {
  if (!t || it >= t->cap) return;
  t->value[it] = value;
  t->energy[it] = energy;
  t->sigma[it] = sigma;
  t->nnz[it] = nnz;
  {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    t->stamp[it] = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
  }
  t->iterations = it + 1;
}

/* ------------------------------------------- helpers mirroring the module API */
static void replace(omat **dst, omat *src) {
  omat_free(*dst);
  *dst = src;
}
/* C = alpha*A*B (+beta*C), in place on *C */
static void ps_mm(const omat *A, const omat *B, omat **C, double alpha, double beta, double thr) {
  replace(C, oracle_ps_multiply(A, B, *C, alpha, beta, thr));
}
/* B <- alpha*A + B */
static void ps_inc(const omat *A, omat **B, double alpha, double thr) {
  replace(B, oracle_increment(A, *B, alpha, thr));
}
static omat *ps_empty(const omat *like) { return omat_new(like->rows, like->cols, like->is_complex, 0); }
static double ps_dot(const omat *A, const omat *B) {
  /* DotMatrix_psr returns the real part for complex operands (PSMatrixAlgebraModule.F90:387-397) */
  double out[2];
  if (A->is_complex != B->is_complex) {
    omat *Ac = omat_to_complex(A), *Bc = omat_to_complex(B);
    oracle_dot(Ac, Bc, out);
    omat_free(Ac);
    omat_free(Bc);
  } else {
    oracle_dot(A, B, out);
  }
  return out[0];
}

/* FillMatrixPermutation (distributed_includes/FillMatrixPermutation.f90:1-35) */
static omat *perm_matrix(const int32_t *lookup, int32_t n, int rows, int is_complex) {
  int32_t *c = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  int32_t *r = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  double *v = (double *)calloc((size_t)n * 2, sizeof(double));
  for (int32_t i = 0; i < n; ++i) {
    if (rows) {
      c[i] = lookup[i];
      r[i] = i + 1;
    } else {
      c[i] = i + 1;
      r[i] = lookup[i];
    }
    v[is_complex ? 2 * i : i] = 1.0;
  }
  omat *m = omat_from_triplets(n, n, n, c, r, v, is_complex);
  free(c);
  free(r);
  free(v);
  return m;
}
/* PermuteMatrix / UndoPermuteMatrix (LoadBalancerModule.F90:14-92) */
static void permute(omat **M, const int32_t *lookup, int undo) {
  const int32_t n = (*M)->rows;
  omat *PR = perm_matrix(lookup, n, 1, (*M)->is_complex);
  omat *PC = perm_matrix(lookup, n, 0, (*M)->is_complex);
  omat *tmp = NULL, *out = NULL;
  if (!undo) {
    ps_mm(PR, *M, &tmp, 1.0, 0.0, 0.0);
    ps_mm(tmp, PC, &out, 1.0, 0.0, 0.0);
  } else {
    ps_mm(PC, *M, &tmp, 1.0, 0.0, 0.0);
    ps_mm(tmp, PR, &out, 1.0, 0.0, 0.0);
  }
  omat_free(PR);
  omat_free(PC);
  omat_free(tmp);
  replace(M, out);
}
/* SimilarityTransform (PSMatrixAlgebraModule.F90:603-654) */
static omat *similarity(const omat *A, const omat *P, const omat *PInv, double thr) {
  if (oracle_is_identity(P)) return omat_copy(A);
  omat *tmp = NULL, *res = NULL;
  ps_mm(P, A, &tmp, 1.0, 0.0, thr);
  ps_mm(tmp, PInv, &res, 1.0, 0.0, thr);
  omat_free(tmp);
  return res;
}

/* ---------------------------------------------------------------------- TRS2 */
/* DensityMatrixSolversModule.F90:285-481 */
omat *oracle_trs2(const omat *H, const omat *ISQ, double trace, const oparams *p, double *energy,
                  double *mu, otrace *tr) {
  omonitor mon;
  omonitor_init(&mon, p->monitor_convergence, p->converge_diff);
  double *sigma_array = (double *)calloc((size_t)p->max_iterations + 1, sizeof(double));
  omat *IMat = omat_identity(H->rows, H->is_complex);
  omat *ISQT = omat_transpose(ISQ);                       /* :352 */
  omat *WH = similarity(H, ISQ, ISQT, p->threshold);      /* :353-354 */
  if (p->do_load_balancing) {                             /* :357-362 */
    permute(&WH, p->perm, 0);
    permute(&IMat, p->perm, 0);
  }
  double e_min, e_max;
  oracle_gershgorin(WH, &e_min, &e_max);                  /* :365 */
  omat *X = omat_copy(WH);                                /* :368-371 */
  oracle_scale(X, -1.0);
  ps_inc(IMat, &X, e_max, 0.0);
  oracle_scale(X, 1.0 / (e_max - e_min));
  omat *X2 = NULL;
  double energy_value = 0.0, energy_old;
  int II;
  for (II = 1; II <= p->max_iterations; ++II) {           /* :380-413 */
    const double trace_value = oracle_trace(X);
    sigma_array[II] = (trace - trace_value < 0.0) ? -1.0 : 1.0;
    ps_mm(X, X, &X2, 1.0, 0.0, p->threshold);
    if (sigma_array[II] > 0.0) {
      oracle_scale(X, 2.0);
      ps_inc(X2, &X, -1.0, p->threshold);
    } else {
      replace(&X, omat_copy(X2));
    }
    energy_old = energy_value;
    energy_value = ps_dot(X, WH);
    omonitor_append(&mon, energy_value - energy_old);
    trace_rec(tr, II - 1, energy_value - energy_old, energy_value, sigma_array[II], X->nnz);
    if (omonitor_converged(&mon)) break;
  }
  const int total_iterations = II - 1;                    /* :414 */
  if (energy) *energy = energy_value;
  if (p->do_load_balancing) permute(&X, p->perm, 1);      /* :427-430 */
  omat *K = similarity(X, ISQT, ISQ, p->threshold);       /* :433-434 */
  if (mu) {                                               /* :444-472 */
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p->max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double zero_value = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ) {
        if (sigma_array[JJ] < 0.0)
          zero_value = zero_value * zero_value;
        else
          zero_value = 2.0 * zero_value - zero_value * zero_value;
      }
      if (zero_value < 0.5)
        interval_a = midpoint;
      else
        interval_b = midpoint;
      if (fabs(zero_value - 0.5) < p->converge_diff) break;
    }
    *mu = e_max + (e_min - e_max) * midpoint;
  }
  omat_free(WH);
  omat_free(ISQT);
  omat_free(X);
  omat_free(X2);
  omat_free(IMat);
  free(sigma_array);
  return K;
}

/* ---------------------------------------------------------------------- TRS4 */
/* DensityMatrixSolversModule.F90:485-716 */
omat *oracle_trs4(const omat *H, const omat *ISQ, double trace, const oparams *p, double *energy,
                  double *mu, otrace *tr) {
  const double sigma_min = 0.0, sigma_max = 6.0;
  omonitor mon;
  omonitor_init(&mon, p->monitor_convergence, p->converge_diff);
  double *sigma_array = (double *)calloc((size_t)p->max_iterations + 1, sizeof(double));
  omat *IMat = omat_identity(H->rows, H->is_complex);
  omat *ISQT = omat_transpose(ISQ);
  omat *WH = similarity(H, ISQ, ISQT, p->threshold);
  if (p->do_load_balancing) {
    permute(&WH, p->perm, 0);
    permute(&IMat, p->perm, 0);
  }
  double e_min, e_max;
  oracle_gershgorin(WH, &e_min, &e_max);
  omat *X = omat_copy(WH);
  oracle_scale(X, -1.0);
  ps_inc(IMat, &X, e_max, 0.0);
  oracle_scale(X, 1.0 / (e_max - e_min));
  omat *X2 = NULL, *Fx = NULL, *Gx = NULL, *Temp = NULL;
  double energy_value = 0.0, energy_old;
  int II;
  for (II = 1; II <= p->max_iterations; ++II) {           /* :586-638 */
    ps_mm(X, X, &X2, 1.0, 0.0, p->threshold);
    replace(&Fx, omat_copy(X2));
    oracle_scale(Fx, -3.0);
    ps_inc(X, &Fx, 4.0, 0.0);
    replace(&Gx, omat_copy(IMat));
    ps_inc(X, &Gx, -2.0, 0.0);
    ps_inc(X2, &Gx, 1.0, 0.0);
    const double trace_fx = ps_dot(X2, Fx);
    const double trace_gx = ps_dot(X2, Gx);
    if (fabs(trace_gx) < 1.0e-14)
      sigma_array[II] = 0.5 * (sigma_max - sigma_min);
    else
      sigma_array[II] = (trace - trace_fx) / trace_gx;
    if (sigma_array[II] > sigma_max) {
      replace(&Temp, omat_copy(X));
      oracle_scale(Temp, 2.0);
      ps_inc(X2, &Temp, -1.0, 0.0);
    } else if (sigma_array[II] < sigma_min) {
      replace(&Temp, omat_copy(X2));
    } else {
      oracle_scale(Gx, sigma_array[II]);
      ps_inc(Fx, &Gx, 1.0, 0.0);
      ps_mm(X2, Gx, &Temp, 1.0, 0.0, p->threshold);
    }
    /* :630-631: IncrementMatrix(TempMat, X_k, -1) is immediately overwritten by the copy */
    replace(&X, omat_copy(Temp));
    energy_old = energy_value;
    energy_value = ps_dot(X, WH);
    omonitor_append(&mon, energy_value - energy_old);
    trace_rec(tr, II - 1, energy_value - energy_old, energy_value, sigma_array[II], X->nnz);
    if (omonitor_converged(&mon)) break;
  }
  const int total_iterations = II - 1;
  if (energy) *energy = energy_value;
  if (p->do_load_balancing) permute(&X, p->perm, 1);
  omat *K = similarity(X, ISQT, ISQ, p->threshold);
  if (mu) {                                               /* :669-704 */
    double interval_a = 0.0, interval_b = 1.0, midpoint = 0.0;
    for (int it = 1; it <= p->max_iterations; ++it) {
      midpoint = (interval_b - interval_a) / 2.0 + interval_a;
      double z = midpoint;
      for (int JJ = 1; JJ <= total_iterations; ++JJ) {
        if (sigma_array[JJ] > sigma_max)
          z = 2.0 * z - z * z;
        else if (sigma_array[JJ] < sigma_min)
          z = z * z;
        else {
          const double tempfx = (z * z) * (4.0 * z - 3.0 * z * z);
          const double tempgx = (z * z) * (1.0 - z) * (1.0 - z);
          z = tempfx + sigma_array[JJ] * tempgx;
        }
      }
      if (z < 0.5)
        interval_a = midpoint;
      else
        interval_b = midpoint;
      if (fabs(z - 0.5) < p->converge_diff) break;
    }
    *mu = e_max + (e_min - e_max) * midpoint;
  }
  omat_free(WH);
  omat_free(ISQT);
  omat_free(X);
  omat_free(X2);
  omat_free(Fx);
  omat_free(Gx);
  omat_free(Temp);
  omat_free(IMat);
  free(sigma_array);
  return K;
}

/* ---------------------------------------------------------------------- Sign */
/* SignSolversModule.F90:150-258 (CoreComputation, needs_transpose = false) */
omat *oracle_sign(const omat *A, const oparams *p, otrace *tr) {
  const double alpha = 1.69770248526;
  omonitor mon;
  omonitor_init(&mon, p->monitor_convergence, p->converge_diff);
  omat *Identity = omat_identity(A->rows, A->is_complex);
  omat *Out = omat_copy(A);
  if (p->do_load_balancing) {
    permute(&Identity, p->perm, 0);
    permute(&Out, p->perm, 0);
  }
  double e_min, e_max;
  oracle_gershgorin(A, &e_min, &e_max);                   /* :193 (on InMat) */
  double xk = fabs(e_min / e_max);
  oracle_scale(Out, 1.0 / fabs(e_max));
  omat *Temp1 = NULL, *Temp2 = NULL;
  for (int II = 1; II <= p->max_iterations; ++II) {       /* :204-237 */
    const double alpha_k = fmin(sqrt(3.0 / (1.0 + xk + xk * xk)), alpha);
    xk = 0.5 * alpha_k * xk * (3.0 - (alpha_k * alpha_k) * (xk * xk));
    ps_mm(Out, Out, &Temp1, -1.0 * (alpha_k * alpha_k), 0.0, p->threshold);
    ps_inc(Identity, &Temp1, 3.0, 0.0);
    ps_mm(Out, Temp1, &Temp2, 0.5 * alpha_k, 0.0, p->threshold);
    ps_inc(Temp2, &Out, -1.0, 0.0);
    const double norm_value = oracle_norm(Out);
    replace(&Out, omat_copy(Temp2));
    omonitor_append(&mon, norm_value);
    trace_rec(tr, II - 1, norm_value, 0.0, alpha_k, Out->nnz);
    if (omonitor_converged(&mon)) break;
  }
  if (p->do_load_balancing) permute(&Out, p->perm, 1);
  omat_free(Temp1);
  omat_free(Temp2);
  omat_free(Identity);
  return Out;
}

/* -------------------------------------------------------------------- Invert */
/* InverseSolversModule.F90:29-149 */
omat *oracle_invert(const omat *A, const oparams *p, otrace *tr) {
  omonitor mon;
  omonitor_init(&mon, p->monitor_convergence, p->converge_diff);
  omat *Identity = omat_identity(A->rows, A->is_complex);
  omat *Balanced = omat_copy(A);
  if (p->do_load_balancing) {
    permute(&Identity, p->perm, 0);
    permute(&Balanced, p->perm, 0);
  }
  const double sigma = oracle_sigma(Balanced);            /* :88 */
  omat *Out = omat_copy(Balanced);
  oracle_scale(Out, sigma);
  omat *Temp1 = NULL, *Temp2 = NULL;
  for (int II = 1; II <= p->max_iterations; ++II) {       /* :101-127 */
    ps_mm(Out, Balanced, &Temp1, 1.0, 0.0, p->threshold);
    replace(&Temp2, omat_copy(Identity));
    ps_inc(Temp1, &Temp2, -1.0, 0.0);
    const double norm_value = oracle_norm(Temp2);
    replace(&Temp2, NULL);
    ps_mm(Temp1, Out, &Temp2, -1.0, 0.0, p->threshold);
    oracle_scale(Out, 2.0);
    ps_inc(Temp2, &Out, 1.0, p->threshold);
    omonitor_append(&mon, norm_value);
    trace_rec(tr, II - 1, norm_value, 0.0, sigma, Out->nnz);
    if (omonitor_converged(&mon)) break;
  }
  if (p->do_load_balancing) permute(&Out, p->perm, 1);
  omat_free(Temp1);
  omat_free(Temp2);
  omat_free(Balanced);
  omat_free(Identity);
  return Out;
}

/* --------------------------------------------- (Inverse)SquareRoot, Taylor order 5 */
/* SquareRootSolversModule.F90:342-531 with taylor_order = 5 (the default, :179-183) */
static omat *isr_taylor5(const omat *A, const oparams *p, int compute_inverse, otrace *tr) {
  omonitor mon;
  omonitor_init(&mon, p->monitor_convergence, p->converge_diff);
  omat *Identity = omat_identity(A->rows, A->is_complex);
  double e_min, e_max;
  oracle_gershgorin(A, &e_min, &e_max);                   /* :389-391 */
  const double max_between = fmax(fabs(e_min), fabs(e_max));
  const double lambda = 1.0 / max_between;
  omat *ISR = omat_identity(A->rows, A->is_complex);      /* :394-396 */
  omat *SR = omat_copy(A);
  oracle_scale(SR, lambda);
  if (p->do_load_balancing) {                             /* :399-406 */
    permute(&SR, p->perm, 0);
    permute(&Identity, p->perm, 0);
    permute(&ISR, p->perm, 0);
  }
  omat *X = NULL, *Temp = NULL, *Temp2 = NULL;
  const double aa = -40.0 / 35.0, bb = 48.0 / 35.0, cc = -64.0 / 35.0, dd = 128.0 / 35.0;
  const double a = (aa - 1.0) / 2.0;                      /* :453-456 */
  const double b = bb * (a + 1.0) - cc - a * ((a + 1.0) * (a + 1.0));
  const double c = bb - b - a * (a + 1.0);
  const double d = dd - b * c;
  for (int II = 1; II <= p->max_iterations; ++II) {       /* :415-497 */
    ps_mm(ISR, SR, &X, 1.0, 0.0, p->threshold);
    ps_inc(Identity, &X, -1.0, 0.0);
    const double norm_value = oracle_norm(X);
    ps_mm(X, X, &Temp, 1.0, 0.0, p->threshold);           /* :459-462 */
    ps_inc(X, &Temp, a, 0.0);
    replace(&Temp2, omat_copy(Identity));                 /* :465-468 */
    oracle_scale(Temp2, b);
    ps_inc(X, &Temp2, 1.0, 0.0);
    ps_inc(Temp, &Temp2, 1.0, 0.0);
    ps_inc(Identity, &Temp, c, 0.0);                      /* :471 */
    ps_mm(Temp2, Temp, &X, 1.0, 0.0, p->threshold);       /* :474-476 */
    ps_inc(Identity, &X, d, 0.0);
    oracle_scale(X, 35.0 / 128.0);                        /* :479 */
    replace(&Temp, omat_copy(ISR));                       /* :483-485 */
    ps_mm(X, Temp, &ISR, 1.0, 0.0, p->threshold);
    replace(&Temp, omat_copy(SR));                        /* :488-490 */
    ps_mm(Temp, X, &SR, 1.0, 0.0, p->threshold);
    omonitor_append(&mon, norm_value);
    trace_rec(tr, II - 1, norm_value, 0.0, lambda, ISR->nnz);
    if (omonitor_converged(&mon)) break;
  }
  omat *Out;
  if (compute_inverse) {                                  /* :505-511 */
    oracle_scale(ISR, sqrt(lambda));
    Out = omat_copy(ISR);
  } else {
    oracle_scale(SR, 1.0 / sqrt(lambda));
    Out = omat_copy(SR);
  }
  if (p->do_load_balancing) permute(&Out, p->perm, 1);
  omat_free(X);
  omat_free(Temp);
  omat_free(Temp2);
  omat_free(ISR);
  omat_free(SR);
  omat_free(Identity);
  return Out;
}
omat *oracle_inverse_square_root(const omat *A, const oparams *p, otrace *tr) {
  return isr_taylor5(A, p, 1, tr);
}
omat *oracle_square_root(const omat *A, const oparams *p, otrace *tr) {
  return isr_taylor5(A, p, 0, tr);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
