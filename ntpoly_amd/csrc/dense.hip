// Dense side of the engine: sparse <-> dense conversion, entry filter, Hermitian eigendecomposition (a parallel
// two-sided Jacobi method written for this engine) and the (pivoted) Cholesky factorisations.  These serve the
// reference's "dense" solver family (EigenSolversModule.F90, FermiOperatorModule.F90, LinearSolversModule.F90,
// AnalysisModule.F90), which gathers a matrix, factors it densely and sparsifies the result; none of it is on the
// SpGEMM hot path.  (rocSOLVER would do the eigenproblem, but loading its 0.9 GB library costs minutes per process.)
#include <algorithm>
#include <cmath>
#include <numeric>

#include "kernels.hpp"

namespace ntp {
namespace {
constexpr int WAVE = 64;
inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

__device__ inline double mag_of(double v) { return fabs(v); }
__device__ inline double mag_of(double2 v) { return hypot(v.x, v.y); }  // ABS of a Fortran COMPLEX

// ---- filter: count / write the entries of every column with |v| > threshold (FilterMatrix.f90:6-13)
template <typename T>
__global__ void k_filter_count(const int64_t* __restrict__ outer, const T* __restrict__ val, int cols, double threshold,
                               int64_t* __restrict__ cnt) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = threadIdx.x % WAVE;
  if (j >= cols) return;
  int64_t c = 0;
  for (int64_t p = outer[j] + lane; p < outer[j + 1]; p += WAVE) c += mag_of(val[p]) > threshold ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
  if (lane == 0) cnt[j] = c;
}
template <typename T>
__global__ void k_filter_write(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                               const T* __restrict__ val, int cols, double threshold,
                               const int64_t* __restrict__ new_outer, int32_t* __restrict__ out_inner,
                               T* __restrict__ out_val) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = threadIdx.x % WAVE;
  if (j >= cols) return;
  int64_t base = new_outer[j];
  const int64_t s = outer[j], e = outer[j + 1];
  for (int64_t p0 = s; p0 < e; p0 += WAVE) {
    const int64_t p = p0 + lane;
    const bool keep = p < e && mag_of(val[p]) > threshold;
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int64_t q = base + __popcll(m & ((1ull << lane) - 1ull));
      out_inner[q] = inner[p];
      out_val[q] = val[p];
    }
    base += __popcll(m);
  }
}

// ---- sparse -> dense (column major, leading dimension ld; the target is zeroed by the caller)
template <typename T>
__global__ void k_to_dense(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                           const T* __restrict__ val, int cols, T* __restrict__ dense, int64_t ld) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = threadIdx.x % WAVE;
  if (j >= cols) return;
  for (int64_t p = outer[j] + lane; p < outer[j + 1]; p += WAVE) dense[(int64_t)j * ld + inner[p]] = val[p];
}
// ---- dense -> sparse: columns [c0, c0 + cols) of a column-major array, entries with |v| > threshold
// (ConstructMatrixSFromD, DMatrixModule.F90)
template <typename T>
__global__ void k_dense_count(const T* __restrict__ dense, int64_t ld, int rows, int c0, int cols, double threshold,
                              int64_t* __restrict__ cnt) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = threadIdx.x % WAVE;
  if (j >= cols) return;
  const T* __restrict__ col = dense + (int64_t)(c0 + j) * ld;
  int64_t c = 0;
  for (int r = lane; r < rows; r += WAVE) c += mag_of(col[r]) > threshold ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, WAVE);
  if (lane == 0) cnt[j] = c;
}
template <typename T>
__global__ void k_dense_write(const T* __restrict__ dense, int64_t ld, int rows, int c0, int cols, double threshold,
                              const int64_t* __restrict__ outer, int32_t* __restrict__ out_inner,
                              T* __restrict__ out_val) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, lane = threadIdx.x % WAVE;
  if (j >= cols) return;
  const T* __restrict__ col = dense + (int64_t)(c0 + j) * ld;
  int64_t base = outer[j];
  for (int r0 = 0; r0 < rows; r0 += WAVE) {
    const int r = r0 + lane;
    const bool keep = r < rows && mag_of(col[r]) > threshold;
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int64_t q = base + __popcll(m & ((1ull << lane) - 1ull));
      out_inner[q] = r;
      out_val[q] = col[r];
    }
    base += __popcll(m);
  }
}

// A <- (A + A^H) / 2 is NOT applied: like LAPACK with uplo, only one triangle is read.
// zero the trailing columns / entries that EigenSerial.f90:10-14 discards (nvals < n)
template <typename T>
__global__ void k_zero_columns(T* __restrict__ dense, int64_t ld, int rows, int c_first, int c_end) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)(c_end - c_first) * rows;
  if (i >= total) return;
  T z{};
  dense[(int64_t)(c_first + i / rows) * ld + i % rows] = z;
}

template <typename T>
DevMat from_dense_t(const T* dense, int64_t ld, int32_t rows, int32_t c0, int32_t cols, double threshold) {
  DevMat R;
  DevBuf<int64_t> cnt((size_t)cols + 1);
  const bool cplx = sizeof(T) == 16;
  if (cols == 0) {
    R.reset_empty(rows, cols, cplx);
    return R;
  }
  hipLaunchKernelGGL((k_dense_count<T>), dim3(cdiv((int64_t)cols * WAVE, 256)), dim3(256), 0, stream(), dense, ld, rows,
                     c0, cols, threshold, cnt.p);
  DevBuf<int64_t> outer((size_t)cols + 1);
  const int64_t nnz = exclusive_scan_i64(cnt.p, outer.p, cols);
  R.alloc(rows, cols, cplx, nnz);
  HIP_CHECK(hipMemcpyAsync(R.outer.p, outer.p, sizeof(int64_t) * ((size_t)cols + 1), hipMemcpyDeviceToDevice, stream()));
  if (nnz)
    hipLaunchKernelGGL((k_dense_write<T>), dim3(cdiv((int64_t)cols * WAVE, 256)), dim3(256), 0, stream(), dense, ld, rows,
                       c0, cols, threshold, R.outer.p, R.inner.p, reinterpret_cast<T*>(R.val.p));
  return R;
}

// ---- Hermitian eigenproblem: cyclic two-sided Jacobi, round-robin ordering (m/2 disjoint pairs per round, m - 1
// rounds per sweep).  For a pair (p, q) with a_pq = |a_pq| e^{i phi}: J = diag(1, e^{-i phi}) [[c, s], [-s, c]],
// tau = (a_qq - a_pp) / (2 |a_pq|), t = sign(tau) / (|tau| + sqrt(1 + tau^2)), c = 1 / sqrt(1 + t^2), s = t c
// (Golub & Van Loan 8.5; real matrices: e^{-i phi} = sign(a_pq)).  A <- J^H A J, V <- V J.
__device__ inline double re_of(double v) { return v; }
__device__ inline double re_of(double2 v) { return v.x; }
__device__ inline double cj(double v) { return v; }
__device__ inline double2 cj(double2 v) { return make_double2(v.x, -v.y); }
__device__ inline double mul_t(double a, double b) { return a * b; }
__device__ inline double2 mul_t(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ inline double scl(double a, double f) { return a * f; }
__device__ inline double2 scl(double2 a, double f) { return make_double2(a.x * f, a.y * f); }
__device__ inline double add_t(double a, double b) { return a + b; }
__device__ inline double2 add_t(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double sub_t(double a, double b) { return a - b; }
__device__ inline double2 sub_t(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double one_t(double) { return 1.0; }
__device__ inline double2 one_t(double2) { return make_double2(1.0, 0.0); }

template <typename T>
struct JacobiRot {
  double c, s;
  T ph;  // e^{-i phi}
  int p, q, active;
};

// pair k of round r among m players (m even): player m-1 stays, the others rotate
__device__ inline void jacobi_pair(int m, int r, int k, int* p, int* q) {
  int a, b;
  if (k == 0) {
    a = m - 1;
    b = r;
  } else {
    a = (r + k) % (m - 1);
    b = (r - k + (m - 1)) % (m - 1);
  }
  *p = min(a, b);
  *q = max(a, b);
}

// one block per pair: rotation from (a_pp, a_qq, a_pq), then columns p, q of A and of V
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_columns(T* __restrict__ A, T* __restrict__ V, int n, int m, int round,
                                                        double floor_abs, JacobiRot<T>* __restrict__ rots,
                                                        double* __restrict__ mass) {
  __shared__ JacobiRot<T> rs;
  const int k = blockIdx.x;
  if (threadIdx.x == 0) {
    int p, q;
    jacobi_pair(m, round, k, &p, &q);
    JacobiRot<T> r;
    r.p = p;
    r.q = q;
    r.active = 0;
    r.c = 1.0;
    r.s = 0.0;
    r.ph = one_t(T{});
    if (q < n) {
      const T apq = A[(int64_t)q * n + p];
      const double app = re_of(A[(int64_t)p * n + p]), aqq = re_of(A[(int64_t)q * n + q]);
      const double g = mag_of(apq);
      if (g > floor_abs && g > 1e-16 * sqrt(fabs(app) * fabs(aqq))) {
        const double tau = (aqq - app) / (2.0 * g);
        const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        r.c = 1.0 / sqrt(1.0 + t * t);
        r.s = t * r.c;
        r.ph = scl(cj(apq), 1.0 / g);
        r.active = 1;
        mass[k] += g * g;  // off-diagonal weight removed by this pair slot during the sweep (slot k is this block's)
      }
    }
    rs = r;
    rots[k] = r;
  }
  __syncthreads();
  if (!rs.active) return;
  const int p = rs.p, q = rs.q;
  const double c = rs.c, s = rs.s;
  const T ph = rs.ph;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    {
      const T xp = A[(int64_t)p * n + i], xq = mul_t(A[(int64_t)q * n + i], ph);
      A[(int64_t)p * n + i] = sub_t(scl(xp, c), scl(xq, s));
      A[(int64_t)q * n + i] = add_t(scl(xp, s), scl(xq, c));
    }
    {
      const T xp = V[(int64_t)p * n + i], xq = mul_t(V[(int64_t)q * n + i], ph);
      V[(int64_t)p * n + i] = sub_t(scl(xp, c), scl(xq, s));
      V[(int64_t)q * n + i] = add_t(scl(xp, s), scl(xq, c));
    }
  }
}
// rows p, q of A: row_p' = c row_p - s e^{i phi} row_q, row_q' = s row_p + c e^{i phi} row_q; the rotated pair of
// off-diagonal entries is set to exactly zero and the diagonal made real
template <typename T>
__global__ __launch_bounds__(256) void k_jacobi_rows(T* __restrict__ A, int n, const JacobiRot<T>* __restrict__ rots) {
  const JacobiRot<T> r = rots[blockIdx.x];
  if (!r.active) return;
  const int p = r.p, q = r.q;
  const T phc = cj(r.ph);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const T xp = A[(int64_t)j * n + p], xq = mul_t(A[(int64_t)j * n + q], phc);
    T yp = sub_t(scl(xp, r.c), scl(xq, r.s));
    T yq = add_t(scl(xp, r.s), scl(xq, r.c));
    if (j == q) yp = T{};
    if (j == p) yq = T{};
    A[(int64_t)j * n + p] = yp;
    A[(int64_t)j * n + q] = yq;
  }
}
template <typename T>
__global__ void k_set_identity(T* __restrict__ V, int n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * n) return;
  V[i] = (i / n == i % n) ? one_t(T{}) : T{};
}
template <typename T>
__global__ void k_diag_real(const T* __restrict__ A, int n, double* __restrict__ w) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = re_of(A[(int64_t)i * n + i]);
}
template <typename T>
__global__ void k_permute_columns(const T* __restrict__ V, T* __restrict__ out, int n, const int* __restrict__ perm) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)n * n) return;
  out[i] = V[(int64_t)perm[i / n] * n + i % n];
}
__global__ __launch_bounds__(256) void k_sum_small(const double* __restrict__ v, int n, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) acc += v[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = sh[0];
}
template <typename T>
__global__ void k_frob2(const T* __restrict__ A, int64_t total, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const double g = mag_of(A[i]);
    acc += g * g;
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

template <typename T>
void jacobi_eigh(T* A, int32_t n, double* d_W) {
  const int m = (n + 1) & ~1;  // an odd order gets a bye
  const int npairs = m / 2;
  DevBuf<T> V((size_t)n * (size_t)n);
  DevBuf<JacobiRot<T>> rots((size_t)npairs);
  DevBuf<double> mass((size_t)npairs), part(64);
  const int64_t total = (int64_t)n * n;
  hipLaunchKernelGGL((k_set_identity<T>), dim3(cdiv(total, 256)), dim3(256), 0, stream(), V.p, n);
  hipLaunchKernelGGL((k_frob2<T>), dim3(64), dim3(256), 0, stream(), A, total, part.p);
  double hp[64];
  part.download(hp, 64);
  double fro2 = 0.0;
  for (double x : hp) fro2 += x;
  const double floor_abs = 1e-18 * std::sqrt(fro2) + 1e-300;  // entries this small no longer move an eigenvalue
  // a sweep that removed less off-diagonal weight than (1e-15 ||A||_F)^2 ends the iteration: convergence is
  // quadratic, what is left after that sweep is far below the rounding of the eigenvalues
  const double done2 = 1e-30 * fro2;
  constexpr int kMaxSweeps = 40;
  int sweep = 0;
  for (; sweep < kMaxSweeps && m > 1; ++sweep) {
    mass.zero();
    for (int r = 0; r < m - 1; ++r) {
      hipLaunchKernelGGL((k_jacobi_columns<T>), dim3(npairs), dim3(256), 0, stream(), A, V.p, n, m, r, floor_abs, rots.p,
                         mass.p);
      hipLaunchKernelGGL((k_jacobi_rows<T>), dim3(npairs), dim3(256), 0, stream(), A, n, rots.p);
    }
    hipLaunchKernelGGL(k_sum_small, dim3(1), dim3(256), 0, stream(), mass.p, npairs, part.p);
    double removed = 0.0;
    part.download(&removed, 1);
    if (!(removed > done2)) break;
  }
  if (sweep == kMaxSweeps) NTP_FATAL("dense eigensolver (Jacobi) did not converge in 40 sweeps");
  // ascending eigenvalues, vectors permuted along (LAPACK's order)
  hipLaunchKernelGGL((k_diag_real<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream(), A, n, d_W);
  std::vector<double> w((size_t)n);
  HIP_CHECK(hipMemcpyAsync(w.data(), d_W, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, stream()));
  sync_stream();
  std::vector<int> perm((size_t)n);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return w[(size_t)a] < w[(size_t)b]; });
  std::vector<double> ws((size_t)n);
  for (int i = 0; i < n; ++i) ws[(size_t)i] = w[(size_t)perm[(size_t)i]];
  DevBuf<int> dperm((size_t)n);
  dperm.upload(perm.data(), (size_t)n);
  HIP_CHECK(hipMemcpyAsync(d_W, ws.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL((k_permute_columns<T>), dim3(cdiv(total, 256)), dim3(256), 0, stream(), V.p, A, n, dperm.p);
  sync_stream();
}

// ---- Cholesky (real), column at a time on a dense copy; L is column major n x n, zero initialised.
// Step j of CholeskyDecomposition (LinearSolversModule.F90:232-277): d = sqrt(A(j,j) - sum_k L(j,k)^2),
// L(i,j) = (A(j,i) - sum_k L(i,k) L(j,k)) / d kept if |.| > threshold; sums run over k ascending (pruned
// entries are zeros of the dense copy, adding them is exact).
__global__ void k_chol_column(const double* __restrict__ A, double* __restrict__ L, int n, int j, double threshold) {
  const int i = j + blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double dj = 0.0, s = 0.0;
  for (int k = 0; k < j; ++k) {
    const double ljk = L[(int64_t)k * n + j];
    dj = __dadd_rn(dj, __dmul_rn(ljk, ljk));
    s = __dadd_rn(s, __dmul_rn(L[(int64_t)k * n + i], ljk));
  }
  const double insert = sqrt(A[(int64_t)j * n + j] - dj);
  if (i == j) {
    L[(int64_t)j * n + j] = insert;
    return;
  }
  const double inv = 1.0 / insert;
  const double v = inv * (A[(int64_t)i * n + j] - s);
  if (fabs(v) > threshold) L[(int64_t)j * n + i] = v;
}

// Pivoted variant (AnalysisModule.F90:110-160, GetPivot.f90): state = pivot order `piv`, running diagonal `diag`.
// k_pchol_pivot (one block): position of the largest remaining diagonal (strict >, first position wins, must be
// > 0), swap it to position j, publish (pivot index, sqrt, 1/sqrt) -- or pivot -1 when nothing positive is left.
__global__ __launch_bounds__(1024) void k_pchol_pivot(int* __restrict__ piv, const double* __restrict__ diag, int n, int j,
                                                      double* __restrict__ L, int* __restrict__ cur,
                                                      double* __restrict__ curv) {
  __shared__ double sv[1024];
  __shared__ int sp[1024];
  double best = 0.0;
  int pos = -1;
  for (int q = j + threadIdx.x; q < n; q += blockDim.x) {
    const double d = diag[piv[q]];
    if (d > best) { best = d; pos = q; }
  }
  sv[threadIdx.x] = best;
  sp[threadIdx.x] = pos;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const double b = sv[threadIdx.x + o];
      const int p = sp[threadIdx.x + o];
      const bool take = p >= 0 && (b > sv[threadIdx.x] || (b == sv[threadIdx.x] && (sp[threadIdx.x] < 0 || p < sp[threadIdx.x])));
      if (take) { sv[threadIdx.x] = b; sp[threadIdx.x] = p; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int q = sp[0];
    if (q < 0) {
      cur[0] = -1;
      return;
    }
    const int pi = piv[q];
    piv[q] = piv[j];
    piv[j] = pi;
    const double root = sqrt(sv[0]);
    cur[0] = pi;
    curv[0] = 1.0 / root;
    L[(int64_t)j * n + pi] = root;
  }
}
__global__ void k_pchol_column(const double* __restrict__ A, double* __restrict__ L, const int* __restrict__ piv,
                               double* __restrict__ diag, int n, int j, double threshold, const int* __restrict__ cur,
                               const double* __restrict__ curv) {
  const int q = j + 1 + blockIdx.x * blockDim.x + threadIdx.x;
  const int pj = cur[0];
  if (q >= n || pj < 0) return;
  const int pi = piv[q];
  double s = 0.0;
  for (int k = 0; k < j; ++k) s = __dadd_rn(s, __dmul_rn(L[(int64_t)k * n + pi], L[(int64_t)k * n + pj]));
  const double v = curv[0] * (A[(int64_t)pj * n + pi] - s);
  if (fabs(v) > threshold) L[(int64_t)j * n + pi] = v;
  diag[pi] = diag[pi] - v * v;
}
__global__ void k_diag_of(const double* __restrict__ A, int n, double* __restrict__ diag, int* __restrict__ piv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  diag[i] = A[(int64_t)i * n + i];
  piv[i] = i;
}
}  // namespace

DevMat filter(const DevMat& A, double threshold) {
  DevMat R;
  if (A.nnz == 0 || A.cols == 0) return A.clone();
  DevBuf<int64_t> cnt((size_t)A.cols + 1), outer((size_t)A.cols + 1);
  const dim3 grid(cdiv((int64_t)A.cols * WAVE, 256)), block(256);
  int64_t nnz = 0;
  if (A.cplx) {
    const double2* v = reinterpret_cast<const double2*>(A.val.p);
    hipLaunchKernelGGL((k_filter_count<double2>), grid, block, 0, stream(), A.outer.p, v, A.cols, threshold, cnt.p);
    nnz = exclusive_scan_i64(cnt.p, outer.p, A.cols);
    R.alloc(A.rows, A.cols, true, nnz);
    if (nnz)
      hipLaunchKernelGGL((k_filter_write<double2>), grid, block, 0, stream(), A.outer.p, A.inner.p, v, A.cols, threshold,
                         outer.p, R.inner.p, reinterpret_cast<double2*>(R.val.p));
  } else {
    hipLaunchKernelGGL((k_filter_count<double>), grid, block, 0, stream(), A.outer.p, A.val.p, A.cols, threshold, cnt.p);
    nnz = exclusive_scan_i64(cnt.p, outer.p, A.cols);
    R.alloc(A.rows, A.cols, false, nnz);
    if (nnz)
      hipLaunchKernelGGL((k_filter_write<double>), grid, block, 0, stream(), A.outer.p, A.inner.p, A.val.p, A.cols,
                         threshold, outer.p, R.inner.p, R.val.p);
  }
  HIP_CHECK(hipMemcpyAsync(R.outer.p, outer.p, sizeof(int64_t) * ((size_t)A.cols + 1), hipMemcpyDeviceToDevice, stream()));
  sync_stream();  // cnt / outer are released
  return R;
}

void to_dense(const DevMat& A, double* d_dense, int64_t ld) {
  const size_t w = A.wval();
  HIP_CHECK(hipMemsetAsync(d_dense, 0, sizeof(double) * w * (size_t)ld * (size_t)A.cols, stream()));
  if (A.nnz == 0) return;
  const dim3 grid(cdiv((int64_t)A.cols * WAVE, 256)), block(256);
  if (A.cplx)
    hipLaunchKernelGGL((k_to_dense<double2>), grid, block, 0, stream(), A.outer.p, A.inner.p,
                       reinterpret_cast<const double2*>(A.val.p), A.cols, reinterpret_cast<double2*>(d_dense), ld);
  else
    hipLaunchKernelGGL((k_to_dense<double>), grid, block, 0, stream(), A.outer.p, A.inner.p, A.val.p, A.cols, d_dense, ld);
}

DevMat from_dense(const double* d_dense, int64_t ld, int32_t rows, int32_t c0, int32_t cols, bool cplx, double threshold) {
  if (cplx) return from_dense_t<double2>(reinterpret_cast<const double2*>(d_dense), ld, rows, c0, cols, threshold);
  return from_dense_t<double>(d_dense, ld, rows, c0, cols, threshold);
}

void dense_zero_columns(double* d_dense, int64_t ld, int32_t rows, int32_t c_first, int32_t c_end, bool cplx) {
  if (c_end <= c_first || rows == 0) return;
  const int64_t total = (int64_t)(c_end - c_first) * rows;
  if (cplx)
    hipLaunchKernelGGL((k_zero_columns<double2>), dim3(cdiv(total, 256)), dim3(256), 0, stream(),
                       reinterpret_cast<double2*>(d_dense), ld, rows, c_first, c_end);
  else
    hipLaunchKernelGGL((k_zero_columns<double>), dim3(cdiv(total, 256)), dim3(256), 0, stream(), d_dense, ld, rows,
                       c_first, c_end);
}

// Hermitian eigendecomposition of a dense n x n matrix (column major, both triangles given): on return d_A holds
// the eigenvectors (columns) and d_W the eigenvalues in ascending order -- LAPACK's DSYEVD / ZHEEVD contract,
// which is what DMatrixModule.F90's EigenDecomposition calls.
void dense_eigh(double* d_A, int32_t n, bool cplx, double* d_W) {
  if (n == 0) return;
  if (cplx) jacobi_eigh<double2>(reinterpret_cast<double2*>(d_A), n, d_W);
  else jacobi_eigh<double>(d_A, n, d_W);
}

// L (n x n column major, zeroed here) from the symmetric positive definite dense A; rank < 0: plain Cholesky
// over all columns, otherwise `rank` steps of the pivoted factorisation (row index = original index, column =
// step).  Everything stays on the device; no read-back per step.
void dense_cholesky(const double* d_A, double* d_L, int32_t n, double threshold, int32_t rank) {
  HIP_CHECK(hipMemsetAsync(d_L, 0, sizeof(double) * (size_t)n * (size_t)n, stream()));
  if (n == 0) return;
  if (rank < 0) {
    for (int j = 0; j < n; ++j)
      hipLaunchKernelGGL(k_chol_column, dim3(cdiv(n - j, 256)), dim3(256), 0, stream(), d_A, d_L, n, j, threshold);
    return;
  }
  DevBuf<double> diag((size_t)n), curv(2);
  DevBuf<int> piv((size_t)n), cur(2);
  hipLaunchKernelGGL(k_diag_of, dim3(cdiv(n, 256)), dim3(256), 0, stream(), d_A, n, diag.p, piv.p);
  for (int j = 0; j < std::min(rank, n); ++j) {
    hipLaunchKernelGGL(k_pchol_pivot, dim3(1), dim3(1024), 0, stream(), piv.p, diag.p, n, j, d_L, cur.p, curv.p);
    if (j + 1 < n)
      hipLaunchKernelGGL(k_pchol_column, dim3(cdiv(n - j - 1, 256)), dim3(256), 0, stream(), d_A, d_L, piv.p, diag.p, n, j,
                         threshold, cur.p, curv.p);
  }
  sync_stream();  // diag / piv are released
}

}  // namespace ntp
