// Two vocabulary operations of the solver loops on compressed columns (real and complex) without the merge pass:
//   B <- B + alpha I   (IncrementMatrix(Identity, B, alpha); SignSolversModule.F90:150-258, SquareRootSolversModule.F90:342-531)
//                      in place when every column of B stores its diagonal entry -- then the merge of AddSparseVectors.f90
//                      changes one value per column and nothing else
//   || alpha A + B ||  (the convergence norm of a loop that forms the difference of two iterates only to take its norm:
//                      IncrementMatrix + MatrixNorm) from a dense LDS window per column, without forming the difference
// Same element arithmetic as the merge kernel (kernels.hip inc_decide: alpha * a rounded, then added to b); column sums in
// a different (fixed) order than MatrixNorm's pass over the merged columns: norms agree to 1e-15 relative.
#include <hip/hip_runtime.h>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

// column j (global column c0 + j): position of the diagonal entry and its new value; flag bit 0: a column without
// diagonal entry, bit 1: a stored zero above the diagonal (the merge would drop it: AddSparseVectors thresholds what lies
// inside the other vector's range), bit 2: the new diagonal value is zero (the merge would drop it)
template <typename T>
__global__ __launch_bounds__(256) void k_diag_find(Csc B, int c0, double alpha, int64_t* __restrict__ pos, T* __restrict__ newval, int* __restrict__ flag) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= B.cols) return;
  const int lane = lane_id();
  const int d = c0 + j;
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  const int64_t s = B.outer[j], e = B.outer[j + 1];
  bool found = false, bad = false;
  for (int64_t p = s + lane; p < e; p += WAVE) {
    const int r = B.inner[p];
    const T v = Bv[p];
    if (r == d) {
      T one = Sc<T>::zero();
      reinterpret_cast<double*>(&one)[0] = 1.0;
      const T nv = Sc<T>::add(Sc<T>::scale(alpha, one), v);
      pos[j] = p;
      newval[j] = nv;
      found = true;
      if (Sc<T>::is_zero(nv) || !(Sc<T>::mag(nv) > 0.0)) bad = true;
    } else if (r < d && !(Sc<T>::mag(v) > 0.0)) {
      bad = true;
    }
  }
  const bool any_found = __ballot(found) != 0ull;
  const bool any_bad = __ballot(bad) != 0ull;
  if (lane == 0) {
    if (!any_found) atomicOr(flag, 1);
    if (any_bad) atomicOr(flag, 2);
  }
}
template <typename T>
__global__ void k_diag_apply(int cols, const int64_t* __restrict__ pos, const T* __restrict__ newval, T* __restrict__ val) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < cols) val[pos[j]] = newval[j];
}

constexpr int NORM_NW = 4;
// one wave per column: window over the union of the two columns' row extents; sums[j] = sum over the rows of |alpha a + b|
template <typename T>
__global__ __launch_bounds__(NORM_NW* WAVE) void k_norm_axpy(Csc A, Csc B, double alpha, int wmax, double* __restrict__ sums, int* __restrict__ flag, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int blk = xcd_block(nblocks);
  if (blk < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = blk * NORM_NW + wave;
  if (j >= A.cols) return;
  T* win = reinterpret_cast<T*>(smem) + (size_t)wave * wmax;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  const int64_t as = A.outer[j], ae = A.outer[j + 1], bs = B.outer[j], be = B.outer[j + 1];
  int lo = INT_MAX, hi = -1;
  if (ae > as) { lo = min(lo, A.inner[as]); hi = max(hi, A.inner[ae - 1]); }
  if (be > bs) { lo = min(lo, B.inner[bs]); hi = max(hi, B.inner[be - 1]); }
  if (hi < lo) {
    if (lane == 0) sums[j] = 0.0;
    return;
  }
  const int ext = hi - lo + 1;
  if (ext > wmax) {   // (not this kernel's column: the caller forms the difference)
    if (lane == 0) atomicOr(flag, 1);
    return;
  }
  for (int s = lane; s < ext; s += WAVE) win[s] = Sc<T>::zero();
  __builtin_amdgcn_wave_barrier();
  for (int64_t p = as + lane; p < ae; p += WAVE) win[A.inner[p] - lo] = Sc<T>::scale(alpha, Av[p]);
  __builtin_amdgcn_wave_barrier();
  for (int64_t p = bs + lane; p < be; p += WAVE) {
    const int r = B.inner[p] - lo;
    win[r] = Sc<T>::add(win[r], Bv[p]);
  }
  __builtin_amdgcn_wave_barrier();
  double sum = 0.0;
  for (int s = lane; s < ext; s += WAVE) sum = __dadd_rn(sum, Sc<T>::mag(win[s]));
  sum = wave_sum_f64(sum);
  if (lane == 0) sums[j] = sum;
}

}  // namespace

// B <- B + alpha I in place (local columns c0 .. c0 + B.cols of the global matrix); false: B untouched, the caller merges
bool add_identity_inplace(DevMat& B, double alpha, int32_t c0) {
  if (B.loose() || B.expanded() || B.blocked() || B.cols == 0 || B.nnz < B.cols) return false;
  const int n = B.cols;
  DevBuf<int64_t> pos((size_t)n);
  DevBuf<double> newval((size_t)n * B.wval());
  DevBuf<int> flag(2);
  flag.zero();
  dispatch_type(B.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_diag_find<T>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(B), (int)c0, alpha, pos.p,
                       reinterpret_cast<T*>(newval.p), flag.p);
  });
  long long hf = 0;
  {
    ScalarFetch f;
    f.add(flag.p, 1, &hf);
    f.run();
  }
  if ((int)(hf & 0xffffffffll) != 0) return false;
  bump_matrix_value_epoch();
  dispatch_type(B.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_diag_apply<T>), dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, pos.p, reinterpret_cast<const T*>(newval.p),
                       reinterpret_cast<T*>(B.val.p));
  });
  return true;
}

// max over the local columns of sum |alpha a + b| (MatrixNorm of B + alpha A, IncrementMatrix rules, threshold 0);
// false: not taken
bool norm_axpy_columns(const DevMat& A, const DevMat& B, double alpha, double* norm) {
  if (A.loose() || B.loose() || A.expanded() || B.expanded() || A.blocked() || B.blocked()) return false;
  if (A.cplx != B.cplx || A.cols != B.cols || A.rows != B.rows || A.cols == 0) return false;
  const int n = A.cols;
  const size_t esz = A.cplx ? 16 : 8;
  const int wmax = (int)(64 * 1024 / (NORM_NW * esz));   // 1024 complex / 2048 real rows per column
  DevBuf<double> sums((size_t)n);
  DevBuf<int> flag(2);
  flag.zero();
  const int nblocks = cdiv(n, NORM_NW);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_norm_axpy<T>), dim3(xcd_grid(nblocks)), dim3(NORM_NW * WAVE), (size_t)wmax * esz * NORM_NW, stream(), view(A), view(B),
                       alpha, wmax, sums.p, flag.p, nblocks);
  });
  const double v = max_of(sums, (size_t)n);   // (synchronises: the flag is read behind it)
  int hf[2] = {0, 0};
  flag.download(hf, 2);
  if (hf[0] != 0) return false;
  *norm = v;
  return true;
}

}  // namespace ntp
