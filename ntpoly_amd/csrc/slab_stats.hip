// Small companions of the slab algebra (kernels.hip, last section).
// (1) Statistics that are only computed when the kernel timers are on: the intermediate products of C = A B for
//     operands in slab form -- sum over the entries B(k, j) of the entries of A(:, k) -- which the compressed-column
//     paths get from their plans (products per second).
// (2) IncrementMatrix(Identity, B, alpha) with threshold 0 on a slab-form B whose diagonal lies inside its runs: one
//     value per column changes, in place, instead of a merge pass over the whole matrix (AddSparseVectors rules for the
//     one row both columns can share: both present -> alpha + b kept unless exactly zero; B has a hole there -> alpha).
#include <hip/hip_runtime.h>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {
__global__ __launch_bounds__(256) void k_sa_products(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                     const int64_t* __restrict__ off, const double* __restrict__ val,
                                                     const int32_t* __restrict__ acount, int acols, unsigned long long* __restrict__ out) {
  __shared__ long long red[4];
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  long long p = 0;
  if (j < n) {
    const int f = first[j], l = last[j];
    if (l >= f) {
      const double* __restrict__ v = val + (off[j] - f);
      for (int k = f + lane_id(); k <= l; k += WAVE)
        if (v[k] != 0.0 && k < acols) p += acount[k];
    }
  }
  p = wave_sum_i64(p);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(out, (unsigned long long)t);
  }
}
}  // namespace

namespace {
// pass 1 (apply = 0): what would happen, without touching anything -- st[0] |= 1: a diagonal outside its column's run
// (the run would have to grow), |= 2: a diagonal that cancels at the end of its run (the run would have to shrink);
// st[1] += change of the entry count.  pass 2 (apply = 1): the values and the per-column counts.
__global__ __launch_bounds__(256) void k_sa_add_diagonal(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                         const int64_t* __restrict__ off, double* __restrict__ val,
                                                         int32_t* __restrict__ count, int col_offset, double alpha, int apply,
                                                         unsigned long long* __restrict__ st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = 0, delta = 0;
  if (j < n) {
    const int d = j + col_offset, f = first[j], l = last[j];
    if (l < f || d < f || d > l) {
      flag = 1;
    } else {
      double* p = val + (off[j] + (d - f));
      const double old = *p;
      const bool hb = old != 0.0;
      const double s = hb ? __dadd_rn(alpha, old) : alpha;
      const bool keep = fabs(s) > 0.0;
      if (!keep && (d == f || d == l)) flag = 2;
      delta = (keep ? 1 : 0) - (hb ? 1 : 0);
      if (apply) {
        *p = keep ? s : 0.0;
        count[j] += delta;
      }
    }
  }
  if (!apply) {
    const unsigned long long any = __ballot(flag != 0);
    if (any) {
      int fl = flag;
      for (int o = 32; o > 0; o >>= 1) fl |= __shfl_xor(fl, o, WAVE);
      if (lane_id() == 0) atomicOr(st, (unsigned long long)fl);
    }
    const long long dsum = wave_sum_i64(delta);
    if (lane_id() == 0 && dsum) atomicAdd(st + 1, (unsigned long long)dsum);
  }
}
}  // namespace

// B <- B + alpha I (IncrementMatrix(Identity, B, alpha, 0)); false: not done (B untouched) -- the caller merges
bool slab_add_diagonal(DevMat& B, double alpha, int32_t col_offset) {
  if (!B.expanded() || B.cplx || B.rows != B.cols || B.slab->labelled() || B.slab->origin || B.zero_free != 1 || alpha == 0.0) return false;
  SlabForm& f = *B.slab;
  const int n = B.cols;
  DevBuf<unsigned long long> st(2);
  st.zero();
  hipLaunchKernelGGL(k_sa_add_diagonal, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, f.val.p, f.count.p,
                     col_offset, alpha, 0, st.p);
  unsigned long long h[2] = {0, 0};
  {
    ScalarFetch ft;
    ft.add(st.p, 2, h);
    ft.run();
  }
  if (h[0] != 0) return false;
  hipLaunchKernelGGL(k_sa_add_diagonal, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, f.val.p, f.count.p,
                     col_offset, alpha, 1, st.p);
  B.nnz += (long long)h[1];
  f.tiles.release();      // (the multiplier tiles held the old diagonal; the next step's plan depends on the extents only)
  f.tile_off.release();
  return true;
}

long long slab_product_count(const DevMat& A, const DevMat& B) {
  if (!A.expanded() || !B.expanded() || A.cplx || B.cplx) return 0;
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  DevBuf<unsigned long long> acc(1);
  acc.zero();
  hipLaunchKernelGGL(k_sa_products, dim3(cdiv((int64_t)B.cols * WAVE, 256)), dim3(256), 0, stream(), B.cols, fb.first.p, fb.last.p,
                     fb.off.p, fb.val.p, fa.count.p, A.cols, acc.p);
  unsigned long long h = 0;
  ScalarFetch f;
  f.add(acc.p, 1, &h);
  f.run();
  return (long long)h;
}

}  // namespace ntp
