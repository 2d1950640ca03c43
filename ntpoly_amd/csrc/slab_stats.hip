// Statistics of the slab algebra (kernels.hip, last section) that are only computed when the kernel timers are on:
// the intermediate products of C = A B for operands in slab form -- sum over the entries B(k, j) of the entries of
// A(:, k) -- which the compressed-column paths get from their plans (products per second).
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
