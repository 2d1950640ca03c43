// Rate and layout of v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 products per instruction) against v_mfma_f64_16x16x4_f64
// on gfx950: cycles per instruction with independent and with dependent accumulators (one wave per SIMD), and which lane holds
// which element of A, B and D (probed with unit vectors).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KIND, int NACC>
__global__ void k_rate(double* out, int iters, long long* cyc) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d acc[NACC];
  double d[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = v4d{0, 0, 0, 0}; d[i] = 0.0; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else d[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d[i], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + d[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// layout probe: A = indicator of lane la, B = indicator of lane lb -> which lanes of D become 1
__global__ void k_layout(int la, int lb, double* dout) {
  const int l = threadIdx.x;
  const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  dout[l] = d;
}
int main() {
  double* dout; long long* dc;
  hipMalloc(&dout, 256 * 512 * 8); hipMalloc(&dc, 16);
  const int iters = 20000;
  auto run = [&](auto kern, int nacc, const char* name) {
    long long c = 0;
    hipLaunchKernelGGL(kern, dim3(1), dim3(256), 0, 0, dout, iters, dc);
    hipDeviceSynchronize();
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %.1f cycles (s_memtime ticks x 24: 100 MHz -> 2.4 GHz) per instruction\n", name, (double)c * 24.0 / iters / nacc);
  };
  run(k_rate<0, 1>, 1, "16x16x4, 1 accumulator (dependent chain)");
  run(k_rate<0, 4>, 4, "16x16x4, 4 accumulators");
  run(k_rate<1, 1>, 1, "4x4x4 x 4 blocks, 1 accumulator (dependent)");
  run(k_rate<1, 4>, 4, "4x4x4 x 4 blocks, 4 accumulators");
  run(k_rate<1, 8>, 8, "4x4x4 x 4 blocks, 8 accumulators");
  // layout: for a few (la, lb) print the lanes of D that are non-zero
  std::vector<double> h(64);
  const int probes[][2] = {{0, 0}, {1, 0}, {0, 1}, {4, 0}, {0, 4}, {16, 16}, {17, 16}, {16, 20}, {5, 1}, {21, 17}, {0, 16}, {48, 49}};
  for (auto& p : probes) {
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, p[0], p[1], dout);
    hipMemcpy(h.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
    printf("A lane %2d, B lane %2d -> D lanes:", p[0], p[1]);
    for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf(" %d", l);
    printf("\n");
  }
  return 0;
}
