// Does v_mfma_f64_16x16x4_f64 share its execution unit with ordinary VALU work?  Per iteration one MFMA plus NV independent
// VALU instructions (32-bit integer adds / 64-bit fp adds), one wave per SIMD; and MFMA-only waves next to VALU-only waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NV, int KIND>
__global__ void k_mix(double* out, int iters, long long* cyc) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d acc = {0, 0, 0, 0};
  int x[8] = {l, l + 1, l + 2, l + 3, l + 4, l + 5, l + 6, l + 7};
  double y[8] = {1.0 * l, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i % 8]) : "v"(l));
      else asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[i % 8]) : "v"(a));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 8; ++i) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// waves 0-3 of a block: MFMA only; waves 4-7: VALU only (one of each per SIMD)
template <int KIND>
__global__ void k_pair(double* out, int iters, long long* cyc) {
  const int l = threadIdx.x & 63, w = threadIdx.x / 64;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d acc = {0, 0, 0, 0};
  int x[8] = {l, l + 1, l + 2, l + 3, l + 4, l + 5, l + 6, l + 7};
  double y[8] = {1.0 * l, 2.0, 3.0, 4.0, 5.0, 6.0, 7.0, 8.0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (w < 4) {
    for (int it = 0; it < iters; ++it) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i % 8]) : "v"(l));
        else asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[i % 8]) : "v"(a));
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = acc[0] + acc[1] + acc[2] + acc[3];
  for (int i = 0; i < 8; ++i) s += x[i] + y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x == 0 || threadIdx.x == 256) && blockIdx.x == 0) cyc[w / 4] = t1 - t0;
}
int main() {
  double* dout; long long* dc;
  hipMalloc(&dout, 256 * 512 * 8); hipMalloc(&dc, 16);
  const int iters = 20000;
  auto run = [&](auto kern, int threads, const char* name) {
    long long c[2] = {0, 0};
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, dout, 100, dc);
    hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, dout, iters, dc);
    hipDeviceSynchronize();
    hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    printf("%-60s cycles/iter: %.1f  (second group %.1f)\n", name, (double)c[0] / iters, (double)c[1] / iters);
  };
  run(k_mix<0, 0>, 256, "1 MFMA");
  run(k_mix<4, 0>, 256, "1 MFMA + 4 v_add_u32");
  run(k_mix<8, 0>, 256, "1 MFMA + 8 v_add_u32");
  run(k_mix<16, 0>, 256, "1 MFMA + 16 v_add_u32");
  run(k_mix<32, 0>, 256, "1 MFMA + 32 v_add_u32");
  run(k_mix<4, 1>, 256, "1 MFMA + 4 v_add_f64");
  run(k_mix<8, 1>, 256, "1 MFMA + 8 v_add_f64");
  run(k_mix<16, 1>, 256, "1 MFMA + 16 v_add_f64");
  run(k_pair<0>, 512, "MFMA waves | 16 v_add_u32 waves (same SIMDs)");
  run(k_pair<1>, 512, "MFMA waves | 16 v_add_f64 waves (same SIMDs)");
  return 0;
}
