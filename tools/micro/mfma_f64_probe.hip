// v_mfma_f64_16x16x4_f64 on gfx950: (1) fragment layout, (2) is D = C + sum_k A(i,k) B(k,j) rounded like a chain of
// fma() in ascending k?  (3) issue rate with independent / dependent accumulators.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip && ./mfma_f64_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_probe(const double* A, const double* B, const double* C, double* D) {
  // A: 16 x 4 row-major, B: 4 x 16 row-major, C/D: 16 x 16 row-major
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];
  const double b = B[(l / 16) * 16 + (l % 16)];
  v4d c;
  for (int v = 0; v < 4; ++v) c[v] = C[(4 * v + l / 16) * 16 + (l % 16)];
  v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int v = 0; v < 4; ++v) D[(4 * v + l / 16) * 16 + (l % 16)] = d[v];
}

template <int NACC>
__global__ void k_rate(double* out, int iters, long long* cycles) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = v4d{0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

// FMA vector rate for comparison: 16 independent chains
__global__ void k_fma_rate(double* out, int iters, long long* cycles) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l;
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, acc[i], 1e-30);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  std::mt19937_64 rng(7);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  std::vector<double> A(64), B(64), C(256), D(256);
  double *dA, *dB, *dC, *dD;
  CK(hipMalloc(&dA, 64 * 8)); CK(hipMalloc(&dB, 64 * 8)); CK(hipMalloc(&dC, 256 * 8)); CK(hipMalloc(&dD, 256 * 8));
  int bad_layout = 0, bad_fma = 0, bad_fma_rev = 0, bad_unfused = 0, total = 0;
  for (int trial = 0; trial < 200; ++trial) {
    for (auto& x : A) x = U(rng) * std::pow(2.0, (int)(U(rng) * 20));
    for (auto& x : B) x = U(rng) * std::pow(2.0, (int)(U(rng) * 20));
    for (auto& x : C) x = U(rng) * std::pow(2.0, (int)(U(rng) * 20));
    if (trial % 5 == 0) for (int i = 0; i < 64; i += 3) A[i] = 0.0;   // zero padding as the kernel will have it
    CK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dC, C.data(), 256 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    CK(hipMemcpy(D.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double f = C[i * 16 + j], fr = C[i * 16 + j], u = C[i * 16 + j];
        long double ex = C[i * 16 + j];
        for (int k = 0; k < 4; ++k) {
          f = std::fma(A[i * 4 + k], B[k * 16 + j], f);
          fr = std::fma(A[i * 4 + 3 - k], B[(3 - k) * 16 + j], fr);
          volatile double p = A[i * 4 + k] * B[k * 16 + j];
          u = u + p;
          ex += (long double)A[i * 4 + k] * B[k * 16 + j];
        }
        const double d = D[i * 16 + j];
        ++total;
        if (std::fabs(d - (double)ex) > 1e-12 * (std::fabs((double)ex) + 1e-300) + 1e-9) ++bad_layout;
        if (std::memcmp(&d, &f, 8)) ++bad_fma;
        if (std::memcmp(&d, &fr, 8)) ++bad_fma_rev;
        if (std::memcmp(&d, &u, 8)) ++bad_unfused;
      }
  }
  printf("elements %d: layout mismatches %d; != fma chain (k ascending) %d; != fma chain (k descending) %d; != unfused %d\n",
         total, bad_layout, bad_fma, bad_fma_rev, bad_unfused);

  double* dout;
  long long* dcyc;
  CK(hipMalloc(&dout, 1024 * 256 * 8));
  CK(hipMalloc(&dcyc, 8));
  const int iters = 20000;
  auto run = [&](auto kern, int blocks, int threads, const char* name, double mfma_per_iter) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, 100, dcyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, dout, iters, dcyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long cyc = 0;
    hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks %4d x %4d thr: %.3f ms, memtime ticks/iter %.1f (100 MHz ticks -> ns*0.1), per op %.2f ns\n", name, blocks, threads, ms,
           (double)cyc / iters, (double)ms * 1e6 / iters / mfma_per_iter);
  };
  // one wave per SIMD on every CU: 256 blocks of 256 threads
  run(k_rate<1>, 256, 256, "mfma f64 16x16x4, 1 dependent acc, 1 wave/SIMD", 1);
  run(k_rate<2>, 256, 256, "mfma f64 16x16x4, 2 acc, 1 wave/SIMD", 2);
  run(k_rate<4>, 256, 256, "mfma f64 16x16x4, 4 acc, 1 wave/SIMD", 4);
  run(k_rate<1>, 512, 256, "mfma f64 16x16x4, 1 acc, 2 waves/SIMD", 1);
  run(k_rate<1>, 1024, 256, "mfma f64 16x16x4, 1 acc, 4 waves/SIMD", 1);
  run(k_fma_rate, 256, 256, "v_fma_f64 x16 chains, 1 wave/SIMD", 16);
  run(k_fma_rate, 1024, 256, "v_fma_f64 x16 chains, 4 waves/SIMD", 16);
  return 0;
}
