// Microbenchmark: throughput of LDS f64 accumulate forms on gfx950 (cycles per wave-instruction).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/micro/lds_atomic_bench.hip -o lds_bench && ./lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: ds_add_f64 (no return); 1: ds_read_b64 + v_add_f64 + ds_write_b64; 2: ds_add_f64 + v_mul_f64 + v_sub
// PAT 0: 64 contiguous rows, 1: stride 2, 2: scattered (multiplicative hash)
template <int MODE, int PAT>
__global__ void k(double* out, long long* cycles, int iters, int window) {
  extern __shared__ double acc[];
  const int wave = threadIdx.x / 64, lane = threadIdx.x & 63;
  double* a = acc + wave * window;
  for (int s = lane; s < window; s += 64) a[s] = 0.0;
  __syncthreads();
  const unsigned mask = window - 1;
  unsigned s0[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (PAT == 0) s0[u] = (lane + u * 64) & mask;
    else if (PAT == 1) s0[u] = (lane * 2 + u * 128 + (u & 1)) & mask;
    else s0[u] = (((lane + 64 * u) * 2654435761u) >> 9) & mask;
  }
  double v = 1.0 + lane;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned s = (s0[u] + it) & mask;
      if (MODE == 0) atomicAdd(&a[s], v);
      else if (MODE == 2) atomicAdd(&a[s], __dmul_rn(v, 1.0000001 + u));
      else { a[s] = __dadd_rn(a[s], v); __builtin_amdgcn_wave_barrier(); }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  __syncthreads();
  double sum = 0;
  for (int s = lane; s < window; s += 64) sum += a[s];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (lane == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int MODE, int PAT>
int run(int waves, int blocks_per_cu, double* out, long long* cyc) {
  const int blocks = 256 * blocks_per_cu, iters = 4000, window = 1024;
  std::vector<long long> h(blocks * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)waves * window * sizeof(double);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(blocks), dim3(64 * waves), lds, 0, out, cyc, iters, window);
    hipEventRecord(e1);
    CHECK(hipEventSynchronize(e1));
    hipEventElapsedTime(&ms, e0, e1);
  }
  CHECK(hipMemcpy(h.data(), cyc, blocks * 16 * sizeof(long long), hipMemcpyDeviceToHost));
  double avg = 0; for (int b = 0; b < blocks; ++b) avg += h[b * 16]; avg /= blocks;
  const double ops = (double)blocks * waves * iters * 8;
  printf("mode %d pat %d waves/CU %2d: %.3f ms | %.1f memtime-ticks per wave-op | %.2f wave-ops/us/CU | %.2e elem/s chip\n", MODE, PAT,
         waves * blocks_per_cu, ms, avg / (iters * 8.0), ops / (ms * 1e3) / 256, ops * 64 / (ms * 1e-3));
  return 0;
}

int main() {
  double* out; long long* cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 1024 * sizeof(double)));
  CHECK(hipMalloc(&cyc, 256 * 8 * 16 * sizeof(long long)));
  for (int cfg = 0; cfg < 4; ++cfg) {
    const int waves = cfg == 0 ? 1 : 4, bpc = cfg == 0 ? 1 : cfg == 1 ? 1 : cfg == 2 ? 2 : 4;
    run<0, 0>(waves, bpc, out, cyc); run<0, 1>(waves, bpc, out, cyc); run<0, 2>(waves, bpc, out, cyc);
    run<1, 0>(waves, bpc, out, cyc); run<1, 2>(waves, bpc, out, cyc);
    run<2, 0>(waves, bpc, out, cyc); run<2, 2>(waves, bpc, out, cyc);
  }
  return 0;
}
