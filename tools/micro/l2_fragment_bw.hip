// What the L2 delivers to a CU for the access pattern of the MFMA tile kernel's A operand (csrc/spgemm_tile.hip): one
// global_load_dwordx4 per lane = 1 KB per wave made of FOUR segments of 256 contiguous bytes (32 rows of four consecutive
// columns whose slots lie 5 KB apart), PF loads in flight per wave, 4 .. 16 waves per CU, every workgroup a "block" of 16
// columns that walks its 21 tiles x 84 k groups, consecutive blocks on one XCD (their operands overlap as a band's do, so the
// XCD's L2 serves most of it).  Compared with the same bytes as ONE contiguous KB per load.  No arithmetic, no LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/l2_fragment_bw.hip -o tools/micro/l2_fragment_bw && tools/micro/l2_fragment_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int TILES = 21, GROUPS = 84, NBLOCKS = 16384;
#ifndef PITCH
#define PITCH 640
#endif

typedef double v4d __attribute__((ext_vector_type(4)));
template <int PF, bool FRAG, int MIS = 0, int MF = 0>
__global__ __launch_bounds__(256) void k_stream(const double* __restrict__ base, int reps, double* out) {
  extern __shared__ double lds[];
  [[maybe_unused]] v4d c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  if (MF) { for (int i = threadIdx.x; i < 4 * GROUPS * 16; i += 256) lds[i] = 1.0; __syncthreads(); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int per = NBLOCKS / 8;
  const int b = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);      // (the kernel's XCD mapping: contiguous ranges per XCD)
  const double* p0 = base + (size_t)b * 16 * PITCH;
  v2d acc = {0.0, 0.0};
  v2d ring[PF];
  const int total = reps * ((TILES + 3 - wave) / 4) * GROUPS;
  auto addr = [&](int i) -> const v2d* {
    const int g = i % GROUPS, t = wave + 4 * ((i / GROUPS) % ((TILES + 3 - wave) / 4));
    if (FRAG) {
      // MIS: the 256-byte segments as a band's slabs have them -- column c's rows shifted by (len - 1) * c elements against the
      // lines: 1 = alternately 0 / 64 bytes (runs of 201), 2 = any multiple of 16 bytes (R = 2 keeps 16-byte alignment ... of
      // the LOAD only if the shift is even; odd shifts make it two 8-byte halves in the kernel too), 3 = any multiple of 8
      const int c = 4 * g + (lane >> 4);
      const int sh = MIS == 0 ? 0 : MIS == 1 ? 8 * (c & 1) : MIS == 2 ? 2 * ((c * 5) & 7) : ((c * 5) & 15);
      return reinterpret_cast<const v2d*>(p0 + (size_t)c * PITCH + 32 * t + 2 * (lane & 15) + sh);
    }
    return reinterpret_cast<const v2d*>(p0 + (size_t)(4 * g) * PITCH + 128 * (t % 5) + 2 * lane);
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) ring[u] = *addr(u);
  int i = PF;
  for (; i + PF <= total; i += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (MF) {   // MF 1: the kernel's slot -- one LDS read of the multiplier, two matrix instructions on the loaded fragment; 2: without the LDS read
        const double bv = MF == 1 ? lds[(((i + u) % GROUPS) * 4 + (lane >> 4)) * 16 + (lane & 15)] : 1.0;
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][0], bv, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][1], bv, c1, 0, 0, 0);
      } else acc += ring[u];
      ring[u] = *addr(i + u);
    }
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) acc += ring[u];
  if (MF) acc += v2d{c0[0] + c0[1] + c0[2] + c0[3], c1[0] + c1[1] + c1[2] + c1[3]};
  if (acc[0] == 1.2345e300) out[0] = acc[1];
}

template <int PF, bool FRAG, int MIS = 0, int MF = 0>
double run(const double* d, int wgs_per_cu, double* out) {
  // (occupancy: dynamic LDS sized so that exactly wgs_per_cu workgroups of four waves fit the 160 KB of a CU)
  const size_t lds = (size_t)(160 * 1024 / wgs_per_cu) - 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stream<PF, FRAG, MIS, MF>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int reps = 8, grid = 256 * wgs_per_cu * 4;     // four rounds of workgroups per CU slot
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((k_stream<PF, FRAG, MIS, MF>), dim3(grid), dim3(256), lds, 0, d, 1, out);
  (void)hipEventRecord(a);
  hipLaunchKernelGGL((k_stream<PF, FRAG, MIS, MF>), dim3(grid), dim3(256), lds, 0, d, reps, out);
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
  double loads = 0;
  for (int w = 0; w < 4; ++w) loads += (double)reps * ((TILES + 3 - w) / 4) * GROUPS;
  const double bytes = (double)grid * loads * 1024.0;
  return bytes / (ms * 1e-3) / 2.4e9 / 256.0;   // bytes per clock per CU at 2.4 GHz
}

int main() {
  const size_t total = ((size_t)NBLOCKS * 16 + 4 * GROUPS + 64) * PITCH + 4096;
  double *d, *out;
  CK(hipMalloc(&d, total * 8)); CK(hipMalloc(&out, 64));
  CK(hipMemset(d, 0, total * 8));
  std::printf("# bytes per clock per CU (wall clock x 2.4 GHz).  FRAG = 4 x 256 B per load (the kernel's A operand), LINE = 1 KB contiguous\n");
  for (int wg = 2; wg <= 6; ++wg) {
    std::printf("waves/CU %2d:  FRAG  pf3 %5.1f  pf6 %5.1f  pf12 %5.1f     LINE  pf3 %5.1f  pf6 %5.1f  pf12 %5.1f\n", 4 * wg,
                run<3, true>(d, wg, out), run<6, true>(d, wg, out), run<12, true>(d, wg, out),
                run<3, false>(d, wg, out), run<6, false>(d, wg, out), run<12, false>(d, wg, out));
  }
  std::printf("# FRAG aligned, 12 waves per CU: loads only / + one LDS read and two f64 matrix instructions per load (the kernel's slot) / matrix instructions without the LDS read\n");
  std::printf("pf3  %5.1f %5.1f %5.1f   pf6  %5.1f %5.1f %5.1f   pf12 %5.1f %5.1f %5.1f\n",
              run<3, true, 0, 0>(d, 3, out), run<3, true, 0, 1>(d, 3, out), run<3, true, 0, 2>(d, 3, out),
              run<6, true, 0, 0>(d, 3, out), run<6, true, 0, 1>(d, 3, out), run<6, true, 0, 2>(d, 3, out),
              run<12, true, 0, 0>(d, 3, out), run<12, true, 0, 1>(d, 3, out), run<12, true, 0, 2>(d, 3, out));
  std::printf("# FRAG with the segments shifted against the 128-byte lines: aligned / 0|64 B alternating / multiples of 16 B / multiples of 8 B (two 8-byte halves per lane)\n");
  for (int wg = 2; wg <= 4; ++wg) {
    std::printf("waves/CU %2d:  pf6  %5.1f %5.1f %5.1f %5.1f    pf12 %5.1f %5.1f %5.1f %5.1f\n", 4 * wg,
                run<6, true, 0>(d, wg, out), run<6, true, 1>(d, wg, out), run<6, true, 2>(d, wg, out), run<6, true, 3>(d, wg, out),
                run<12, true, 0>(d, wg, out), run<12, true, 1>(d, wg, out), run<12, true, 2>(d, wg, out), run<12, true, 3>(d, wg, out));
  }
  return 0;
}
