// Hand-written gfx950 (CDNA4, wave64) kernels of the NTPoly hot path.
//
// Numerical contract (DESIGN.md "Parity"): every C(i,j) of the SpGEMM is accumulated in
// ascending-k order with an UNFUSED multiply and add (two roundings), exactly as the reference's
// MultiplyBlock loop does on baseline x86-64 (sparse_includes/MultiplyBlock.f90:9-36), and is
// pruned with PruneList's strict `|alpha*v| > threshold` (sparse_includes/PruneList.f90:22).
// One wavefront owns one output column; lanes stride over the entries of one operand column at a
// time, so no two lanes ever touch the same accumulator slot inside one step and no atomics are
// needed.  The result is bit-identical to the reference's sparse branch and independent of the
// GPU count.  This file must be compiled with -ffp-contract=off.
#include "kernels.hpp"
#include "device_util.hpp"
#include "spgemm_grouped.hpp"
#include "slab_types.hpp"
#include "spgemm_tile.hpp"
#include "spgemm_block.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <climits>
#include <cmath>
#include <cstring>
#include <type_traits>
#include <numeric>
#include <utility>

#include <rocprim/device/device_radix_sort.hpp>

namespace ntp {

namespace {

// ------------------------------------------------------------------ scans / reductions
// single-workgroup exclusive scan of n int64 values (n <= a few million); out[n] = total
__global__ __launch_bounds__(1024) void k_scan_excl_i64(const int64_t* __restrict__ in,
                                                        int64_t* __restrict__ out, int64_t n) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = min(n, (int64_t)t * chunk), e = min(n, b + chunk);
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = (t == 0) ? 0 : part[t - 1];
  for (int64_t i = b; i < e; ++i) {
    const int64_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 1023) out[n] = part[1023];
}

// int32 counts -> int64 exclusive offsets (same structure)
__global__ __launch_bounds__(1024) void k_scan_excl_i32(const int32_t* __restrict__ in,
                                                        int64_t* __restrict__ out, int64_t n) {
  __shared__ int64_t part[1024];
  const int t = threadIdx.x;
  const int64_t chunk = (n + 1023) / 1024;
  const int64_t b = min(n, (int64_t)t * chunk), e = min(n, b + chunk);
  int64_t s = 0;
  for (int64_t i = b; i < e; ++i) s += in[i];
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int64_t v = (t >= off) ? part[t - off] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int64_t run = (t == 0) ? 0 : part[t - 1];
  for (int64_t i = b; i < e; ++i) {
    const int64_t v = in[i];
    out[i] = run;
    run += v;
  }
  if (t == 1023) out[n] = part[1023];
}

// three-kernel scan for long arrays: per-block sums, single-block scan of the sums, per-block rescan
template <typename TIn>
__global__ __launch_bounds__(256) void k_scan_block_sums(const TIn* __restrict__ in, int64_t n, int64_t* __restrict__ sums) {
  __shared__ int64_t ws[4];
  const int64_t base = (int64_t)blockIdx.x * 2048;
  int64_t s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int64_t i = base + u * 256 + threadIdx.x;
    if (i < n) s += (int64_t)in[i];
  }
  s = wave_sum_i64(s);
  if (lane_id() == 0) ws[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
template <typename TIn>
__global__ __launch_bounds__(256) void k_scan_block_apply(const TIn* __restrict__ in, int64_t n,
                                                          const int64_t* __restrict__ block_off, int64_t* __restrict__ out) {
  // each thread owns 8 consecutive elements; block-level exclusive scan of the per-thread sums
  __shared__ int64_t part[256];
  const int64_t base = (int64_t)blockIdx.x * 2048 + (int64_t)threadIdx.x * 8;
  int64_t v[8];
  int64_t s = 0;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    v[u] = (base + u < n) ? (int64_t)in[base + u] : 0;
    s += v[u];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int64_t t = ((int)threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  int64_t run = block_off[blockIdx.x] + (threadIdx.x ? part[threadIdx.x - 1] : 0);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (base + u < n) out[base + u] = run;
    run += v[u];
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = block_off[gridDim.x];
}

// deterministic final reduction of per-block partials (sum of pairs / min / max)
// block g reduces pairs [g*chunk, min(n, (g+1)*chunk)) in a fixed order -> out[2g], out[2g+1]
__global__ __launch_bounds__(256) void k_reduce_sum2(const double* __restrict__ part, int n,
                                                     double* __restrict__ out, int chunk) {
  __shared__ double sx[256], sy[256];
  double x = 0, y = 0;
  const int i0 = blockIdx.x * chunk, i1 = min(n, i0 + chunk);
  for (int i = i0 + threadIdx.x; i < i1; i += 256) {
    x = __dadd_rn(x, part[2 * i]);
    y = __dadd_rn(y, part[2 * i + 1]);
  }
  sx[threadIdx.x] = x;
  sy[threadIdx.x] = y;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sx[threadIdx.x] = __dadd_rn(sx[threadIdx.x], sx[threadIdx.x + o]);
      sy[threadIdx.x] = __dadd_rn(sy[threadIdx.x], sy[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = sx[0];
    out[2 * blockIdx.x + 1] = sy[0];
  }
}

__global__ __launch_bounds__(256) void k_reduce_minmax(const double* __restrict__ vmin,
                                                       const double* __restrict__ vmax, int64_t n,
                                                       double* __restrict__ out) {
  __shared__ double smin[256], smax[256];
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    if (vmin) mn = fmin(mn, vmin[i]);
    if (vmax) mx = fmax(mx, vmax[i]);
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = smin[0];
    out[1] = smax[0];
  }
}

// first stage for long vectors: block b reduces the elements b * 256 + t, (b + G) * 256 + t, ... to pmin[b] / pmax[b]
// (min and max are exact in any grouping: the two stages return what the single block returns)
__global__ __launch_bounds__(256) void k_reduce_minmax_part(const double* __restrict__ vmin, const double* __restrict__ vmax, int64_t n,
                                                            double* __restrict__ pmin, double* __restrict__ pmax) {
  __shared__ double smin[256], smax[256];
  double mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (vmin) mn = fmin(mn, vmin[i]);
    if (vmax) mx = fmax(mx, vmax[i]);
  }
  smin[threadIdx.x] = mn;
  smax[threadIdx.x] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pmin[blockIdx.x] = smin[0];
    pmax[blockIdx.x] = smax[0];
  }
}
// out[0] = min of vmin (+inf without), out[1] = max of vmax (-inf without)
void launch_reduce_minmax(const double* vmin, const double* vmax, int64_t n, double* out) {
  if (n <= 8192) {
    hipLaunchKernelGGL(k_reduce_minmax, dim3(1), dim3(256), 0, stream(), vmin, vmax, n, out);
    return;
  }
  const int G = (int)std::min<int64_t>(256, (n + 1023) / 1024);
  DevBuf<double> part((size_t)2 * G);
  hipLaunchKernelGGL(k_reduce_minmax_part, dim3(G), dim3(256), 0, stream(), vmin, vmax, n, part.p, part.p + G);
  hipLaunchKernelGGL(k_reduce_minmax, dim3(1), dim3(256), 0, stream(), vmin ? part.p : (const double*)nullptr,
                     vmax ? part.p + G : (const double*)nullptr, (int64_t)G, out);
}

// ------------------------------------------------------------------ SpGEMM: planning
// first/last stored row and length of every column of A (compact arrays that stay L2-resident
// while the plan kernel gathers them)
__global__ void k_col_extent(Csc A, int32_t* __restrict__ cmin, int32_t* __restrict__ cmax,
                             int32_t* __restrict__ clen) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= A.cols) return;
  const int64_t s = A.outer[k], e = col_end(A, k);
  clen[k] = (int32_t)(e - s);
  cmin[k] = (e > s) ? A.inner[s] : INT_MAX;
  cmax[k] = (e > s) ? A.inner[e - 1] : -1;
}

// span of every column run
__global__ void k_span_of(const int32_t* __restrict__ cmin, const int32_t* __restrict__ cmax, int32_t* __restrict__ span, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) span[k] = cmax[k] >= cmin[k] ? cmax[k] - cmin[k] + 1 : 0;
}

// stats[16] = max window rows, stats[17] = max k range over the column blocks, stats[18] = number of non-empty
// columns of A: a few blocks, one atomic per block and statistic (same-address atomics from every block of the
// plan kernels would cost more than this pass)
__global__ __launch_bounds__(256) void k_slab_reduce(const int32_t* __restrict__ blk_w, const int32_t* __restrict__ blk_kn,
                                                     int nblocks, const int32_t* __restrict__ aspan, int acols,
                                                     unsigned long long* __restrict__ stats) {
  __shared__ int sw[4], sk[4], sn[4];
  int w = 0, kn = 0, ne = 0;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  for (int i = t0; i < nblocks; i += stride) {
    w = max(w, blk_w[i]);
    kn = max(kn, blk_kn[i]);
  }
  for (int i = t0; i < acols; i += stride) ne += aspan[i] > 0 ? 1 : 0;
  w = wave_max_i32(w);
  kn = wave_max_i32(kn);
  for (int o = 32; o > 0; o >>= 1) ne += __shfl_xor(ne, o, WAVE);
  if (lane_id() == 0) {
    sw[threadIdx.x / WAVE] = w;
    sk[threadIdx.x / WAVE] = kn;
    sn[threadIdx.x / WAVE] = ne;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMax(&stats[16], (unsigned long long)max(max(sw[0], sw[1]), max(sw[2], sw[3])));
    atomicMax(&stats[17], (unsigned long long)max(max(sk[0], sk[1]), max(sk[2], sk[3])));
    atomicAdd(&stats[18], (unsigned long long)(sn[0] + sn[1] + sn[2] + sn[3]));
  }
}

// register-slab kernel geometry: J output columns per workgroup, SL slabs of 64 rows per wave, NW waves
constexpr int SLAB_CJ = 8, SLAB_CSL = 2, SLAB_CNW = 6;  // complex operands

// bins: 0 empty | 1..4 LDS direct window of 512/1024/2048/4096 rows | 5 LDS hash | 6 HBM accumulator
constexpr int BIN_EMPTY = 0, BIN_HASH = 5, BIN_HBM = 6, BIN_HASH_BIG = 7;  // 7: overflowed the small table
constexpr int HASH_MAX_FILL = 3072;   // distinct rows per column the LDS hash takes (3/4 of its large table); beyond: HBM accumulator

__host__ __device__ inline int window_bin(int span) {
  return span <= 512 ? 1 : span <= 1024 ? 2 : span <= 2048 ? 3 : span <= 4096 ? 4 : BIN_HASH;
}

// one wave per output column j: row window [lo, lo+span) that column j of A*B can touch, the
// number of intermediate products, an upper bound of its nnz and the kernel bin.
// stats: [0..6] bin histogram, [7] total products
__global__ __launch_bounds__(256) void k_spgemm_plan(Csc B, const int32_t* __restrict__ cmin,
                                                     const int32_t* __restrict__ cmax,
                                                     const int32_t* __restrict__ clen,
                                                     int32_t* __restrict__ lo_arr,
                                                     int32_t* __restrict__ span_arr,
                                                     uint8_t* __restrict__ bin_arr,
                                                     int64_t* __restrict__ ub_arr,
                                                     int64_t* __restrict__ ip_arr,
                                                     unsigned long long* __restrict__ stats,
                                                     int force_bin) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= B.cols) return;
  const int lane = lane_id();
  int lo = INT_MAX, hi = -1;
  int64_t ip = 0;
  for (int64_t p = B.outer[j] + lane; p < B.outer[j + 1]; p += WAVE) {
    const int k = B.inner[p];
    const int len = clen[k];
    if (len > 0) {
      lo = min(lo, cmin[k]);
      hi = max(hi, cmax[k]);
      ip += len;
    }
  }
  lo = wave_min_i32(lo);
  hi = wave_max_i32(hi);
  ip = wave_sum_i64(ip);
  if (lane == 0) {
    const int span = (hi >= lo) ? (hi - lo + 1) : 0;
    int bin = span == 0 ? BIN_EMPTY : window_bin(span);
    if (span > 0 && force_bin > 0) {
      if (force_bin >= BIN_HASH) bin = force_bin;
      else bin = max(bin, force_bin);  // a window can be forced larger, never smaller
    }
    int64_t ub = min((int64_t)span, ip);
    if (bin == BIN_HASH) ub = min(ub, (int64_t)HASH_MAX_FILL);
    lo_arr[j] = (span > 0) ? lo : 0;
    span_arr[j] = span;
    bin_arr[j] = (uint8_t)bin;
    ub_arr[j] = ub;
    ip_arr[j] = ip;
  }
}

// The column-pair kernel processes columns (2g, 2g+1) together: give both the larger window class so
// that a pair is never split over two launches.
__global__ void k_pair_bins(uint8_t* __restrict__ bin_arr, int n) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = 2 * g, j1 = j0 + 1;
  if (j1 >= n) return;
  const int b0 = bin_arr[j0], b1 = bin_arr[j1];
  if (b0 >= 1 && b0 <= 3 && b1 >= 1 && b1 <= 3 && (b0 == 3) != (b1 == 3)) {
    bin_arr[j0] = 3;
    bin_arr[j1] = 3;
  }
}

// histogram of the per-column bins (+ total of an optional int64 array) without hammering one
// address per column: LDS partials per block, 8 atomics per block.
// stats[0..6] += bin counts, stats[7] += sum(extra)
// stats[9 + b] = max over columns of bin b of span (when span_arr is given)
__global__ __launch_bounds__(256) void k_bin_hist(const uint8_t* __restrict__ bin_arr, const int64_t* __restrict__ extra,
                                                  const int32_t* __restrict__ span_arr, int n,
                                                  unsigned long long* __restrict__ stats) {
  __shared__ unsigned long long h[8];
  __shared__ unsigned int hmax[8];
  if (threadIdx.x < 8) {
    h[threadIdx.x] = 0ull;
    hmax[threadIdx.x] = 0u;
  }
  __syncthreads();
  unsigned int local[7] = {0, 0, 0, 0, 0, 0, 0};
  long long sum = 0;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
    const int b = bin_arr[j];
#pragma unroll
    for (int k = 0; k < 7; ++k) local[k] += (b == k) ? 1u : 0u;
    if (extra) sum += extra[j];
    if (span_arr) atomicMax(&hmax[b], (unsigned int)span_arr[j]);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    unsigned int v = local[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    if (lane_id() == 0 && v) atomicAdd(&h[k], (unsigned long long)v);
  }
  sum = wave_sum_i64(sum);
  if (lane_id() == 0 && sum) atomicAdd(&h[7], (unsigned long long)sum);
  __syncthreads();
  if (threadIdx.x < 8 && h[threadIdx.x]) atomicAdd(&stats[threadIdx.x], h[threadIdx.x]);
  if (threadIdx.x < 7 && hmax[threadIdx.x]) atomicMax(&stats[9 + threadIdx.x], (unsigned long long)hmax[threadIdx.x]);
}

// ------------------------------------------------------------------ SpGEMM: numeric, LDS window
// Column j of C = alpha * A * B.  The wave walks column j of B in storage order (ascending k,
// fetched 64 entries at a time and broadcast with v_readlane); for each (k, b) its lanes stride
// over column k of A with coalesced index/value loads and update the direct-mapped LDS
// accumulator acc[row - lo] (a collision-free hash: the bucket of row i is i - lo).  The
// epilogue scans the window in row order, applies the prune rule and compacts the survivors with
// ballot + popcount prefix, so the column comes out sorted with no sort.
template <typename T, int W, int NW>
__global__ __launch_bounds__(NW* WAVE) void k_spgemm_window(
    Csc A, Csc B, const int32_t* __restrict__ lo_arr, const int32_t* __restrict__ span_arr,
    const uint8_t* __restrict__ bin_arr, int my_bin, const int64_t* __restrict__ tmpoff,
    int32_t* __restrict__ out_inner, T* __restrict__ out_val, int32_t* __restrict__ count,
    double alpha, double threshold, int dense_rule, int nblocks) {
  __shared__ T acc_all[NW * W];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = b * NW + wave;
  if (j >= B.cols) return;
  if (bin_arr[j] != my_bin) return;
  T* acc = acc_all + wave * W;
  const bool fma = (dense_rule & 2) != 0;   // (option spgemm_fma: one rounding per product)
  const int span = span_arr[j], lo = lo_arr[j];
  for (int s = lane; s < span; s += WAVE) acc[s] = Sc<T>::zero();

  const int32_t* __restrict__ Ai = A.inner;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  const int64_t bs = B.outer[j], be = B.outer[j + 1];
  for (int64_t p0 = bs; p0 < be; p0 += WAVE) {
    const int np = (int)min((int64_t)WAVE, be - p0);
    int k_l = 0;
    T b_l = Sc<T>::zero();
    int64_t as_l = 0;
    int len_l = 0;
    if (lane < np) {
      k_l = B.inner[p0 + lane];
      b_l = Bv[p0 + lane];
      as_l = A.outer[k_l];
      len_l = (int)(A.outer[k_l + 1] - as_l);
    }
    for (int t = 0; t < np; ++t) {
      const int64_t as = readlane_i64(as_l, t);
      const int len = readlane_i32(len_l, t);
      const T bk = readlane_T(b_l, t);
      for (int q = lane; q < len; q += 2 * WAVE) {
        const int q1 = q + WAVE;
        const bool v1 = q1 < len;
        const int i0 = Ai[as + q];
        const T a0 = Av[as + q];
        const int i1 = v1 ? Ai[as + q1] : lo;
        const T a1 = v1 ? Av[as + q1] : Sc<T>::zero();
        acc[i0 - lo] = Sc<T>::fmadd(a0, bk, acc[i0 - lo], fma);
        if (v1) acc[i1 - lo] = Sc<T>::fmadd(a1, bk, acc[i1 - lo], fma);
      }
      // the next k may hit the same rows: keep this wave's LDS traffic in program order
      __builtin_amdgcn_wave_barrier();
    }
  }

  const int64_t base = tmpoff[j];
  int cnt = 0;
  for (int s0 = 0; s0 < span; s0 += WAVE) {
    const int s = s0 + lane;
    const T v = (s < span) ? acc[s] : Sc<T>::zero();
    const T sv = Sc<T>::scale(alpha, v);
    const bool keep = (s < span) && ((dense_rule & 1) ? (Sc<T>::mag(v) > threshold) : (Sc<T>::mag(sv) > threshold));
    const unsigned long long m = __ballot(keep);
    if (keep) {
      const int64_t pos = base + cnt + __popcll(m & lanemask_lt());
      out_inner[pos] = lo + s;
      out_val[pos] = sv;
    }
    cnt += __popcll(m);
  }
  if (lane == 0) count[j] = cnt;
}

// ------------------------------------------------------------------ SpGEMM: numeric, column-pair kernel (LDS accumulators)
// General fallback of the register-slab kernel (operands whose columns are not run-like).  Design notes from its
// profiles (profiles/README.md items 2-5): operand loads use the SGPR-base + 32-bit VGPR offset form with the chunk
// stride as an immediate (lanes past the end of a column over-read into the DevMat slack instead of being clamped),
// multipliers and row descriptors are pinned to SGPRs, the LDS address of a product is one v_lshl_add_u32:
// (row << 3) + (window base - lo*8), and the accumulate is one ds_add_f64 per product.
template <int MAXCH>
struct PairRegs {
  int idx[MAXCH];
  double val[MAXCH];
};
template <int C>
__device__ inline void pair_issue_one(int& idx, double& val, unsigned vo_i, unsigned vo_v, const int32_t* Ai,
                                      const double* Av) {
  asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=&v"(idx) : "v"(vo_i), "s"(Ai), "n"(C * 256) : "memory");
  asm volatile("global_load_dwordx2 %0, %1, %2 offset:%3" : "=&v"(val) : "v"(vo_v), "s"(Av), "n"(C * 512) : "memory");
}
template <int MAXCH, int... Cs>
__device__ inline void pair_issue(PairRegs<MAXCH>& st, unsigned vo_i, unsigned vo_v, const int32_t* Ai, const double* Av,
                                  std::integer_sequence<int, Cs...>) {
  (pair_issue_one<Cs>(st.idx[Cs], st.val[Cs], vo_i, vo_v, Ai, Av), ...);
}
// FL: 1 = column 0 only, 2 = column 1 only, 3 = both
template <int FL, int ABL = 0>
__device__ inline void pair_apply(int idx, double val, double b0, double b1, int cb0, int cb1, char* smem) {
  if constexpr (ABL & 1) {  // ablation: no LDS accumulate
    if constexpr (FL & 1) { const double p = __dmul_rn(val, b0); const int a = (idx << 3) + cb0; asm volatile("" :: "v"(p), "v"(a)); }
    if constexpr (FL & 2) { const double p = __dmul_rn(val, b1); const int a = (idx << 3) + cb1; asm volatile("" :: "v"(p), "v"(a)); }
  } else {
    if constexpr (FL & 1) atomicAdd(reinterpret_cast<double*>(smem + ((idx << 3) + cb0)), __dmul_rn(val, b0));
    if constexpr (FL & 2) atomicAdd(reinterpret_cast<double*>(smem + ((idx << 3) + cb1)), __dmul_rn(val, b1));
  }
}
template <int FL, int MAXCH, int ABL = 0>
__device__ inline void pair_process(PairRegs<MAXCH>& st, int len, double b0, double b1, int cb0, int cb1, char* smem,
                                    int lane) {
#pragma unroll
  for (int c = 0; c < MAXCH; ++c) {
    if ((c + 1) * WAVE <= len) {
      pair_apply<FL, ABL>(st.idx[c], st.val[c], b0, b1, cb0, cb1, smem);
    } else if (c * WAVE < len) {
      if (lane < len - c * WAVE) pair_apply<FL, ABL>(st.idx[c], st.val[c], b0, b1, cb0, cb1, smem);
    }
  }
}

// One wave owns two adjacent output columns.  An earlier generation merged the two B columns with a serial scalar
// merge (7.5 of its 13.3 ms, profiles/README.md item 4); here the union of the two columns
// is walked in ROW WINDOWS of 64 consecutive k.  Inside a window the slot of a row is simply
// k - kw, i.e. a LANE: each column scatters its multipliers to the slot lanes with one ds_permute
// (cross-lane, no LDS memory), A's column pointers for the 64 slots are one coalesced load of
// A.outer[kw .. kw+64], and the occupied slots are walked with s_ff1 over the ballot mask.  The next
// window's B entries are requested before the current window is processed.
template <int MAXCH, int NW>
__global__ __launch_bounds__(NW* WAVE) void k_spgemm_pair3(
    Csc A, Csc B, const int32_t* __restrict__ lo_arr, const int32_t* __restrict__ span_arr,
    const uint8_t* __restrict__ bin_arr, int bin_lo, int bin_hi, const int64_t* __restrict__ tmpoff,
    int32_t* __restrict__ out_inner, double* __restrict__ out_val, int32_t* __restrict__ count,
    double alpha, double threshold, int dense_rule, int nblocks, int wrt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = uni_i32(threadIdx.x / WAVE), lane = lane_id();
  const int g = b * NW + wave;
  const int base0 = (wave * 2) * wrt * 8, base1 = base0 + wrt * 8;
  double* acc0 = reinterpret_cast<double*>(smem + base0);
  double* acc1 = reinterpret_cast<double*>(smem + base1);
  const int32_t* __restrict__ Ai = A.inner;
  const double* __restrict__ Av = static_cast<const double*>(A.val);
  const double* __restrict__ Bv = static_cast<const double*>(B.val);
  const int32_t* __restrict__ Bi = B.inner;
  const int64_t* __restrict__ Ao = A.outer;

  const int j0 = 2 * g, j1 = 2 * g + 1;
  const int bb0 = bin_arr[min(j0, B.cols - 1)], bb1 = bin_arr[min(j1, B.cols - 1)];
  const bool act0 = uni_i32((j0 < B.cols && bb0 >= bin_lo && bb0 <= bin_hi) ? 1 : 0) != 0;
  const bool act1 = uni_i32((j1 < B.cols && bb1 >= bin_lo && bb1 <= bin_hi) ? 1 : 0) != 0;
  if (!act0 && !act1) return;
  const int lo0 = uni_i32(act0 ? lo_arr[j0] : 0), span0 = uni_i32(act0 ? span_arr[j0] : 0);
  const int lo1 = uni_i32(act1 ? lo_arr[j1] : 0), span1 = uni_i32(act1 ? span_arr[j1] : 0);
  const int cb0 = base0 - lo0 * 8, cb1 = base1 - lo1 * 8;
  int p0 = uni_i32(act0 ? (int)B.outer[j0] : 0), e0 = uni_i32(act0 ? (int)B.outer[j0 + 1] : 0);
  int p1 = uni_i32(act1 ? (int)B.outer[j1] : 0), e1 = uni_i32(act1 ? (int)B.outer[j1 + 1] : 0);
  for (int s = lane; s < span0; s += WAVE) acc0[s] = 0.0;
  for (int s = lane; s < span1; s += WAVE) acc1[s] = 0.0;
  const unsigned lane4 = (unsigned)lane * 4u, lane8 = (unsigned)lane * 8u;
  const int ncolsA = A.cols;

  // speculative batches: the next 64 entries of each B column (over-read past the column end is
  // harmless: DevMat keeps kIndexSlack entries of slack and the lanes are masked by position)
  int kv0 = Bi[p0 + lane], kv1 = Bi[p1 + lane];
  double bv0 = Bv[p0 + lane], bv1 = Bv[p1 + lane];

  PairRegs<MAXCH> sa, sb;
  for (;;) {
    // ---- window start = smallest pending row of the two columns
    const int f0 = (p0 < e0) ? readlane_i32(kv0, 0) : INT_MAX;
    const int f1 = (p1 < e1) ? readlane_i32(kv1, 0) : INT_MAX;
    const int kw = min(f0, f1);
    if (kw == INT_MAX) break;
    // column pointers of A for the 64 slots of the window (one coalesced load)
    const int ks = min(kw + lane, ncolsA - 1);
    const int64_t o0 = Ao[ks], o1 = Ao[ks + 1];
    // entries of each column that fall into [kw, kw+64)
    const bool in0 = (p0 + lane < e0) && (kv0 < kw + WAVE);
    const bool in1 = (p1 + lane < e1) && (kv1 < kw + WAVE);
    const int n0 = __popcll(__ballot(in0)), n1 = __popcll(__ballot(in1));
    // Each column's in-window entries are sorted and sit in lanes 0..n-1, so the entry of slot s (row kw+s)
    // is held by lane rank(s) = number of in-window entries with a smaller row: a gather (ds_bpermute).
    unsigned long long m0 = 0ull, m1 = 0ull;  // occupancy masks by slot
    {
      // occupancy: bit (kv - kw) for every in-window entry, OR-reduced over the wave
      unsigned long long x0 = in0 ? (1ull << (kv0 - kw)) : 0ull, x1 = in1 ? (1ull << (kv1 - kw)) : 0ull;
      for (int o = 32; o > 0; o >>= 1) {
        x0 |= __shfl_xor(x0, o, WAVE);
        x1 |= __shfl_xor(x1, o, WAVE);
      }
      m0 = x0;
      m1 = x1;
    }
    const unsigned long long below = lanemask_lt();
    const int r0 = __popcll(m0 & below), r1 = __popcll(m1 & below);  // source lane of slot `lane`
    const double bs0 = __shfl(bv0, r0, WAVE), bs1 = __shfl(bv1, r1, WAVE);
    const int as_s = (int)o0, len_s = (int)(o1 - o0);
    // advance the column cursors and request the next batches now (latency hides behind the window)
    p0 += n0;
    p1 += n1;
    const int nk0 = Bi[p0 + lane], nk1 = Bi[p1 + lane];
    const double nb0 = Bv[p0 + lane], nb1 = Bv[p1 + lane];

    // ---- walk the occupied slots, A column double-buffered in registers
    unsigned long long mu = m0 | m1;
    int s_cur = __builtin_ctzll(mu);
    mu &= mu - 1;
    pair_issue<MAXCH>(sa, ((unsigned)readlane_i32(as_s, s_cur) << 2) + lane4,
                      ((unsigned)readlane_i32(as_s, s_cur) << 3) + lane8, Ai, Av, std::make_integer_sequence<int, MAXCH>{});
    bool use_a = true;
    for (;;) {
      const bool more = mu != 0ull;
      const int s_nxt = more ? __builtin_ctzll(mu) : s_cur;
      mu &= mu - 1;
      const unsigned an = (unsigned)readlane_i32(as_s, s_nxt);
      // descriptor of the current slot
      const int len = readlane_i32(len_s, s_cur);
      const int fl = (int)((m0 >> s_cur) & 1ull) | ((int)((m1 >> s_cur) & 1ull) << 1);
      const double b0 = readlane_f64(bs0, s_cur), b1 = readlane_f64(bs1, s_cur);
      const int as_c = readlane_i32(as_s, s_cur);
      if (use_a) {
        pair_issue<MAXCH>(sb, (an << 2) + lane4, (an << 3) + lane8, Ai, Av, std::make_integer_sequence<int, MAXCH>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MAXCH) : "memory");
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) asm volatile("" : "+v"(sa.idx[c]), "+v"(sa.val[c]));
        if (fl == 3) pair_process<3, MAXCH>(sa, len, b0, b1, cb0, cb1, smem, lane);
        else if (fl == 1) pair_process<1, MAXCH>(sa, len, b0, b1, cb0, cb1, smem, lane);
        else pair_process<2, MAXCH>(sa, len, b0, b1, cb0, cb1, smem, lane);
      } else {
        pair_issue<MAXCH>(sa, (an << 2) + lane4, (an << 3) + lane8, Ai, Av, std::make_integer_sequence<int, MAXCH>{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * MAXCH) : "memory");
#pragma unroll
        for (int c = 0; c < MAXCH; ++c) asm volatile("" : "+v"(sb.idx[c]), "+v"(sb.val[c]));
        if (fl == 3) pair_process<3, MAXCH>(sb, len, b0, b1, cb0, cb1, smem, lane);
        else if (fl == 1) pair_process<1, MAXCH>(sb, len, b0, b1, cb0, cb1, smem, lane);
        else pair_process<2, MAXCH>(sb, len, b0, b1, cb0, cb1, smem, lane);
      }
      if (len > MAXCH * WAVE) {  // rare: column longer than the register set
        for (int q = MAXCH * WAVE; q < len; q += WAVE) {
          if (q + lane < len) {
            const int i = Ai[as_c + q + lane];
            const double v = Av[as_c + q + lane];
            if (fl & 1) atomicAdd(acc0 + (i - lo0), __dmul_rn(v, b0));
            if (fl & 2) atomicAdd(acc1 + (i - lo1), __dmul_rn(v, b1));
          }
        }
      }
      if (!more) break;
      s_cur = s_nxt;
      use_a = !use_a;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the trailing (dummy) set before its registers are re-issued
    kv0 = nk0; kv1 = nk1;
    bv0 = nb0; bv1 = nb1;
  }
  __builtin_amdgcn_wave_barrier();

#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const bool act = r == 0 ? act0 : act1;
    if (!act) continue;
    const int j = r == 0 ? j0 : j1;
    const int lo = r == 0 ? lo0 : lo1, span = r == 0 ? span0 : span1;
    const double* acc = r == 0 ? acc0 : acc1;
    const int64_t base = tmpoff[j];
    int cnt = 0;
    for (int s0 = 0; s0 < span; s0 += WAVE) {
      const int s = s0 + lane;
      const double v = (s < span) ? acc[s] : 0.0;
      const double sv = __dmul_rn(alpha, v);
      const bool keep = (s < span) && ((dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold));
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = base + cnt + __popcll(m & lanemask_lt());
        out_inner[pos] = lo + s;
        out_val[pos] = sv;
      }
      cnt += __popcll(m);
    }
    if (lane == 0) count[j] = cnt;
  }
}

// ------------------------------------------------------------------ SpGEMM: numeric, register-slab kernel
// For operands whose columns are (nearly) contiguous runs of rows -- banded Hamiltonians and the density
// matrices purified from them -- the accumulator does not need LDS at all.  A workgroup owns a block of J
// consecutive output columns and the row window [lo, lo+W) they can touch; the window is cut into slabs of 64
// rows, slab m belongs to wave m % NW, and inside a slab lane l owns row lo + 64 m + l: the partial sums live
// in REGISTERS, acc[slab][column].  Walking k = kmin..kmax in ascending order (the reference's accumulation
// order, SMatrixAlgebraModule SparseBranch), a wave loads the 64 values A(rows of its slab, k) with one
// coalesced load -- A is pre-expanded so that a column is a dense run over its span, holes = 0 -- and the J
// multipliers B(k, j..j+J-1) arrive as wave-uniform SGPRs from a pre-expanded per-block tile, so every
// product is v_mul_f64 v, s, v followed by v_add_f64: no atomics, no index loads, 16 B of L1 traffic per
// 2 J flops.  Padding with zeros is exact: x + (+-0 * b) == x for every x that survives the prune
// (|v| > threshold >= 0 drops the zeros themselves), so results stay bit-identical to the sparse walk.
//
// expanded A: aexp[aeoff[k] + (r - afirst[k])] = A(r, k) for afirst[k] <= r <= alast[k]
template <typename T>
__global__ __launch_bounds__(256) void k_slab_expand_a(Csc A, const int32_t* __restrict__ afirst,
                                                       const int64_t* __restrict__ aeoff, T* __restrict__ aexp) {
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (k >= A.cols) return;
  const int lane = lane_id();
  const int64_t s = A.outer[k], e = A.outer[k + 1];
  if (e <= s) return;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  T* __restrict__ dst = aexp + aeoff[k] - afirst[k];
  for (int64_t p = s + lane; p < e; p += WAVE) {
    const int r = A.inner[p];
    const int prev = (p > s) ? A.inner[p - 1] : r - 1;
    dst[r] = Av[p];
    for (int h = prev + 1; h < r; ++h) dst[h] = Sc<T>::zero();
  }
}

// Expanded columns for the MFMA tile kernel (spgemm_tile.hpp, rows per lane > 1): the run of column k sits in a slot that
// starts at a multiple of `al` rows below its first row and ends at one above its last, position = row (mod al), the
// pads zero -- a lane may then read R consecutive rows that only touch the run.  span[k] = slot size; after the scan of
// the slot sizes k_aligned_offsets moves every offset to the position of the column's first row and zeroes the pads.
__global__ void k_span_aligned(const int32_t* __restrict__ cmin, const int32_t* __restrict__ cmax, int32_t* __restrict__ span, int n, int al) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int f = cmin[k], l = cmax[k];
  span[k] = l >= f ? (l / al + 1) * al - f / al * al : 0;
}
template <typename T>
__global__ __launch_bounds__(256) void k_aligned_offsets(const int32_t* __restrict__ cmin, const int32_t* __restrict__ cmax,
                                                         int64_t* __restrict__ off, T* __restrict__ exp, int n, int al) {
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (k >= n) return;
  const int lane = lane_id();
  const int f = cmin[k], l = cmax[k];
  if (l < f) return;
  const int64_t slot = off[k];
  const int head = f - f / al * al, tail = (l / al + 1) * al - 1 - l;
  for (int i = lane; i < head; i += WAVE) exp[slot + i] = Sc<T>::zero();
  for (int i = lane; i < tail; i += WAVE) exp[slot + head + (l - f + 1) + i] = Sc<T>::zero();
  if (lane == 0) off[k] = slot + head;
}

// per block of J output columns (one wave each): k range = union of the columns' row ranges in B, row window =
// union of the runs of A(:, k) over that k range (a superset of the rows actually touched when B has holes --
// harmless, the window only has to contain them), sizes of the B tile and of the output slots.
template <int J>
__global__ __launch_bounds__(256) void k_slab_plan(int ncols, const int32_t* __restrict__ bfirst,
                                                   const int32_t* __restrict__ blast, const int32_t* __restrict__ cmin,
                                                   const int32_t* __restrict__ cmax, int32_t* __restrict__ blk_lo,
                                                   int32_t* __restrict__ blk_w, int32_t* __restrict__ blk_kmin,
                                                   int32_t* __restrict__ blk_kn, int64_t* __restrict__ bsz,
                                                   int64_t* __restrict__ tsz, int nblocks, int align16 = 0) {
  // align16 (MFMA tile kernel): the window starts at a multiple of 16 rows and is a multiple of 16 rows long, so that a
  // column's slot keeps every row r at a position = r (mod 16): a 16-row tile segment of a run is one 128-byte line
  const int b = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int lane = lane_id();
  if (b < nblocks) {
    const int j = b * J + lane;
    const bool has = lane < J && j < ncols && blast[min(j, ncols - 1)] >= 0;
    const int kmin = wave_min_i32(has ? bfirst[j] : INT_MAX), kmax = wave_max_i32(has ? blast[j] : -1);
    if (kmax >= kmin && kmax - kmin >= 4096) {
      // a k range far beyond any multiplier tile (at most 1023 rows): the operand is not run-like and the slab kernels
      // will be declined -- do not walk the range (for a relabelled matrix it is the whole dimension, in every block)
      if (lane == 0) {
        blk_lo[b] = 0;
        blk_w[b] = 1 << 30;
        blk_kmin[b] = kmin;
        blk_kn[b] = kmax - kmin + 1;
        bsz[b] = 0;
        tsz[b] = 0;
      }
    } else {
      int lo = INT_MAX, hi = -1;
      // (a block whose columns are all empty has kmin = INT_MAX: kmin + lane must not be formed)
      for (int k = (kmax >= kmin) ? kmin + lane : 0; k <= kmax; k += WAVE) {
        const int c0 = cmin[k], c1 = cmax[k];
        if (c1 >= c0) {
          lo = min(lo, c0);
          hi = max(hi, c1 + 1);
        }
      }
      lo = wave_min_i32(lo);
      hi = wave_max_i32(hi);
      if (lane == 0) {
        if (align16 > 0 && hi > lo) {   // (a multiple of 16: the rows of a tile of the MFMA kernel)
          lo = lo / align16 * align16;
          hi = (hi + align16 - 1) / align16 * align16;
        }
        const int w = (hi > lo) ? hi - lo : 0;
        const int kn = (w > 0 && kmax >= kmin) ? kmax - kmin + 1 : 0;
        blk_lo[b] = w > 0 ? lo : 0;
        blk_w[b] = w;
        blk_kmin[b] = kn > 0 ? kmin : 0;
        blk_kn[b] = kn;
        bsz[b] = (int64_t)((kn + 1) & ~1) * J;  // an all-zero row pads odd k ranges
        tsz[b] = (int64_t)w * J;
      }
    }
  }
}

// The maxima of the windows and k ranges (stats[16], stats[17]) and the exclusive prefix sums of the slot sizes in ONE launch
// without any dependency between workgroups: workgroup g owns a contiguous part of the blocks and sums everything BEFORE its
// part itself (at most a few hundred KB from L2, every load of a thread in flight together) -- redundant reads instead of a
// second and a third launch.  toff/boff hold nblocks + 1 entries.
constexpr int kOffParts = 64;
__global__ __launch_bounds__(256) void k_slab_offsets(const int32_t* __restrict__ blk_w, const int32_t* __restrict__ blk_kn,
                                                      const int64_t* __restrict__ tsz, const int64_t* __restrict__ bsz,
                                                      int nblocks, int part, int64_t* __restrict__ toff,
                                                      int64_t* __restrict__ boff, unsigned long long* __restrict__ stats) {
  __shared__ long long wsum[2][4];
  __shared__ int wmax[2][4];
  const int lane = lane_id(), wave = threadIdx.x / WAVE;
  const int p0 = blockIdx.x * part, p1 = min(nblocks, p0 + part);
  if (p0 >= nblocks) return;
  auto block_sum = [&](long long& a, long long& b) {   // both sums over the workgroup, returned to every thread
    a = wave_sum_i64(a);
    b = wave_sum_i64(b);
    __syncthreads();
    if (lane == 0) { wsum[0][wave] = a; wsum[1][wave] = b; }
    __syncthreads();
    a = wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
    b = wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
  };
  long long carry_t = 0, carry_b = 0;
  {   // (eight loads of a thread in flight together: the loop is a chain of memory round trips otherwise -- 19 us at 16 384 blocks)
    int i = threadIdx.x;
    for (; i + 7 * 256 < p0; i += 8 * 256) {
      long long t[8], u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        t[e] = tsz[i + e * 256];
        u[e] = boff ? (long long)bsz[i + e * 256] : 0;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) { carry_t += t[e]; carry_b += u[e]; }
    }
    for (; i < p0; i += 256) {
      carry_t += tsz[i];
      if (boff) carry_b += bsz[i];
    }
  }
  block_sum(carry_t, carry_b);
  int mw = 0, mk = 0;
  for (int base = p0; base < p1; base += 256) {
    const int i = base + threadIdx.x;
    const long long t = i < p1 ? (long long)tsz[i] : 0, b = (boff && i < p1) ? (long long)bsz[i] : 0;
    if (i < p1) {
      mw = max(mw, blk_w[i]);
      mk = max(mk, blk_kn[i]);
    }
    long long xt = t, xb = b;
    for (int o = 1; o < WAVE; o <<= 1) {
      const long long at = __shfl_up(xt, o, WAVE), ab = __shfl_up(xb, o, WAVE);
      if (lane >= o) { xt += at; xb += ab; }
    }
    __syncthreads();
    if (lane == WAVE - 1) { wsum[0][wave] = xt; wsum[1][wave] = xb; }
    __syncthreads();
    long long pt = carry_t + xt - t, pb = carry_b + xb - b;
    for (int q = 0; q < wave; ++q) { pt += wsum[0][q]; pb += wsum[1][q]; }
    if (i < p1) {
      toff[i] = pt;
      if (boff) boff[i] = pb;
    }
    carry_t += wsum[0][0] + wsum[0][1] + wsum[0][2] + wsum[0][3];
    carry_b += wsum[1][0] + wsum[1][1] + wsum[1][2] + wsum[1][3];
  }
  if (p1 == nblocks && threadIdx.x == 0) {
    toff[nblocks] = carry_t;
    if (boff) boff[nblocks] = carry_b;
  }
  mw = wave_max_i32(mw);
  mk = wave_max_i32(mk);
  __syncthreads();
  if (lane == 0) { wmax[0][wave] = mw; wmax[1][wave] = mk; }
  __syncthreads();
  if (threadIdx.x == 0 && stats) {
    atomicMax(&stats[16], (unsigned long long)max(max(wmax[0][0], wmax[0][1]), max(wmax[0][2], wmax[0][3])));
    atomicMax(&stats[17], (unsigned long long)max(max(wmax[1][0], wmax[1][1]), max(wmax[1][2], wmax[1][3])));
  }
}

// B tile of a block: bblk[boff + (k - kmin) * J + jj] = B(k, b*J + jj), zeros elsewhere.  Staged through LDS
// (transposed, odd pitch) so that both the scatter and the write-out are conflict-free / coalesced.  When A and
// B are the same matrix (X*X in every purification step) the same pass also writes the expanded runs of A
// (fuse_a), so the operand is read from HBM once.
template <typename T, int J>
__global__ __launch_bounds__(256) void k_slab_expand_b(Csc B, const int32_t* __restrict__ blk_kmin,
                                                       const int32_t* __restrict__ blk_kn,
                                                       const int64_t* __restrict__ blk_boff, T* __restrict__ bblk,
                                                       int nblocks, int pitch, int fuse_a,
                                                       const int64_t* __restrict__ aeoff, T* __restrict__ aexp,
                                                       const int32_t* __restrict__ clen,
                                                       int64_t* __restrict__ blk_prod) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* tile = reinterpret_cast<T*>(smem);  // [J][pitch]
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int kn = blk_kn[b], kmin = blk_kmin[b];
  if (kn == 0 && !fuse_a) return;  // (fuse_a: the runs of these columns may still be needed as A columns)
  const bool tiled = kn > 0;
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  constexpr int CH = 5, NCOL = J / 4;  // columns per wave
  // request the first CH*64 entries of this wave's columns before anything else (over-read stays in the slack)
  int idx[NCOL][CH];
  T val[NCOL][CH];
#pragma unroll
  for (int q = 0; q < NCOL; ++q) {
    const int j = min(b * J + wave + 4 * q, B.cols - 1);
    const int64_t s = B.outer[j];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      idx[q][c] = B.inner[s + c * WAVE + lane];
      val[q][c] = Bv[s + c * WAVE + lane];
    }
  }
  const int kne = (kn + 1) & ~1;
  for (int jj = 0; jj < J; ++jj)
    for (int i = threadIdx.x; i < kne; i += blockDim.x) tile[jj * pitch + i] = Sc<T>::zero();
  __syncthreads();
  long long nprod = 0;  // products of this wave's columns: sum over their entries of nnz(A(:, row))
#pragma unroll
  for (int q = 0; q < NCOL; ++q) {
    const int jj = wave + 4 * q;
    const int j = b * J + jj;
    if (j >= B.cols) break;
    const int64_t s = B.outer[j], e = col_end(B, j);
    if (e <= s) continue;
    const int first = readlane_i32(idx[q][0], 0);
    T* __restrict__ dst = fuse_a ? aexp + aeoff[j] - first : nullptr;
    int carry = first - 1;  // row of the entry before the current chunk
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int64_t p = s + c * WAVE + lane;
      const int r = idx[q][c];
      int prev = __shfl_up(r, 1, WAVE);
      if (lane == 0) prev = carry;
      if (p < e) {
        if (clen) nprod += clen[r];
        if (tiled) tile[jj * pitch + (r - kmin)] = val[q][c];
        if (fuse_a) {
          dst[r] = val[q][c];
          for (int h = prev + 1; h < r; ++h) dst[h] = Sc<T>::zero();
        }
      }
      carry = readlane_i32(r, WAVE - 1);
    }
    for (int64_t p = s + CH * WAVE + lane; p < e; p += WAVE) {
      const int r = B.inner[p];
      const T v = Bv[p];
      if (clen) nprod += clen[r];
      if (tiled) tile[jj * pitch + (r - kmin)] = v;
      if (fuse_a) {
        const int prev = B.inner[p - 1];
        dst[r] = v;
        for (int h = prev + 1; h < r; ++h) dst[h] = Sc<T>::zero();
      }
    }
  }
  nprod = wave_sum_i64(nprod);
  __shared__ long long sprod[4];
  if (lane == 0) sprod[wave] = nprod;
  __syncthreads();
  if (threadIdx.x == 0) blk_prod[b] = sprod[0] + sprod[1] + sprod[2] + sprod[3];
  T* __restrict__ out = bblk + blk_boff[b];
  const int total = kne * J;
  for (int i = threadIdx.x; i < total; i += blockDim.x) out[i] = tile[(i % J) * pitch + (i / J)];
}

// upper-bound output slot of every column: blk_toff[b] + jj * W_b
template <int J>
__global__ void k_slab_tmpoff(int ncols, const int32_t* __restrict__ blk_w, const int64_t* __restrict__ blk_toff,
                              int64_t* __restrict__ tmpoff) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > ncols) return;
  if (j == ncols) {
    const int b = (ncols - 1) / J;
    tmpoff[j] = blk_toff[b] + (int64_t)blk_w[b] * J;
    return;
  }
  const int b = j / J;
  tmpoff[j] = blk_toff[b] + (int64_t)(j % J) * blk_w[b];
}

// n + 4 records: the tail is empty (pipelined prefetch past the last column)
__global__ void k_slab_runs(const int32_t* __restrict__ cmin, const int32_t* __restrict__ cmax,
                            const int64_t* __restrict__ aeoff, const char* __restrict__ aexp, int elem_bytes,
                            SlabRun* __restrict__ runs, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n + 4) return;
  const bool any = k < n && cmax[k] >= cmin[k];
  SlabRun r;
  const uint64_t addr = reinterpret_cast<uint64_t>(aexp + (any ? aeoff[k] * elem_bytes : 0));
  r.addr_lo = (uint32_t)addr;
  r.addr_hi = (uint32_t)(addr >> 32) & 0xffffu;
  r.nbytes = any ? (uint32_t)(cmax[k] - cmin[k] + 1) * (uint32_t)elem_bytes : 0u;
  r.flags = kBufferFlags;
  r.first = any ? cmin[k] : (1 << 30);
  r.first8 = any ? cmin[k] * elem_bytes : 0;  // first row * element size
  r.span62 = any ? (cmax[k] - cmin[k] + 1) + 62 : 0;
  r.pad = 0;
  runs[k] = r;
}

// plast[j] = largest label among the rows of column j (lab[row]): label-ordered steps (SlabForm::plast)
__global__ __launch_bounds__(256) void k_col_plast(Csc A, const int32_t* __restrict__ lab, int32_t* __restrict__ plast) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  int mx = -1;
  for (int64_t p = A.outer[j] + lane, e = A.outer[j + 1]; p < e; p += WAVE) mx = max(mx, lab[A.inner[p]]);
  mx = wave_max_i32(mx);
  if (lane == 0) plast[j] = mx;
}

constexpr int SLAB_DTILE = 4096;   // doubles of LDS for the D columns of a block (fused epilogues): 32 KB, four workgroups per CU
typedef double v8d __attribute__((ext_vector_type(8)));
#include "slab_loop.inc"

// Pipeline of one wave (steps = consecutive k, two register sets A/B): while step t is multiplied, the slab
// loads of step t+1 are in flight (buffer loads: rows outside the run of column k read as 0.0 through the
// descriptor's bounds check, so every step issues exactly SL loads and the waits are static), and so are the
// scalar loads of B(k+1, :) and of the run record of step t+2.  Products are issued as four independent
// multiply -> add chains (mul x4, add x4; unfused like the reference) so that no instruction waits on the one
// before it.  The whole loop is ONE inline-asm block over fixed physical registers (slab_loop.inc, generated
// by tools/gen_slab_asm.py, register map there): with separate asm statements the compiler is free to copy a
// register between them -- including one whose asynchronous load has not landed yet.
//
// EPI (fused epilogues of the purification steps, A = B = X real, one rank): the partial sums never leave the
// registers as a product --
//   1: result = the pruned product; its dot with D and its trace are accumulated on the way (TRS2, sigma < 0)
//   2: result = am * product + bm * X merged by the AddSparseVectors rules (inc_decide), plus dot and trace (sigma > 0):
//      X(r, j) is read back from the expanded runs the loop has just multiplied with (cache-hot), D(r, j) from the
//      expanded copy of D kept for the whole solve.
// Both leave the result LOOSE in the upper-bound slots.  A zero of the expanded X is read as "no entry": the host
// makes sure X stores no zero value (DevMat::zero_free).  Where a column of X leaves its block's row window the kernel
// raises fz.flag and the host repeats the step on the unfused path.

template <int J, int SL, int NW, int MODE, int EPI = 0>  // MODE 0 unfused, 1 fma, 2..4 ablations (timing experiments, wrong results)
__global__ __launch_bounds__(NW* WAVE) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_spgemm_slab(
    const SlabRun* __restrict__ runs, const double* __restrict__ bblk,
    const int64_t* __restrict__ blk_boff, const int32_t* __restrict__ blk_kmin, const int32_t* __restrict__ blk_kn,
    const int32_t* __restrict__ blk_lo, const int32_t* __restrict__ blk_w, const int64_t* __restrict__ blk_toff,
    int32_t* __restrict__ out_inner, double* __restrict__ out_val, int32_t* __restrict__ count, double alpha,
    double threshold, int dense_rule, int ncols, int nblocks, const SlabFuseArgs* __restrict__ fzp) {
  static_assert(J == 16 && SL == 3 && (NW == 4 || ((NW == 6 || NW == 8) && (MODE == 0 || MODE == 9))), "register map / wave rotation of slab_loop.inc");
  __shared__ int cnt_s[NW * SL][J];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = uni_i32(threadIdx.x / WAVE), lane = lane_id();
  const int lo = blk_lo[b], kmin = blk_kmin[b], kn = blk_kn[b], w = blk_w[b];
  if (kn == 0) {
    if constexpr (EPI != 0) {   // no product entries in these columns
      const int j = b * J + threadIdx.x;
      if (threadIdx.x < J && j < ncols) {
        fzp->ofirst[j] = INT_MAX;
        fzp->olast[j] = -1;
        if (fzp->oplast) fzp->oplast[j] = -1;
        // (EPI 2: ... and none in the result only if the columns of X are empty as well; otherwise -- columns of X
        // whose rows name empty columns, an unsymmetric pattern -- the step is not this kernel's)
        if (EPI == 2 && fzp->xmax[j] >= fzp->xmin[j]) atomicOr(fzp->flag, 1);
      }
    }
    return;
  }
  // fused epilogues: what they read per column is staged in LDS before the loop starts (the loads overlap with the
  // other workgroups' loops; after the loop the registers are too few to hide sixteen dependent round trips) --
  // the extents of the X and D columns and the expanded D columns themselves (SLAB_DTILE doubles, checked on the host)
  __shared__ double dtile[EPI != 0 ? SLAB_DTILE : 1];
  __shared__ int cs_dmin[J], cs_dn[J], cs_dofs[J], cs_xmin[J], cs_xn[J], cs_xl[J], cs_xpl[J];
  __shared__ long long cs_xoff[J];
  if constexpr (EPI != 0) {
    const SlabFuseArgs fz0 = *fzp;
    if (threadIdx.x < J) {
      const int j = b * J + threadIdx.x, jc = min(j, ncols - 1);
      const int df = fz0.dmin[jc], dl = fz0.dmax[jc];
      cs_dmin[threadIdx.x] = df;
      cs_dn[threadIdx.x] = (j < ncols && dl >= df) ? dl - df + 1 : 0;
      if constexpr (EPI == 2) {
        const int xf = fz0.xmin[jc], xl = j < ncols ? fz0.xmax[jc] : -1;
        cs_xmin[threadIdx.x] = xf;
        cs_xl[threadIdx.x] = xl;
        // position of the last entry of X(:, j) in the order the rules are stated in: its row, or its largest label
        cs_xpl[threadIdx.x] = (fz0.lab && j < ncols && xl >= xf) ? fz0.xplast[jc] : xl;
        cs_xn[threadIdx.x] = xl >= xf ? xl - xf + 1 : 0;
        cs_xoff[threadIdx.x] = fz0.xoff[jc];
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int ofs = 0;
      for (int jj = 0; jj < J; ++jj) { cs_dofs[jj] = ofs; ofs += cs_dn[jj]; }
    }
    __syncthreads();
    for (int jj = 0; jj < J; ++jj) {
      const int dn = cs_dn[jj], ofs = cs_dofs[jj];
      const double* src = fz0.dexp + fz0.doff[min(b * J + jj, ncols - 1)];
      for (int i = threadIdx.x; i < dn; i += NW * WAVE) dtile[ofs + i] = src[i];
    }
    if (fz0.prod) {   // products of this block = sum over the tile rows k of (non-zero multipliers) * (entries of A(:, k));
                      // always run when the operand is in slab form: reading the tile here also pulls it into L2 before
                      // the loop asks for its rows one by one (2.5 % of the step; without in_count the sum is not used)
      const double* __restrict__ tile = bblk + blk_boff[b];
      long long p = 0;
      for (int k = threadIdx.x; k < kn; k += NW * WAVE) {
        int c = 0;
#pragma unroll
        for (int q = 0; q < J; ++q) c += tile[(size_t)k * J + q] != 0.0 ? 1 : 0;
        p += (long long)c * (fz0.in_count ? fz0.in_count[fz0.steps ? fz0.steps[blk_boff[b] / J + 4 * (int64_t)b + k] : kmin + k] : 1);
      }
      p = wave_sum_i64(p);
      __shared__ long long prod_s[NW];
      if (lane == 0) prod_s[wave] = p;
      __syncthreads();
      if (threadIdx.x == 0) {
        long long t = 0;
        for (int q = 0; q < NW; ++q) t += prod_s[q];
        fz0.prod[b] = t;
      }
    }
  }
  const int rbase = lo + WAVE * wave;  // slab s of this wave starts at row rbase + 64*NW*s
  const SlabRun* rp = runs + kmin;        // record of step kk: rp[kk]
  if constexpr (EPI != 0) {               // (label-ordered steps: the block's own record list)
    const SlabRun* br = fzp->blkruns;
    if (br) rp = br + (blk_boff[b] / J + 4 * (int64_t)b);
  }
  const double* bq = bblk + (MODE == 5 ? 0 : blk_boff[b]);  // multipliers of step kk: bq[kk*J .. kk*J+J)
  const unsigned r0 = (unsigned)(rbase + lane) * 8u;
  const int e0 = rbase + WAVE - 1, e1 = e0 + WAVE * NW, e2 = e1 + WAVE * NW;
  v8d accL0, accH0, accL1, accH1, accL2, accH2;
#define SLAB_LOOP_OPERANDS                                                                                        \
  : "=&{v[2:17]}"(accL0), "=&{v[18:33]}"(accH0), "=&{v[34:49]}"(accL1), "=&{v[50:65]}"(accH1),                   \
    "=&{v[66:81]}"(accL2), "=&{v[82:97]}"(accH2)                                                                   \
  : [rp] "s"(rp), [bq] "s"(bq), [kn] "s"(kn), [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [r0] "v"(r0),            \
    [c1] "n"(WAVE * NW * 8), [c2] "n"(2 * WAVE * NW * 8), [wv] "s"(wave)                                         \
  : SLAB_LOOP_CLOBBERS
  if constexpr (MODE == 1) {
    asm volatile(SLAB_LOOP_ASM_FMA SLAB_LOOP_OPERANDS);  // option spgemm_fma: one rounding per product (v_fma_f64)
#ifdef NTP_ABLATIONS
  } else if constexpr (MODE == 2) {
    asm volatile(SLAB_LOOP_ASM_ABL1 SLAB_LOOP_OPERANDS);  // no slab loads
  } else if constexpr (MODE == 3) {
    asm volatile(SLAB_LOOP_ASM_ABL2 SLAB_LOOP_OPERANDS);  // no multiplier loads
  } else if constexpr (MODE == 4) {
    asm volatile(SLAB_LOOP_ASM_ABL3 SLAB_LOOP_OPERANDS);  // no arithmetic
#endif
  } else if constexpr (MODE == 9) {
    // label-ordered steps: the multiplier row of a step sits at the byte offset its run record names
    asm volatile(SLAB_LOOP_ASM_ROWOFF
                 : "=&{v[2:17]}"(accL0), "=&{v[18:33]}"(accH0), "=&{v[34:49]}"(accL1), "=&{v[50:65]}"(accH1),
                   "=&{v[66:81]}"(accL2), "=&{v[82:97]}"(accH2)
                 : [rp] "s"(rp), [bq] "s"(bq), [kn] "s"(kn), [e0] "s"(e0), [e1] "s"(e1), [e2] "s"(e2), [r0] "v"(r0),
                   [c1] "n"(WAVE * NW * 8), [c2] "n"(2 * WAVE * NW * 8), [wv] "s"(wave)
                 : SLAB_LOOP_ROWOFF_CLOBBERS);
  } else if constexpr (MODE == 8) {
    asm volatile(SLAB_LOOP_ASM_LEANPF SLAB_LOOP_OPERANDS);  // lean periods + rotating scalar-cache prefetch
  } else if constexpr (MODE == 7) {
    asm volatile(SLAB_LOOP_ASM_LEAN SLAB_LOOP_OPERANDS);  // whole periods without pointer arithmetic / exit tests, then the plain loop
  } else if constexpr (MODE == 6) {
    asm volatile(SLAB_LOOP_ASM_PF SLAB_LOOP_OPERANDS);  // unfused, with the rotating scalar-cache prefetch of the fused loop
  } else {
    asm volatile(SLAB_LOOP_ASM SLAB_LOOP_OPERANDS);  // MODE 5 (experiment): every block reads the first tile (cache-hot multipliers)
  }
#undef SLAB_LOOP_OPERANDS
  double acc[SL][J];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) {
    acc[0][jj] = accL0[jj]; acc[0][jj + 8] = accH0[jj];
    acc[1][jj] = accL1[jj]; acc[1][jj + 8] = accH1[jj];
    acc[2][jj] = accL2[jj]; acc[2][jj + 8] = accH2[jj];
  }

  if constexpr (EPI != 0) {
    // ---- fused epilogues (see above).  The arguments come from memory only now: held in SGPRs across the loop they
    // would not fit beside the registers the loop owns.  The code below is branch-free per element and walks the
    // columns one at a time (fences keep the per-column scalars from being hoisted together): the 96 partial sums
    // leave the compiler very few registers of either kind.
    asm volatile("" ::: "memory");
    const SlabFuseArgs fz = *fzp;
    __shared__ int amax_s[NW * SL][J], first_s[NW * SL][J], plast_s[NW * SL][J];
    __shared__ int amax_f[J], col_first[J], col_last[J];
    __shared__ double red_s[2 * NW];
    __shared__ long long pn_s[NW];
    // (lane s * J + jj of amaxv / cntv keeps the scalar of slab s, column jj)
    const bool labelled = fz.lab != nullptr;   // positions are labels, not rows (label-ordered steps, SlabFuseArgs)
    int prow[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const int r = lo + WAVE * (wave + NW * s) + lane;
      prow[s] = labelled ? fz.lab[min(r, ncols - 1)] : r;
    }
    if constexpr (EPI == 2) {   // first pass: last kept row of every product column, entries of the product
      int pn = 0, amaxv = -1;
#pragma unroll
      for (int jj = 0; jj < J; ++jj) {
        int lmax = -1;   // (labelled: this lane's largest kept label of the column, reduced over the wave once per column)
#pragma unroll
        for (int s = 0; s < SL; ++s) {
          const double v = acc[s][jj];
          const double sv = __dmul_rn(alpha, v);
          const bool ha = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
          const unsigned long long m = __ballot(ha);
          pn += __popcll(m);
          const int last = m ? lo + WAVE * (wave + NW * s) + 63 - __clzll((long long)m) : -1;
          if (!labelled) asm("v_writelane_b32 %0, %1, %2" : "+v"(amaxv) : "s"(last), "n"(s * J + jj));
          lmax = max(lmax, ha ? prow[s] : -1);
        }
        if (labelled) {
          const int last = uni_i32(wave_max_i32(lmax));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(amaxv) : "s"(last), "n"(jj));
        }
      }
      if (lane < SL * J) amax_s[wave + NW * (lane / J)][lane % J] = amaxv;
      if (lane == 0) pn_s[wave] = pn;
      __syncthreads();
      if (threadIdx.x < J) {
        int mx = -1;
        for (int m = 0; m < NW * SL; ++m) mx = max(mx, amax_s[m][threadIdx.x]);
        amax_f[threadIdx.x] = mx;
        const int j = b * J + threadIdx.x;
        if (j < ncols) {  // every stored row of X(:, j) must be a row of this block's window
          const int f = cs_xmin[threadIdx.x], l = cs_xl[threadIdx.x];
          if (l >= f && (f < lo || l >= lo + w)) atomicOr(fz.flag, 1);
        }
      }
      if (threadIdx.x == 0) {
        long long t = 0;
        for (int q = 0; q < NW; ++q) t += pn_s[q];
        fz.pnnz[b] = t;
      }
      __syncthreads();
    }
    double dsum = 0.0, tsum = 0.0;
    int cntv = 0, firstv = INT_MAX, lastv = -1;   // (lane s * J + jj: entries / first / last kept row of that slab and column)
    int plastv = -1;                              // (... and, label-ordered, the largest label among the kept rows)
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      __builtin_amdgcn_sched_barrier(0);
      const int j = b * J + jj;
      int xf = 0, xn = 0, xl = -1, amax = -1;   // xn: rows of the run of X(:, j), 0 = empty
      const double* xcol = fz.xexp;
      if constexpr (EPI == 2) {
        xf = cs_xmin[jj];
        xl = cs_xpl[jj];      // (position of the column's last entry: row or label, as prow)
        xn = cs_xn[jj];
        xcol = fz.xexp + cs_xoff[jj];
        amax = amax_f[jj];
      }
      const int df = cs_dmin[jj], dn = cs_dn[jj];
      const double* dcol = dtile + cs_dofs[jj];
      int lmaxk = -1;
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        const int r = lo + WAVE * (wave + NW * s) + lane;
        double v = acc[s][jj];
        asm volatile("" : "+v"(v));   // (recomputed, not 48 lane masks carried over from the first pass)
        const double sv = __dmul_rn(alpha, v);
        const bool ha = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
        bool keep;
        double o;
        if constexpr (EPI == 1) {
          keep = ha;
          o = sv;
        } else {
          const unsigned off = (unsigned)(r - xf);
          const bool inb = off < (unsigned)xn;
          const double braw = xcol[inb ? off : 0u];    // (an empty column reads the first word of its neighbour)
          const double bv = inb ? braw : 0.0;
          const bool hb = bv != 0.0;
          const double bs = __dmul_rn(fz.bm, bv);
          const double wa = __dmul_rn(fz.am, sv);
          const double both = __dadd_rn(wa, bs);
          o = ha ? (hb ? both : wa) : bs;                       // (neither: bs = 0)
          const bool tail = ha ? (!hb && prow[s] > xl) : (prow[s] > amax);   // the rest of one column beyond the other's end
          keep = (ha || hb) && (tail || fabs(o) > fz.thr_m);
        }
        const unsigned doffs = (unsigned)(r - df);
        const bool ind = doffs < (unsigned)dn;
        const double draw = dcol[ind ? doffs : 0u];
        const double dv = (ind && keep) ? draw : 0.0;
        dsum = __dadd_rn(dsum, __dmul_rn(keep ? o : 0.0, dv));
        tsum = __dadd_rn(tsum, (keep && r == j + fz.col_offset) ? o : 0.0);
        acc[s][jj] = keep ? o : 0.0;   // the slab form of the result: a zero is "no entry"
        const unsigned long long m = __ballot(keep);
        {
          const int pc = (int)__popcll(m);
          const int row0 = lo + WAVE * (wave + NW * s);
          const int fr = m ? row0 + (int)__builtin_ctzll(m) : INT_MAX, lr = m ? row0 + 63 - __clzll((long long)m) : -1;
          asm("v_writelane_b32 %0, %1, %2" : "+v"(cntv) : "s"(pc), "n"(s * J + jj));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(firstv) : "s"(fr), "n"(s * J + jj));
          asm("v_writelane_b32 %0, %1, %2" : "+v"(lastv) : "s"(lr), "n"(s * J + jj));
        }
        lmaxk = max(lmaxk, keep ? prow[s] : -1);
      }
      if (labelled) {   // the largest label among the kept rows of the column: one reduction over the wave per column
        const int pl = uni_i32(wave_max_i32(lmaxk));
        asm("v_writelane_b32 %0, %1, %2" : "+v"(plastv) : "s"(pl), "n"(jj));
      }
      // (everything of this column is consumed here: its lane masks and loaded values do not outlive it)
      asm volatile("" : "+v"(dsum), "+v"(tsum), "+v"(cntv), "+v"(firstv), "+v"(lastv), "+v"(plastv));
    }
    asm volatile("" ::: "memory");
    if (lane < SL * J) {
      cnt_s[wave + NW * (lane / J)][lane % J] = cntv;
      first_s[wave + NW * (lane / J)][lane % J] = firstv;
      amax_s[wave + NW * (lane / J)][lane % J] = lastv;    // (the array of the first pass, free again)
      plast_s[wave + NW * (lane / J)][lane % J] = plastv;
    }
    dsum = wave_sum_f64(dsum);
    tsum = wave_sum_f64(tsum);
    if (lane == 0) { red_s[2 * wave] = dsum; red_s[2 * wave + 1] = tsum; }
    __syncthreads();
    if (threadIdx.x < J) {   // entries, first and last row of every column of the block
      int run = 0, cf = INT_MAX, cl = -1, pl = -1;
      for (int m = 0; m < NW * SL; ++m) {
        run += cnt_s[m][threadIdx.x];
        cf = min(cf, first_s[m][threadIdx.x]);
        cl = max(cl, amax_s[m][threadIdx.x]);
        pl = max(pl, plast_s[m][threadIdx.x]);
      }
      col_first[threadIdx.x] = cf;
      col_last[threadIdx.x] = cl;
      if constexpr (EPI == 1) amax_f[threadIdx.x] = run;   // (the product is the result: its entries per column)
      const int j = b * J + threadIdx.x;
      if (j < ncols) {
        count[j] = run;
        fz.ofirst[j] = cf;
        fz.olast[j] = cl;
        if (labelled) fz.oplast[j] = pl;
      }
    }
    if (threadIdx.x == 64) {
      double x = 0.0, y = 0.0;
      for (int q = 0; q < NW; ++q) { x = __dadd_rn(x, red_s[2 * q]); y = __dadd_rn(y, red_s[2 * q + 1]); }
      fz.part[2 * b] = x;
      fz.part[2 * b + 1] = y;
    }
    __syncthreads();
    if constexpr (EPI == 1) {
      if (threadIdx.x == 0) {
        long long t = 0;
        for (int jj = 0; jj < J; ++jj) t += amax_f[jj];
        fz.pnnz[b] = t;
      }
    }
    // ---- the result in slab form (SlabForm): every column as a dense run, the block as a row-major tile
    const int64_t tbase = blk_toff[b];
    int tk0 = INT_MAX, tk1 = -1;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      tk0 = min(tk0, col_first[jj]);
      tk1 = max(tk1, col_last[jj]);
    }
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int cf = col_first[jj], cn = col_last[jj] - cf;   // (empty: cn < 0)
      double* __restrict__ dst = out_val + tbase + (int64_t)jj * w;
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        const unsigned o = (unsigned)(lo + WAVE * (wave + NW * s) + lane - cf);
        if (cn >= 0 && o <= (unsigned)cn) dst[o] = acc[s][jj];
      }
    }
    double2* __restrict__ tile = reinterpret_cast<double2*>(fz.tiles + tbase);
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const unsigned o = (unsigned)(lo + WAVE * (wave + NW * s) + lane - tk0);
      if (tk1 >= tk0 && o <= (unsigned)(tk1 - tk0)) {
#pragma unroll
        for (int q = 0; q < J / 2; ++q) tile[(size_t)o * (J / 2) + q] = make_double2(acc[s][2 * q], acc[s][2 * q + 1]);
      }
    }
    return;
  }
  // ---- epilogue: prune, count per (slab, column), prefix over slabs, write in row order
#pragma unroll
  for (int s = 0; s < SL; ++s) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const double v = acc[s][jj];
      const double sv = __dmul_rn(alpha, v);
      const bool keep = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
      const unsigned long long m = __ballot(keep);
      if (lane == 0) cnt_s[wave + NW * s][jj] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < J) {
    int run = 0;
    for (int m = 0; m < NW * SL; ++m) {
      const int c = cnt_s[m][threadIdx.x];
      cnt_s[m][threadIdx.x] = run;
      run += c;
    }
    const int j = b * J + threadIdx.x;
    if (j < ncols) count[j] = run;
  }
  __syncthreads();
  const int64_t tbase = blk_toff[b];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int r = lo + WAVE * (wave + NW * s) + lane;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const double v = acc[s][jj];
      const double sv = __dmul_rn(alpha, v);
      const bool keep = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = tbase + (int64_t)jj * w + cnt_s[wave + NW * s][jj] + __popcll(m & lanemask_lt());
        out_inner[pos] = r;
        out_val[pos] = sv;
      }
    }
  }
}

// Narrow row windows (<= 256 / 512 rows): the same loop with ONE / TWO slabs per wave.  A third / two thirds of the
// partial-sum registers means 6 / 5 resident waves per SIMD instead of 4, and the loop is latency bound (time ~ 1 /
// occupancy, profiles/README.md item 14).  Register maps: tools/gen_slab_asm.py geometry().
template <int SL>
struct SlabOcc;
template <>
struct SlabOcc<1> { static constexpr int waves = 6; };  // (7+ waves per SIMD would leave a wave fewer SGPRs than the loop uses)
template <>
struct SlabOcc<2> { static constexpr int waves = 5; };

template <int SL>
__global__ __launch_bounds__(4 * WAVE) __attribute__((amdgpu_waves_per_eu(SlabOcc<SL>::waves, SlabOcc<SL>::waves)))
void k_spgemm_slab_n(const SlabRun* __restrict__ runs, const double* __restrict__ bblk,
                     const int64_t* __restrict__ blk_boff, const int32_t* __restrict__ blk_kmin,
                     const int32_t* __restrict__ blk_kn, const int32_t* __restrict__ blk_lo,
                     const int32_t* __restrict__ blk_w, const int64_t* __restrict__ blk_toff,
                     int32_t* __restrict__ out_inner, double* __restrict__ out_val, int32_t* __restrict__ count,
                     double alpha, double threshold, int dense_rule, int ncols, int nblocks) {
  constexpr int J = 16, NW = 4;
  static_assert(SL == 1 || SL == 2, "register maps of slab_loop.inc");
  __shared__ int cnt_s[NW * SL][J];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = uni_i32(threadIdx.x / WAVE), lane = lane_id();
  const int lo = blk_lo[b], kmin = blk_kmin[b], kn = blk_kn[b], w = blk_w[b];
  if (kn == 0) return;
  const int rbase = lo + WAVE * wave;
  const SlabRun* rp = runs + kmin;
  const double* bq = bblk + blk_boff[b];
  const unsigned r0 = (unsigned)(rbase + lane) * 8u;
  const int e0 = rbase + WAVE - 1, e1 = e0 + WAVE * NW;
  double acc[SL][J];
  if constexpr (SL == 1) {
    v8d aL0, aH0;
    asm volatile(SLAB_LOOP_ASM_SL1
                 : "=&{v[2:17]}"(aL0), "=&{v[18:33]}"(aH0)
                 : [rp] "s"(rp), [bq] "s"(bq), [kn] "s"(kn), [e0] "s"(e0), [r0] "v"(r0)
                 : SLAB_LOOP_SL1_CLOBBERS);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) { acc[0][jj] = aL0[jj]; acc[0][jj + 8] = aH0[jj]; }
  } else {
    v8d aL0, aH0, aL1, aH1;
    asm volatile(SLAB_LOOP_ASM_SL2
                 : "=&{v[2:17]}"(aL0), "=&{v[18:33]}"(aH0), "=&{v[34:49]}"(aL1), "=&{v[50:65]}"(aH1)
                 : [rp] "s"(rp), [bq] "s"(bq), [kn] "s"(kn), [e0] "s"(e0), [e1] "s"(e1), [r0] "v"(r0),
                   [c1] "n"(WAVE * NW * 8)
                 : SLAB_LOOP_SL2_CLOBBERS);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      acc[0][jj] = aL0[jj]; acc[0][jj + 8] = aH0[jj];
      acc[SL - 1][jj] = aL1[jj]; acc[SL - 1][jj + 8] = aH1[jj];
    }
  }
  // ---- epilogue as in k_spgemm_slab: prune, count per (slab, column), prefix over slabs, write in row order
#pragma unroll
  for (int s = 0; s < SL; ++s) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const double v = acc[s][jj];
      const double sv = __dmul_rn(alpha, v);
      const bool keep = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
      const unsigned long long m = __ballot(keep);
      if (lane == 0) cnt_s[wave + NW * s][jj] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < J) {
    int run = 0;
    for (int m = 0; m < NW * SL; ++m) {
      const int c = cnt_s[m][threadIdx.x];
      cnt_s[m][threadIdx.x] = run;
      run += c;
    }
    const int j = b * J + threadIdx.x;
    if (j < ncols) count[j] = run;
  }
  __syncthreads();
  const int64_t tbase = blk_toff[b];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int r = lo + WAVE * (wave + NW * s) + lane;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const double v = acc[s][jj];
      const double sv = __dmul_rn(alpha, v);
      const bool keep = (dense_rule & 1) ? (fabs(v) > threshold) : (fabs(sv) > threshold);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = tbase + (int64_t)jj * w + cnt_s[wave + NW * s][jj] + __popcll(m & lanemask_lt());
        out_inner[pos] = r;
        out_val[pos] = sv;
      }
    }
  }
}

// Complex operands: the same design with 8 complex columns per workgroup (a multiplier set is again 32 SGPRs), two
// slabs per wave and six waves (12 slabs = the same 768-row window), one buffer_load_dwordx4 per slab and step.
// (ar + i ai)(br + i bi) is four products, one subtraction, one addition and the two accumulates, each rounded on
// its own -- the arithmetic of the reference's complex multiply-add (and of Sc<double2>::mul / add here).
template <int NW>
__global__ __launch_bounds__(NW* WAVE) __attribute__((amdgpu_waves_per_eu(5, 5))) void k_spgemm_slab_c(
    const SlabRun* __restrict__ runs, const double2* __restrict__ bblk,
    const int64_t* __restrict__ blk_boff, const int32_t* __restrict__ blk_kmin, const int32_t* __restrict__ blk_kn,
    const int32_t* __restrict__ blk_lo, const int32_t* __restrict__ blk_w, const int64_t* __restrict__ blk_toff,
    int32_t* __restrict__ out_inner, double2* __restrict__ out_val, int32_t* __restrict__ count, double alpha,
    double threshold, int dense_rule, int ncols, int nblocks) {
  constexpr int J = 8, SL = 2;
  static_assert(NW == 6 || NW == 8, "row window of 768 / 1024 rows (the register map does not depend on NW)");
  __shared__ int cnt_s[NW * SL][J];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = uni_i32(threadIdx.x / WAVE), lane = lane_id();
  const int lo = blk_lo[b], kmin = blk_kmin[b], kn = blk_kn[b], w = blk_w[b];
  if (kn == 0) return;
  const int rbase = lo + WAVE * wave;
  const SlabRun* rp = runs + kmin;
  const double2* bq = bblk + blk_boff[b];
  const unsigned r0 = (unsigned)(rbase + lane) * 16u;
  const int e0 = rbase + WAVE - 1, e1 = e0 + WAVE * NW;
  v8d accA, accB, accC, accD;  // slab 0: columns 0..3, 4..7 (re, im pairs); slab 1: the same
  asm volatile(SLAB_LOOP_ASM_CPLX
               : "=&{v[2:17]}"(accA), "=&{v[18:33]}"(accB), "=&{v[34:49]}"(accC), "=&{v[50:65]}"(accD)
               : [rp] "s"(rp), [bq] "s"(bq), [kn] "s"(kn), [e0] "s"(e0), [e1] "s"(e1), [r0] "v"(r0),
                 [c1] "n"(WAVE * NW * 16)
               : SLAB_LOOP_CPLX_CLOBBERS);
  double2 acc[SL][J];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    acc[0][c] = make_double2(accA[2 * c], accA[2 * c + 1]);
    acc[0][c + 4] = make_double2(accB[2 * c], accB[2 * c + 1]);
    acc[1][c] = make_double2(accC[2 * c], accC[2 * c + 1]);
    acc[1][c + 4] = make_double2(accD[2 * c], accD[2 * c + 1]);
  }
  using T = double2;
  unsigned keepbits = 0;  // bit s*J + jj: the magnitude test (a hypot) is evaluated once per value
#pragma unroll
  for (int s = 0; s < SL; ++s) {
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const T v = acc[s][jj];
      const bool keep = Sc<T>::mag((dense_rule & 1) ? v : Sc<T>::scale(alpha, v)) > threshold;
      const unsigned long long m = __ballot(keep);
      keepbits |= keep ? (1u << (s * J + jj)) : 0u;
      if (lane == 0) cnt_s[wave + NW * s][jj] = __popcll(m);
    }
  }
  __syncthreads();
  if (threadIdx.x < J) {
    int run = 0;
    for (int m = 0; m < NW * SL; ++m) {
      const int c = cnt_s[m][threadIdx.x];
      cnt_s[m][threadIdx.x] = run;
      run += c;
    }
    const int j = b * J + threadIdx.x;
    if (j < ncols) count[j] = run;
  }
  __syncthreads();
  const int64_t tbase = blk_toff[b];
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int r = lo + WAVE * (wave + NW * s) + lane;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const bool keep = (keepbits >> (s * J + jj)) & 1u;
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = tbase + (int64_t)jj * w + cnt_s[wave + NW * s][jj] + __popcll(m & lanemask_lt());
        out_inner[pos] = r;
        out_val[pos] = Sc<T>::scale(alpha, acc[s][jj]);
      }
    }
  }
}

// ------------------------------------------------------------------ SpGEMM: numeric, LDS hash
// For columns whose row window does not fit the direct map (e.g. after a load-balancing
// permutation).  4096 LDS buckets per wave, multiplicative hash + linear probing; bucket claim by
// LDS compare-and-swap, value update non-atomic (one lane per row inside a step).  Survivors are
// packed as (row << 13 | slot), bitonic-sorted in LDS and written in row order.  A column that
// fills more than HASH_MAX_FILL buckets is handed to the HBM accumulator kernel (bin 6).
template <typename T, int SLOTS>
__global__ __launch_bounds__(WAVE) void k_spgemm_hash(
    Csc A, Csc B, const int32_t* __restrict__ span_arr, uint8_t* __restrict__ bin_arr,
    const int64_t* __restrict__ tmpoff, int32_t* __restrict__ out_inner, T* __restrict__ out_val,
    int32_t* __restrict__ count, unsigned long long* __restrict__ stats, double alpha,
    double threshold, int dense_rule, int nblocks) {
  // One wave per output column, k ascending (the accumulation order of the reference).  The table is sized for
  // occupancy, not for the worst case: SLOTS = 1024 (12 KB for real values: a dozen waves per CU) takes every column
  // first, a column that fills more than 3/4 of it is handed to the 4096-slot pass, and from there to the HBM
  // accumulator.  Keys and values are compacted and sorted in place (bitonic on pairs), so the table is all the
  // LDS a wave needs.
  constexpr int MAX_FILL = SLOTS * 3 / 4;
  constexpr int SHIFT = SLOTS == 1024 ? 22 : 20;
  static_assert(SLOTS == 1024 || SLOTS == 4096, "table classes");
  __shared__ int keys[SLOTS];
  __shared__ T vals[SLOTS];
  const int j = xcd_block(nblocks);
  if (j < 0 || j >= B.cols) return;
  if (bin_arr[j] != (SLOTS == 1024 ? BIN_HASH : BIN_HASH_BIG)) return;
  const int lane = lane_id();
  for (int s = lane; s < SLOTS; s += WAVE) {
    keys[s] = -1;
    vals[s] = Sc<T>::zero();
  }
  __syncthreads();
  const int32_t* __restrict__ Ai = A.inner;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  int filled = 0;
  bool overflow = false;
  const int64_t bs = B.outer[j], be = B.outer[j + 1];
  for (int64_t p0 = bs; p0 < be && !overflow; p0 += WAVE) {
    // 64 entries of the B column at once: row id, value and the extent of the A column it names
    const bool have = p0 + lane < be;
    const int kk = have ? B.inner[p0 + lane] : 0;
    const T bvv = have ? Bv[p0 + lane] : Sc<T>::zero();
    const int64_t a0 = have ? A.outer[kk] : 0;
    const int alen = have ? (int)(A.outer[kk + 1] - a0) : 0;
    const int m = (int)min((int64_t)WAVE, be - p0);
    for (int t = 0; t < m && !overflow; ++t) {
      const int64_t as = readlane_i64(a0, t);
      const int len = readlane_i32(alen, t);
      const T bk = readlane_T(bvv, t);
      // up to CH chunks (64 entries each) of the A column are requested before the first of them is hashed
      // (measured: 6.7 -> 6.2 ms banded, 13.1 -> 10.7 ms permuted; requesting the next column as well adds nothing)
      constexpr int CH = 5;
      for (int q0 = 0; q0 < len && !overflow; q0 += CH * WAVE) {
        int ci[CH];
        T ca[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          const int q = q0 + c * WAVE + lane;
          ci[c] = -1;
          ca[c] = Sc<T>::zero();
          if (q < len) {
            ci[c] = Ai[as + q];
            ca[c] = Av[as + q];
          }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          if (q0 + c * WAVE >= len) break;
          bool fresh = false;
          if (ci[c] >= 0) {
            const int i = ci[c];
            int h = (int)(((unsigned)i * 2654435761u) >> SHIFT) & (SLOTS - 1);
            for (;;) {
              const int old = atomicCAS(&keys[h], -1, i);
              if (old == -1) { fresh = true; break; }
              if (old == i) break;
              h = (h + 1) & (SLOTS - 1);
            }
            vals[h] = Sc<T>::fmadd(ca[c], bk, vals[h], (dense_rule & 2) != 0);
          }
          filled += __popcll(__ballot(fresh));
          __builtin_amdgcn_wave_barrier();
          if (filled > MAX_FILL) { overflow = true; break; }
        }
      }
    }
  }
  if (overflow) {
    if (lane == 0) {
      if (SLOTS == 1024) {
        bin_arr[j] = BIN_HASH_BIG;
        atomicAdd(&stats[19], 1ull);
      } else {
        bin_arr[j] = BIN_HBM;
        atomicAdd(&stats[8], 1ull);
      }
      count[j] = 0;
    }
    return;
  }
  __syncthreads();
  // occupied buckets to the front, in place (a chunk is read into registers before anything of it is overwritten,
  // and the write position never passes the read position)
  int n = 0;
  for (int s0 = 0; s0 < SLOTS; s0 += WAVE) {
    const int s = s0 + lane;
    const int key = keys[s];
    const T v = vals[s];
    const bool occ = key >= 0;
    const unsigned long long mk = __ballot(occ);
    __syncthreads();
    if (occ) {
      const int d = n + __popcll(mk & lanemask_lt());
      keys[d] = key;
      vals[d] = v;
    }
    n += __popcll(mk);
    __syncthreads();
  }
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  for (int s = n + lane; s < n2; s += WAVE) keys[s] = INT_MAX;
  __syncthreads();
  for (int kk2 = 2; kk2 <= n2; kk2 <<= 1) {
    for (int jj = kk2 >> 1; jj > 0; jj >>= 1) {
      for (int t = lane; t < n2; t += WAVE) {
        const int ixj = t ^ jj;
        if (ixj > t) {
          const int x = keys[t], y = keys[ixj];
          const bool up = (t & kk2) == 0;
          if ((x > y) == up) {
            keys[t] = y;
            keys[ixj] = x;
            const T vx = vals[t];
            vals[t] = vals[ixj];
            vals[ixj] = vx;
          }
        }
      }
      __syncthreads();
    }
  }
  const int64_t base = tmpoff[j];
  int cnt = 0;
  for (int s0 = 0; s0 < n; s0 += WAVE) {
    const int s = s0 + lane;
    const bool in = s < n;
    const int row = in ? keys[s] : 0;
    const T v = in ? vals[s] : Sc<T>::zero();
    const T sv = Sc<T>::scale(alpha, v);
    const bool keep = in && ((dense_rule & 1) ? (Sc<T>::mag(v) > threshold) : (Sc<T>::mag(sv) > threshold));
    const unsigned long long mk = __ballot(keep);
    if (keep) {
      const int64_t pos = base + cnt + __popcll(mk & lanemask_lt());
      out_inner[pos] = row;
      out_val[pos] = sv;
    }
    cnt += __popcll(mk);
  }
  if (lane == 0) count[j] = cnt;
}

// ------------------------------------------------------------------ SpGEMM: numeric, HBM accumulator
// Last resort for very long columns: each persistent wave owns a dense accumulator of `rows`
// entries in HBM (what the reference allocates for EVERY column).  Updates and the final scan
// go through L2 (agent-scope atomics / sc1 loads) so that the wave always sees its own previous
// step.  tmpoff2 gives the column's slot in the second temporary region.
template <typename T>
__global__ __launch_bounds__(WAVE) void k_spgemm_hbm(
    Csc A, Csc B, const int32_t* __restrict__ lo_arr, const int32_t* __restrict__ span_arr,
    const uint8_t* __restrict__ bin_arr, const int64_t* __restrict__ tmpoff2,
    int32_t* __restrict__ out_inner, T* __restrict__ out_val, int32_t* __restrict__ count,
    double* __restrict__ workspace, double alpha, double threshold, int dense_rule) {
  constexpr int WV = Sc<T>::cplx ? 2 : 1;
  double* acc = workspace + (size_t)blockIdx.x * (size_t)A.rows * WV;
  const int lane = lane_id();
  const int32_t* __restrict__ Ai = A.inner;
  const double* __restrict__ Av = static_cast<const double*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  for (int j = blockIdx.x; j < B.cols; j += gridDim.x) {
    if (bin_arr[j] != BIN_HBM) continue;
    const int lo = lo_arr[j], span = span_arr[j];
    for (int64_t p = B.outer[j]; p < B.outer[j + 1]; ++p) {
      const int k = B.inner[p];
      const T bk = Bv[p];
      const int64_t as = A.outer[k];
      const int len = (int)(A.outer[k + 1] - as);
      for (int q = lane; q < len; q += WAVE) {
        const int i = Ai[as + q];
        if constexpr (Sc<T>::cplx) {
          const double2 a = make_double2(Av[2 * (as + q)], Av[2 * (as + q) + 1]);
          const double2 pr = Sc<double2>::mul(a, bk);
          double* slot = acc + 2 * (size_t)i;
          const double ox = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double oy = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot, __dadd_rn(ox, pr.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot + 1, __dadd_rn(oy, pr.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          double* slot = acc + (size_t)i;
          const double o = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot, Sc<double>::fmadd(Av[as + q], bk, o, (dense_rule & 2) != 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // all of this step's stores must have reached L2 before the next step reads them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_wave_barrier();
    }
    const int64_t base = tmpoff2[j];
    int cnt = 0;
    for (int s0 = 0; s0 < span; s0 += WAVE) {
      const int s = s0 + lane;
      const bool in = s < span;
      T v = Sc<T>::zero();
      if (in) {
        double* slot = acc + (size_t)(lo + s) * WV;
        if constexpr (Sc<T>::cplx) {
          v = make_double2(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
          __hip_atomic_store(slot, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot + 1, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
          v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(slot, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      const T sv = Sc<T>::scale(alpha, v);
      const bool keep = in && ((dense_rule & 1) ? (Sc<T>::mag(v) > threshold) : (Sc<T>::mag(sv) > threshold));
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = base + cnt + __popcll(m & lanemask_lt());
        out_inner[pos] = lo + s;
        out_val[pos] = sv;
      }
      cnt += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    if (lane == 0) count[j] = cnt;
  }
}

__global__ void k_hbm_ub(const uint8_t* __restrict__ bin_arr, const int32_t* __restrict__ span_arr,
                         const int64_t* __restrict__ ip_arr, int64_t* __restrict__ ub2, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  ub2[j] = (bin_arr[j] == BIN_HBM) ? min((int64_t)span_arr[j], ip_arr[j]) : 0;
}

// copy the kept entries of every column from its temporary slot to its final place
// (one wave per column, coalesced)
template <typename T>
__global__ __launch_bounds__(256) void k_compact(int ncols, const int64_t* __restrict__ srcoff,
                                                 const int64_t* __restrict__ srcoff2,
                                                 const uint8_t* __restrict__ sel2,
                                                 const int64_t* __restrict__ dstoff,
                                                 const int32_t* __restrict__ src_inner,
                                                 const T* __restrict__ src_val,
                                                 const int32_t* __restrict__ src2_inner,
                                                 const T* __restrict__ src2_val,
                                                 int32_t* __restrict__ dst_inner,
                                                 T* __restrict__ dst_val, int nblocks) {
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int j = b * 4 + threadIdx.x / WAVE;
  if (j >= ncols) return;
  const int lane = lane_id();
  const int64_t d0 = dstoff[j];
  const int n = (int)(dstoff[j + 1] - d0);
  const bool second = sel2 && sel2[j] == BIN_HBM;
  const int64_t s0 = second ? srcoff2[j] : srcoff[j];
  const int32_t* si = second ? src2_inner : src_inner;
  const T* sv = second ? src2_val : src_val;
  for (int t = lane; t < n; t += WAVE) {
    dst_inner[d0 + t] = si[s0 + t];
    dst_val[d0 + t] = sv[s0 + t];
  }
}

// ------------------------------------------------------------------ increment (sparse AXPY)
// B <- alpha*A + B, one wave per column, AddSparseVectors rules
// (sparse_includes/AddSparseVectors.f90:24-68): in the region where both columns still have
// entries an element is kept only if |value| > threshold; once one column is exhausted the rest
// of the other is copied unfiltered.  "Exhausted" is a comparison with the other column's last
// row, so a direct-mapped LDS window over the union row range decides every slot independently.
__global__ void k_inc_plan(Csc A, Csc B, int32_t* __restrict__ lo_arr, int32_t* __restrict__ span_arr,
                           uint8_t* __restrict__ bin_arr, unsigned long long* __restrict__ stats,
                           int32_t* __restrict__ count, int force_seq) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 16) stats[threadIdx.x] = 0ull;  // the histogram kernel that follows adds into it
  if (j >= A.cols) return;
  count[j] = 0;  // columns of the empty bin are never visited by a merge kernel
  const int64_t as = A.outer[j], ae = col_end(A, j), bs = B.outer[j], be = col_end(B, j);
  int lo = INT_MAX, hi = -1;
  if (ae > as) { lo = min(lo, A.inner[as]); hi = max(hi, A.inner[ae - 1]); }
  if (be > bs) { lo = min(lo, B.inner[bs]); hi = max(hi, B.inner[be - 1]); }
  const int span = hi >= lo ? hi - lo + 1 : 0;
  // 1, 2: direct-mapped LDS windows; 3: rank merge (columns scattered over the row range, both together <= 2048
  // entries); 4: sequential two-pointer merge
  int bin = span == 0 ? 0 : span <= 512 ? 1 : span <= 2048 ? 2 : ((ae - as) + (be - bs) <= 2048 ? 3 : 4);
  if (span > 0 && force_seq) bin = force_seq == 2 ? ((ae - as) + (be - bs) <= 2048 ? 3 : 4) : 4;
  lo_arr[j] = span > 0 ? lo : 0;
  span_arr[j] = span;
  bin_arr[j] = (uint8_t)bin;
}

template <typename T>
__device__ inline bool inc_decide(bool ha, bool hb, T a, T b, int row, int amax, int bmax,
                                  double alpha, double threshold, T* out) {
  // a is the raw A value; wa = alpha*a as in AddSparseVectors.f90:28
  if (ha && hb) {
    const T s = Sc<T>::add(Sc<T>::scale(alpha, a), b);
    *out = s;
    return Sc<T>::mag(s) > threshold;
  }
  if (ha) {
    const T wa = Sc<T>::scale(alpha, a);
    *out = wa;
    return (row > bmax) ? true : (Sc<T>::mag(wa) > threshold);  // tail of A copied unfiltered (:57-62)
  }
  *out = b;
  return (row > amax) ? true : (Sc<T>::mag(b) > threshold);      // tail of B copied unfiltered (:63-68)
}

// DOT: additionally scatter column j of a third matrix D into the window and return
// sum conj(result) * D per block (fused energy evaluation of the TRS2 update).
// B's values are scaled by beta first (ScaleMatrix followed by IncrementMatrix in the reference).
template <typename T, int W, int NW, bool DOT>
__global__ __launch_bounds__(NW* WAVE) void k_inc_window(
    Csc A, Csc B, Csc D, const int32_t* __restrict__ lo_arr, const int32_t* __restrict__ span_arr,
    const uint8_t* __restrict__ bin_arr, int my_bin, int32_t* __restrict__ out_inner,
    T* __restrict__ out_val, int32_t* __restrict__ count, double alpha, double beta, double threshold,
    double* __restrict__ dot_partial, int nblocks, double* __restrict__ trace_partial, int col_offset,
    const int64_t* __restrict__ dstoff) {
  __shared__ T wa_all[NW * W];
  __shared__ T wb_all[NW * W];
  __shared__ T wd_all[DOT ? NW * W : 1];
  __shared__ uint8_t fl_all[NW * W];
  __shared__ double red[2 * NW];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = b * NW + wave;
  double dx = 0.0, dy = 0.0, dt = 0.0;
  const bool mine = (j < A.cols) && (bin_arr[j < A.cols ? j : 0] == my_bin);
  if (mine) {
    T* wa = wa_all + wave * W;
    T* wb = wb_all + wave * W;
    T* wd = wd_all + (DOT ? wave * W : 0);
    uint8_t* fl = fl_all + wave * W;
    const int span = span_arr[j], lo = lo_arr[j];
    const T* __restrict__ Av = static_cast<const T*>(A.val);
    const T* __restrict__ Bv = static_cast<const T*>(B.val);
    const T* __restrict__ Dv = static_cast<const T*>(D.val);
    const int64_t as = A.outer[j], ae = col_end(A, j), bs = B.outer[j], be = col_end(B, j);
    const int64_t ds = DOT ? D.outer[j] : 0, de = DOT ? D.outer[j + 1] : 0;
    // All operand loads of the column are requested up front (CH chunks of 64 entries per operand, reading past
    // the column end is harmless: DevMat keeps kIndexSlack entries of slack), so the column costs one memory
    // round trip instead of one per operand; longer columns finish in the loops below.
    constexpr int CH = 5;
    int ai[CH], bi[CH], di[DOT ? CH : 1];
    T av[CH], bv[CH], dv[DOT ? CH : 1];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      ai[c] = A.inner[as + c * WAVE + lane];
      av[c] = Av[as + c * WAVE + lane];
      bi[c] = B.inner[bs + c * WAVE + lane];
      bv[c] = Bv[bs + c * WAVE + lane];
      if constexpr (DOT) {
        di[c] = D.inner[ds + c * WAVE + lane];
        dv[c] = Dv[ds + c * WAVE + lane];
      }
    }
    const int amax = ae > as ? A.inner[ae - 1] : -1;
    const int bmax = be > bs ? B.inner[be - 1] : -1;
    for (int s = lane; s < span; s += WAVE) fl[s] = 0;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (as + c * WAVE + lane < ae) {
        wa[ai[c] - lo] = av[c];
        fl[ai[c] - lo] = 1;
      }
    }
    for (int64_t p = as + CH * WAVE + lane; p < ae; p += WAVE) {
      const int s = A.inner[p] - lo;
      wa[s] = Av[p];
      fl[s] = 1;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (bs + c * WAVE + lane < be) {
        wb[bi[c] - lo] = Sc<T>::scale(beta, bv[c]);
        fl[bi[c] - lo] |= 2;
      }
    }
    for (int64_t p = bs + CH * WAVE + lane; p < be; p += WAVE) {
      const int s = B.inner[p] - lo;
      wb[s] = Sc<T>::scale(beta, Bv[p]);
      fl[s] |= 2;
    }
    __builtin_amdgcn_wave_barrier();
    if constexpr (DOT) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int s = di[c] - lo;
        if (ds + c * WAVE + lane < de && s >= 0 && s < span) {
          wd[s] = dv[c];
          fl[s] |= 4;
        }
      }
      for (int64_t p = ds + CH * WAVE + lane; p < de; p += WAVE) {
        const int s = D.inner[p] - lo;
        if (s >= 0 && s < span) {
          wd[s] = Dv[p];
          fl[s] |= 4;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    const int64_t base = dstoff[j];  // upper-bound slot: every column may keep all of A and B
    int cnt = 0;
    for (int s0 = 0; s0 < span; s0 += WAVE) {
      const int s = s0 + lane;
      const int f = (s < span) ? fl[s] : 0;
      T v = Sc<T>::zero();
      bool keep = false;
      if (f & 3) keep = inc_decide<T>(f & 1, f & 2, (f & 1) ? wa[s] : Sc<T>::zero(), (f & 2) ? wb[s] : Sc<T>::zero(),
                                      lo + s, amax, bmax, alpha, threshold, &v);
      const unsigned long long m = __ballot(keep);
      if (keep) {
        const int64_t pos = base + cnt + __popcll(m & lanemask_lt());
        out_inner[pos] = lo + s;
        out_val[pos] = v;
        if constexpr (DOT) {
          if (lo + s == col_offset + j) dt = Sc<T>::re(v);  // diagonal entry of the result (at most one per column)
          if (f & 4) {
            if constexpr (Sc<T>::cplx) {
              const double2 pr = Sc<double2>::mul(Sc<double2>::conj(v), wd[s]);
              dx = __dadd_rn(dx, pr.x);
              dy = __dadd_rn(dy, pr.y);
            } else {
              dx = __dadd_rn(dx, __dmul_rn(v, wd[s]));
            }
          }
        }
      }
      cnt += __popcll(m);
    }
    if (lane == 0) count[j] = cnt;
  }
  if constexpr (DOT) {
    dx = wave_sum_f64(dx);
    dy = wave_sum_f64(dy);
    if (lane == 0) {
      red[2 * wave] = dx;
      red[2 * wave + 1] = dy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sx = 0.0, sy = 0.0;
      for (int w = 0; w < NW; ++w) {
        sx = __dadd_rn(sx, red[2 * w]);
        sy = __dadd_rn(sy, red[2 * w + 1]);
      }
      dot_partial[2 * b] = sx;
      dot_partial[2 * b + 1] = sy;
    }
    if (trace_partial) {  // block-uniform
      dt = wave_sum_f64(dt);
      __syncthreads();
      if (lane == 0) red[wave] = dt;
      __syncthreads();
      if (threadIdx.x == 0) {
        double st = 0.0;
        for (int w = 0; w < NW; ++w) st = __dadd_rn(st, red[w]);
        trace_partial[2 * b] = st;
        trace_partial[2 * b + 1] = 0.0;
      }
    }
  }
}

// Columns whose rows scatter over the whole range (operands under a load-balancing permutation): one wave per column,
// merge by RANK.  Both row lists go to LDS; every entry finds its rank in the other list by binary search, which gives
// its place p = (#A rows < r) + (#B rows < r) in the merged order (a row present in both lists gets ONE place, written
// by the B side) and decides the AddSparseVectors rule on its own (the "tail" test is a comparison with the other
// column's last row, as in the window kernel).  Kept entries are written at their merged place, a bitmap over the
// places is prefix-summed, and the column is compacted in place, 64 places per round.  With DOT the kept entries are
// also multiplied with D (binary search in its column) and the diagonal entry is picked up for the trace.
constexpr int INC_MERGE_CAP = 2048;
template <typename T, bool DOT>
__global__ __launch_bounds__(4 * WAVE) void k_inc_merge(
    Csc A, Csc B, Csc D, const uint8_t* __restrict__ bin_arr, int my_bin, int32_t* __restrict__ out_inner,
    T* __restrict__ out_val, int32_t* __restrict__ count, double alpha, double beta, double threshold,
    double* __restrict__ dot_partial, int nblocks, double* __restrict__ trace_partial, int col_offset, int row_block,
    const int64_t* __restrict__ dstoff) {
  // row_block > 0: the AddSparseVectors rule is applied per segment of `row_block` rows (the reference adds the
  // matrices block by block, so "the other column is exhausted" is decided inside each row block)
  constexpr int NW = 4, NWORD = INC_MERGE_CAP / 64, DCAP = DOT ? 1024 : 1;   // (longer columns of D are searched in memory)
  __shared__ int rows_all[NW][INC_MERGE_CAP];
  __shared__ int drows_all[NW][DCAP];
  __shared__ unsigned long long bits_all[NW][NWORD];
  __shared__ double red[2 * NW];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = b * NW + wave;
  double dx = 0.0, dy = 0.0, dt = 0.0;
  const bool mine = (j < A.cols) && (bin_arr[j < A.cols ? j : 0] == my_bin);
  if (mine) {
    const T* __restrict__ Av = static_cast<const T*>(A.val);
    const T* __restrict__ Bv = static_cast<const T*>(B.val);
    const T* __restrict__ Dv = static_cast<const T*>(D.val);
    const int64_t as = A.outer[j], ae = col_end(A, j), bs = B.outer[j], be = col_end(B, j);
    const int na = (int)(ae - as), nb = (int)(be - bs);
    int* ra = rows_all[wave];
    int* rb = ra + na;
    unsigned long long* bits = bits_all[wave];
    for (int i = lane; i < na; i += WAVE) ra[i] = A.inner[as + i];
    for (int i = lane; i < nb; i += WAVE) rb[i] = B.inner[bs + i];
    int* rd = drows_all[wave];
    const int64_t ds = DOT ? D.outer[j] : 0;
    const int nd = DOT ? (int)(D.outer[j + 1] - ds) : 0;
    if constexpr (DOT) {
      if (nd <= DCAP)
        for (int i = lane; i < nd; i += WAVE) rd[i] = D.inner[ds + i];
    }
    if (lane < NWORD) bits[lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    const int amax = na ? ra[na - 1] : -1, bmax = nb ? rb[nb - 1] : -1;
    const int64_t base = dstoff[j];
    // last row of list `l` inside the row block of r (or -1): the largest entry below the block's end, if it is in the block
    auto last_in_block = [&](const int* l, int n, int r) -> int {
      const int e = (r / row_block + 1) * row_block;
      int lo = 0, hi = n;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (l[mid] < e) lo = mid + 1;
        else hi = mid;
      }
      if (lo == 0) return -1;
      const int v = l[lo - 1];
      return v >= e - row_block ? v : -1;
    };
    // A side: entries whose row is not in B
    for (int i = lane; i < na; i += WAVE) {
      const int r = ra[i];
      int lo = 0, hi = nb;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (rb[mid] < r) lo = mid + 1;
        else hi = mid;
      }
      if (lo < nb && rb[lo] == r) continue;
      const T v = Sc<T>::scale(alpha, Av[as + i]);
      const int blast = row_block > 0 ? last_in_block(rb, nb, r) : bmax;
      if (r > blast || Sc<T>::mag(v) > threshold) {
        const int p = i + lo;
        out_inner[base + p] = r;
        out_val[base + p] = v;
        atomicOr(&bits[p >> 6], 1ull << (p & 63));
      }
    }
    // B side (scaled first, as AddSparseVectors sees it), with the matching A entry where there is one
    for (int i = lane; i < nb; i += WAVE) {
      const int r = rb[i];
      int lo = 0, hi = na;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ra[mid] < r) lo = mid + 1;
        else hi = mid;
      }
      const bool both = lo < na && ra[lo] == r;
      const T wb = Sc<T>::scale(beta, Bv[bs + i]);
      T v;
      bool keep;
      if (both) {
        v = Sc<T>::add(Sc<T>::scale(alpha, Av[as + lo]), wb);
        keep = Sc<T>::mag(v) > threshold;
      } else {
        v = wb;
        const int alast = row_block > 0 ? last_in_block(ra, na, r) : amax;
        keep = r > alast || Sc<T>::mag(v) > threshold;
      }
      if (keep) {
        const int p = i + lo;
        out_inner[base + p] = r;
        out_val[base + p] = v;
        atomicOr(&bits[p >> 6], 1ull << (p & 63));
      }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();   // the entries written above are read back by other lanes below
    // compaction in place: round k moves places [64k, 64k + 64) down to [pre_k, ...)
    const int nplace = na + nb;
    int run = 0;
    for (int k = 0; k * 64 < nplace; ++k) {
      const unsigned long long w = bits[k];
      const bool kept = (w >> lane) & 1ull;
      int r = 0;
      T v = Sc<T>::zero();
      if (kept) {
        r = out_inner[base + k * 64 + lane];
        v = out_val[base + k * 64 + lane];
      }
      __builtin_amdgcn_wave_barrier();
      if (kept) {
        const int64_t pos = base + run + __popcll(w & lanemask_lt());
        out_inner[pos] = r;
        out_val[pos] = v;
        if constexpr (DOT) {
          if (r == col_offset + j) dt = Sc<T>::re(v);
          int lo = 0, hi = nd;
          if (nd <= DCAP) {
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (rd[mid] < r) lo = mid + 1;
              else hi = mid;
            }
          } else {
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (D.inner[ds + mid] < r) lo = mid + 1;
              else hi = mid;
            }
          }
          if (lo < nd && (nd <= DCAP ? rd[lo] : D.inner[ds + lo]) == r) {
            if constexpr (Sc<T>::cplx) {
              const double2 pr = Sc<double2>::mul(Sc<double2>::conj(v), Dv[ds + lo]);
              dx = __dadd_rn(dx, pr.x);
              dy = __dadd_rn(dy, pr.y);
            } else {
              dx = __dadd_rn(dx, __dmul_rn(v, Dv[ds + lo]));
            }
          }
        }
      }
      run += __popcll(w);
      __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) count[j] = run;
  }
  if constexpr (DOT) {
    dx = wave_sum_f64(dx);
    dy = wave_sum_f64(dy);
    if (lane == 0) {
      red[2 * wave] = dx;
      red[2 * wave + 1] = dy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double sx = 0.0, sy = 0.0;
      for (int w = 0; w < NW; ++w) {
        sx = __dadd_rn(sx, red[2 * w]);
        sy = __dadd_rn(sy, red[2 * w + 1]);
      }
      dot_partial[2 * b] = sx;
      dot_partial[2 * b + 1] = sy;
    }
    if (trace_partial) {  // block-uniform
      dt = wave_sum_f64(dt);
      __syncthreads();
      if (lane == 0) red[wave] = dt;
      __syncthreads();
      if (threadIdx.x == 0) {
        double st = 0.0;
        for (int w = 0; w < NW; ++w) st = __dadd_rn(st, red[w]);
        trace_partial[2 * b] = st;
        trace_partial[2 * b + 1] = 0.0;
      }
    }
  }
}

// sequential two-pointer merge, one thread per column (columns wider than the LDS window)
template <typename T>
__global__ void k_inc_seq(Csc A, Csc B, const uint8_t* __restrict__ bin_arr, int my_bin,
                          int32_t* __restrict__ out_inner, T* __restrict__ out_val,
                          int32_t* __restrict__ count, double alpha, double beta, double threshold, int row_block,
                          const int64_t* __restrict__ dstoff) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.cols) return;
  if (bin_arr[j] != my_bin) return;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  int64_t aa = A.outer[j], bb = B.outer[j];
  const int64_t ea_all = col_end(A, j), eb_all = col_end(B, j);
  int64_t cc = dstoff[j];
  const int64_t c0 = cc;
  // (row_block > 0: the merge runs row block by row block, as the reference adds block by block)
  for (int seg_end = row_block > 0 ? row_block : INT_MAX; aa < ea_all || bb < eb_all;
       seg_end = row_block > 0 ? seg_end + row_block : INT_MAX) {
  int64_t ea = aa, eb = bb;
  if (row_block > 0) {
    while (ea < ea_all && A.inner[ea] < seg_end) ++ea;
    while (eb < eb_all && B.inner[eb] < seg_end) ++eb;
  } else {
    ea = ea_all;
    eb = eb_all;
  }
  while (aa < ea && bb < eb) {
    const int ia = A.inner[aa], ib = B.inner[bb];
    if (ia == ib) {
      const T s = Sc<T>::add(Sc<T>::scale(alpha, Av[aa]), Sc<T>::scale(beta, Bv[bb]));
      if (Sc<T>::mag(s) > threshold) { out_inner[cc] = ia; out_val[cc] = s; ++cc; }
      ++aa; ++bb;
    } else if (ia > ib) {
      const T w = Sc<T>::scale(beta, Bv[bb]);
      if (Sc<T>::mag(w) > threshold) { out_inner[cc] = ib; out_val[cc] = w; ++cc; }
      ++bb;
    } else {
      const T w = Sc<T>::scale(alpha, Av[aa]);
      if (Sc<T>::mag(w) > threshold) { out_inner[cc] = ia; out_val[cc] = w; ++cc; }
      ++aa;
    }
  }
  for (; aa < ea; ++aa) { out_inner[cc] = A.inner[aa]; out_val[cc] = Sc<T>::scale(alpha, Av[aa]); ++cc; }
  for (; bb < eb; ++bb) { out_inner[cc] = B.inner[bb]; out_val[cc] = Sc<T>::scale(beta, Bv[bb]); ++cc; }
  }
  count[j] = (int32_t)(cc - c0);
}

__global__ void k_sum_outer(const int64_t* __restrict__ a, const int64_t* __restrict__ b,
                            int64_t* __restrict__ out, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= n) out[j] = a[j] + b[j];
}
__global__ void k_pick2_i64(const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t* __restrict__ out) {
  out[0] = a[0];
  out[1] = b ? b[0] : 0;
}
// entries of column j of A plus those of column j of B (either may be loose): the tight output slots of a merge
__global__ void k_sum_counts(Csc A, Csc B, int32_t* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < A.cols) out[j] = (int32_t)(col_end(A, j) - A.outer[j]) + (int32_t)(col_end(B, j) - B.outer[j]);
}

// ------------------------------------------------------------------ dot / pairwise / trace / norms
// sum_j sum_i conj(A_ij) B_ij.  One wave per column: lanes stride over A's column and find the
// partner in B's column by binary search (B's column is hot in L1/L2: it is read log2 times by
// neighbouring lanes).  Per-wave partials are reduced by k_reduce_sum2 in a fixed order.
// sum conj(A) .* B: column j of B is scattered into a direct-mapped LDS window over its row range (like the
// SpGEMM / increment windows), column j of A probes it.  Columns of B wider than the window use a binary search.
template <typename T>
__global__ __launch_bounds__(256) void k_dot(Csc A, Csc B, double* __restrict__ partial, int nblocks,
                                             double* __restrict__ trace_partial, int col_offset) {
  constexpr int W = 1024, CH = 5;
  __shared__ T win_all[4 * W];
  __shared__ uint8_t fl_all[4 * W];
  __shared__ double sx[4], sy[4];
  const int b = blockIdx.x;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  T* win = win_all + wave * W;
  uint8_t* fl = fl_all + wave * W;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  double x = 0, y = 0, tr = 0;  // tr: trace of A (diagonal = row col_offset + j of column j), on request
  auto accum = [&](T a, T bval) {
    if constexpr (Sc<T>::cplx) {
      const double2 pr = Sc<double2>::mul(Sc<double2>::conj(a), bval);
      x = __dadd_rn(x, pr.x);
      y = __dadd_rn(y, pr.y);
    } else {
      x = __dadd_rn(x, __dmul_rn(a, bval));
    }
  };
  for (int j = b * 4 + wave; j < A.cols; j += nblocks * 4) {
    const int64_t bs = B.outer[j], be = B.outer[j + 1], as = A.outer[j], ae = col_end(A, j);
    if (ae == as || (be == bs && !trace_partial)) continue;
    int ai[CH], bi[CH];
    T av[CH], bv[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {  // over-read past the column end stays inside the DevMat slack
      bi[c] = B.inner[bs + c * WAVE + lane];
      bv[c] = Bv[bs + c * WAVE + lane];
      ai[c] = A.inner[as + c * WAVE + lane];
      av[c] = Av[as + c * WAVE + lane];
    }
    const int lo = B.inner[bs], span = B.inner[be - 1] - lo + 1;
    if (span <= W) {
      for (int s = lane; s < span; s += WAVE) fl[s] = 0;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        if (bs + c * WAVE + lane < be) {
          win[bi[c] - lo] = bv[c];
          fl[bi[c] - lo] = 1;
        }
      }
      for (int64_t p = bs + CH * WAVE + lane; p < be; p += WAVE) {
        win[B.inner[p] - lo] = Bv[p];
        fl[B.inner[p] - lo] = 1;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int s = ai[c] - lo;
        const bool valid = as + c * WAVE + lane < ae;
        if (valid && s >= 0 && s < span && fl[s]) accum(av[c], win[s]);
        if (valid && ai[c] == col_offset + j) tr = __dadd_rn(tr, Sc<T>::re(av[c]));
      }
      for (int64_t p = as + CH * WAVE + lane; p < ae; p += WAVE) {
        const int s = A.inner[p] - lo;
        if (s >= 0 && s < span && fl[s]) accum(Av[p], win[s]);
        if (A.inner[p] == col_offset + j) tr = __dadd_rn(tr, Sc<T>::re(Av[p]));
      }
      __builtin_amdgcn_wave_barrier();
    } else if (be - bs <= W) {
      // rows scattered over the whole range (relabelled operands): the row ids of B's column go to LDS (in the window's
      // memory) and A's entries search them there
      int* rows = reinterpret_cast<int*>(win);
      const int nb = (int)(be - bs);
#pragma unroll
      for (int c = 0; c < CH; ++c)
        if (c * WAVE + lane < nb) rows[c * WAVE + lane] = bi[c];
      for (int q = CH * WAVE + lane; q < nb; q += WAVE) rows[q] = B.inner[bs + q];
      __builtin_amdgcn_wave_barrier();
      for (int64_t p = as + lane; p < ae; p += WAVE) {
        const int r = A.inner[p];
        int l = 0, h = nb;
        while (l < h) {
          const int mid = (l + h) >> 1;
          if (rows[mid] < r) l = mid + 1; else h = mid;
        }
        if (l < nb && rows[l] == r) accum(Av[p], Bv[bs + l]);
        if (r == col_offset + j) tr = __dadd_rn(tr, Sc<T>::re(Av[p]));
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      for (int64_t p = as + lane; p < ae; p += WAVE) {
        const int r = A.inner[p];
        int64_t l = bs, h = be;
        while (l < h) {
          const int64_t mid = (l + h) >> 1;
          if (B.inner[mid] < r) l = mid + 1; else h = mid;
        }
        if (l < be && B.inner[l] == r) accum(Av[p], Bv[l]);
        if (r == col_offset + j) tr = __dadd_rn(tr, Sc<T>::re(Av[p]));
      }
    }
  }
  x = wave_sum_f64(x);
  y = wave_sum_f64(y);
  tr = wave_sum_f64(tr);
  if (lane == 0) { sx[wave] = x; sy[wave] = y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * b] = __dadd_rn(__dadd_rn(sx[0], sx[1]), __dadd_rn(sx[2], sx[3]));
    partial[2 * b + 1] = __dadd_rn(__dadd_rn(sy[0], sy[1]), __dadd_rn(sy[2], sy[3]));
  }
  if (trace_partial) {
    __syncthreads();
    if (lane == 0) sx[wave] = tr;
    __syncthreads();
    if (threadIdx.x == 0) {
      trace_partial[2 * b] = __dadd_rn(__dadd_rn(sx[0], sx[1]), __dadd_rn(sx[2], sx[3]));
      trace_partial[2 * b + 1] = 0.0;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void k_grand_sum(const T* __restrict__ v, int64_t n,
                                                   double* __restrict__ partial) {
  __shared__ double sx[4], sy[4];
  double x = 0, y = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if constexpr (Sc<T>::cplx) { x = __dadd_rn(x, v[i].x); y = __dadd_rn(y, v[i].y); }
    else x = __dadd_rn(x, v[i]);
  }
  x = wave_sum_f64(x);
  y = wave_sum_f64(y);
  const int wave = threadIdx.x / WAVE;
  if (lane_id() == 0) { sx[wave] = x; sy[wave] = y; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = __dadd_rn(__dadd_rn(sx[0], sx[1]), __dadd_rn(sx[2], sx[3]));
    partial[2 * blockIdx.x + 1] = __dadd_rn(__dadd_rn(sy[0], sy[1]), __dadd_rn(sy[2], sy[3]));
  }
}

// pairwise product, sequential per column; count pass (out_inner == nullptr) then fill pass
template <typename T>
__global__ void k_pairwise(Csc A, Csc B, const int64_t* __restrict__ couter, int32_t* __restrict__ out_inner,
                           T* __restrict__ out_val, int32_t* __restrict__ count, int conj_a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= A.cols) return;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  int64_t aa = A.outer[j], ea = A.outer[j + 1], bb = B.outer[j], eb = B.outer[j + 1];
  int64_t cc = couter ? couter[j] : 0;
  int n = 0;
  while (aa < ea && bb < eb) {
    const int ia = A.inner[aa], ib = B.inner[bb];
    if (ia == ib) {
      if (out_inner) {
        out_inner[cc] = ia;
        out_val[cc] = Sc<T>::mul(conj_a ? Sc<T>::conj(Av[aa]) : Av[aa], Bv[bb]);
        ++cc;
      }
      ++n; ++aa; ++bb;
    } else if (ia > ib) ++bb;
    else ++aa;
  }
  if (count) count[j] = n;
}

// diagonal sum: thread per column, binary search for row == j + col_offset
template <typename T>
__global__ __launch_bounds__(256) void k_trace(Csc A, int col_offset, double* __restrict__ partial) {
  __shared__ double sx[4];
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  double x = 0;
  for (int j = blockIdx.x * 256 + threadIdx.x; j < A.cols; j += gridDim.x * 256) {
    const int r = j + col_offset;
    int64_t l = A.outer[j], h = col_end(A, j);
    const int64_t e = h;
    while (l < h) {
      const int64_t mid = (l + h) >> 1;
      if (A.inner[mid] < r) l = mid + 1; else h = mid;
    }
    if (l < e && A.inner[l] == r) x = __dadd_rn(x, Sc<T>::re(Av[l]));
  }
  x = wave_sum_f64(x);
  const int wave = threadIdx.x / WAVE;
  if (lane_id() == 0) sx[wave] = x;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = __dadd_rn(__dadd_rn(sx[0], sx[1]), __dadd_rn(sx[2], sx[3]));
    partial[2 * blockIdx.x + 1] = 0.0;
  }
}

// per column: sum |v| (mode 0) or Gershgorin disc ends d-r, d+r (mode 1); one wave per column
template <typename T>
__global__ __launch_bounds__(256) void k_colstat(Csc A, int col_offset, int mode,
                                                 double* __restrict__ out0, double* __restrict__ out1) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  double r = 0, d = 0;
  if (mode == 1) {
    // Gershgorin radius: the sum of the column's moduli must not depend on the ORDER the entries are stored in -- a solve
    // that runs on a relabelled copy of the matrix (band scope across ranks, the load balancer) has to start from the same
    // spectral bounds as the solve on the caller's labels, or the two differ in the last bits of every entry from the first
    // iterate on.  Accumulated in double-double (device_util.hpp dd_add), rounded once at the end; k_sa_colstat sums the runs
    // of a slab form the same way, so a session's bounds are these bits too.
    double rl = 0.0;
    for (int64_t p = A.outer[j] + lane; p < A.outer[j + 1]; p += WAVE) {
      if (A.inner[p] == j + col_offset) d = __dadd_rn(d, Sc<T>::re(Av[p]));
      else dd_add(r, rl, Sc<T>::mag(Av[p]), 0.0);
    }
    dd_wave_sum(r, rl);
    d = wave_sum_f64(d);
    if (lane == 0) { out0[j] = d - r; out1[j] = d + r; }
    return;
  }
  for (int64_t p = A.outer[j] + lane; p < A.outer[j + 1]; p += WAVE) r = __dadd_rn(r, Sc<T>::mag(Av[p]));
  r = wave_sum_f64(r);
  if (lane == 0) out0[j] = r;
}

__global__ void k_scale(double* __restrict__ v, int64_t n, double c) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    v[i] = __dmul_rn(c, v[i]);
}
__global__ void k_conj(double* __restrict__ v, int64_t nnz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
    v[2 * i + 1] = -v[2 * i + 1];
}
__global__ void k_to_complex(const double* __restrict__ in, double* __restrict__ out, int64_t nnz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x) {
    out[2 * i] = in[i];
    out[2 * i + 1] = 0.0;
  }
}
__global__ void k_to_real(const double* __restrict__ in, double* __restrict__ out, int64_t nnz) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = in[2 * i];
}
__global__ void k_identity(int64_t* __restrict__ outer, int32_t* __restrict__ inner, double* __restrict__ val,
                           int n, int col_offset, int cols, int cplx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > cols) return;
  // local column j is global column j + col_offset; it holds a one iff that is < n
  const int ones_before = max(0, min(j, n - col_offset));
  outer[j] = ones_before;
  if (j < cols && j + col_offset < n) {
    inner[ones_before] = j + col_offset;
    if (cplx) { val[2 * ones_before] = 1.0; val[2 * ones_before + 1] = 0.0; }
    else val[ones_before] = 1.0;
  }
}
// flags[0] = off-diagonal or non-one entries, flags[1] = diagonal ones
template <typename T>
__global__ void k_identity_check(Csc A, int col_offset, unsigned long long* __restrict__ flags) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  long long bad = 0, good = 0;
  if (j < A.cols) {
    for (int64_t p = A.outer[j]; p < A.outer[j + 1]; ++p) {
      bool one;
      if constexpr (Sc<T>::cplx) one = hypot(Av[p].x - 1.0, Av[p].y) <= 2.2250738585072014e-308;
      else one = fabs(Av[p] - 1.0) <= 2.2250738585072014e-308;
      if (A.inner[p] != j + col_offset || !one) bad += 1;
      else good += 1;
    }
  }
  // (one pair of atomics per wave: every entry adding to the same two counters serialises in L2)
  bad = wave_sum_i64(bad);
  good = wave_sum_i64(good);
  if (lane_id() == 0) {
    if (bad) atomicAdd(&flags[0], (unsigned long long)bad);
    if (good) atomicAdd(&flags[1], (unsigned long long)good);
  }
}

// ------------------------------------------------------------------ re-indexing (transpose / permutation)
// key = new_col << 32 | new_row for every entry, payload = source position; radix sort (rocPRIM)
// then gather.  Used by setup paths only (transpose of ISQ, load-balancing permutation).
__global__ void k_expand_cols(const int64_t* __restrict__ outer, int cols, int32_t* __restrict__ colidx) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= cols) return;
  for (int64_t p = outer[j] + lane_id(); p < outer[j + 1]; p += WAVE) colidx[p] = j;
}
template <typename T>
__global__ void k_remap_keys(const int32_t* __restrict__ colidx, const int32_t* __restrict__ inner,
                             const T* __restrict__ val, int64_t nnz, const int32_t* __restrict__ row_map,
                             const int32_t* __restrict__ col_map, int transpose, int col_lo, int col_hi,
                             int drop_zero, unsigned long long* __restrict__ keys,
                             unsigned int* __restrict__ payload) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= nnz) return;
  int r = inner[p], c = colidx[p];
  if (row_map) r = row_map[r];
  if (col_map) c = col_map[c];
  if (transpose) { const int t = r; r = c; c = t; }
  const bool dropped = c < col_lo || c >= col_hi || (drop_zero && Sc<T>::is_zero(val[p]));
  keys[p] = dropped ? ~0ull : (((unsigned long long)(unsigned)(c - col_lo) << 32) | (unsigned)r);
  payload[p] = (unsigned int)p;
}
template <typename T>
__global__ void k_remap_gather(const unsigned long long* __restrict__ keys, const unsigned int* __restrict__ payload,
                               int64_t nkeep, const T* __restrict__ val, int32_t* __restrict__ inner_out,
                               T* __restrict__ val_out, int32_t* __restrict__ colcount) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = p < nkeep;
  const unsigned long long k = live ? keys[p] : ~0ull;
  if (live) {
    inner_out[p] = (int32_t)(k & 0xffffffffull);
    val_out[p] = val[payload[p]];
  }
  // the keys are sorted by column: one atomic per run of equal columns inside the wave, not one per entry
  const int col = (int)(k >> 32), lane = lane_id();
  const int prev = __shfl_up(col, 1, WAVE);
  const bool leader = live && (lane == 0 || col != prev);
  const unsigned long long heads = __ballot(leader), alive = __ballot(live);
  if (leader) {
    const unsigned long long later = heads & ~((2ull << lane) - 1ull);   // leaders after this lane
    const int end = later ? __builtin_ctzll(later) : (int)__popcll(alive);   // (live lanes are a prefix of the wave)
    atomicAdd(&colcount[col], end - lane);
  }
}
__global__ void k_count_valid(const unsigned long long* __restrict__ keys, int64_t n,
                              unsigned long long* __restrict__ out) {
  // (a few thousand atomics on the one counter, not one per entry or per wave: they serialise in L2)
  __shared__ unsigned long long sc[4];
  unsigned long long c = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
    c += keys[p] != ~0ull ? 1ull : 0ull;
  c = (unsigned long long)wave_sum_i64((int64_t)c);
  if (lane_id() == 0) sc[threadIdx.x / WAVE] = c;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sc[0] + sc[1] + sc[2] + sc[3]);
}

// column slicing / concatenation
__global__ void k_shift_outer(const int64_t* __restrict__ in, int64_t* __restrict__ out, int n, int64_t shift) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j <= n) out[j] = in[j] + shift;
}

// out[i] = in[i] - in[0] + add  (re-base a slice of column offsets)
__global__ void k_rebase_i64(const int64_t* __restrict__ in, int64_t* __restrict__ out, int n, int64_t add) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] - in[0] + add;
}
__global__ void k_linear_i64(int64_t* __restrict__ out, int64_t n, int64_t step) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = i * step;
}
__global__ void k_fill_i64(int64_t* __restrict__ out, int64_t n, int64_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = v;
}
// smallest first row / largest last row over the non-empty columns: mm[0] = min, mm[1] = max
__global__ void k_row_range(Csc A, int* __restrict__ mm) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int lo = INT_MAX, hi = -1;
  if (j < A.cols) {
    const int64_t s = A.outer[j], e = A.outer[j + 1];
    if (e > s) {
      lo = A.inner[s];
      hi = A.inner[e - 1];
    }
  }
  lo = wave_min_i32(lo);
  hi = wave_max_i32(hi);
  if (lane_id() == 0) {
    if (lo != INT_MAX) atomicMin(&mm[0], lo);
    if (hi >= 0) atomicMax(&mm[1], hi);
  }
}

}  // namespace

// =====================================================================================
EngineOptions& options() {
  static EngineOptions* o = [] {
    auto* e = new EngineOptions();
    if (const char* v = std::getenv("NTPOLY_AMD_HALO_OVERLAP")) e->halo_overlap = std::atoi(v);  // see kernels.hpp
    if (const char* v = std::getenv("NTPOLY_AMD_SPGEMM_FMA")) e->spgemm_fma = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_ARITHMETIC")) {   // the documented selector
      const std::string a(v);
      if (a == "fma") e->spgemm_fma = 1;
      else if (a == "unfused") e->spgemm_fma = 0;
      else NTP_FATAL("NTPOLY_AMD_ARITHMETIC must be fma or unfused, not " + a);
    }
    if (const char* v = std::getenv("NTPOLY_AMD_TILE_ROWS")) e->tile_rows = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_TILE_WAVES")) e->tile_waves = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_PLAN_AHEAD")) e->plan_ahead = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_SLAB_ALGEBRA")) e->slab_algebra = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_PANEL_SESSIONS")) e->panel_sessions = std::atoi(v);
    if (const char* v = std::getenv("NTPOLY_AMD_BLOCK_UNFUSED")) e->block_unfused = std::atoi(v);
    return e;
  }();
  return *o;
}
SpgemmStats& last_spgemm_stats() {
  static SpgemmStats* s = new SpgemmStats();
  return *s;
}
SpgemmAccum& spgemm_accum() {
  static SpgemmAccum* a = new SpgemmAccum();
  return *a;
}

namespace {
// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); asynchronous on the engine stream
template <typename TIn>
void scan_async(const TIn* d_in, int64_t* d_out, int64_t n) {
  if (n <= 4096) {
    if constexpr (sizeof(TIn) == 8) hipLaunchKernelGGL(k_scan_excl_i64, dim3(1), dim3(1024), 0, stream(), (const int64_t*)d_in, d_out, n);
    else hipLaunchKernelGGL(k_scan_excl_i32, dim3(1), dim3(1024), 0, stream(), (const int32_t*)d_in, d_out, n);
    return;
  }
  const int nb = (int)((n + 2047) / 2048);
  DevBuf<int64_t> sums((size_t)nb), offs((size_t)nb + 1);
  hipLaunchKernelGGL((k_scan_block_sums<TIn>), dim3(nb), dim3(256), 0, stream(), d_in, n, sums.p);
  hipLaunchKernelGGL(k_scan_excl_i64, dim3(1), dim3(1024), 0, stream(), sums.p, offs.p, (int64_t)nb);
  hipLaunchKernelGGL((k_scan_block_apply<TIn>), dim3(nb), dim3(256), 0, stream(), d_in, n, offs.p, d_out);
}
// deterministic two-level sum of n (x, y) pairs into out_dev[0..1]
void reduce_sum2_async(const double* part, int n, double* out_dev) {
  if (n <= 8192) {
    hipLaunchKernelGGL(k_reduce_sum2, dim3(1), dim3(256), 0, stream(), part, n, out_dev, n);
    return;
  }
  const int chunk = 2048, g = (n + chunk - 1) / chunk;
  DevBuf<double> lvl((size_t)2 * g);
  hipLaunchKernelGGL(k_reduce_sum2, dim3(g), dim3(256), 0, stream(), part, n, lvl.p, chunk);
  hipLaunchKernelGGL(k_reduce_sum2, dim3(1), dim3(256), 0, stream(), lvl.p, g, out_dev, g);
}
}  // namespace

// Totals of a fused purification step in two launches: sum count[0..n) (entries of the result), sum a[0..m) and
// sum b[0..m) (entries of the product, products), sum of the (dot, trace) pairs part[0..2m) -- fixed shapes and orders,
// so the sums are reproducible.  out: 3 x int64 then 2 x double (as raw 8-byte words).
constexpr int FT_BLOCKS = 64;
__global__ __launch_bounds__(256) void k_fused_totals(const int32_t* __restrict__ count, int n, const long long* __restrict__ a,
                                                      const long long* __restrict__ b, const double* __restrict__ part,
                                                      int m, const double* __restrict__ lvl_in, double* __restrict__ out,
                                                      int stage) {
  __shared__ long long si[3][4];
  __shared__ double sd[2][4];
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  long long c = 0, x = 0, y = 0;
  double d = 0.0, t = 0.0;
  if (stage == 0) {   // block g of FT_BLOCKS takes a contiguous chunk of every array
    const int g = blockIdx.x;
    const int n0 = (int)((int64_t)n * g / FT_BLOCKS), n1 = (int)((int64_t)n * (g + 1) / FT_BLOCKS);
    // (the loads of a thread in flight together, the additions in the loop's order)
#pragma unroll 8
    for (int i = n0 + threadIdx.x; i < n1; i += 256) c += count[i];
    const int m0 = (int)((int64_t)m * g / FT_BLOCKS), m1 = (int)((int64_t)m * (g + 1) / FT_BLOCKS);
#pragma unroll 2
    for (int i = m0 + threadIdx.x; i < m1; i += 256) {
      x += a[i];
      if (b) y += b[i];
      d = __dadd_rn(d, part[2 * i]);
      t = __dadd_rn(t, part[2 * i + 1]);
    }
  } else {
    const long long* li = reinterpret_cast<const long long*>(lvl_in);
    for (int g = threadIdx.x; g < FT_BLOCKS; g += 256) {
      c += li[5 * g]; x += li[5 * g + 1]; y += li[5 * g + 2];
      d = __dadd_rn(d, lvl_in[5 * g + 3]);
      t = __dadd_rn(t, lvl_in[5 * g + 4]);
    }
  }
  c = wave_sum_i64(c); x = wave_sum_i64(x); y = wave_sum_i64(y);
  d = wave_sum_f64(d); t = wave_sum_f64(t);
  if (lane == 0) { si[0][wave] = c; si[1][wave] = x; si[2][wave] = y; sd[0][wave] = d; sd[1][wave] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long* oi = reinterpret_cast<long long*>(out) + (stage == 0 ? 5 * blockIdx.x : 0);
    double* od = out + (stage == 0 ? 5 * blockIdx.x : 0);
    for (int q = 0; q < 3; ++q) oi[q] = si[q][0] + si[q][1] + si[q][2] + si[q][3];
    od[3] = __dadd_rn(__dadd_rn(sd[0][0], sd[0][1]), __dadd_rn(sd[0][2], sd[0][3]));
    od[4] = __dadd_rn(__dadd_rn(sd[1][0], sd[1][1]), __dadd_rn(sd[1][2], sd[1][3]));
  }
}

void scan_i32_async(const int32_t* d_in, int64_t* d_out, int64_t n) { scan_async<int32_t>(d_in, d_out, n); }
void scan_i64_async(const int64_t* d_in, int64_t* d_out, int64_t n) { scan_async<int64_t>(d_in, d_out, n); }

namespace {
struct FetchArgs {
  const unsigned long long* src[16];
  int words[16];
  int n;
};
__global__ __launch_bounds__(64) void k_fetch(FetchArgs a, unsigned long long* __restrict__ host_mapped) {
  int off = 0;
  for (int s = 0; s < a.n; ++s) {
    for (int i = threadIdx.x; i < a.words[s]; i += 64) host_mapped[off + i] = a.src[s][i];
    off += a.words[s];
  }
  __threadfence_system();
}
}  // namespace

void ScalarFetch::run() {
  static unsigned long long* host = nullptr;
  static unsigned long long* dev = nullptr;
  constexpr int kMaxWords = 512;
  if (!host) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&host), kMaxWords * 8, hipHostMallocMapped));
    HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0));
  }
  FetchArgs a;
  a.n = n;
  int total = 0;
  for (int s = 0; s < n; ++s) {
    a.src[s] = static_cast<const unsigned long long*>(src[s]);
    a.words[s] = words[s];
    total += words[s];
  }
  if (total > kMaxWords) NTP_FATAL("ScalarFetch: too many words");
  if (total) hipLaunchKernelGGL(k_fetch, dim3(1), dim3(64), 0, stream(), a, dev);
  sync_stream();
  int off = 0;
  for (int s = 0; s < n; ++s) {
    std::memcpy(dst[s], host + off, (size_t)words[s] * 8);
    off += words[s];
  }
  n = 0;
}

static int64_t exclusive_scan_i64_from_i32(const int32_t* d_in, int64_t* d_out, int64_t n) {
  scan_async<int32_t>(d_in, d_out, n);
  int64_t total = 0;
  ScalarFetch f;
  f.add(d_out + n, 1, &total);
  f.run();
  return total;
}
int64_t exclusive_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n) {
  scan_async<int64_t>(d_in, d_out, n);
  int64_t total = 0;
  HIP_CHECK(hipMemcpyAsync(&total, d_out + n, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
  sync_stream();
  return total;
}

namespace {
// Kernel timing with HIP events on the engine's stream (option time_kernels).  The events of a multiply are only
// recorded; they are resolved (hipEventElapsedTime) when the statistics are read, so timing adds no
// synchronisation to the measured region.
struct TimedCall {
  hipEvent_t ev[4];  // total start/stop, numeric start/stop
};
std::vector<hipEvent_t>& event_pool() {
  static auto* p = new std::vector<hipEvent_t>();
  return *p;
}
std::vector<TimedCall>& pending_timings() {
  static auto* p = new std::vector<TimedCall>();
  return *p;
}
hipEvent_t get_event() {
  auto& pool = event_pool();
  if (!pool.empty()) {
    hipEvent_t e = pool.back();
    pool.pop_back();
    return e;
  }
  hipEvent_t e;
  HIP_CHECK(hipEventCreate(&e));
  return e;
}
struct EventTimer {
  hipEvent_t a = nullptr, b = nullptr;
  bool on;
  explicit EventTimer(bool enable) : on(enable) {
    if (on) {
      a = get_event();
      b = get_event();
    }
  }
  void start() { if (on) HIP_CHECK(hipEventRecord(a, stream())); }
  void stop() { if (on) HIP_CHECK(hipEventRecord(b, stream())); }
};
}  // namespace

namespace {
// counts the operations that change the values of a matrix in place (scale, conjugate, scale_columns): part of the key
// of the cache below, together with the serial number of the value buffer's allocation
unsigned long long& value_epoch() {
  static unsigned long long e = 0;
  return e;
}
}  // namespace
unsigned long long matrix_value_epoch() { return value_epoch(); }
void bump_matrix_value_epoch() { value_epoch() += 1; }
namespace {
// D of a fused purification step (SlabFusion), expanded once: dexp[doff[j] + (r - dmin[j])] = D(r, j), zero in the holes
struct DotOperand {
  const void* val = nullptr;
  unsigned long long serial = 0, epoch = 0;
  int64_t nnz = -1;
  int32_t cols = 0;
  DevBuf<int32_t> dmin, dmax;
  DevBuf<int64_t> doff;
  DevBuf<double> dexp;
  int64_t max_tile = 0;   // largest sum of the spans of 16 consecutive columns (the kernel keeps that tile in LDS)
  int align = 1;          // columns in aligned zero-padded slots of that many rows (k_span_aligned), 1: packed back to back
};
// stored values that are exactly zero (real matrices; columns may be loose): one wave per column
__global__ __launch_bounds__(256) void k_count_zero_values(Csc A, unsigned long long* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  const double* __restrict__ v = static_cast<const double*>(A.val);
  int c = 0;
  for (int64_t p = A.outer[j] + lane, e = col_end(A, j); p < e; p += WAVE) c += v[p] == 0.0 ? 1 : 0;
  const unsigned long long m = __ballot(c != 0);
  if (m && lane == 0) atomicAdd(out, 1ull);
}
__global__ void k_span_block_max(const int32_t* __restrict__ span, int n, int J, unsigned long long* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b * J >= n) return;
  unsigned long long t = 0;
  for (int j = b * J; j < min(n, (b + 1) * J); ++j) t += (unsigned long long)span[j];
  atomicMax(out, t);
}
DotOperand*& dot_operand_slot() {
  static DotOperand* c = new DotOperand();
  return c;
}
void drop_dot_operand() { *dot_operand_slot() = DotOperand(); }
const DotOperand& dot_operand(const DevMat& D) {
  DotOperand* c = dot_operand_slot();
  const unsigned long long ser = dev_alloc_serial(D.val.p);
  const int want_al = options().spgemm_fma == 1 ? tile_expand_align() : 1;
  if (c->val == D.val.p && c->serial == ser && ser != 0 && c->epoch == value_epoch() && c->nnz == D.nnz && c->cols == D.cols &&
      c->align == want_al)
    return *c;
  const int n = D.cols;
  c->dmin.alloc((size_t)n); c->dmax.alloc((size_t)n); c->doff.alloc((size_t)n + 1);
  DevBuf<int32_t> dlen((size_t)n), dspan((size_t)n);
  hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(D), c->dmin.p, c->dmax.p, dlen.p);
  const int al = options().spgemm_fma == 1 ? tile_expand_align() : 1;   // (the MFMA tile kernel reads R rows at a time)
  if (al > 1) hipLaunchKernelGGL(k_span_aligned, dim3(cdiv(n, 256)), dim3(256), 0, stream(), c->dmin.p, c->dmax.p, dspan.p, n, al);
  else hipLaunchKernelGGL(k_span_of, dim3(cdiv(n, 256)), dim3(256), 0, stream(), c->dmin.p, c->dmax.p, dspan.p, n);
  scan_async<int32_t>(dspan.p, c->doff.p, (int64_t)n);
  DevBuf<unsigned long long> tmax(1);
  tmax.zero();
  hipLaunchKernelGGL(k_span_block_max, dim3(cdiv(cdiv(n, SLAB_J), 256)), dim3(256), 0, stream(), dspan.p, n, SLAB_J, tmax.p);
  int64_t total = 0;
  unsigned long long hmax = 0;
  {
    ScalarFetch f;
    f.add(c->doff.p + n, 1, &total);
    f.add(tmax.p, 1, &hmax);
    f.run();
  }
  c->max_tile = (int64_t)hmax;
  c->dexp.alloc((size_t)total + 1);
  if (al > 1)
    hipLaunchKernelGGL(k_aligned_offsets<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), c->dmin.p, c->dmax.p,
                       c->doff.p, c->dexp.p, n, al);
  c->align = al;
  hipLaunchKernelGGL(k_slab_expand_a<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(D), c->dmin.p,
                     c->doff.p, c->dexp.p);
  c->val = D.val.p;
  c->serial = ser;
  c->epoch = value_epoch();
  c->nnz = D.nnz;
  c->cols = n;
  return *c;
}
}  // namespace

long long* fusion_counts() {
  static long long c[3] = {0, 0, 0};
  return c;
}
long long* tile2_counts() {
  static long long c[2] = {0, 0};
  return c;
}

void flush_spgemm_timers() {
  auto& pend = pending_timings();
  if (pend.empty()) return;
  SpgemmAccum& acc = spgemm_accum();
  for (size_t i = 0; i < pend.size(); ++i) {
    float t_all = 0.f, t_num = 0.f;
    HIP_CHECK(hipEventSynchronize(pend[i].ev[1]));
    HIP_CHECK(hipEventElapsedTime(&t_all, pend[i].ev[0], pend[i].ev[1]));
    HIP_CHECK(hipEventElapsedTime(&t_num, pend[i].ev[2], pend[i].ev[3]));
    acc.ms_total += t_all;
    acc.ms_numeric += t_num;
    if (i + 1 == pend.size()) {
      last_spgemm_stats().ms_total = t_all;
      last_spgemm_stats().ms_numeric = t_num;
    }
    for (int k = 0; k < 4; ++k) event_pool().push_back(pend[i].ev[k]);
  }
  pend.clear();
}

namespace {
// compressed columns from the slab form: one wave per column, the non-zeros of its run in row order
__global__ __launch_bounds__(256) void k_pack_slab(int ncols, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                   const int64_t* __restrict__ off, const double* __restrict__ val,
                                                   const int64_t* __restrict__ outer, int32_t* __restrict__ inner,
                                                   double* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= ncols) return;
  const int lane = lane_id();
  const int f = first[j], l = last[j];
  int64_t pos = outer[j];
  const double* __restrict__ src = val + off[j];
  for (int r0 = f; r0 <= l; r0 += WAVE) {
    const int r = r0 + lane;
    const double v = r <= l ? src[r - f] : 0.0;
    const unsigned long long m = __ballot(v != 0.0);
    if (v != 0.0) {
      const int64_t q = pos + __popcll(m & lanemask_lt());
      inner[q] = r;
      out[q] = v;
    }
    pos += __popcll(m);
  }
}
__global__ __launch_bounds__(256) void k_pack_slab_c(int ncols, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                     const int64_t* __restrict__ off, const double2* __restrict__ val,
                                                     const int64_t* __restrict__ outer, int32_t* __restrict__ inner,
                                                     double2* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= ncols) return;
  const int lane = lane_id();
  const int f = first[j], l = last[j];
  int64_t pos = outer[j];
  const double2* __restrict__ src = val + off[j];
  for (int r0 = f; r0 <= l; r0 += WAVE) {
    const int r = r0 + lane;
    const double2 v = r <= l ? src[r - f] : make_double2(0.0, 0.0);
    const bool nz = v.x != 0.0 || v.y != 0.0;
    const unsigned long long m = __ballot(nz);
    if (nz) {
      const int64_t q = pos + __popcll(m & lanemask_lt());
      inner[q] = r;
      out[q] = v;
    }
    pos += __popcll(m);
  }
}
template <int MAXCH, int NW>
void launch_pair3(int bin_lo, int bin_hi, int wrt, const DevMat& A, const DevMat& B, const int32_t* lo, const int32_t* span,
                  const uint8_t* binarr, const int64_t* tmpoff, int32_t* out_inner, double* out_val, int32_t* count,
                  double alpha, double thr, int dense_rule) {
  const int ngroups = cdiv(B.cols, 2);
  const int nblocks = cdiv(ngroups, NW);
  const size_t lds = (size_t)NW * 2 * (size_t)wrt * sizeof(double);
  hipLaunchKernelGGL((k_spgemm_pair3<MAXCH, NW>), dim3(xcd_grid(nblocks)), dim3(NW * WAVE), lds, stream(), view(A), view(B),
                     lo, span, binarr, bin_lo, bin_hi, tmpoff, out_inner, out_val, count, alpha, thr, dense_rule, nblocks, wrt);
}

template <typename T, int W, int NW>
void launch_window(int bin, const DevMat& A, const DevMat& B, const int32_t* lo, const int32_t* span,
                   const uint8_t* binarr, const int64_t* tmpoff, int32_t* out_inner, T* out_val,
                   int32_t* count, double alpha, double thr, int dense_rule) {
  const int nblocks = cdiv(B.cols, NW);
  hipLaunchKernelGGL((k_spgemm_window<T, W, NW>), dim3(xcd_grid(nblocks)), dim3(NW * WAVE), 0, stream(),
                     view(A), view(B), lo, span, binarr, bin, tmpoff, out_inner, out_val, count, alpha, thr,
                     dense_rule, nblocks);
}
}  // namespace

// ------------------------------------------------------------------ row strips (products with wide scattered columns)
// Operands whose product columns hold more distinct rows than the grouped kernel's largest table (a 3-D Hamiltonian:
// ~3 000 rows per column scattered over +-40 000) are multiplied in S row STRIPS of A: strip s keeps the rows whose
// block (row >> shift) is = s (mod S), so every column of A * B falls apart into S pieces of ~1/S of its rows, each
// small enough for the tables.  C(i, j) only depends on row i of A: every piece is the exact product restricted to its
// rows (same products, same ascending k, same prune), and the pieces of a column interleave block by block.
namespace {
struct StripCtx { bool active = false, failed = false; };
// the strip count that worked for the last multiply of a dimension that needed strips: [0] dimension, [1] strips
constexpr int kMaxStrips = 16;   // (StripCols: what k_strip_total / k_strip_merge interleave)
int* strips_memory(bool cplx) {
  static int m[2][2] = {{-1, 0}, {-1, 0}};
  return m[cplx ? 1 : 0];
}
StripCtx& strip_ctx() {
  static StripCtx c;
  return c;
}
template <typename T>
__global__ __launch_bounds__(256) void k_strip_count(Csc A, int shift, int S, int s, int32_t* __restrict__ cnt) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  int c = 0;
  for (int64_t p = A.outer[j] + lane, e = A.outer[j + 1]; p < e; p += WAVE) c += ((A.inner[p] >> shift) % S) == s ? 1 : 0;
  c = (int)wave_sum_i64(c);
  if (lane == 0) cnt[j] = c;
}
template <typename T>
__global__ __launch_bounds__(256) void k_strip_fill(Csc A, int shift, int S, int s, const int64_t* __restrict__ outer,
                                                    int32_t* __restrict__ inner, T* __restrict__ val) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  int64_t pos = outer[j];
  for (int64_t p0 = A.outer[j], e = A.outer[j + 1]; p0 < e; p0 += WAVE) {
    const int64_t p = p0 + lane;
    const bool in = p < e && ((A.inner[p] >> shift) % S) == s;
    const unsigned long long m = __ballot(in);
    if (in) {
      const int64_t q = pos + __popcll(m & lanemask_lt());
      inner[q] = A.inner[p];
      val[q] = Av[p];
    }
    pos += __popcll(m);
  }
}
struct StripCols { const int64_t* outer[16]; const int32_t* inner[16]; const void* val[16]; };
__global__ void k_strip_total(StripCols P, int S, int n, int32_t* __restrict__ total) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int t = 0;
  for (int s = 0; s < S; ++s) t += (int)(P.outer[s][j + 1] - P.outer[s][j]);
  total[j] = t;
}
// column j of the result = its S pieces interleaved block by block (blocks of 1 << shift rows; block b belongs to piece
// b % S and its entries sit together there, in row order): one wave per column, a lane per block
template <typename T>
__global__ __launch_bounds__(256) void k_strip_merge(StripCols P, int S, int shift, int n, const int64_t* __restrict__ outer,
                                                     int32_t* __restrict__ inner, T* __restrict__ val) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  int rmin = INT_MAX, rmax = -1;
  for (int s = 0; s < S; ++s) {
    const int64_t a = P.outer[s][j], e = P.outer[s][j + 1];
    if (e > a) {
      rmin = min(rmin, P.inner[s][a]);
      rmax = max(rmax, P.inner[s][e - 1]);
    }
  }
  if (rmax < rmin) return;
  int64_t base = outer[j];
  for (int b0 = rmin >> shift; b0 <= (rmax >> shift); b0 += WAVE) {
    const int b = b0 + lane;
    const int s = b % S;
    const int64_t a = P.outer[s][j], e = P.outer[s][j + 1];
    // entries of piece s with row >> shift == b: [lo, hi)
    int64_t lo = a, hi = e;
    if (b <= (rmax >> shift)) {
      const int r0 = b << shift;
      int64_t x = a, y = e;
      while (x < y) { const int64_t m = (x + y) >> 1; if (P.inner[s][m] < r0) x = m + 1; else y = m; }
      lo = x;
      y = e;
      const long long r1 = ((long long)(b + 1)) << shift;
      while (x < y) { const int64_t m = (x + y) >> 1; if ((long long)P.inner[s][m] < r1) x = m + 1; else y = m; }
      hi = x;
    } else {
      lo = hi = a;
    }
    const int c = (int)(hi - lo);
    int incl = c;   // inclusive prefix over the lanes
    for (int o = 1; o < WAVE; o <<= 1) {
      const int t = __shfl_up(incl, o, WAVE);
      if (lane >= o) incl += t;
    }
    const int64_t dst = base + incl - c;
    const T* __restrict__ sv = static_cast<const T*>(P.val[s]);
    for (int i = 0; i < c; ++i) {
      inner[dst + i] = P.inner[s][lo + i];
      val[dst + i] = sv[lo + i];
    }
    base += __shfl(incl, WAVE - 1, WAVE);
  }
}
__global__ void k_span_sum(const int32_t* __restrict__ cmin, const int32_t* __restrict__ cmax, int n, unsigned long long* __restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  long long v = (k < n && cmax[k] >= cmin[k]) ? (cmax[k] - cmin[k] + 1) : 0;
  v = wave_sum_i64(v);
  if (lane_id() == 0 && v) atomicAdd(out, (unsigned long long)v);
}
}  // namespace

// C = alpha * A * B through S row strips of A (see above).  false: a strip still overflowed the grouped tables (C unset).
static bool spgemm_striped(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule, int S,
                           int shift) {
  const int n = B.cols;
  std::vector<DevMat> piece((size_t)S);
  StripCtx& ctx = strip_ctx();
  for (int s = 0; s < S; ++s) {
    DevMat As;
    As.rows = A.rows; As.cols = A.cols; As.cplx = A.cplx; As.zero_free = A.zero_free;
    DevBuf<int32_t> cnt((size_t)A.cols);
    As.outer.alloc((size_t)A.cols + 1);
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_strip_count<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), view(A), shift, S, s, cnt.p);
    });
    As.nnz = exclusive_scan_i64_from_i32(cnt.p, As.outer.p, A.cols);
    As.inner.alloc((size_t)As.nnz + kIndexSlack);
    As.val.alloc(((size_t)As.nnz + kIndexSlack) * As.wval());
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_strip_fill<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), view(A), shift, S, s,
                         As.outer.p, As.inner.p, reinterpret_cast<T*>(As.val.p));
    });
    ctx.active = true;
    ctx.failed = false;
    spgemm(As, B, piece[(size_t)s], alpha, threshold, dense_rule, nullptr, nullptr, nullptr);
    ctx.active = false;
    if (ctx.failed) return false;
  }
  StripCols P;
  for (int s = 0; s < 16; ++s) {
    const DevMat& m = piece[(size_t)std::min(s, S - 1)];
    P.outer[s] = m.outer.p; P.inner[s] = m.inner.p; P.val[s] = m.val.p;
  }
  DevBuf<int32_t> total((size_t)n);
  hipLaunchKernelGGL(k_strip_total, dim3(cdiv(n, 256)), dim3(256), 0, stream(), P, S, n, total.p);
  C.rows = A.rows; C.cols = n; C.cplx = A.cplx;
  C.cnt.release();
  C.slab.reset();
  C.outer.alloc((size_t)n + 1);
  C.nnz = exclusive_scan_i64_from_i32(total.p, C.outer.p, n);
  C.inner.alloc((size_t)C.nnz + kIndexSlack);
  C.val.alloc(((size_t)C.nnz + kIndexSlack) * C.wval());
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_strip_merge<T>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), P, S, shift, n, C.outer.p,
                       C.inner.p, reinterpret_cast<T*>(C.val.p));
  });
  sync_stream();   // (the pieces are released on return)
  return true;
}

namespace {
// the block path (spgemm_block.hip) with spgemm()'s book-keeping around it
int g_block_keep = 0;           // > 0: a caller that understands DevMat::blk is waiting for the product (BlockKeepScope)
bool block_eligible(const DevMat& A, const DevMat& B, const ColRange* arange) {
  return !arange && !A.cplx && !B.cplx && block_arithmetic_ok() && options().block_path != 0 && options().spgemm_variant < 0 &&
         options().spgemm_force_bin <= 0 && A.rows == A.cols && B.rows == B.cols && A.cols == B.rows;
}
bool try_block_path(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule) {
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  BlockInfo bi;
  SpgemmStats st;
  st.nnz_a = A.nnz;
  st.nnz_b = B.nnz;
  const int32_t n = B.cols;
  const int64_t nnz_a = A.nnz, nnz_b = B.nnz;   // (C may be one of the operands)
  const int32_t acols = A.cols, bcols = B.cols;
  if (!spgemm_block(A, B, C, alpha, threshold, dense_rule, &bi, timing ? t_num.a : nullptr, timing ? t_num.b : nullptr, g_block_keep > 0)) {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    A.block_hint = 0;
    B.block_hint = 0;
    return false;
  }
  A.block_hint = 1;   // (harmless when A or B is C: the result is new)
  B.block_hint = 1;
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  st.block = 1;
  st.block_fill = bi.fill_a;
  st.block_tile_products = bi.tile_products;
  st.block_cand = bi.cand;
  st.products = bi.products;
  st.nnz_c = C.nnz;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.products += st.products;
  acc.nnz_c += C.nnz;
  acc.alg_bytes += 12.0 * (double)(nnz_a + nnz_b + C.nnz) + 4.0 * ((double)acols + bcols + n + 3);
  return true;
}
// the thin-left kernel (spgemm_thin.hip) with spgemm()'s book-keeping around it
bool try_thin_left(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule) {
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  SpgemmStats st;
  st.nnz_a = A.nnz;
  st.nnz_b = B.nnz;
  const int64_t nnz_a = A.nnz, nnz_b = B.nnz;   // (C may be one of the operands)
  const int32_t acols = A.cols, bcols = B.cols, n = B.cols;
  const bool cplx = A.cplx;
  const int dr = (dense_rule ? 1 : 0) | ((options().spgemm_fma && !A.cplx) ? 2 : 0);
  int64_t products = 0;
  if (!spgemm_thin_left(A, B, C, alpha, threshold, dr, timing ? &products : nullptr, timing ? t_num.a : nullptr, timing ? t_num.b : nullptr)) {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  }
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  st.thin = 1;
  st.products = products;
  st.nnz_c = C.nnz;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.products += st.products;
  acc.nnz_c += C.nnz;
  acc.alg_bytes += (cplx ? 20.0 : 12.0) * (double)(nnz_a + nnz_b + C.nnz) + 4.0 * ((double)acols + bcols + n + 3);
  return true;
}
}  // namespace
// a TRS2 step in block form (spgemm_block.hip block_trs2_step) with spgemm()'s book-keeping
bool trs2_block_step(DevMat& X, int mode, double threshold, bool dense_rule, const DevMat& D, double out[4]) {
  if (!X.blocked() && !X.block_hint) return false;
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  BlockInfo bi;
  SpgemmStats st;
  st.nnz_a = X.nnz;
  st.nnz_b = X.nnz;
  const int64_t nnz_x = X.nnz;
  const int32_t n = X.cols;
  if (!block_trs2_step(X, mode, threshold, dense_rule, D, out, &bi, timing ? t_num.a : nullptr, timing ? t_num.b : nullptr)) {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  }
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  st.block = 1;
  st.fused = mode;
  st.block_fill = bi.fill_a;
  st.block_tile_products = bi.tile_products;
  st.block_cand = bi.cand;
  st.products = bi.products;
  st.nnz_c = bi.nnz_c;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.products += st.products;
  acc.nnz_c += bi.nnz_c;
  acc.alg_bytes += 12.0 * (double)(2 * nnz_x + bi.nnz_c) + 4.0 * (3.0 * n + 3);
  fusion_counts()[mode == 1 ? 0 : 1] += 1;
  X.block_hint = 1;
  return true;
}
BlockKeepScope::BlockKeepScope() { g_block_keep += 1; }
BlockKeepScope::~BlockKeepScope() { g_block_keep -= 1; }

void spgemm(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule,
            LooseProduct* loose, const ColRange* arange, SlabFusion* fuse) {
  if (loose) loose->valid = false;
  if (fuse) fuse->done = false;
  // operands in block form, or in compressed columns of a dimension whose last product the block path computed: the
  // block path first (no run statistics, no per-column plan)
  if ((A.blocked() || B.blocked() || A.block_hint || B.block_hint) && !A.loose() && !B.loose() && !A.expanded() && !B.expanded() &&
      A.nnz > 0 && B.nnz > 0 && block_eligible(A, B, arange) && !strip_ctx().active) {
    if (try_block_path(A, B, C, alpha, threshold, dense_rule)) return;
  }
  if (A.expanded() || B.expanded() || A.blocked() || B.blocked()) {   // (the steps on the slab form are slab_step's; everything else wants columns)
    DevMat Ap = packed_copy(A);
    if (&A == &B) {
      spgemm(Ap, Ap, C, alpha, threshold, dense_rule, loose, arange, fuse);
    } else {
      DevMat Bp = packed_copy(B);
      spgemm(Ap, Bp, C, alpha, threshold, dense_rule, loose, arange, fuse);
    }
    return;
  }
  if (A.cols != B.rows) NTP_FATAL("spgemm: inner dimensions differ");
  if (A.cplx != B.cplx) NTP_FATAL("spgemm: mixed scalar types must be up-cast by the caller");
  const int32_t m = A.rows, n = B.cols;
  SpgemmStats st;
  st.nnz_a = A.nnz;
  st.nnz_b = B.nnz;
  if (A.nnz == 0 || B.nnz == 0 || n == 0 || m == 0) {
    C.reset_empty(m, n, A.cplx);
    last_spgemm_stats() = st;
    return;
  }
  // a thin left operand (an identity, a near-diagonal factor): the output-driven gather kernel, whatever the right operand's shape
  // (not with a fused purification step asked for: the gather kernel has no such epilogue, and a fused step's left operand
  // is the iterate itself -- never thin)
  if (!loose && !arange && !fuse && !A.loose() && !B.loose() && options().thin_left != 0 && options().spgemm_variant < 0 && options().spgemm_force_bin <= 0 &&
      !strip_ctx().active && A.rows == A.cols && A.nnz <= 8 * (int64_t)A.cols && B.nnz >= A.nnz) {
    if (try_thin_left(A, B, C, alpha, threshold, dense_rule)) return;
  }
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  // columns of A that the rows of B can name: all of them, or the range the caller knows (a gathered operand is
  // dim wide but populated only over the halo range: the per-column work arrays then cover that range only and are
  // addressed with absolute column ids through pointers biased by -ka)
  const int32_t ka = arange ? arange->a : 0, kb = arange ? arange->b : A.cols;
  const int32_t nka = std::max(0, kb - ka);
  DevBuf<int32_t> cmin_own((size_t)nka), cmax_own((size_t)nka), clen_own((size_t)nka);
  struct Biased { int32_t* p; };
  const Biased cmin{cmin_own.p - ka}, cmax{cmax_own.p - ka}, clen{clen_own.p - ka};
  // loose operands (an iterate left in its merge slots): the register-slab path of X * X reads them as they are,
  // every other path gets packed copies
  const bool loose_in = A.loose() || B.loose();
  Csc Ar = lview(A);
  Ar.outer += ka;
  if (Ar.cnt) Ar.cnt += ka;
  Ar.cols = nka;
  if (nka) hipLaunchKernelGGL(k_col_extent, dim3(cdiv(nka, 256)), dim3(256), 0, stream(), Ar, cmin_own.p, cmax_own.p, clen_own.p);
  DevBuf<int32_t> lo, span, count(n);
  DevBuf<uint8_t> bin;
  DevBuf<int64_t> ub, ip, tmpoff(n + 1);
  DevBuf<unsigned long long> stats(24);
  stats.zero();
  count.zero();
  unsigned long long hstats[24] = {0};
  int64_t tmp_total = 0;

  // ---- candidate for the register-slab kernel (real operands, run-like columns): its plan needs only the column
  // extents, so the per-column walk over B (k_spgemm_plan) is skipped when the slab kernel is taken
  // (complex operands: 8 columns per workgroup, two slabs per wave, six waves -- the same 768-row window and the
  // same 128-byte multiplier row)
  const int SJ = A.cplx ? SLAB_CJ : SLAB_J;
  // widest row window the slab kernels take: real 6 waves x 3 slabs = 1152 rows, complex 8 waves x 2 slabs = 1024
  // (the usual geometries are 4 x 3 and 6 x 2 = 768 rows; the wider workgroups serve operands with longer runs)
  // (option spgemm_fma: 1 = the MFMA tile kernel, spgemm_tile.hip -- any window; 3 = the v_fma_f64 loop of the slab kernel, 4 waves)
  // (complex operands under FMA arithmetic, option complex_tile: the MFMA tile kernel of spgemm_tile_c.hip -- any window)
  const bool ctile_try = A.cplx && options().spgemm_fma == 1 && options().complex_tile != 0;
  const int slab_rows = A.cplx ? (ctile_try ? 16384 : 8 * SLAB_CSL * WAVE) : (options().spgemm_fma == 1 ? 16384 : (options().spgemm_fma ? SLAB_NW : 8) * SLAB_SL * WAVE);
  const size_t esz = A.cplx ? 16 : 8;
  const int sv_opt = options().spgemm_variant;
  // ---- grouped LDS-hash kernel FIRST when the previous multiply of this dimension was computed by it without a
  // single column handed back and its kept min-hash column order still fits this operand (spgemm_grouped mode 2 checks
  // that): no slab planning, no per-column plan -- fixed output slots of the largest table class per column
  DevBuf<int32_t> tmp_inner;
  DevBuf<double> tmp_val;
  static int grouped_first_n[2] = {-1, -1};
  bool grouped_done = false;
  // (bit 0: the dense branch's order of threshold and alpha; bit 1: FMA accumulation, option spgemm_fma, real operands)
  const int dr = (dense_rule ? 1 : 0) | ((options().spgemm_fma && !A.cplx) ? 2 : 0);
  if (!loose_in && grouped_first_n[A.cplx ? 1 : 0] == n && sv_opt < 0 && options().spgemm_force_bin <= 0 && m == A.cols && (int64_t)n * 1536 < (1ll << 33)) {
    constexpr int64_t kSlot = 1536;   // rows of the largest table class: no column of a finished group holds more
    tmp_total = (int64_t)n * kSlot;
    tmp_inner.alloc((size_t)tmp_total + kIndexSlack);
    tmp_val.alloc(((size_t)tmp_total + kIndexSlack) * A.wval());
    bin.alloc(n);
    hipLaunchKernelGGL(k_linear_i64, dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), tmpoff.p, (int64_t)n + 1, kSlot);
    GroupedInfo gi;
    t_num.start();
    hipEvent_t late = timing ? get_event() : nullptr;
    const bool ok = spgemm_grouped(A, B, tmpoff.p, tmp_inner.p, tmp_val.p, count.p, bin.p, alpha, threshold, dr, 2, &gi, late);
    if (ok && gi.failed_cols == 0) {
      grouped_done = true;
      if (late) std::swap(t_num.a, late);
      st.grouped = 1;
      st.products = gi.products;
      st.bin_cols[5] = n;
      st.gh_failed_cols = 0;
      st.gh_groups = gi.groups;
      st.gh_level = gi.level;
      st.gh_minhash = gi.minhash;
      st.gh_union_ratio = gi.union_ratio;
      st.gh_tile_rows = gi.tile_rows;
    } else {
      grouped_first_n[A.cplx ? 1 : 0] = -1;   // (columns it may have written are recomputed below)
      count.zero();
      tmp_inner.release();
      tmp_val.release();
    }
    if (late) event_pool().push_back(late);
  }
  const bool slab_try = !grouped_done && options().spgemm_force_bin <= 0 && (sv_opt < 0 || sv_opt / 100 == 4 || sv_opt / 100 == 6) &&
                        A.nnz < 1000000000LL && B.nnz < 1000000000LL;
  const int snb = cdiv(n, SJ);
  const bool tile_expand = !A.cplx && options().spgemm_fma == 1;
  DevBuf<int32_t> bfirst_own, blast_own, blen_own, aspan, blk_lo, blk_w, blk_kmin, blk_kn;
  DevBuf<int64_t> aeoff, bsz, tsz, blk_boff, blk_toff;
  DevBuf<char> runs;
  int64_t slab_tot[3] = {0, 0, 0};
  bool use_slab = false;
  if (slab_try) {
    const int32_t *bfirst = cmin.p, *blast = cmax.p;
    if (&A != &B) {
      bfirst_own.alloc(n); blast_own.alloc(n); blen_own.alloc(n);
      hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), lview(B), bfirst_own.p, blast_own.p, blen_own.p);
      bfirst = bfirst_own.p; blast = blast_own.p;
    }
    aspan.alloc((size_t)nka); aeoff.alloc((size_t)nka + 1);   // local ids 0 .. nka (column ka + id)
    // (real operands under option spgemm_fma = 1: aligned, zero-padded slots for the MFMA tile kernel)
    if (tile_expand)
      hipLaunchKernelGGL(k_span_aligned, dim3(cdiv(nka, 256)), dim3(256), 0, stream(), cmin_own.p, cmax_own.p, aspan.p, nka, tile_expand_align());
    else
      hipLaunchKernelGGL(k_span_of, dim3(cdiv(nka, 256)), dim3(256), 0, stream(), cmin_own.p, cmax_own.p, aspan.p, nka);
    scan_async<int32_t>(aspan.p, aeoff.p, (int64_t)nka);
    blk_lo.alloc(snb); blk_w.alloc(snb); blk_kmin.alloc(snb); blk_kn.alloc(snb);
    bsz.alloc(snb); tsz.alloc(snb); blk_boff.alloc((size_t)snb + 1); blk_toff.alloc((size_t)snb + 1);
    if (A.cplx)
      hipLaunchKernelGGL((k_slab_plan<SLAB_CJ>), dim3(cdiv((int64_t)snb * WAVE, 256)), dim3(256), 0, stream(), n, bfirst,
                         blast, cmin.p, cmax.p, blk_lo.p, blk_w.p, blk_kmin.p, blk_kn.p, bsz.p, tsz.p, snb, ctile_try ? 16 : 0);
    else
      hipLaunchKernelGGL((k_slab_plan<SLAB_J>), dim3(cdiv((int64_t)snb * WAVE, 256)), dim3(256), 0, stream(), n, bfirst,
                         blast, cmin.p, cmax.p, blk_lo.p, blk_w.p, blk_kmin.p, blk_kn.p, bsz.p, tsz.p, snb,
                         options().spgemm_fma == 1 ? tile_expand_align() : 0);
    hipLaunchKernelGGL(k_slab_reduce, dim3(64), dim3(256), 0, stream(), blk_w.p, blk_kn.p, snb, aspan.p, nka, stats.p);
    scan_async<int64_t>(bsz.p, blk_boff.p, (int64_t)snb);
    scan_async<int64_t>(tsz.p, blk_toff.p, (int64_t)snb);
    ScalarFetch f;
    f.add(aeoff.p + nka, 1, &slab_tot[0]);
    f.add(blk_boff.p + snb, 1, &slab_tot[1]);
    f.add(blk_toff.p + snb, 1, &slab_tot[2]);
    f.add(stats.p, 24, hstats);
    f.run();
    // use it when the window fits the register slabs and the zero padding stays small
    const int64_t max_w = (int64_t)hstats[16], max_kn = (int64_t)hstats[17];
    // (the multiplier tile of a block is staged in LDS by the expansion kernel: up to 128 KB, requested explicitly)
    const bool fits = max_w > 0 && max_w <= slab_rows && ((max_kn + 1) | 1) * SJ * (int64_t)esz <= 128 * 1024;
    // (aligned slots, tile_expand: up to two alignment units of zero padding per column are not holes)
    const double pad_allow = tile_expand ? 2.0 * (double)tile_expand_align() * (double)nka : 0.0;
    // (the tile kernel's work follows the runs of A and the k range, not the entries of B: a sparse B inside wide column
    // extents -- 3 I - X^2 near convergence -- only has to keep its multiplier tiles within twice the size of A)
    const double btile_allow = tile_expand ? std::max(1.5 * (double)B.nnz + 4096.0, 2.0 * (double)A.nnz) : 1.5 * (double)B.nnz + 4096.0;
    const bool dense_runs = (double)slab_tot[0] <= 1.5 * (double)A.nnz + pad_allow && (double)slab_tot[1] <= btile_allow &&
                            (double)slab_tot[0] >= 48.0 * (double)hstats[18];  // mean run of the non-empty columns >= 48 rows
    use_slab = fits && (dense_runs || sv_opt / 100 == 4);
    if (!use_slab && std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
      std::fprintf(stderr, "spgemm: no slab path: n %d nnzA %lld nnzB %lld max_w %lld max_kn %lld fits %d runsA %lld tilesB %lld nonempty %llu same %d\n", n,
                   (long long)A.nnz, (long long)B.nnz, (long long)max_w, (long long)max_kn, (int)fits, (long long)slab_tot[0],
                   (long long)slab_tot[1], hstats[18], (int)(&A == &B));
  }
  if (loose_in && !(use_slab && &A == &B)) {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    DevMat Ap = packed_copy(A);
    if (&A == &B) {
      spgemm(Ap, Ap, C, alpha, threshold, dense_rule, loose, arange, nullptr);
    } else {
      DevMat Bp = packed_copy(B);
      spgemm(Ap, Bp, C, alpha, threshold, dense_rule, loose, arange, nullptr);
    }
    return;
  }
  // ---- operands without run structure, FMA arithmetic, real, square, one rank: 16 x 16 blocks of a clustered index order
  // on the FP64 matrix cores (spgemm_block.hip); declined (false) when the clustering finds no blocks worth it
  // (only operands whose row windows lie beyond the direct-mapped LDS kernels -- 4096 rows: those kernels multiply in label
  // order and serve banded operands that merely failed the run-density test; the block path multiplies in its own order)
  if (!use_slab && !grouped_done && !loose_in && block_eligible(A, B, arange) && !strip_ctx().active &&
      (options().block_path == 2 || (slab_try && (int64_t)hstats[16] > 4096))) {
    if (timing) {   // (the helper brings its own timers)
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    if (try_block_path(A, B, C, alpha, threshold, dense_rule)) return;
    t_all = EventTimer(timing);
    t_num = EventTimer(timing);
    t_all.start();
  }
  // a dimension whose products needed row strips last time (columns with more distinct rows than the grouped kernel's
  // tables hold): straight to the strips, without the attempt on the whole operand
  if (!use_slab && !grouped_done && !loose_in && !arange && sv_opt < 0 && options().spgemm_force_bin <= 0 && !strip_ctx().active &&
      strips_memory(A.cplx)[0] == n && strips_memory(A.cplx)[1] >= 2 && m == A.cols) {
    DevBuf<unsigned long long> ssum(1);
    ssum.zero();
    if (nka) hipLaunchKernelGGL(k_span_sum, dim3(cdiv(nka, 256)), dim3(256), 0, stream(), cmin_own.p, cmax_own.p, nka, ssum.p);
    unsigned long long span_sum = 0;
    {
      ScalarFetch f;
      f.add(ssum.p, 1, &span_sum);
      f.run();
    }
    const double avg_span = (double)span_sum / (double)std::max(1, nka);
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    // (the iterates of a solve fill in: when the remembered strip count no longer fits, twice the strips -- up to the 16
    // the merge kernel interleaves -- before the whole-operand path, whose upper-bound slots are tens of GB here)
    for (int S = strips_memory(A.cplx)[1]; S <= kMaxStrips; S *= 2) {
      int shift = 6;
      while ((1 << (shift + 1)) * 8.0 * S <= avg_span && shift < 24) ++shift;
      if (spgemm_striped(A, B, C, alpha, threshold, dense_rule, S, shift)) {
        strips_memory(A.cplx)[1] = S;
        SpgemmStats& ls = last_spgemm_stats();
        ls.nnz_a = A.nnz;
        ls.nnz_c = C.nnz;
        ls.strips = S;
        SpgemmAccum& ac = spgemm_accum();
        ac.calls -= S - 1;
        ac.alg_bytes -= (double)(S - 1) * ((A.cplx ? 20.0 : 12.0) * (double)B.nnz + 4.0 * (2.0 * n + A.cols + 3.0));
        return;
      }
      if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM")) std::fprintf(stderr, "spgemm: %d strips no longer fit\n", S);
    }
    strips_memory(A.cplx)[0] = -1;   // (did not fit this time: the ordinary path decides again)
    t_all = EventTimer(timing);
    t_num = EventTimer(timing);
    t_all.start();
  }
  st.slab = use_slab ? 1 : 0;
  // real run-like operands under option spgemm_fma = 1: the same plan and operands, numeric phase on the matrix cores
  const bool use_tile = use_slab && !A.cplx && options().spgemm_fma == 1 && (sv_opt < 0 || sv_opt / 100 == 6) &&
                        spgemm_tile_fits((int)hstats[17], (int)hstats[16]);
  if (use_slab && !A.cplx && options().spgemm_fma == 1 && !use_tile && (int64_t)hstats[16] > SLAB_NW * SLAB_SL * WAVE) {
    // (neither the tile kernel -- k range beyond its LDS tile -- nor the four-wave FMA loop: general kernels)
    use_slab = false;
    st.slab = 0;
  }
  // complex run-like operands under FMA arithmetic: the same plan and operands, numeric phase on the matrix cores
  // (a geometry it does not take goes to the register-slab kernel when the window fits that one, to the general kernels otherwise)
  const bool use_ctile = use_slab && ctile_try && (sv_opt < 0 || sv_opt == 400) && spgemm_tile_c_fits((int)hstats[17], (int)hstats[16]);
  if (use_slab && A.cplx && !use_ctile && (int64_t)hstats[16] > 8 * SLAB_CSL * WAVE) {
    use_slab = false;
    st.slab = 0;
  }
  if (use_slab) {
    tmp_total = slab_tot[2];
    if (A.cplx)
      hipLaunchKernelGGL((k_slab_tmpoff<SLAB_CJ>), dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), n, blk_w.p, blk_toff.p, tmpoff.p);
    else
      hipLaunchKernelGGL((k_slab_tmpoff<SLAB_J>), dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), n, blk_w.p, blk_toff.p, tmpoff.p);
  } else if (!grouped_done) {
    // ---- general plan: per output column its row window, product count, upper bound and kernel bin
    lo.alloc(n); span.alloc(n); bin.alloc(n); ub.alloc((size_t)n + 1); ip.alloc(n);
    hipLaunchKernelGGL(k_spgemm_plan, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(B), cmin.p,
                       cmax.p, clen.p, lo.p, span.p, bin.p, ub.p, ip.p, stats.p, options().spgemm_force_bin);
    if (!A.cplx && options().spgemm_force_bin <= 0)
      hipLaunchKernelGGL(k_pair_bins, dim3(cdiv((n + 1) / 2, 256)), dim3(256), 0, stream(), bin.p, n);
    hipLaunchKernelGGL(k_bin_hist, dim3(std::min(cdiv(n, 256), 512)), dim3(256), 0, stream(), bin.p, ip.p, span.p, n, stats.p);
    scan_async<int64_t>(ub.p, tmpoff.p, (int64_t)n);
    ScalarFetch f;
    f.add(stats.p, 24, hstats);
    f.add(tmpoff.p + n, 1, &tmp_total);
    f.run();
    for (int i = 0; i < 6; ++i) st.bin_cols[i] = (int64_t)hstats[i];
    st.bin_cols[5] += (int64_t)hstats[6];
    st.products = (int64_t)hstats[7];
  }
  st.tmp_entries = tmp_total;

  // + slack: a loose product is read by kernels that fetch whole 64-entry chunks past a column's end
  if (!grouped_done) {
    tmp_inner.alloc((size_t)tmp_total + kIndexSlack);
    tmp_val.alloc(((size_t)tmp_total + kIndexSlack) * A.wval());
  }
  DevBuf<int32_t> tmp2_inner;
  DevBuf<double> tmp2_val;
  DevBuf<int64_t> tmpoff2;
  DevBuf<double> aexp, bblk;
  // one zeroed block of 8-byte words: [flag 2 | entries of the product per block snb + 1 | products per block snb |
  // (dot, trace) per block 2 snb] -- blocks without work write nothing
  DevBuf<int64_t> zwords, blk_prod_scan;
  int64_t *blk_prod = nullptr, *fz_pnnz = nullptr, *fz_flag = nullptr;
  double* fz_part = nullptr;
  DevBuf<char> fz_args;
  DevBuf<int32_t> fz_first, fz_last;
  DevBuf<double> fz_tiles;
  DevBuf<int64_t> tile_ooff, tile_otoff;   // (tile kernel: where every column's run / every block's tile rows start)
  bool fuse_now = false;
  DevBuf<int32_t> sc_xplast, sc_oplast;   // (label-aware panel step inside a band scope: largest label per column of X / of the result)
  if (use_slab) {
    zwords.alloc((size_t)4 * snb + 4);
    zwords.zero();
    fz_flag = zwords.p;
    fz_pnnz = zwords.p + 2;
    blk_prod = zwords.p + 3 + snb;
    fz_part = reinterpret_cast<double*>(zwords.p + 3 + 2 * (size_t)snb);
    blk_prod_scan.alloc((size_t)snb + 1);
    aexp.alloc(((size_t)slab_tot[0] + 1) * A.wval());
    bblk.alloc(((size_t)slab_tot[1] + 16 * SJ) * A.wval());  // slack: the loop prefetches a few rows past the last tile
    const bool same = (&A == &B);
    if (tile_expand && nka)
      hipLaunchKernelGGL(k_aligned_offsets<double>, dim3(cdiv((int64_t)nka * WAVE, 256)), dim3(256), 0, stream(), cmin_own.p,
                         cmax_own.p, aeoff.p, aexp.p, nka, tile_expand_align());
    runs.alloc(((size_t)nka + 4) * sizeof(SlabRun));
    hipLaunchKernelGGL(k_slab_runs, dim3(cdiv(nka + 4, 256)), dim3(256), 0, stream(), cmin_own.p, cmax_own.p, aeoff.p,
                       reinterpret_cast<const char*>(aexp.p), (int)esz, reinterpret_cast<SlabRun*>(runs.p), nka);
    const int pitch = ((int)hstats[17] + 1) | 1;
    if ((size_t)pitch * SJ * esz > 64 * 1024) {   // tiles beyond the default dynamic-LDS limit
      static bool raised = false;
      if (!raised) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_slab_expand_b<double, SLAB_J>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_slab_expand_b<double2, SLAB_CJ>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
        raised = true;
      }
    }
    const int32_t* clen_opt = timing ? clen.p : (const int32_t*)nullptr;  // product count: statistics only
    if (A.cplx) {
      double2* ae = reinterpret_cast<double2*>(aexp.p);
      if (!same)
        hipLaunchKernelGGL(k_slab_expand_a<double2>, dim3(cdiv((int64_t)nka * WAVE, 256)), dim3(256), 0, stream(),
                           Ar, cmin_own.p, aeoff.p, ae);
      hipLaunchKernelGGL((k_slab_expand_b<double2, SLAB_CJ>), dim3(xcd_grid(snb)), dim3(256), (size_t)pitch * SJ * esz,
                         stream(), lview(B), blk_kmin.p, blk_kn.p, blk_boff.p, reinterpret_cast<double2*>(bblk.p), snb, pitch,
                         same ? 1 : 0, aeoff.p, ae, clen_opt, blk_prod);
    } else {
      if (!same)
        hipLaunchKernelGGL(k_slab_expand_a<double>, dim3(cdiv((int64_t)nka * WAVE, 256)), dim3(256), 0, stream(),
                           Ar, cmin_own.p, aeoff.p, aexp.p);
      hipLaunchKernelGGL((k_slab_expand_b<double, SLAB_J>), dim3(xcd_grid(snb)), dim3(256), (size_t)pitch * SJ * esz,
                         stream(), lview(B), blk_kmin.p, blk_kn.p, blk_boff.p, bblk.p, snb, pitch, same ? 1 : 0, aeoff.p,
                         aexp.p, clen_opt, blk_prod);
    }
  }
  if (!grouped_done) t_num.start();
  if (use_ctile) {   // the product as dense complex column runs in the slots; compressed columns by the pack pass below
    fz_first.alloc((size_t)n); fz_last.alloc((size_t)n);
    tile_ooff.alloc((size_t)n + 1);
    TileLaunch tl;
    tl.runs = reinterpret_cast<const SlabRun*>(runs.p) - ka;
    tl.bblk = bblk.p; tl.blk_boff = blk_boff.p; tl.blk_kmin = blk_kmin.p; tl.blk_kn = blk_kn.p; tl.blk_lo = blk_lo.p;
    tl.blk_w = blk_w.p; tl.blk_toff = blk_toff.p; tl.out_val = tmp_val.p; tl.count = count.p;
    tl.ofirst = fz_first.p; tl.olast = fz_last.p; tl.ooff = tile_ooff.p;
    tl.alpha = alpha; tl.threshold = threshold; tl.dense_rule = dr; tl.ncols = n; tl.nblocks = snb;
    tl.max_kn = (int)hstats[17]; tl.max_w = (int)hstats[16]; tl.epi = 0;
    launch_spgemm_tile_c(tl);
    for (int i = 0; i < 7; ++i) hstats[i] = 0;
  } else if (use_slab && A.cplx) {
    if ((int64_t)hstats[16] <= SLAB_CNW * SLAB_CSL * WAVE)
      hipLaunchKernelGGL((k_spgemm_slab_c<SLAB_CNW>), dim3(xcd_grid(snb)), dim3(SLAB_CNW * WAVE), 0, stream(),
                         reinterpret_cast<const SlabRun*>(runs.p) - ka, reinterpret_cast<const double2*>(bblk.p), blk_boff.p,
                         blk_kmin.p, blk_kn.p, blk_lo.p, blk_w.p, blk_toff.p, tmp_inner.p,
                         reinterpret_cast<double2*>(tmp_val.p), count.p, alpha, threshold, dr, n, snb);
    else
      hipLaunchKernelGGL((k_spgemm_slab_c<8>), dim3(xcd_grid(snb)), dim3(8 * WAVE), 0, stream(),
                         reinterpret_cast<const SlabRun*>(runs.p) - ka, reinterpret_cast<const double2*>(bblk.p), blk_boff.p,
                         blk_kmin.p, blk_kn.p, blk_lo.p, blk_w.p, blk_toff.p, tmp_inner.p,
                         reinterpret_cast<double2*>(tmp_val.p), count.p, alpha, threshold, dr, n, snb);
    for (int i = 0; i < 7; ++i) hstats[i] = 0;
  } else if (use_slab) {
    // occupancy experiments (408 / 409): unused dynamic LDS limits the workgroups per CU to 2 / 3 instead of 4
    const size_t occ_lds = sv_opt == 408 ? 60 * 1024 : sv_opt == 409 ? 50 * 1024 : 0;
    auto launch_slab = [&](auto fma_tag) {
      hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, SLAB_NW, decltype(fma_tag)::value>), dim3(xcd_grid(snb)),
                         dim3(SLAB_NW * WAVE), occ_lds, stream(), reinterpret_cast<const SlabRun*>(runs.p) - ka, bblk.p, blk_boff.p,
                         blk_kmin.p, blk_kn.p, blk_lo.p, blk_w.p, blk_toff.p, tmp_inner.p, tmp_val.p, count.p, alpha,
                         threshold, dr, n, snb, (const SlabFuseArgs*)nullptr);
    };
    int abl = (sv_opt / 100 == 4) ? sv_opt % 100 : 0;  // 401..404: ablations for timing experiments (WRONG results)
#ifndef NTP_ABLATIONS
    // (the wrong-result loop variants are compiled only into the experiment build, -DNTP_ABLATIONS -- tools/bench_spgemm.py;
    // in the product library the option values select the ordinary loop)
    if (abl >= 1 && abl <= 4) abl = 0;
#endif
    // narrow windows: one / two slabs per wave (more resident waves); 410 keeps three slabs for comparison
    const int64_t max_w_now = (int64_t)hstats[16];
    auto launch_narrow = [&](auto sl_tag) {
      constexpr int SLN = decltype(sl_tag)::value;
      hipLaunchKernelGGL((k_spgemm_slab_n<SLN>), dim3(xcd_grid(snb)), dim3(SLAB_NW * WAVE), 0, stream(),
                         reinterpret_cast<const SlabRun*>(runs.p) - ka, bblk.p, blk_boff.p, blk_kmin.p, blk_kn.p, blk_lo.p,
                         blk_w.p, blk_toff.p, tmp_inner.p, tmp_val.p, count.p, alpha, threshold, dr, n, snb);
    };
    // fused epilogue of a purification step (SlabFusion): the same loop, the result leaves the registers merged
    // (one rank: A and B are the iterate itself; a rank of several: B is its column panel [panel_c0, panel_c0 + n) of
    // the iterate and A holds -- at least -- those columns, as the gathered halo operand does)
    const bool whole = &A == &B && !arange && m == n && fuse && fuse->panel_c0 < 0;
    const bool panel = fuse && fuse->panel_c0 >= 0 && &A != &B && A.cols == m && ka <= fuse->panel_c0 && fuse->panel_c0 + n <= kb;
    const int xshift = panel ? fuse->panel_c0 - ka : 0;   // own column j = local column xshift + j of the A-side arrays
    fuse_now = fuse != nullptr && fuse->mode != 0 && (whole || panel) && abl == 0 && (!options().spgemm_fma || use_tile) &&
               (sv_opt < 0 || (use_tile && sv_opt / 100 == 6)) && fuse->D && !fuse->D->cplx && !fuse->D->loose() && !fuse->D->expanded() && fuse->D->rows == m &&
               fuse->D->cols == n;
    if (fuse_now && fuse->mode == 2 && B.zero_free != 1) {
      // the merge reads a zero of the expanded columns of X as "no entry": make sure no stored value is one (once per
      // solve: the results of the fused steps are zero-free by construction)
      DevBuf<unsigned long long> zc(1);
      zc.zero();
      hipLaunchKernelGGL(k_count_zero_values, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), lview(B), zc.p);
      unsigned long long hz = 0;
      ScalarFetch f;
      f.add(zc.p, 1, &hz);
      f.run();
      if (hz == 0) B.zero_free = 1;
      else fuse_now = false;
    }
    const DotOperand* dop_p = fuse_now ? &dot_operand(*fuse->D) : nullptr;
    if (fuse_now && !use_tile && dop_p->max_tile > SLAB_DTILE) fuse_now = false;   // (D columns too wide for the LDS tile)
    if (fuse_now) {
      const DotOperand& dop = *dop_p;
      SlabFuseArgs fz;
      fz.am = fuse->am; fz.bm = fuse->bm; fz.thr_m = fuse->threshold;
      fz.xexp = aexp.p; fz.xoff = aeoff.p + xshift; fz.xmin = cmin_own.p + xshift; fz.xmax = cmax_own.p + xshift;
      fz.dexp = dop.dexp.p; fz.doff = dop.doff.p; fz.dmin = dop.dmin.p; fz.dmax = dop.dmax.p;
      fz.part = fz_part; fz.pnnz = reinterpret_cast<long long*>(fz_pnnz); fz.flag = reinterpret_cast<int*>(fz_flag);
      fz.col_offset = fuse->col_offset;
      // (a panel step of a solve in a recovered band order across ranks: the merge's "beyond the last entry" tests on the caller's
      // labels -- scope_labels(), kernels.hpp; the largest label of every column of X from its compressed columns)
      if (panel && use_tile && scope_labels() != nullptr) {
        sc_xplast.alloc((size_t)n); sc_oplast.alloc((size_t)n);
        hipLaunchKernelGGL(k_col_plast, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), lview(B), scope_labels(), sc_xplast.p);
        fz.lab = scope_labels(); fz.xplast = sc_xplast.p; fz.oplast = sc_oplast.p;
      }
      fz_first.alloc((size_t)n); fz_last.alloc((size_t)n);
      fz_tiles.alloc((size_t)tmp_total + kIndexSlack);
      fz.ofirst = fz_first.p; fz.olast = fz_last.p; fz.tiles = fz_tiles.p;
      fz_args.alloc(sizeof(SlabFuseArgs));
      fz_args.upload(reinterpret_cast<const char*>(&fz), sizeof(SlabFuseArgs));
      auto launch_fused = [&](auto nw_tag, auto mode_tag, auto epi_tag) {
        constexpr int FNW = decltype(nw_tag)::value;
        hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, FNW, decltype(mode_tag)::value, decltype(epi_tag)::value>),
                           dim3(xcd_grid(snb)), dim3(FNW * WAVE), 0, stream(), reinterpret_cast<const SlabRun*>(runs.p) - ka,
                           bblk.p, blk_boff.p, blk_kmin.p, blk_kn.p, blk_lo.p, blk_w.p, blk_toff.p, tmp_inner.p, tmp_val.p,
                           count.p, alpha, threshold, dr, n, snb, reinterpret_cast<const SlabFuseArgs*>(fz_args.p));
      };
      auto by_mode = [&](auto nw_tag, auto mode_tag) {
        if (fuse->mode == 1) launch_fused(nw_tag, mode_tag, std::integral_constant<int, 1>{});
        else launch_fused(nw_tag, mode_tag, std::integral_constant<int, 2>{});
      };
      if (use_tile) {
        tile_ooff.alloc((size_t)n + 1);
        tile_otoff.alloc((size_t)snb + 1);
        TileLaunch tl;
        tl.runs = reinterpret_cast<const SlabRun*>(runs.p) - ka;
        tl.bblk = bblk.p; tl.blk_boff = blk_boff.p; tl.blk_kmin = blk_kmin.p; tl.blk_kn = blk_kn.p; tl.blk_lo = blk_lo.p;
        tl.blk_w = blk_w.p; tl.blk_toff = blk_toff.p; tl.out_val = tmp_val.p; tl.count = count.p;
        tl.ofirst = fz_first.p; tl.olast = fz_last.p; tl.ooff = tile_ooff.p; tl.otoff = tile_otoff.p;
        tl.alpha = alpha; tl.threshold = threshold; tl.dense_rule = dr; tl.ncols = n; tl.nblocks = snb;
        tl.max_kn = (int)hstats[17]; tl.max_w = (int)hstats[16]; tl.epi = fuse->mode; tl.fz = &fz; tl.rows = tile_rows();
        tl.labelled = fz.lab != nullptr; tl.nrows = m;
        tl.abase = aexp.p; tl.abytes = aexp.n * sizeof(double);
        tl.dbase = dop.dexp.p; tl.dbytes = dop.dexp.n * sizeof(double);
        launch_spgemm_tile(tl);   // (writes the end markers of tile_ooff / tile_otoff as well)
      } else if (max_w_now > 6 * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{});
      else if (max_w_now > SLAB_NW * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
      else by_mode(std::integral_constant<int, SLAB_NW>{}, std::integral_constant<int, 8>{});
    } else if (use_tile) {   // the product as dense column runs in the slots; compressed columns by the pack pass below
      fz_first.alloc((size_t)n); fz_last.alloc((size_t)n);
      tile_ooff.alloc((size_t)n + 1);
      TileLaunch tl;
      tl.runs = reinterpret_cast<const SlabRun*>(runs.p) - ka;
      tl.bblk = bblk.p; tl.blk_boff = blk_boff.p; tl.blk_kmin = blk_kmin.p; tl.blk_kn = blk_kn.p; tl.blk_lo = blk_lo.p;
      tl.blk_w = blk_w.p; tl.blk_toff = blk_toff.p; tl.out_val = tmp_val.p; tl.count = count.p;
      tl.ofirst = fz_first.p; tl.olast = fz_last.p; tl.ooff = tile_ooff.p;
      tl.alpha = alpha; tl.threshold = threshold; tl.dense_rule = dr; tl.ncols = n; tl.nblocks = snb;
      tl.max_kn = (int)hstats[17]; tl.max_w = (int)hstats[16]; tl.epi = 0; tl.rows = tile_rows();
      tl.abase = aexp.p; tl.abytes = aexp.n * sizeof(double);
      launch_spgemm_tile(tl);
    } else if (max_w_now > 6 * SLAB_SL * WAVE)         // 1153 .. 1536 rows: eight waves per workgroup
      hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, 8, 0>), dim3(xcd_grid(snb)), dim3(8 * WAVE), 0, stream(),
                         reinterpret_cast<const SlabRun*>(runs.p) - ka, bblk.p, blk_boff.p, blk_kmin.p, blk_kn.p, blk_lo.p,
                         blk_w.p, blk_toff.p, tmp_inner.p, tmp_val.p, count.p, alpha, threshold, dr, n, snb, (const SlabFuseArgs*)nullptr);
    else if (max_w_now > SLAB_NW * SLAB_SL * WAVE)   // 769 .. 1152 rows: six waves per workgroup
      hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, 6, 0>), dim3(xcd_grid(snb)), dim3(6 * WAVE), 0, stream(),
                         reinterpret_cast<const SlabRun*>(runs.p) - ka, bblk.p, blk_boff.p, blk_kmin.p, blk_kn.p, blk_lo.p,
                         blk_w.p, blk_toff.p, tmp_inner.p, tmp_val.p, count.p, alpha, threshold, dr, n, snb, (const SlabFuseArgs*)nullptr);
    else if (abl == 0 && !options().spgemm_fma && sv_opt != 410 && max_w_now <= SLAB_NW * WAVE) launch_narrow(std::integral_constant<int, 1>{});
    else if (abl == 0 && !options().spgemm_fma && sv_opt != 410 && max_w_now <= 2 * SLAB_NW * WAVE) launch_narrow(std::integral_constant<int, 2>{});
#ifdef NTP_ABLATIONS
    else if (abl == 1) launch_slab(std::integral_constant<int, 2>{});
    else if (abl == 2) launch_slab(std::integral_constant<int, 3>{});
    else if (abl == 3) launch_slab(std::integral_constant<int, 4>{});
    else if (abl == 4) launch_slab(std::integral_constant<int, 5>{});
#endif
    else if (abl == 5) launch_slab(std::integral_constant<int, 6>{});
    else if (abl == 6) launch_slab(std::integral_constant<int, 7>{});
    else if (abl == 7) launch_slab(std::integral_constant<int, 8>{});
    else if (options().spgemm_fma) launch_slab(std::integral_constant<int, 1>{});
    else if (sv_opt == 410) launch_slab(std::integral_constant<int, 0>{});  // the plain loop (comparison)
    else launch_slab(std::integral_constant<int, 8>{});  // whole periods without pointer arithmetic + rotating prefetch
    for (int i = 0; i < 7; ++i) hstats[i] = 0;
  }
  // ---- operands without run structure: groups of similar columns share the fetch and the hashing of the A columns
  // (spgemm_grouped.hip); taken when most columns are beyond the direct-mapped LDS windows
  if (!use_slab && !grouped_done) {
    unsigned long long nonempty = 0;
    for (int i = 1; i <= 6; ++i) nonempty += hstats[i];
    const bool want = nonempty > 0 && options().spgemm_force_bin <= 0 &&
                      (sv_opt == 500 || (sv_opt >= 511 && sv_opt <= 541) || (sv_opt < 0 && hstats[5] * 2 >= nonempty));
    if (want) {
      GroupedInfo gi;
      hipEvent_t late = timing ? get_event() : nullptr;   // "numeric" = the grouped kernel launches, not its planning passes
      if (spgemm_grouped(A, B, tmpoff.p, tmp_inner.p, tmp_val.p, count.p, bin.p, alpha, threshold, dr, sv_opt >= 500 ? 1 : 0, &gi, late)) {
        for (int i = 1; i <= 6; ++i) hstats[i] = 0;
        hstats[5] = (unsigned long long)gi.failed_cols;   // what the grouped kernel handed back: per-column LDS hash below
        st.grouped = 1;
        if (late) std::swap(t_num.a, late);
        // the next multiply of this dimension goes to the grouped kernel first (see the top of this function)
        if (gi.failed_cols == 0 && gi.minhash && sv_opt < 0) grouped_first_n[A.cplx ? 1 : 0] = n;
      }
      if (late) event_pool().push_back(late);
      st.gh_failed_cols = gi.failed_cols;
      st.gh_groups = gi.groups;
      st.gh_level = gi.level;
      st.gh_minhash = gi.minhash;
      st.gh_union_ratio = gi.union_ratio;
      st.gh_tile_rows = gi.tile_rows;
      // Most groups outgrew the largest table: the columns of the product hold too many distinct rows (a 3-D
      // Hamiltonian).  Inside a strip multiply that is the answer; otherwise multiply in row strips of A (above), with
      // twice the strips until every piece fits; the strip count that worked is remembered per dimension.
      static const int fail_pct = std::getenv("NTPOLY_AMD_STRIP_FAIL_PCT") ? std::atoi(std::getenv("NTPOLY_AMD_STRIP_FAIL_PCT")) : 10;
      if (gi.failed_cols * 100 > (int64_t)n * fail_pct && sv_opt < 0 && !arange && !loose_in) {
        StripCtx& sc = strip_ctx();
        if (sc.active) {
          sc.failed = true;
          if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
            std::fprintf(stderr, "spgemm strip: %lld of %d columns failed (groups %lld, table class %d, union ratio %.2f, nnzA %lld)\n",
                         (long long)gi.failed_cols, n, (long long)gi.groups, gi.level, gi.union_ratio, (long long)A.nnz);
          if (timing) {
            event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
            event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
          }
          C.reset_empty(m, n, A.cplx);
          return;
        }
        int* mem = strips_memory(A.cplx);
        DevBuf<unsigned long long> ssum(1);
        ssum.zero();
        if (nka) hipLaunchKernelGGL(k_span_sum, dim3(cdiv(nka, 256)), dim3(256), 0, stream(), cmin_own.p, cmax_own.p, nka, ssum.p);
        unsigned long long span_sum = 0;
        {
          ScalarFetch f;
          f.add(ssum.p, 1, &span_sum);
          f.run();
        }
        const double avg_span = (double)span_sum / (double)std::max(1, nka);
        if (timing) {   // (the strip multiplies time themselves)
          event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
          event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
        }
        for (int S = (mem[0] == n && mem[1] >= 2) ? mem[1] : 2; S <= kMaxStrips; S *= 2) {
          // blocks of rows: about eight per strip inside a column's extent, at least 64 rows
          int shift = 6;
          while ((1 << (shift + 1)) * 8.0 * S <= avg_span && shift < 24) ++shift;
          if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
            std::fprintf(stderr, "spgemm: %lld of %d columns beyond the grouped tables; trying %d strips, blocks of %d rows (mean extent %.0f)\n",
                         (long long)gi.failed_cols, n, S, 1 << shift, avg_span);
          if (spgemm_striped(A, B, C, alpha, threshold, dense_rule, S, shift)) {
            mem[0] = n;
            mem[1] = S;
            SpgemmStats& ls = last_spgemm_stats();
            ls.nnz_a = A.nnz;
            ls.nnz_c = C.nnz;
            ls.strips = S;
            // (the strip multiplies accounted for themselves: one multiply, B read once as far as the algorithm goes)
            SpgemmAccum& ac = spgemm_accum();
            ac.calls -= S - 1;
            ac.alg_bytes -= (double)(S - 1) * ((A.cplx ? 20.0 : 12.0) * (double)B.nnz + 4.0 * (2.0 * n + A.cols + 3.0));
            return;
          }
        }
        // (not even sixteen strips: the per-column kernels below take the columns; timers were handed back above)
        t_all = EventTimer(timing);
        t_num = EventTimer(timing);
        t_all.start();
        t_num.start();
      }
    }
  }
  unsigned long long hash_big = 0;  // columns that outgrew the small hash table
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    T* tv = reinterpret_cast<T*>(tmp_val.p);
    int variant = options().spgemm_variant;
    if (variant / 100 == 4 || variant / 100 == 5 || variant / 100 == 6) variant = -1;  // slab / grouped kernel requested but not applicable
    if constexpr (!Sc<T>::cplx) {
      // real operands that are not run-like: column-pair kernel, register-set depth from the mean column length of A
      // (its accumulate is an LDS atomic add of the rounded product: no FMA form -- under option spgemm_fma the window
      // kernels below take these columns)
      if (variant < 0 && !options().spgemm_fma && A.nnz < 500000000LL && B.nnz < 2000000000LL) {
        const double avg = (double)A.nnz / (double)std::max(1, nka);
        const int need = (int)std::ceil(avg / 64.0);
        variant = 300 + 10 * std::min(6, std::max(2, need)) + 1;
      }
    }
    auto wrt_of = [&](int b) { return (int)((hstats[9 + b] + 63) / 64 * 64); };
    if constexpr (!Sc<T>::cplx) {
      if (variant >= 300 && variant < 400 && A.nnz < 500000000LL && B.nnz < 2000000000LL) {  // column-pair kernel
        const int maxch = (variant / 10) % 10, nw = variant % 10;  // variant = 3<MAXCH><NW>
        // window classes 1 and 2 (spans <= 1024) share one launch sized by the largest span present; class 3
        // (<= 2048) gets its own so that a few wide columns do not cost everybody LDS occupancy
        const int groups[2][2] = {{1, 2}, {3, 3}};
        for (int gi = 0; gi < 2; ++gi) {
          const int blo = groups[gi][0], bhi = groups[gi][1];
          unsigned long long cols_here = 0;
          int w = 64;
          for (int bsel = blo; bsel <= bhi; ++bsel) {
            cols_here += hstats[bsel];
            if (hstats[bsel]) w = std::max(w, wrt_of(bsel));
          }
          if (!cols_here) continue;
          bool launched = false;
#define PAIR3_CASE(M, N) \
  if (maxch == M && nw == N) { launch_pair3<M, N>(blo, bhi, w, A, B, lo.p, span.p, bin.p, tmpoff.p, tmp_inner.p, tv, count.p, alpha, threshold, dr); launched = true; }
          PAIR3_CASE(2, 1) PAIR3_CASE(3, 1) PAIR3_CASE(4, 1) PAIR3_CASE(5, 1) PAIR3_CASE(6, 1) PAIR3_CASE(5, 2) PAIR3_CASE(5, 4)
#undef PAIR3_CASE
          if (!launched) NTP_FATAL("unknown spgemm_variant");
          for (int bsel = blo; bsel <= bhi; ++bsel) hstats[bsel] = 0;
        }
      }
    }
    if (hstats[1]) launch_window<T, 512, 4>(1, A, B, lo.p, span.p, bin.p, tmpoff.p, tmp_inner.p, tv, count.p, alpha, threshold, dr);
    if (hstats[2]) launch_window<T, 1024, 4>(2, A, B, lo.p, span.p, bin.p, tmpoff.p, tmp_inner.p, tv, count.p, alpha, threshold, dr);
    if (hstats[3]) launch_window<T, 2048, 2>(3, A, B, lo.p, span.p, bin.p, tmpoff.p, tmp_inner.p, tv, count.p, alpha, threshold, dr);
    if (hstats[4]) launch_window<T, 4096, 1>(4, A, B, lo.p, span.p, bin.p, tmpoff.p, tmp_inner.p, tv, count.p, alpha, threshold, dr);
    if (hstats[5]) {
      // small tables first (occupancy); the columns that outgrow them go through the large-table pass
      hipLaunchKernelGGL((k_spgemm_hash<T, 1024>), dim3(xcd_grid(n)), dim3(WAVE), 0, stream(), view(A), view(B), span.p,
                         bin.p, tmpoff.p, tmp_inner.p, tv, count.p, stats.p, alpha, threshold, dr, n);
      unsigned long long big = 0;
      {
        ScalarFetch f;
        f.add(stats.p + 19, 1, &big);
        f.run();
      }
      if (big)
        hipLaunchKernelGGL((k_spgemm_hash<T, 4096>), dim3(xcd_grid(n)), dim3(WAVE), 0, stream(), view(A), view(B), span.p,
                           bin.p, tmpoff.p, tmp_inner.p, tv, count.p, stats.p, alpha, threshold, dr, n);
      hash_big = big;
    }
  });
  // columns that overflowed the LDS hash (or were forced) go through the HBM accumulator
  unsigned long long overflow = hstats[6];
  if (hstats[5] && hash_big) {
    unsigned long long ov = 0;
    ScalarFetch f;
    f.add(stats.p + 8, 1, &ov);
    f.run();
    overflow += ov;
  }
  st.overflow_cols = (int64_t)overflow;
  if (overflow) {
    DevBuf<int64_t> ub2(n + 1);
    tmpoff2.alloc(n + 1);
    hipLaunchKernelGGL(k_hbm_ub, dim3(cdiv(n, 256)), dim3(256), 0, stream(), bin.p, span.p, ip.p, ub2.p, n);
    const int64_t total2 = exclusive_scan_i64(ub2.p, tmpoff2.p, n);
    tmp2_inner.alloc((size_t)total2);
    tmp2_val.alloc((size_t)total2 * A.wval());
    const int nwaves = (int)std::min<int64_t>(overflow, 1024);
    DevBuf<double> ws((size_t)nwaves * (size_t)m * A.wval());
    ws.zero();
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_spgemm_hbm<T>), dim3(nwaves), dim3(WAVE), 0, stream(), view(A), view(B), lo.p, span.p,
                         bin.p, tmpoff2.p, tmp2_inner.p, reinterpret_cast<T*>(tmp2_val.p), count.p, ws.p, alpha,
                         threshold, dr);
    });
    sync_stream();  // ws is released below
  }
  t_num.stop();
  if (use_slab && !fuse_now) scan_async<int64_t>(blk_prod, blk_prod_scan.p, (int64_t)snb);  // total = products of this multiply

  if (fuse_now) {
    // one read-back: entries of the result, (dot, trace), the kernel's objections, the product's own entry count
    DevBuf<double> lvl((size_t)5 * FT_BLOCKS), tot(5);
    hipLaunchKernelGGL(k_fused_totals, dim3(FT_BLOCKS), dim3(256), 0, stream(), count.p, n,
                       reinterpret_cast<const long long*>(fz_pnnz), reinterpret_cast<const long long*>(blk_prod),
                       fz_part, snb, (const double*)nullptr, lvl.p, 0);
    hipLaunchKernelGGL(k_fused_totals, dim3(1), dim3(256), 0, stream(), (const int32_t*)nullptr, 0, (const long long*)nullptr,
                       (const long long*)nullptr, (const double*)nullptr, 0, lvl.p, tot.p, 1);
    int64_t nnz = 0, flagv[2] = {0, 0}, pnz = 0;
    unsigned long long slab_products = 0;
    double hd[2] = {0, 0};
    {
      unsigned long long raw[5] = {0, 0, 0, 0, 0};
      ScalarFetch f;
      f.add(tot.p, 5, raw);
      f.add(fz_flag, 1, flagv);
      f.run();
      nnz = (int64_t)raw[0];
      pnz = (int64_t)raw[1];
      slab_products = raw[2];
      std::memcpy(hd, &raw[3], 2 * sizeof(double));
    }
    t_all.stop();
    if (timing) {
      if (pending_timings().size() >= 4096) flush_spgemm_timers();
      pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
    }
    if ((int32_t)flagv[0] != 0) {   // (see SlabFuseArgs) the step is repeated without the fusion
      fuse->done = false;
      fuse->refused += 1;
      fusion_counts()[2] += 1;
      spgemm(A, B, C, alpha, threshold, dense_rule, loose, arange, nullptr);
      return;
    }
    DevMat R;
    R.rows = m;
    R.cols = n;
    R.cplx = false;
    R.nnz = nnz;
    R.zero_free = 1;   // kept entries passed |v| > threshold (> = 0) or are scaled copies of entries that did
    R.slab.reset(new SlabForm());
    R.slab->first = std::move(fz_first);
    R.slab->last = std::move(fz_last);
    R.slab->count = std::move(count);
    R.slab->row_pad = use_tile ? tile_expand_align() : 1;
    R.slab->off = use_tile ? std::move(tile_ooff) : std::move(tmpoff);
    R.slab->tile_off = use_tile ? std::move(tile_otoff) : std::move(blk_toff);
    R.slab->val = std::move(tmp_val);
    R.slab->tiles = std::move(fz_tiles);
    R.slab->slots = tmp_total;
    if (sc_oplast.p) R.slab->plast = std::move(sc_oplast);
    fuse->result = std::move(R);
    fuse->done = true;
    fusion_counts()[fuse->mode == 1 ? 0 : 1] += 1;
    fuse->dot = hd[0];
    fuse->trace = hd[1];
    fuse->product_nnz = pnz;
    st.products = (int64_t)slab_products;
    st.nnz_c = pnz;
    st.fused = fuse->mode;
    last_spgemm_stats() = st;
    SpgemmAccum& acc = spgemm_accum();
    acc.calls += 1;
    acc.products += st.products;
    acc.nnz_c += pnz;
    acc.alg_bytes += 12.0 * (double)(A.nnz + B.nnz + pnz) + 4.0 * ((double)A.cols + B.cols + n + 3);
    return;
  }

  if (loose && use_slab && !use_tile && !use_ctile) {
    // hand the slots over as they are: the consumer (axpby) reads the columns in place and reports the exact nnz
    loose->valid = true;
    loose->rows = m;
    loose->cols = n;
    loose->slots = tmp_total;
    loose->start = std::move(tmpoff);
    loose->count = std::move(count);
    loose->inner = std::move(tmp_inner);
    loose->val = std::move(tmp_val);
    t_all.stop();
    if (timing) {
      if (pending_timings().size() >= 4096) flush_spgemm_timers();
      pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
    }
    loose->prod_index = -1;
    if (timing) {  // the product count of the slab path comes from the expansion pass; read together with the nnz later
      loose->prod_scan = std::move(blk_prod_scan);
      loose->prod_index = snb;
    }
    st.nnz_c = -1;  // not known yet
    last_spgemm_stats() = st;
    SpgemmAccum& acc = spgemm_accum();
    acc.calls += 1;
    acc.products += st.products;
    acc.alg_bytes += (4.0 + (double)esz) * (double)(A.nnz + B.nnz) + 4.0 * ((double)A.cols + B.cols + n + 3);  // + 12 * nnzC: axpby
    return;
  }

  // exact column pointers, then move every column to its final place
  C.rows = m;
  C.cols = n;
  C.cplx = A.cplx;
  C.outer.alloc((size_t)n + 1);
  scan_async<int32_t>(count.p, C.outer.p, (int64_t)n);
  int64_t nnz = 0;
  unsigned long long slab_products = 0;
  {
    ScalarFetch f;
    f.add(C.outer.p + n, 1, &nnz);
    if (use_slab) f.add(blk_prod_scan.p + snb, 1, &slab_products);
    f.run();
  }
  if (use_slab) st.products = (int64_t)slab_products;
  C.nnz = nnz;
  C.inner.alloc((size_t)nnz + kIndexSlack);
  C.val.alloc(((size_t)nnz + kIndexSlack) * C.wval());
  const int nblocks = cdiv(n, 4);
  if (use_tile)   // (the tile kernel left dense runs: one wave per column collects the non-zeros in row order)
    hipLaunchKernelGGL(k_pack_slab, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fz_first.p, fz_last.p,
                       tile_ooff.p, tmp_val.p, C.outer.p, C.inner.p, C.val.p);
  else if (use_ctile)
    hipLaunchKernelGGL(k_pack_slab_c, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fz_first.p, fz_last.p,
                       tile_ooff.p, reinterpret_cast<const double2*>(tmp_val.p), C.outer.p, C.inner.p, reinterpret_cast<double2*>(C.val.p));
  else dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_compact<T>), dim3(xcd_grid(nblocks)), dim3(256), 0, stream(), n, tmpoff.p,
                       overflow ? tmpoff2.p : nullptr, overflow ? bin.p : nullptr, C.outer.p, tmp_inner.p,
                       reinterpret_cast<const T*>(tmp_val.p), tmp2_inner.p, reinterpret_cast<const T*>(tmp2_val.p),
                       C.inner.p, reinterpret_cast<T*>(C.val.p), nblocks);
  });
  t_all.stop();
  st.nnz_c = nnz;
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.products += st.products;
  acc.nnz_c += nnz;
  const double per = A.cplx ? 20.0 : 12.0;
  acc.alg_bytes += per * (double)(A.nnz + B.nnz + nnz) + 4.0 * ((double)A.cols + B.cols + n + 3);
}


// -------------------------------------------------------------------------------------
// A purification step on an iterate that is already in slab form (SlabForm): plan from the column extents, run
// records, the fused kernel, the totals -- no pass over the entries outside the kernel itself.
namespace {
}  // namespace

namespace {
// run records from run addresses (panel steps: the runs of the halo columns sit in the receive buffer)
__global__ void k_pack_reduce4(const double* __restrict__ tot, const int* __restrict__ flag, double* __restrict__ out) {
  out[0] = tot[3];
  out[1] = 0.0;
  out[2] = tot[4];
  out[3] = flag[0] == 0 ? 1.0 : 0.0;
}
__global__ void k_slab_runs_addr(const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                 const unsigned long long* __restrict__ addr, unsigned long long fallback,
                                 SlabRun* __restrict__ runs, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n + 4) return;
  const bool any = k < n && last[k] >= first[k];
  SlabRun r;
  const unsigned long long a = any ? addr[k] : fallback;
  r.addr_lo = (uint32_t)a;
  r.addr_hi = (uint32_t)(a >> 32) & 0xffffu;
  r.nbytes = any ? (uint32_t)(last[k] - first[k] + 1) * 8u : 0u;
  r.flags = kBufferFlags;
  r.first = any ? first[k] : (1 << 30);
  r.first8 = any ? first[k] * 8 : 0;
  r.span62 = any ? (last[k] - first[k] + 1) + 62 : 0;
  r.pad = 0;
  runs[k] = r;
}

// label-ordered steps: the columns kmin .. kmin + kn - 1 of a block sorted by label give the order of its k steps; the
// block's run records, its step list and its multiplier tile rows are written in that order (records and steps at
// tile_off[b] / J + 4 b, four empty records behind the last for the loop's look-ahead)
__global__ __launch_bounds__(256) void k_slab_order_steps(const int32_t* __restrict__ lab, const int32_t* __restrict__ blk_kmin,
                                                          const int32_t* __restrict__ blk_kn, const int64_t* __restrict__ tile_off,
                                                          const double* __restrict__ tiles_in, const SlabRun* __restrict__ runs,
                                                          int ncols, double* __restrict__ tiles_out, SlabRun* __restrict__ blkruns,
                                                          int32_t* __restrict__ steps, int nblocks) {
  // tiles_out == nullptr: the tile stays as it is and every record names its row (pad = byte offset of row k - kmin):
  // the row-offset variant of the loop (SLAB_LOOP_ASM_ROWOFF) reads the multipliers there
  extern __shared__ unsigned long long sk[];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int kn = blk_kn[b], kmin = blk_kmin[b];
  if (kn == 0) return;
  int m = 1;
  while (m < kn) m <<= 1;
  for (int i = threadIdx.x; i < m; i += blockDim.x)
    sk[i] = i < kn ? (((unsigned long long)(unsigned)lab[kmin + i] << 32) | (unsigned)i) : ~0ull;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const unsigned long long x = sk[i], y = sk[l];
          if ((x > y) == up) { sk[i] = y; sk[l] = x; }
        }
      }
      __syncthreads();
    }
  }
  const int64_t toff = tile_off[b];
  const int64_t rbase = toff / SLAB_J + 4 * (int64_t)b;
  for (int t = threadIdx.x; t < kn + 4; t += blockDim.x) {
    if (t < kn) {
      const int i = (int)(unsigned)(sk[t] & 0xffffffffull);
      const int k = kmin + i;
      if (steps) steps[rbase + t] = k;
      SlabRun r = runs[k];
      r.pad = i * (int)(SLAB_J * sizeof(double));
      blkruns[rbase + t] = r;
    } else {
      if (steps) steps[rbase + t] = ncols;
      blkruns[rbase + t] = runs[ncols];   // (an empty record: the tail of the record array)
    }
  }
  if (!tiles_out) return;
  const double2* __restrict__ src = reinterpret_cast<const double2*>(tiles_in + toff);
  double2* __restrict__ dst = reinterpret_cast<double2*>(tiles_out + toff);
  for (int i = threadIdx.x; i < kn * (SLAB_J / 2); i += blockDim.x) {
    const int t = i / (SLAB_J / 2), q = i % (SLAB_J / 2);
    dst[(size_t)t * (SLAB_J / 2) + q] = src[(size_t)(unsigned)(sk[t] & 0xffffffffull) * (SLAB_J / 2) + q];
  }
}
// largest label among the entries of every column
}  // namespace

namespace {
// the plan of a step X * X from the column extents of X (device arrays of n columns; a panel step: the extents of the
// columns ka .. of the distributed iterate on the A side); sizes are left on the device in stats[16..18] / blk_toff[snb]
void launch_slab_plan(SlabPlan& P, int n, const int32_t* first, const int32_t* last, const int32_t* afirst, const int32_t* alast,
                      int align, unsigned long long* stats, DevBuf<int64_t>* tile_offsets = nullptr) {
  const int snb = cdiv(n, SLAB_J);
  P.blk_lo.alloc(snb); P.blk_w.alloc(snb); P.blk_kmin.alloc(snb); P.blk_kn.alloc(snb);
  P.blk_toff.alloc((size_t)snb + 1);
  P.align = align;
  DevBuf<int64_t> bsz(snb), tsz(snb);
  if (tile_offsets) tile_offsets->alloc((size_t)snb + 1);
  hipLaunchKernelGGL((k_slab_plan<SLAB_J>), dim3(cdiv((int64_t)snb * WAVE, 256)), dim3(256), 0, stream(), n, first, last, afirst,
                     alast, P.blk_lo.p, P.blk_w.p, P.blk_kmin.p, P.blk_kn.p, bsz.p, tsz.p, snb, align);
  if (options().plan_fused != 0) {
    // maxima and both prefix sums in ONE launch (k_slab_offsets) instead of four to seven
    const int part = cdiv(cdiv(snb, kOffParts), 256) * 256;
    hipLaunchKernelGGL(k_slab_offsets, dim3(cdiv(snb, part)), dim3(256), 0, stream(), P.blk_w.p, P.blk_kn.p, tsz.p, bsz.p, snb, part,
                       P.blk_toff.p, tile_offsets ? tile_offsets->p : (int64_t*)nullptr, stats);
    return;
  }
  hipLaunchKernelGGL(k_slab_reduce, dim3(64), dim3(256), 0, stream(), P.blk_w.p, P.blk_kn.p, snb, (const int32_t*)nullptr, 0, stats);
  scan_async<int64_t>(tsz.p, P.blk_toff.p, (int64_t)snb);
  if (tile_offsets) scan_async<int64_t>(bsz.p, tile_offsets->p, (int64_t)snb);   // (where the multiplier tile of every block starts)
}
}  // namespace

bool slab_step(DevMat& X, SlabFusion& fu, double threshold, bool dense_rule, const SlabHalo* halo) {
  fu.done = false;
  if (!X.expanded() || X.cplx || (!halo && X.rows != X.cols) || !fu.D || fu.D->cplx || fu.D->loose() || fu.D->expanded() ||
      fu.D->rows != X.rows || fu.D->cols != X.cols || (fu.mode != 1 && fu.mode != 2))
    return false;
  const bool tile = options().spgemm_fma == 1;   // FMA arithmetic: the MFMA tile kernel (spgemm_tile.hip)
  if ((options().spgemm_variant >= 0 && !(tile && options().spgemm_variant / 100 == 6)) || (options().spgemm_fma && !tile) || options().spgemm_force_bin > 0 || !options().fused_update)
    return false;
  const SlabForm& in = *X.slab;
  // rows per lane of the tile kernel: what the option asks for, if the runs of X (and of D) are padded for it; the runs
  // of a halo sit packed in the receive buffer: one row per lane there
  int trows = 1;
  if (tile) {
    trows = tile_rows();
    // (row groups start at multiples of R: the pads must cover them -- of the iterate's own runs and of the runs a halo
    // brought: the senders pack them into aligned zero-padded slots, SlabHalo::row_pad)
    while (trows > 1 && (in.row_pad % (16 * trows) != 0 || (halo && halo->row_pad % (16 * trows) != 0))) trows >>= 1;
  }
  const int n = X.cols, snb = cdiv(n, SLAB_J);
  const DotOperand& dop = dot_operand(*fu.D);
  if (!tile && dop.max_tile > SLAB_DTILE) return false;
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  SpgemmStats st;
  st.nnz_a = st.nnz_b = X.nnz;
  // ---- plan (column extents only): left behind by the step that produced X (option plan_ahead), or made here and
  // read back before the launch
  DevBuf<int32_t> count((size_t)n), ofirst((size_t)n), olast((size_t)n);
  DevBuf<int64_t> tmpoff((size_t)n + 1);
  // [flag 2 | product entries per block snb + 1 | (unused) snb | (dot, trace) per block 2 snb | plan statistics 24 |
  //  statistics of the next step's plan 24]
  DevBuf<int64_t> zwords((size_t)4 * snb + 4 + 48);
  zwords.zero();
  if (!tile) count.zero();   // (the tile kernel writes the count of every column)
  int64_t* fz_flag = zwords.p;
  int64_t* fz_pnnz = zwords.p + 2;
  double* fz_part = reinterpret_cast<double*>(zwords.p + 3 + 2 * (size_t)snb);
  unsigned long long* stats = reinterpret_cast<unsigned long long*>(zwords.p + 4 * (size_t)snb + 4);
  unsigned long long* next_stats = stats + 24;
  // (A side: the iterate's own columns, or -- a panel step -- the columns ka .. kb of the distributed iterate,
  // addressed with global column numbers through biased pointers)
  const int ka = halo ? halo->ka : 0, nka = halo ? halo->kb - halo->ka : n;
  const int plan_align = tile ? 16 * tile_rows() : 0;
  SlabPlan own_plan;
  SlabPlan* plan = &own_plan;
  if (!halo && in.next_plan && in.next_plan->align == plan_align && (int64_t)in.next_plan->blk_lo.n == snb) {
    plan = in.next_plan.get();
  } else if (halo && halo->plan && halo->plan->align == plan_align && (int64_t)halo->plan->blk_lo.n == snb) {
    plan = halo->plan;
  } else {
    const int32_t* afirst = halo ? halo->first - ka : in.first.p;
    const int32_t* alast = halo ? halo->last - ka : in.last.p;
    launch_slab_plan(own_plan, n, in.first.p, in.last.p, afirst, alast, plan_align, stats);
    unsigned long long hs[3] = {0, 0, 0};
    ScalarFetch f;
    f.add(own_plan.blk_toff.p + snb, 1, &own_plan.total);
    f.add(stats + 16, 3, hs);
    f.run();
    own_plan.max_w = (int)hs[0];
    own_plan.max_kn = (int)hs[1];
  }
  DevBuf<int32_t>&blk_lo = plan->blk_lo, &blk_w = plan->blk_w, &blk_kmin = plan->blk_kmin, &blk_kn = plan->blk_kn;
  DevBuf<int64_t>& blk_toff = plan->blk_toff;
  const int64_t tmp_total = plan->total;
  const unsigned long long hst[2] = {(unsigned long long)plan->max_w, (unsigned long long)plan->max_kn};
  const int64_t max_w = (int64_t)hst[0];
  auto give_up = [&]() {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  };
  if (max_w <= 0 || (!tile && max_w > 8 * SLAB_SL * WAVE)) return give_up();
  if (tile && !spgemm_tile_fits((int)hst[1], (int)max_w)) return give_up();
  if (in.labelled() && (int64_t)hst[1] > 2048) return give_up();   // (the per-block sort of the steps)
  if (!tile)   // (the tile kernel writes the result's own offsets)
    hipLaunchKernelGGL((k_slab_tmpoff<SLAB_J>), dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), n, blk_w.p, blk_toff.p, tmpoff.p);
  DevBuf<char> runs(((size_t)nka + 4) * sizeof(SlabRun));
  if (halo)
    hipLaunchKernelGGL(k_slab_runs_addr, dim3(cdiv(nka + 4, 256)), dim3(256), 0, stream(), halo->first, halo->last, halo->addr,
                       (unsigned long long)reinterpret_cast<uintptr_t>(in.val.p), reinterpret_cast<SlabRun*>(runs.p), nka);
  else
    hipLaunchKernelGGL(k_slab_runs, dim3(cdiv(n + 4, 256)), dim3(256), 0, stream(), in.first.p, in.last.p, in.off.p,
                       reinterpret_cast<const char*>(in.val.p), 8, reinterpret_cast<SlabRun*>(runs.p), n);
  // (a fifth more than needed: the iterates fill in over the first steps of a solve, and a block that is too small
  // next time costs a hipMalloc inside the loop)
  const size_t oslots = (size_t)tmp_total + (size_t)tmp_total / 5 + kIndexSlack;
  // (FMA arithmetic on one rank: the result carries runs only -- the next step's multiplier tiles come from the runs, which
  // that step reads as its left operand anyway: a third less written per launch and no tile read at all; the unfused
  // loop and the panel steps take their multiplier rows from tiles in memory and keep them)
  const bool runs_only = tile && !halo && options().tile_runs_only != 0;
  DevBuf<double> oval(oslots), otiles(runs_only ? 0 : oslots);
  SlabFuseArgs fz;
  fz.am = fu.am; fz.bm = fu.bm; fz.thr_m = fu.threshold;
  fz.xexp = in.val.p; fz.xoff = in.off.p; fz.xmin = in.first.p; fz.xmax = in.last.p;
  fz.dexp = dop.dexp.p; fz.doff = dop.doff.p; fz.dmin = dop.dmin.p; fz.dmax = dop.dmax.p;
  fz.ofirst = ofirst.p; fz.olast = olast.p; fz.tiles = runs_only ? nullptr : otiles.p;
  fz.part = fz_part; fz.pnnz = reinterpret_cast<long long*>(fz_pnnz); fz.flag = reinterpret_cast<int*>(fz_flag);
  fz.col_offset = fu.col_offset;
  int64_t* blk_prod = zwords.p + 3 + snb;
  // label-ordered steps (SlabForm::lab): the block's records and tile rows sorted by label
  const bool labelled = in.labelled() && !halo && !tile;   // (the tile kernel walks the k steps in position order: no sorted records)
  if (in.labelled() && halo) return give_up();
  DevBuf<double> tiles_ord;
  DevBuf<char> blkruns;
  DevBuf<int32_t> steps, oplast;
  // (the loop variant that takes a step's multiplier row from its record needs no copy of the tiles in step order)
  const bool rowoff = labelled && options().label_rowoff != 0;
  if (labelled) {
    const size_t nrec = in.tiles.n / SLAB_J + 4 * (size_t)snb + 8;
    blkruns.alloc(nrec * sizeof(SlabRun));
    oplast.alloc((size_t)n);
    if (!rowoff) {
      tiles_ord.alloc(in.tiles.n);
      steps.alloc(nrec);
    }
    int m = 1;
    while (m < (int)hst[1]) m <<= 1;
    hipLaunchKernelGGL(k_slab_order_steps, dim3(xcd_grid(snb)), dim3(256), (size_t)m * 8, stream(), in.lab.p, blk_kmin.p, blk_kn.p,
                       in.tile_off.p, in.tiles.p, reinterpret_cast<const SlabRun*>(runs.p), n, tiles_ord.p,
                       reinterpret_cast<SlabRun*>(blkruns.p), steps.p, snb);
    fz.lab = in.lab.p;
    fz.blkruns = reinterpret_cast<const SlabRun*>(blkruns.p);
    fz.steps = steps.p;
    fz.xplast = in.plast.p;
    fz.oplast = oplast.p;
  }
  // (labels only steer the epilogue's "beyond the last entry" tests.  A panel step of a solve that runs in a recovered band
  // order across ranks takes them from the scope -- one array for all rows -- and the largest label of every column of the
  // iterate from the step before, SlabForm::plast)
  const bool scope_lab = tile && halo && !in.labelled() && scope_labels() != nullptr && in.plast.p != nullptr && fu.mode != 0;
  const bool tile_labelled = tile && ((in.labelled() && !halo) || scope_lab);
  if (tile_labelled) {
    oplast.alloc((size_t)n);
    fz.lab = scope_lab ? scope_labels() : in.lab.p;
    fz.xplast = in.plast.p;
    fz.oplast = oplast.p;
  }
  // the block's pass over its multiplier tile before the loop: always (it warms L2); with the entry counts of the
  // columns at hand it also counts the products (the tile rows are global column numbers)
  fz.prod = reinterpret_cast<long long*>(blk_prod);
  if (!halo || halo->count) fz.in_count = halo ? halo->count - ka : in.count.p;
  DevBuf<char> fz_args;
  if (!tile) {   // (the slab loop reads them from memory after its loop; the tile kernel takes them by value)
    fz_args.alloc(sizeof(SlabFuseArgs));
    fz_args.upload(reinterpret_cast<const char*>(&fz), sizeof(SlabFuseArgs));
  }
  const int dr = dense_rule ? 1 : 0;
  t_num.start();
  auto launch = [&](auto nw_tag, auto mode_tag, auto epi_tag) {
    constexpr int FNW = decltype(nw_tag)::value;
    hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, FNW, decltype(mode_tag)::value, decltype(epi_tag)::value>),
                       dim3(xcd_grid(snb)), dim3(FNW * WAVE), 0, stream(), reinterpret_cast<const SlabRun*>(runs.p) - ka,
                       (labelled && !rowoff) ? tiles_ord.p : in.tiles.p, in.tile_off.p, blk_kmin.p, blk_kn.p, blk_lo.p, blk_w.p, blk_toff.p, (int32_t*)nullptr,
                       oval.p, count.p, 1.0, threshold, dr, n, snb, reinterpret_cast<const SlabFuseArgs*>(fz_args.p));
  };
  auto by_mode = [&](auto nw_tag, auto mode_tag) {
    if (fu.mode == 1) launch(nw_tag, mode_tag, std::integral_constant<int, 1>{});
    else launch(nw_tag, mode_tag, std::integral_constant<int, 2>{});
  };
  DevBuf<int64_t> tile_ooff, tile_otoff;
  bool used_tile2 = false;
  if (tile) {
    tile_ooff.alloc((size_t)n + 1);
    tile_otoff.alloc((size_t)snb + 1);
    TileLaunch tl;
    tl.runs = reinterpret_cast<const SlabRun*>(runs.p) - ka;
    tl.bblk = in.tiles.p; tl.blk_boff = in.tile_off.p;
    if (in.tiles.p == nullptr || in.tile_off.p == nullptr) {   // (the iterate carries runs only: the right operand from its runs)
      tl.bblk = nullptr; tl.blk_boff = nullptr;
      tl.brun_first = in.first.p; tl.brun_last = in.last.p; tl.brun_off = in.off.p; tl.brun_val = in.val.p; tl.bbytes = in.val.n * sizeof(double); tl.brun_pad = in.row_pad;
    }
    tl.blk_kmin = blk_kmin.p; tl.blk_kn = blk_kn.p; tl.blk_lo = blk_lo.p;
    tl.blk_w = blk_w.p; tl.blk_toff = blk_toff.p; tl.out_val = oval.p; tl.count = count.p;
    tl.ofirst = ofirst.p; tl.olast = olast.p; tl.ooff = tile_ooff.p; tl.otoff = tile_otoff.p;
    tl.alpha = 1.0; tl.threshold = threshold; tl.dense_rule = dr; tl.ncols = n; tl.nblocks = snb;
    tl.max_kn = (int)hst[1]; tl.max_w = (int)max_w; tl.epi = fu.mode; tl.fz = &fz; tl.rows = trows; tl.labelled = tile_labelled;
    tl.nrows = X.rows;
    if (!halo) {   // (every run of A -- and X itself -- in the iterate's own value buffer, D in its expansion: 32-bit offsets)
      tl.abase = in.val.p; tl.abytes = in.val.n * sizeof(double);
      tl.dbase = dop.dexp.p; tl.dbytes = dop.dexp.n * sizeof(double);
    }
    // (the fused epilogue's arguments travel by value; the kernel writes the end markers of the offsets.)  The two-block
    // geometry first where it can apply: one rank, runs only, two rows per lane; a pair that does not fit after all leaves a
    // mark that comes back with the totals -- the step is then repeated on k_spgemm_tile
    if (!halo && options().tile2 != 0 && !in.no_tile2) used_tile2 = launch_spgemm_tile2(tl, reinterpret_cast<int*>(fz_flag) + 2);
    if (!used_tile2) launch_spgemm_tile(tl);
  } else if (rowoff) {
    if (max_w > 6 * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 8>{}, std::integral_constant<int, 9>{});
    else if (max_w > SLAB_NW * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 6>{}, std::integral_constant<int, 9>{});
    else by_mode(std::integral_constant<int, SLAB_NW>{}, std::integral_constant<int, 9>{});
  } else if (max_w > 6 * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{});
  else if (max_w > SLAB_NW * SLAB_SL * WAVE) by_mode(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
  else by_mode(std::integral_constant<int, SLAB_NW>{}, std::integral_constant<int, 8>{});
  t_num.stop();
  DevBuf<double> lvl((size_t)5 * FT_BLOCKS), tot(5);
  hipLaunchKernelGGL(k_fused_totals, dim3(FT_BLOCKS), dim3(256), 0, stream(), count.p, n,
                     reinterpret_cast<const long long*>(fz_pnnz), reinterpret_cast<const long long*>(blk_prod), fz_part, snb,
                     (const double*)nullptr, lvl.p, 0);
  hipLaunchKernelGGL(k_fused_totals, dim3(1), dim3(256), 0, stream(), (const int32_t*)nullptr, 0, (const long long*)nullptr,
                     (const long long*)nullptr, (const double*)nullptr, 0, lvl.p, tot.p, 1);
  unsigned long long raw[5] = {0, 0, 0, 0, 0};
  int64_t flagv[2] = {0, 0};
  DevBuf<double> red4;
  if (halo && halo->reduce) {   // (dot, 0, trace, this rank succeeded) summed over the ranks, read back with the totals
    red4.alloc(4);
    hipLaunchKernelGGL(k_pack_reduce4, dim3(1), dim3(1), 0, stream(), tot.p, reinterpret_cast<const int*>(fz_flag), red4.p);
    halo->reduce->allreduce(red4.p);
  }
  // the plan of the next step on the result, right behind this one's kernel: its sizes come back with the totals
  std::unique_ptr<SlabPlan> next;
  unsigned long long next_hs[3] = {0, 0, 0};
  if (tile && !halo && options().plan_ahead != 0) {
    next.reset(new SlabPlan());
    launch_slab_plan(*next, n, ofirst.p, olast.p, ofirst.p, olast.p, plan_align, next_stats);
  }
  // the result as it stands on the device (its entry count comes back below)
  DevMat R;
  R.rows = X.rows;
  R.cols = n;
  R.cplx = false;
  R.zero_free = 1;
  R.slab.reset(new SlabForm());
  R.slab->first = std::move(ofirst);
  R.slab->last = std::move(olast);
  R.slab->count = std::move(count);
  R.slab->row_pad = tile ? 16 * trows : 1;
  R.slab->no_tile2 = in.no_tile2;
  R.slab->off = tile ? std::move(tile_ooff) : std::move(tmpoff);
  if (tile) R.slab->tile_off = std::move(tile_otoff);   // (the slab loop's result takes the plan's tile offsets below, once the step has succeeded)
  R.slab->val = std::move(oval);
  if (runs_only) { R.slab->tile_off.release(); } else R.slab->tiles = std::move(otiles);
  R.slab->slots = tmp_total;
  {
    ScalarFetch f;
    f.add(tot.p, 5, raw);
    f.add(fz_flag, 2, flagv);
    if (red4.p) f.add(red4.p, 4, halo->reduce->reduced);
    if (next) {
      f.add(next->blk_toff.p + snb, 1, &next->total);
      f.add(next_stats + 16, 3, next_hs);
    }
    // (a panel step: what the caller wants to know about the NEXT step -- its exchange layout and plan, from this result's
    // extents -- rides on this read-back)
    if (halo && halo->before_fetch) halo->before_fetch(R, reinterpret_cast<const long long*>(tot.p), f);
    f.run();
    if (red4.p) halo->reduce->done = true;
  }
  t_all.stop();
  if (next) {
    next->max_w = (int)next_hs[0];
    next->max_kn = (int)next_hs[1];
  }
  if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
    std::fprintf(stderr, "[slab_step] mode %d slots %lld w %llu kn %llu in-slots %lld nnz-in %lld plan %s\n", fu.mode, (long long)tmp_total,
                 hst[0], hst[1], (long long)in.slots, (long long)X.nnz, plan == &own_plan ? "made here" : "left by the step before");
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  if (used_tile2 && (int32_t)flagv[1] != 0) {   // a pair of blocks did not fit the two-block geometry: X is untouched, the same step on k_spgemm_tile
    X.slab->no_tile2 = true;
    tile2_counts()[1] += 1;
    return slab_step(X, fu, threshold, dense_rule, halo);
  }
  if (used_tile2) tile2_counts()[0] += 1;
  if ((int32_t)flagv[0] != 0) {   // (SlabFuseArgs) X is untouched: the caller repeats the step on the unfused path
    fu.refused += 1;
    fusion_counts()[2] += 1;
    return false;
  }
  const int64_t nnz = (int64_t)raw[0], pnz = (int64_t)raw[1];
  double hd[2];
  std::memcpy(hd, &raw[3], 2 * sizeof(double));
  const int64_t nnz_in = X.nnz;
  R.nnz = nnz;
  R.slab->next_plan = std::move(next);
  if (!tile) R.slab->tile_off = std::move(plan->blk_toff);   // (the slab loop's result keeps the plan's tile offsets)
  if (labelled || tile_labelled) {
    R.slab->lab = std::move(X.slab->lab);
    R.slab->plast = std::move(oplast);
  }
  if (halo) fu.result = std::move(R);   // (the ranks agree first: the caller installs it)
  else X = std::move(R);
  fu.done = true;
  fu.dot = hd[0];
  fu.trace = hd[1];
  fu.product_nnz = pnz;
  fusion_counts()[fu.mode == 1 ? 0 : 1] += 1;
  st.slab = 1;
  st.fused = fu.mode;
  st.nnz_c = pnz;
  st.tmp_entries = tmp_total;
  const bool counted = !halo || halo->count;
  st.products = counted ? (int64_t)raw[2] : -1;   // (panel steps: only when the entry counts travelled with the extents)
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  if (counted) acc.products += (int64_t)raw[2];
  acc.nnz_c += pnz;
  acc.alg_bytes += 12.0 * (double)(2 * nnz_in + pnz) + 4.0 * (3.0 * n + 3);
  return true;
}

// Compressed columns -> slab form, with labels: Xs is the matrix in a bandwidth-reducing order (relabel.hip), lab[index]
// = the caller's index.  false: the columns are not run-like enough for the register-slab kernel (nothing changed).
bool slab_from_csc(DevMat& Xs, DevBuf<int32_t>& lab) {
  if (Xs.cplx || Xs.loose() || Xs.expanded() || Xs.rows != Xs.cols || Xs.nnz == 0) return false;
  const int n = Xs.cols, snb = cdiv(n, SLAB_J);
  std::unique_ptr<SlabForm> f(new SlabForm());
  f->first.alloc((size_t)n); f->last.alloc((size_t)n); f->count.alloc((size_t)n); f->off.alloc((size_t)n + 1);
  f->tile_off.alloc((size_t)snb + 1); f->plast.alloc((size_t)n);
  DevBuf<int32_t> span((size_t)n), blk_lo(snb), blk_w(snb), blk_kmin(snb), blk_kn(snb);
  DevBuf<int64_t> bsz(snb), tsz(snb), dummy(snb);
  DevBuf<unsigned long long> stats(24);
  stats.zero();
  hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(Xs), f->first.p, f->last.p, f->count.p);
  const int al = options().spgemm_fma == 1 ? tile_expand_align() : 1;   // (aligned zero-padded slots for the MFMA tile kernel)
  if (al > 1) hipLaunchKernelGGL(k_span_aligned, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f->first.p, f->last.p, span.p, n, al);
  else hipLaunchKernelGGL(k_span_of, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f->first.p, f->last.p, span.p, n);
  scan_async<int32_t>(span.p, f->off.p, (int64_t)n);
  hipLaunchKernelGGL((k_slab_plan<SLAB_J>), dim3(cdiv((int64_t)snb * WAVE, 256)), dim3(256), 0, stream(), n, f->first.p, f->last.p,
                     f->first.p, f->last.p, blk_lo.p, blk_w.p, blk_kmin.p, blk_kn.p, bsz.p, tsz.p, snb);
  hipLaunchKernelGGL(k_slab_reduce, dim3(64), dim3(256), 0, stream(), blk_w.p, blk_kn.p, snb, (const int32_t*)nullptr, 0, stats.p);
  scan_async<int64_t>(bsz.p, f->tile_off.p, (int64_t)snb);
  int64_t tot_val = 0, tot_tiles = 0;
  unsigned long long hst[2] = {0, 0};
  {
    ScalarFetch ft;
    ft.add(f->off.p + n, 1, &tot_val);
    ft.add(f->tile_off.p + snb, 1, &tot_tiles);
    ft.add(stats.p + 16, 2, hst);
    ft.run();
  }
  const int64_t max_w = (int64_t)hst[0], max_kn = (int64_t)hst[1];
  const int pitch = ((int)max_kn + 1) | 1;
  if (max_w <= 0 || max_w > 8 * SLAB_SL * WAVE || (size_t)pitch * SLAB_J * 8 > 128 * 1024) return false;
  if ((double)tot_val > 2.0 * (double)Xs.nnz + (al > 1 ? 2.0 * al * (double)n : 0.0)) return false;   // (mostly holes: no band was recovered)
  f->val.alloc((size_t)tot_val + kIndexSlack);
  if (al > 1)
    hipLaunchKernelGGL(k_aligned_offsets<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), f->first.p, f->last.p,
                       f->off.p, f->val.p, n, al);
  f->row_pad = al;
  f->tiles.alloc((size_t)tot_tiles + 16 * SLAB_J + kIndexSlack);
  hipLaunchKernelGGL(k_slab_expand_a<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(Xs), f->first.p,
                     f->off.p, f->val.p);
  if ((size_t)pitch * SLAB_J * 8 > 64 * 1024) {
    static bool raised = false;
    if (!raised) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_slab_expand_b<double, SLAB_J>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
      raised = true;
    }
  }
  hipLaunchKernelGGL((k_slab_expand_b<double, SLAB_J>), dim3(xcd_grid(snb)), dim3(256), (size_t)pitch * SLAB_J * 8, stream(), view(Xs),
                     blk_kmin.p, blk_kn.p, f->tile_off.p, f->tiles.p, snb, pitch, 0, (const int64_t*)nullptr, (double*)nullptr,
                     (const int32_t*)nullptr, dummy.p);
  hipLaunchKernelGGL(k_col_plast, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(Xs), lab.p, f->plast.p);
  f->slots = std::max(tot_val, tot_tiles);
  f->lab = std::move(lab);
  DevMat R;
  R.rows = Xs.rows;
  R.cols = n;
  R.cplx = false;
  R.nnz = Xs.nnz;
  R.zero_free = Xs.zero_free;
  R.slab = std::move(f);
  Xs = std::move(R);
  return true;
}

// -------------------------------------------------------------------------------------
void increment(const DevMat& A, DevMat& B, double alpha, double threshold) {
  axpby(A, B, alpha, 1.0, threshold, nullptr, nullptr, nullptr, 0);
}

namespace {
// first operand of the merge: a packed matrix or a loose product (columns in upper-bound slots)
struct MergeOperand {
  Csc v;
  int32_t rows, cols;
  bool cplx;
  int64_t slots;  // entries addressable through v.outer (nnz for a packed operand)
  bool loose;
};
void axpby_impl(const MergeOperand& A, DevMat& B, double alpha, double beta, double threshold, const DevMat* D,
                double* dot_out, double* trace_out, int32_t trace_col_offset, int64_t* a_nnz_out,
                const int64_t* d_extra = nullptr, int64_t* extra_out = nullptr, int row_block = 0,
                bool keep_loose = false) {
  // B may be loose.  keep_loose: the result stays in the (tight) slots the merge kernels wrote it to -- no compaction
  // pass; B is a loose matrix on return
  if (A.rows != B.rows || A.cols != B.cols) NTP_FATAL("increment: shape mismatch");
  if (A.cplx != B.cplx) NTP_FATAL("increment: mixed scalar types must be up-cast by the caller");
  if (D && (D->rows != A.rows || D->cols != A.cols || D->cplx != A.cplx)) NTP_FATAL("increment: dot operand mismatch");
  if (dot_out) dot_out[0] = dot_out[1] = 0.0;
  if (a_nnz_out) *a_nnz_out = A.loose ? 0 : A.slots;
  const int n = A.cols;
  if (n == 0) return;
  if (!A.loose && A.slots == 0 && B.nnz == 0) return;
  DevBuf<int32_t> lo(n), span(n), count(n);
  DevBuf<uint8_t> bin(n);
  DevBuf<unsigned long long> stats(16);
  const Csc Bv = lview(B);
  hipLaunchKernelGGL(k_inc_plan, dim3(cdiv(n, 256)), dim3(256), 0, stream(), A.v, Bv, lo.p, span.p, bin.p,
                     stats.p, count.p, row_block > 0 ? 2 : options().increment_force_seq);   // (blocked rule: rank / sequential merge)
  hipLaunchKernelGGL(k_bin_hist, dim3(std::min(cdiv(n, 256), 512)), dim3(256), 0, stream(), bin.p, (const int64_t*)nullptr,
                     (const int32_t*)nullptr, n, stats.p);
  unsigned long long hs[16];
  {
    ScalarFetch f;
    f.add(stats.p, 16, hs);
    f.run();
  }
  // output slot of column j = A.outer[j] + B.outer[j] (room for all of both columns; for a loose A the slots are
  // simply further apart)
  // (with a loose B or a result that stays loose the slots are placed by the actual column lengths: they would
  // otherwise move further apart with every step)
  const int64_t cap = A.slots + B.nnz;
  const size_t wv = A.cplx ? 2 : 1;
  DevBuf<int32_t> tmp_inner((size_t)cap + kIndexSlack);
  DevBuf<double> tmp_val(((size_t)cap + kIndexSlack) * wv);
  DevBuf<int64_t> srcoff((size_t)n + 1);
  if (keep_loose || B.loose()) {
    DevBuf<int32_t> both((size_t)n);
    hipLaunchKernelGGL(k_sum_counts, dim3(cdiv(n, 256)), dim3(256), 0, stream(), A.v, Bv, both.p);
    scan_async<int32_t>(both.p, srcoff.p, (int64_t)n);
  } else {
    hipLaunchKernelGGL(k_sum_outer, dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), A.v.outer, B.outer.p, srcoff.p, n);
  }
  DevBuf<int64_t> a_total;  // exact nnz of a loose first operand
  if (A.loose) {
    a_total.alloc((size_t)n + 1);
    scan_async<int32_t>(A.v.cnt, a_total.p, (int64_t)n);
  }
  const bool fuse_dot = D != nullptr && dot_out != nullptr && hs[4] == 0;
  const int nb1 = cdiv(n, 4), nb2 = n;
  DevBuf<double> part1, part2, part3, tpart1, tpart2, tpart3;
  const bool fuse_trace = fuse_dot && trace_out != nullptr;
  if (fuse_dot) {  // every (non-padding) block of the merge kernels writes its slot: no zero fill needed
    if (hs[1]) part1.alloc((size_t)2 * nb1);
    if (hs[2]) part2.alloc((size_t)2 * nb2);
    if (hs[3]) part3.alloc((size_t)2 * nb1);
    if (fuse_trace) {
      if (hs[1]) tpart1.alloc((size_t)2 * nb1);
      if (hs[2]) tpart2.alloc((size_t)2 * nb2);
      if (hs[3]) tpart3.alloc((size_t)2 * nb1);
    }
  }
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    T* tv = reinterpret_cast<T*>(tmp_val.p);
    const Csc dv = D ? view(*D) : A.v;
    if (hs[1]) {
      if (fuse_dot)
        hipLaunchKernelGGL((k_inc_window<T, 512, 4, true>), dim3(xcd_grid(nb1)), dim3(256), 0, stream(), A.v, Bv, dv,
                           lo.p, span.p, bin.p, 1, tmp_inner.p, tv, count.p, alpha, beta, threshold, part1.p, nb1, tpart1.p, trace_col_offset, srcoff.p);
      else
        hipLaunchKernelGGL((k_inc_window<T, 512, 4, false>), dim3(xcd_grid(nb1)), dim3(256), 0, stream(), A.v, Bv, dv,
                           lo.p, span.p, bin.p, 1, tmp_inner.p, tv, count.p, alpha, beta, threshold, (double*)nullptr, nb1, (double*)nullptr, 0, srcoff.p);
    }
    if (hs[2]) {
      if (fuse_dot)
        hipLaunchKernelGGL((k_inc_window<T, 2048, 1, true>), dim3(xcd_grid(nb2)), dim3(WAVE), 0, stream(), A.v, Bv, dv,
                           lo.p, span.p, bin.p, 2, tmp_inner.p, tv, count.p, alpha, beta, threshold, part2.p, nb2, tpart2.p, trace_col_offset, srcoff.p);
      else
        hipLaunchKernelGGL((k_inc_window<T, 2048, 1, false>), dim3(xcd_grid(nb2)), dim3(WAVE), 0, stream(), A.v, Bv, dv,
                           lo.p, span.p, bin.p, 2, tmp_inner.p, tv, count.p, alpha, beta, threshold, (double*)nullptr, nb2, (double*)nullptr, 0, srcoff.p);
    }
    if (hs[3]) {
      if (fuse_dot)
        hipLaunchKernelGGL((k_inc_merge<T, true>), dim3(xcd_grid(nb1)), dim3(256), 0, stream(), A.v, Bv, dv, bin.p, 3,
                           tmp_inner.p, tv, count.p, alpha, beta, threshold, part3.p, nb1, tpart3.p, trace_col_offset, row_block, srcoff.p);
      else
        hipLaunchKernelGGL((k_inc_merge<T, false>), dim3(xcd_grid(nb1)), dim3(256), 0, stream(), A.v, Bv, dv, bin.p, 3,
                           tmp_inner.p, tv, count.p, alpha, beta, threshold, (double*)nullptr, nb1, (double*)nullptr, 0, row_block, srcoff.p);
    }
    if (hs[4]) {
      hipLaunchKernelGGL((k_inc_seq<T>), dim3(cdiv(n, 64)), dim3(64), 0, stream(), A.v, Bv, bin.p, 4,
                         tmp_inner.p, tv, count.p, alpha, beta, threshold, row_block, srcoff.p);
    }
  });
  DevBuf<double> dres;
  if (fuse_dot) {
    dres.alloc(12);  // pairs that are not produced are not read either (see the sums below)
    if (hs[1]) reduce_sum2_async(part1.p, nb1, dres.p);
    if (hs[2]) reduce_sum2_async(part2.p, nb2, dres.p + 2);
    if (fuse_trace && hs[1]) reduce_sum2_async(tpart1.p, nb1, dres.p + 4);
    if (fuse_trace && hs[2]) reduce_sum2_async(tpart2.p, nb2, dres.p + 6);
    if (hs[3]) reduce_sum2_async(part3.p, nb1, dres.p + 8);
    if (fuse_trace && hs[3]) reduce_sum2_async(tpart3.p, nb1, dres.p + 10);
  }
  DevMat R;
  R.rows = A.rows;
  R.cols = n;
  R.cplx = A.cplx;
  R.outer.alloc((size_t)n + 1);
  scan_async<int32_t>(count.p, R.outer.p, (int64_t)n);
  int64_t nnz = 0;
  double hd[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  {
    ScalarFetch f;
    f.add(R.outer.p + n, 1, &nnz);
    if (fuse_dot) f.add(dres.p, 12, hd);
    int64_t a_nnz = 0;
    if (A.loose) f.add(a_total.p + n, 1, &a_nnz);
    if (d_extra && extra_out) f.add(d_extra, 1, extra_out);
    f.run();
    if (A.loose && a_nnz_out) *a_nnz_out = a_nnz;
  }
  R.nnz = nnz;
  if (keep_loose) {
    R.outer = std::move(srcoff);
    R.cnt = std::move(count);
    R.inner = std::move(tmp_inner);
    R.val = std::move(tmp_val);
    R.slots = cap;
  } else {
    R.inner.alloc((size_t)nnz + kIndexSlack);
    R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
    const int nblocks = cdiv(n, 4);
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_compact<T>), dim3(xcd_grid(nblocks)), dim3(256), 0, stream(), n, srcoff.p, nullptr, nullptr,
                         R.outer.p, tmp_inner.p, reinterpret_cast<const T*>(tmp_val.p), nullptr, nullptr, R.inner.p,
                         reinterpret_cast<T*>(R.val.p), nblocks);
    });
  }
  B = std::move(R);
  if (dot_out && D) {
    if (fuse_dot) {
      dot_out[0] = (hs[1] ? hd[0] : 0.0) + (hs[2] ? hd[2] : 0.0) + (hs[3] ? hd[8] : 0.0);
      dot_out[1] = (hs[1] ? hd[1] : 0.0) + (hs[2] ? hd[3] : 0.0) + (hs[3] ? hd[9] : 0.0);
    } else {
      dot(B, *D, dot_out);
    }
  }
  if (trace_out)
    *trace_out = fuse_trace ? (hs[1] ? hd[4] : 0.0) + (hs[2] ? hd[6] : 0.0) + (hs[3] ? hd[10] : 0.0) : trace(B, trace_col_offset);
}
}  // namespace

void increment_blocked(const DevMat& A, DevMat& B, double alpha, double threshold, int32_t row_block) {
  const MergeOperand a{view(A), A.rows, A.cols, A.cplx, A.nnz, false};
  axpby_impl(a, B, alpha, 1.0, threshold, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, row_block);
}

void axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold, const DevMat* D, double* dot_out,
           double* trace_out, int32_t trace_col_offset) {
  const MergeOperand a{view(A), A.rows, A.cols, A.cplx, A.nnz, false};
  axpby_impl(a, B, alpha, beta, threshold, D, dot_out, trace_out, trace_col_offset, nullptr);
}

namespace {
void dot_trace_impl(const DevMat& A, const DevMat& B, double out[2], double* trace_out, int32_t col_offset,
                    const int64_t* d_extra, int n_extra, int64_t* extra_out);
}
DevMat packed_copy(const DevMat& M) {
  if (M.blocked()) return block_unpack(M);   // (spgemm_block.hip)
  if (M.expanded()) {
    const SlabForm& f = *M.slab;
    if (f.origin) return f.origin->clone();
    DevMat R;
    R.rows = M.rows;
    R.cols = M.cols;
    R.cplx = M.cplx;
    R.zero_free = 1;
    const int n = M.cols;
    R.outer.alloc((size_t)n + 1);
    scan_async<int32_t>(f.count.p, R.outer.p, (int64_t)n);
    int64_t nnz = 0;
    {
      ScalarFetch ft;
      ft.add(R.outer.p + n, 1, &nnz);
      ft.run();
    }
    R.nnz = nnz;
    R.inner.alloc((size_t)nnz + kIndexSlack);
    R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
    if (n && M.cplx) {   // (complex slab sessions: (re, im) interleaved runs)
      hipLaunchKernelGGL(k_pack_slab_c, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p,
                         reinterpret_cast<const double2*>(f.val.p), R.outer.p, R.inner.p, reinterpret_cast<double2*>(R.val.p));
      return R;
    }
    if (n)
      hipLaunchKernelGGL(k_pack_slab, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p,
                         f.off.p, f.val.p, R.outer.p, R.inner.p, R.val.p);
    if (f.labelled()) {   // back to the caller's labels (entry (i, j) -> (lab[i], lab[j]), columns sorted again)
      DevMat Q = remap_general(R, f.lab.p, f.lab.p, M.rows, 0, n, false);
      Q.zero_free = 1;
      return Q;
    }
    return R;
  }
  if (!M.loose()) return M.clone();
  DevMat R;
  R.rows = M.rows;
  R.cols = M.cols;
  R.cplx = M.cplx;
  R.zero_free = M.zero_free;
  const int n = M.cols;
  R.outer.alloc((size_t)n + 1);
  scan_async<int32_t>(M.cnt.p, R.outer.p, (int64_t)n);
  int64_t nnz = 0;
  {
    ScalarFetch f;
    f.add(R.outer.p + n, 1, &nnz);
    f.run();
  }
  R.nnz = nnz;
  R.inner.alloc((size_t)nnz + kIndexSlack);
  R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
  const int nblocks = cdiv(n, 4);
  if (n)
    dispatch_type(M.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_compact<T>), dim3(xcd_grid(nblocks)), dim3(256), 0, stream(), n, M.outer.p, nullptr, nullptr,
                         R.outer.p, M.inner.p, reinterpret_cast<const T*>(M.val.p), nullptr, nullptr, R.inner.p,
                         reinterpret_cast<T*>(R.val.p), nblocks);
    });
  return R;
}

void pack(DevMat& M) {
  if (M.loose() || M.expanded() || M.blocked()) M = packed_copy(M);
}

bool square_keep_loose(DevMat& X, double threshold, bool dense_rule, const DevMat& D, double out[2], double* trace_out,
                       int32_t col_offset) {
  if (X.cplx || D.cplx || X.rows != X.cols) return false;
  LooseProduct L;
  DevMat AB;
  SlabFusion fu;
  fu.mode = options().fused_update ? 1 : 0;
  fu.D = &D;
  fu.col_offset = col_offset;
  if (fu.mode && !X.expanded() && !X.loose()) relabel_enter(X, D);   // (a band hidden under the labels)
  if (X.expanded() && X.slab->labelled()) {
    fu.D = relabelled_operand(D);
    if (!fu.D) {
      pack(X);
      fu.D = &D;
    }
  }
  if (X.expanded()) {
    if (slab_step(X, fu, threshold, dense_rule)) {
      out[0] = fu.dot;
      out[1] = 0.0;
      if (trace_out) *trace_out = fu.trace;
      return true;
    }
    if (X.slab->labelled()) relabel_giveup(D);
    pack(X);
    fu.D = &D;
  }
  spgemm(X, X, AB, 1.0, threshold, dense_rule, &L, nullptr, fu.mode ? &fu : nullptr);
  if (fu.done) {  // dot and trace came out of the multiply's epilogue
    X = std::move(fu.result);
    out[0] = fu.dot;
    out[1] = 0.0;
    if (trace_out) *trace_out = fu.trace;
    return true;
  }
  if (!L.valid) {  // another kernel computed it (packed)
    X = std::move(AB);
    if (last_spgemm_stats().block) X.block_hint = 1;   // (the next TRS2 step takes the iterate in block form)
    dot_trace(X, D, out, trace_out, col_offset);
    return true;
  }
  const int n = L.cols;
  DevBuf<int64_t> total((size_t)n + 1);
  scan_async<int32_t>(L.count.p, total.p, (int64_t)n);
  DevMat R;
  R.rows = L.rows;
  R.cols = n;
  R.cplx = false;
  R.slots = L.slots;
  R.nnz = L.slots;  // (not known yet; non-zero keeps the reduction from returning early)
  R.outer = std::move(L.start);
  R.cnt = std::move(L.count);
  R.inner = std::move(L.inner);
  R.val = std::move(L.val);
  int64_t extra[2] = {-1, 0};
  DevBuf<int64_t> ex(2);
  hipLaunchKernelGGL(k_pick2_i64, dim3(1), dim3(1), 0, stream(), total.p + n,
                     L.prod_index >= 0 ? L.prod_scan.p + L.prod_index : (const int64_t*)nullptr, ex.p);
  dot_trace_impl(R, D, out, trace_out, col_offset, ex.p, 2, extra);
  if (extra[0] < 0) {
    ScalarFetch f;
    f.add(ex.p, 2, extra);
    f.run();
  }
  R.nnz = extra[0];
  X = std::move(R);
  SpgemmAccum& acc = spgemm_accum();
  acc.nnz_c += extra[0];
  acc.alg_bytes += 12.0 * (double)extra[0];
  acc.products += extra[1];
  last_spgemm_stats().nnz_c = extra[0];
  last_spgemm_stats().products = extra[1];
  return true;
}

void axpby(const LooseProduct& A, DevMat& B, double alpha, double beta, double threshold, const DevMat* D, double* dot_out,
           double* trace_out, int32_t trace_col_offset, int64_t* a_nnz_out, bool keep_loose) {
  if (!A.valid) NTP_FATAL("axpby: invalid loose product");
  Csc v{A.rows, A.cols, A.start.p, A.inner.p, A.val.p};
  v.cnt = A.count.p;
  const MergeOperand a{v, A.rows, A.cols, false, A.slots, true};
  int64_t a_nnz = 0, products = 0;
  axpby_impl(a, B, alpha, beta, threshold, D, dot_out, trace_out, trace_col_offset, &a_nnz,
             A.prod_index >= 0 ? A.prod_scan.p + A.prod_index : nullptr, &products, 0, keep_loose);
  if (a_nnz_out) *a_nnz_out = a_nnz;
  // the multiply that produced A could not account for its output: do it now
  SpgemmAccum& acc = spgemm_accum();
  acc.nnz_c += a_nnz;
  acc.alg_bytes += 12.0 * (double)a_nnz;
  acc.products += products;
  last_spgemm_stats().nnz_c = a_nnz;
  last_spgemm_stats().products = products;
}

void pairwise(const DevMat& A, const DevMat& B, DevMat& C, bool conj_a) {
  if (A.rows != B.rows || A.cols != B.cols || A.cplx != B.cplx) NTP_FATAL("pairwise: operand mismatch");
  const int n = A.cols;
  DevMat R;
  R.rows = A.rows;
  R.cols = n;
  R.cplx = A.cplx;
  R.outer.alloc((size_t)n + 1);
  DevBuf<int32_t> count((size_t)n);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_pairwise<T>), dim3(cdiv(n, 64)), dim3(64), 0, stream(), view(A), view(B), nullptr, nullptr,
                       (T*)nullptr, count.p, 0);
  });
  scan_async<int32_t>(count.p, R.outer.p, (int64_t)n);
  int64_t nnz = 0;
  HIP_CHECK(hipMemcpyAsync(&nnz, R.outer.p + n, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
  sync_stream();
  R.nnz = nnz;
  R.inner.alloc((size_t)nnz + kIndexSlack);
  R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_pairwise<T>), dim3(cdiv(n, 64)), dim3(64), 0, stream(), view(A), view(B), R.outer.p,
                       R.inner.p, reinterpret_cast<T*>(R.val.p), nullptr, conj_a ? 1 : 0);
  });
  C = std::move(R);
}

namespace {
void finish_sum2(DevBuf<double>& partial, int nb, double out[2]) {
  DevBuf<double> res(2);
  reduce_sum2_async(partial.p, nb, res.p);
  res.download(out, 2);
}
}  // namespace

void dot(const DevMat& A, const DevMat& B, double out[2]) { dot_trace(A, B, out, nullptr, 0); }

void dot_trace(const DevMat& A, const DevMat& B, double out[2], double* trace_out, int32_t col_offset) {
  dot_trace_impl(A, B, out, trace_out, col_offset, nullptr, 0, nullptr);
}
namespace {
// A may be loose.  d_extra: device scalars fetched with the result (one read-back for everything)
void dot_trace_impl(const DevMat& A, const DevMat& B, double out[2], double* trace_out, int32_t col_offset,
                    const int64_t* d_extra, int n_extra, int64_t* extra_out) {
  if (A.rows != B.rows || A.cols != B.cols || A.cplx != B.cplx) NTP_FATAL("dot: operand mismatch");
  if (A.expanded()) {
    DevMat P = packed_copy(A);
    dot_trace_impl(P, B, out, trace_out, col_offset, d_extra, n_extra, extra_out);
    return;
  }
  out[0] = out[1] = 0;
  if (trace_out) *trace_out = 0.0;
  if (A.nnz == 0) return;
  if (B.nnz == 0) {
    if (trace_out) *trace_out = trace(A, col_offset);
    return;
  }
  const int nb = std::min(cdiv(A.cols, 4), 8192);
  DevBuf<double> partial((size_t)2 * nb), tpartial;
  if (trace_out) tpartial.alloc((size_t)2 * nb);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_dot<T>), dim3(nb), dim3(256), 0, stream(), lview(A), view(B), partial.p, nb,
                       trace_out ? tpartial.p : (double*)nullptr, col_offset);
  });
  DevBuf<double> res(4);
  reduce_sum2_async(partial.p, nb, res.p);
  if (trace_out) reduce_sum2_async(tpartial.p, nb, res.p + 2);
  double h[4] = {0, 0, 0, 0};
  ScalarFetch f;
  f.add(res.p, trace_out ? 4 : 2, h);
  if (d_extra && n_extra) f.add(d_extra, n_extra, extra_out);
  f.run();
  out[0] = h[0];
  out[1] = h[1];
  if (trace_out) *trace_out = h[2];
}
}  // namespace

void grand_sum(const DevMat& A, double out[2]) {
  out[0] = out[1] = 0;
  if (A.nnz == 0) return;
  const int nb = std::min(cdiv(A.nnz, 256), 2048);
  DevBuf<double> partial((size_t)2 * nb);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_grand_sum<T>), dim3(nb), dim3(256), 0, stream(), reinterpret_cast<const T*>(A.val.p), A.nnz,
                       partial.p);
  });
  finish_sum2(partial, nb, out);
}

double trace(const DevMat& A, int32_t col_offset) {
  if (A.nnz == 0) return 0.0;
  if (A.expanded()) {
    DevMat P = packed_copy(A);
    return trace(P, col_offset);
  }
  const int nb = std::min(cdiv(A.cols, 256), 1024);
  DevBuf<double> partial((size_t)2 * nb);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_trace<T>), dim3(nb), dim3(256), 0, stream(), lview(A), col_offset, partial.p);
  });
  double out[2];
  finish_sum2(partial, nb, out);
  return out[0];
}

void column_abs_sums(const DevMat& A, DevBuf<double>& out) {
  out.alloc((size_t)A.cols);
  if (A.cols == 0) return;
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_colstat<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), view(A), 0, 0,
                       out.p, (double*)nullptr);
  });
}

double max_of(const DevBuf<double>& v, size_t n) {
  if (n == 0) return 0.0;
  DevBuf<double> res(2);
  launch_reduce_minmax(nullptr, v.p, (int64_t)n, res.p);
  double h[2];
  res.download(h, 2);
  return h[1];
}

void gershgorin(const DevMat& A, int32_t col_offset, double* mn, double* mx) {
  if (A.cols == 0) { *mn = INFINITY; *mx = -INFINITY; return; }
  DevBuf<double> lo((size_t)A.cols), hi((size_t)A.cols), res(2);
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_colstat<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), view(A),
                       col_offset, 1, lo.p, hi.p);
  });
  launch_reduce_minmax(lo.p, hi.p, (int64_t)A.cols, res.p);
  double h[2];
  res.download(h, 2);
  *mn = h[0];
  *mx = h[1];
}

void scale(DevMat& A, double c) {
  pack(A);
  value_epoch() += 1;
  const int64_t n = A.nnz * (int64_t)A.wval();
  if (n == 0) return;
  hipLaunchKernelGGL(k_scale, dim3(std::min(cdiv(n, 256), 8192)), dim3(256), 0, stream(), A.val.p, n, c);
}

namespace {
template <typename T>
__global__ __launch_bounds__(256) void k_scale_columns(const int64_t* __restrict__ outer, T* __restrict__ val,
                                                       const T* __restrict__ factor, int cols) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= cols) return;
  const T f = factor[j];
  for (int64_t p = outer[j] + lane_id(); p < outer[j + 1]; p += WAVE) val[p] = Sc<T>::mul(f, val[p]);
}
}  // namespace

// values of column j *= factor[j] (factor: cols scalars of the matrix' type, device memory)
void scale_columns(DevMat& A, const double* d_factor) {
  pack(A);
  value_epoch() += 1;
  if (A.nnz == 0) return;
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_scale_columns<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.outer.p,
                       reinterpret_cast<T*>(A.val.p), reinterpret_cast<const T*>(d_factor), A.cols);
  });
}

void conjugate(DevMat& A) {
  pack(A);
  value_epoch() += 1;
  if (!A.cplx || A.nnz == 0) return;
  hipLaunchKernelGGL(k_conj, dim3(std::min(cdiv(A.nnz, 256), 8192)), dim3(256), 0, stream(), A.val.p, A.nnz);
}

DevMat to_complex(const DevMat& A) {
  if (A.cplx) return A.clone();
  DevMat R;
  R.alloc(A.rows, A.cols, true, A.nnz);
  HIP_CHECK(hipMemcpyAsync(R.outer.p, A.outer.p, sizeof(int64_t) * ((size_t)A.cols + 1), hipMemcpyDeviceToDevice, stream()));
  if (A.nnz) {
    HIP_CHECK(hipMemcpyAsync(R.inner.p, A.inner.p, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToDevice, stream()));
    hipLaunchKernelGGL(k_to_complex, dim3(std::min(cdiv(A.nnz, 256), 8192)), dim3(256), 0, stream(), A.val.p, R.val.p, A.nnz);
  }
  return R;
}

DevMat to_real(const DevMat& A) {
  if (!A.cplx) return A.clone();
  DevMat R;
  R.alloc(A.rows, A.cols, false, A.nnz);
  HIP_CHECK(hipMemcpyAsync(R.outer.p, A.outer.p, sizeof(int64_t) * ((size_t)A.cols + 1), hipMemcpyDeviceToDevice, stream()));
  if (A.nnz) {
    HIP_CHECK(hipMemcpyAsync(R.inner.p, A.inner.p, sizeof(int32_t) * (size_t)A.nnz, hipMemcpyDeviceToDevice, stream()));
    hipLaunchKernelGGL(k_to_real, dim3(std::min(cdiv(A.nnz, 256), 8192)), dim3(256), 0, stream(), A.val.p, R.val.p, A.nnz);
  }
  return R;
}

DevMat identity(int32_t n, int32_t col_offset, int32_t cols, bool cplx) {
  const int64_t ones = std::max<int64_t>(0, std::min<int64_t>(cols, (int64_t)n - col_offset));
  DevMat R;
  R.alloc(n, cols, cplx, ones);
  hipLaunchKernelGGL(k_identity, dim3(cdiv(cols + 1, 256)), dim3(256), 0, stream(), R.outer.p, R.inner.p, R.val.p, n,
                     col_offset, cols, cplx ? 1 : 0);
  return R;
}

int64_t identity_check(const DevMat& A, int32_t col_offset) {
  if (A.cols == 0) return 0;
  DevBuf<unsigned long long> flags(2);
  flags.zero();
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_identity_check<T>), dim3(cdiv(A.cols, 256)), dim3(256), 0, stream(), view(A), col_offset, flags.p);
  });
  unsigned long long h[2];
  flags.download(h, 2);
  return h[0] ? -1 : (int64_t)h[1];
}

namespace {
DevMat remap_impl(const DevMat& A, const int32_t* d_row_map, const int32_t* d_col_map, bool transpose_after,
                  int32_t new_rows, int32_t col_lo, int32_t col_hi, bool drop_zero) {
  const int32_t new_cols = col_hi - col_lo;
  DevMat R;
  if (A.nnz == 0) {
    R.reset_empty(new_rows, new_cols, A.cplx);
    return R;
  }
  const int64_t nnz = A.nnz;
  DevBuf<int32_t> colidx((size_t)nnz);
  hipLaunchKernelGGL(k_expand_cols, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.outer.p, A.cols, colidx.p);
  DevBuf<unsigned long long> keys((size_t)nnz), keys2((size_t)nnz), nvalid(1);
  DevBuf<unsigned int> pay((size_t)nnz), pay2((size_t)nnz);
  nvalid.zero();
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_remap_keys<T>), dim3(cdiv(nnz, 256)), dim3(256), 0, stream(), colidx.p, A.inner.p,
                       reinterpret_cast<const T*>(A.val.p), nnz, d_row_map, d_col_map, transpose_after ? 1 : 0, col_lo,
                       col_hi, drop_zero ? 1 : 0, keys.p, pay.p);
  });
  hipLaunchKernelGGL(k_count_valid, dim3(std::min(cdiv(nnz, 256), 2048)), dim3(256), 0, stream(), keys.p, nnz, nvalid.p);
  size_t tmp_bytes = 0;
  HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys.p, keys2.p, pay.p, pay2.p, (size_t)nnz, 0, 64, stream()));
  DevBuf<char> tmp(tmp_bytes);
  HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, keys.p, keys2.p, pay.p, pay2.p, (size_t)nnz, 0, 64, stream()));
  unsigned long long nkeep = 0;
  nvalid.download(&nkeep, 1);
  R.alloc(new_rows, new_cols, A.cplx, (int64_t)nkeep);
  DevBuf<int32_t> colcount((size_t)new_cols + 1);
  colcount.zero();
  if (nkeep) {
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_remap_gather<T>), dim3(cdiv((int64_t)nkeep, 256)), dim3(256), 0, stream(), keys2.p, pay2.p,
                         (int64_t)nkeep, reinterpret_cast<const T*>(A.val.p), R.inner.p, reinterpret_cast<T*>(R.val.p),
                         colcount.p);
    });
  }
  scan_async<int32_t>(colcount.p, R.outer.p, (int64_t)new_cols);
  sync_stream();
  return R;
}
}  // namespace

DevMat transpose(const DevMat& A) { return remap_impl(A, nullptr, nullptr, true, A.cols, 0, A.rows, false); }

DevMat remap_general(const DevMat& A, const int32_t* d_row_map, const int32_t* d_col_map, int32_t new_rows,
                     int32_t col_lo, int32_t col_hi, bool drop_exact_zeros) {
  return remap_impl(A, d_row_map, d_col_map, false, new_rows, col_lo, col_hi, drop_exact_zeros);
}
DevMat transpose_slice(const DevMat& A, int32_t col_lo, int32_t col_hi) {
  return remap_impl(A, nullptr, nullptr, true, A.cols, col_lo, col_hi, false);
}

void copy_shift_i64(const int64_t* d_src, int64_t* d_dst, int64_t count, int64_t shift) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_shift_outer, dim3(cdiv(count, 256)), dim3(256), 0, stream(), d_src, d_dst, (int)(count - 1), shift);
}

void rebase_i64(const int64_t* d_src, int64_t* d_dst, int64_t count, int64_t add) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_rebase_i64, dim3(cdiv(count, 256)), dim3(256), 0, stream(), d_src, d_dst, (int)count, add);
}
void fill_i64(int64_t* d_dst, int64_t count, int64_t v) {
  if (count <= 0) return;
  hipLaunchKernelGGL(k_fill_i64, dim3(cdiv(count, 256)), dim3(256), 0, stream(), d_dst, count, v);
}
namespace {
// request record of one rank for the halo exchange: (first row, last row of the local B panel, nnz(A_loc), nnz(B_loc))
__global__ void k_halo_request(Csc B, long long nnz_a, long long nnz_b, long long* __restrict__ out4) {
  // out4[0], out4[1] were preset to (INT_MAX, -1); one block per 256 columns reduces with atomics (few blocks)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int lo = INT_MAX, hi = -1;
  if (j < B.cols) {
    const int64_t s = B.outer[j], e = B.outer[j + 1];
    if (e > s) {
      lo = B.inner[s];
      hi = B.inner[e - 1];
    }
  }
  lo = wave_min_i32(lo);
  hi = wave_max_i32(hi);
  if (lane_id() == 0) {
    if (lo != INT_MAX) atomicMin(&out4[0], (long long)lo);
    if (hi >= 0) atomicMax(&out4[1], (long long)hi);
  }
  if (j == 0) {
    out4[2] = nnz_a;
    out4[3] = nnz_b;
  }
}
// per requester q: entry offsets of my panel at the boundaries of the segment [sa[q], sb[q]) it needs, and the
// number of entries in between (one row of the P x P count matrix)
__global__ void k_halo_bounds(const int64_t* __restrict__ outer, int c0, const int32_t* __restrict__ sa,
                              const int32_t* __restrict__ sb, int P, long long* __restrict__ bound,
                              long long* __restrict__ cnt_row) {
  const int q = threadIdx.x;
  if (q >= P) return;
  const long long a = outer[sa[q] - c0], b = outer[sb[q] - c0];
  bound[2 * q] = a;
  bound[2 * q + 1] = b;
  cnt_row[q] = b - a;
}
// the whole P x P count matrix from the gathered column offsets of all panels: cnt[s * P + q] = entries of rank s's
// panel inside the column range rank q asked for; bound[2q], bound[2q + 1] = entry offsets of MY panel at the ends of
// the segment requester q gets from me.  req = gathered (kmin, kmax, nnzA, nnzB) records, outer_all = gathered panel
// offsets, `pitch` entries per rank.  Same arithmetic as halo_segment / panel_range on the host.
__global__ void k_halo_counts(const long long* __restrict__ req, const long long* __restrict__ outer_all, int pitch, int dim, int P,
                              int me, long long* __restrict__ cnt, long long* __restrict__ bound) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P * P) return;
  const int s = t / P, q = t % P;
  const int c0 = (int)(((long long)dim * s) / P), c1 = (int)(((long long)dim * (s + 1)) / P);
  const long long klo = req[4 * q], khi = req[4 * q + 1];
  int a = c0, b = c0;
  if (khi >= klo) {
    const int lo = max(c0, (int)klo), hi = min(c1, (int)khi + 1);
    if (hi > lo) {
      a = lo;
      b = hi;
    }
  }
  const long long oa = outer_all[(size_t)s * pitch + (a - c0)], ob = outer_all[(size_t)s * pitch + (b - c0)];
  cnt[t] = ob - oa;
  if (s == me) {
    bound[2 * q] = oa;
    bound[2 * q + 1] = ob;
  }
}
}  // namespace

void halo_counts_async(const int64_t* d_req, const int64_t* d_outer_all, int pitch, int32_t dim, int P, int me, int64_t* d_cnt,
                       int64_t* d_bound) {
  hipLaunchKernelGGL(k_halo_counts, dim3(cdiv((int64_t)P * P, 256)), dim3(256), 0, stream(),
                     reinterpret_cast<const long long*>(d_req), reinterpret_cast<const long long*>(d_outer_all), pitch, dim, P, me,
                     reinterpret_cast<long long*>(d_cnt), reinterpret_cast<long long*>(d_bound));
}

namespace {
__global__ void k_halo_interior(Csc B, int c0, int c1, long long* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  bool interior = false;
  if (j < B.cols) {
    const int64_t s = B.outer[j], e = B.outer[j + 1];
    interior = (e <= s) || (B.inner[s] >= c0 && B.inner[e - 1] < c1);
  }
  const int lo = wave_min_i32(interior ? j : INT_MAX), hi = wave_max_i32(interior ? j : -1);
  const unsigned long long m = __ballot(interior);
  if (lane_id() == 0 && m) {
    atomicMin(&out[0], (long long)lo);
    atomicMax(&out[1], (long long)hi);
    atomicAdd(reinterpret_cast<unsigned long long*>(&out[2]), (unsigned long long)__popcll(m));
  }
}
__global__ void k_halo_interior_offsets(const int64_t* __restrict__ outer, long long* __restrict__ out) {
  if (out[1] >= out[0]) {
    out[3] = outer[out[0]];
    out[4] = outer[out[1] + 1];
  } else {
    out[3] = out[4] = 0;
  }
}
}  // namespace

void halo_interior_async(const DevMat& B, int32_t c0, int32_t c1, int64_t* d_out5) {
  const long long init[5] = {INT_MAX, -1, 0, 0, 0};
  HIP_CHECK(hipMemcpyAsync(d_out5, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  if (B.cols == 0) return;
  hipLaunchKernelGGL(k_halo_interior, dim3(cdiv(B.cols, 256)), dim3(256), 0, stream(), view(B), c0, c1,
                     reinterpret_cast<long long*>(d_out5));
  hipLaunchKernelGGL(k_halo_interior_offsets, dim3(1), dim3(1), 0, stream(), B.outer.p, reinterpret_cast<long long*>(d_out5));
}

void halo_request_async(const DevMat& B, int64_t nnz_a, int64_t* d_out4) {
  const long long init[2] = {INT_MAX, -1};
  HIP_CHECK(hipMemcpyAsync(d_out4, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL(k_halo_request, dim3(std::max(1, cdiv(B.cols, 256))), dim3(256), 0, stream(), view(B), (long long)nnz_a,
                     (long long)B.nnz, reinterpret_cast<long long*>(d_out4));
}

void halo_bounds_async(const DevMat& A, int32_t c0, const int32_t* d_sa, const int32_t* d_sb, int P, int64_t* d_bound,
                       int64_t* d_cnt_row) {
  if (P > 1024) NTP_FATAL("halo_bounds: too many ranks");
  hipLaunchKernelGGL(k_halo_bounds, dim3(1), dim3(1024), 0, stream(), A.outer.p, c0, d_sa, d_sb, P,
                     reinterpret_cast<long long*>(d_bound), reinterpret_cast<long long*>(d_cnt_row));
}

// ------------------------------------------------------------------ halo exchange of a panel in slab form
namespace {
__global__ void k_slab_request(const int32_t* __restrict__ first, const int32_t* __restrict__ last, int n, long long nnz,
                               const long long* __restrict__ d_nnz, long long* __restrict__ out4) {
  // out4[0], out4[1] were preset to (INT_MAX, -1)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool has = j < n && last[j] >= first[j];
  const int lo = wave_min_i32(has ? first[j] : INT_MAX), hi = wave_max_i32(has ? last[j] : -1);
  if (lane_id() == 0 && hi >= lo) {
    atomicMin(&out4[0], (long long)lo);
    atomicMax(&out4[1], (long long)hi);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out4[2] = out4[3] = d_nnz ? d_nnz[0] : nnz;
}
// (al > 1: the runs travel in the aligned zero-padded slots the MFMA tile kernel reads several rows per lane from --
// span = slot size, SlabForm::row_pad)
__global__ void k_slab_extents(const int32_t* __restrict__ first, const int32_t* __restrict__ last, int n,
                               long long* __restrict__ ext, int32_t* __restrict__ span, int al) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int f = first[j], l = last[j];
  ext[j] = (long long)(unsigned)f | ((long long)l << 32);
  span[j] = l >= f ? (l / al + 1) * al - f / al * al : 0;
}
__global__ __launch_bounds__(256) void k_slab_pack_runs(const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                        const int64_t* __restrict__ off, const double* __restrict__ val,
                                                        const int64_t* __restrict__ pre, int ja, int jb, double* __restrict__ dst, int al) {
  const int j = ja + (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= jb) return;
  const int lane = lane_id();
  const int f = first[j], l = last[j];
  if (l < f) return;
  const double* __restrict__ src = val + off[j];
  double* __restrict__ d = dst + (pre[j] - pre[ja]);
  const int a0 = f / al * al, a1 = (l / al + 1) * al;   // the slot: zeros around the run
  for (int r = a0 + lane; r < a1; r += WAVE) d[r - a0] = (r >= f && r <= l) ? src[r - f] : 0.0;
}
// extents and run addresses of the columns ka .. kb a rank needs: its own from its buffers, the others from the
// receive buffer (source s: its segment [ra_s, rb_s) packed back to back at recv + zoff[s])
// (the segments of the receive buffer -- first column and offset per owner -- travel as kernel arguments when there are at most
// 16 ranks: two small uploads less on the host's critical path behind a step's read-back)
struct HaloSegs {
  int32_t ra[16];
  int64_t zoff[16];
};
__global__ void k_slab_halo_layout(const long long* __restrict__ ext_all, const long long* __restrict__ pre_all, int pitch,
                                   int dim, int P, int me, int ka, int kb, const int32_t* __restrict__ ra_p,
                                   const int64_t* __restrict__ zoff_p, const HaloSegs segs, const double* __restrict__ recv,
                                   const int64_t* __restrict__ own_off, const double* __restrict__ own_val,
                                   int32_t* __restrict__ first, int32_t* __restrict__ last,
                                   unsigned long long* __restrict__ addr, const long long* __restrict__ cnt_all,
                                   int32_t* __restrict__ count, int al) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kb - ka) return;
  const int k = ka + i;
  int s = (int)(((long long)k * P) / dim);   // owner: the panel whose range holds k
  while (s > 0 && (int)(((long long)dim * s) / P) > k) --s;
  while (s + 1 < P && (int)(((long long)dim * (s + 1)) / P) <= k) ++s;
  const int c0 = (int)(((long long)dim * s) / P);
  const long long e = ext_all[(size_t)s * pitch + (k - c0)];
  first[i] = (int)(unsigned)(e & 0xffffffffll);
  last[i] = (int)(e >> 32);
  const double* p;
  if (s == me) p = own_val + own_off[k - c0];
  else {
    const int ras = ra_p ? ra_p[s] : segs.ra[s];
    const int64_t zs = zoff_p ? zoff_p[s] : segs.zoff[s];
    p = recv + zs + (pre_all[(size_t)s * pitch + (k - c0)] - pre_all[(size_t)s * pitch + (ras - c0)]) + (first[i] - first[i] / al * al);
  }
  addr[i] = (unsigned long long)reinterpret_cast<uintptr_t>(p);
  if (count) count[i] = (int32_t)cnt_all[(size_t)s * pitch + (k - c0)];
}
__global__ void k_widen_i32(const int32_t* __restrict__ src, long long* __restrict__ dst, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) dst[j] = src[j];
}
}  // namespace

namespace {
__global__ void k_unpack_extents(const long long* __restrict__ ext_all, int pitch, int dim, int P, int32_t* __restrict__ gfirst,
                                 int32_t* __restrict__ glast) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= dim) return;
  int s = (int)(((long long)k * P) / dim);   // owner: the panel whose range holds k
  while (s > 0 && (int)(((long long)dim * s) / P) > k) --s;
  while (s + 1 < P && (int)(((long long)dim * (s + 1)) / P) <= k) ++s;
  const int c0 = (int)(((long long)dim * s) / P);
  const long long e = ext_all[(size_t)s * pitch + (k - c0)];
  gfirst[k] = (int)(unsigned)(e & 0xffffffffll);
  glast[k] = (int)(e >> 32);
}
}  // namespace

void slab_plan_panel_async(const DevMat& X, const int64_t* d_ext_all, int pitch, int32_t dim, int P, SlabPlan& plan,
                           DevBuf<int32_t>& gfirst, DevBuf<int32_t>& glast, unsigned long long* stats24) {
  gfirst.alloc((size_t)dim);
  glast.alloc((size_t)dim);
  hipLaunchKernelGGL(k_unpack_extents, dim3(cdiv(dim, 256)), dim3(256), 0, stream(), reinterpret_cast<const long long*>(d_ext_all),
                     pitch, dim, P, gfirst.p, glast.p);
  const bool tile = options().spgemm_fma == 1;
  launch_slab_plan(plan, X.cols, X.slab->first.p, X.slab->last.p, gfirst.p, glast.p, tile ? 16 * tile_rows() : 0, stats24);
}

void slab_request_async(const DevMat& X, int64_t* d_out4, const long long* d_nnz) {
  const long long init[2] = {INT_MAX, -1};
  HIP_CHECK(hipMemcpyAsync(d_out4, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  hipLaunchKernelGGL(k_slab_request, dim3(std::max(1, cdiv(X.cols, 256))), dim3(256), 0, stream(), X.slab->first.p,
                     X.slab->last.p, X.cols, (long long)X.nnz, d_nnz, reinterpret_cast<long long*>(d_out4));
}

namespace {
// request (first / last row over the columns, entry count), packed extents, aligned spans and -- statistics -- the entry counts
// of the columns in ONE pass: what k_slab_request + k_slab_extents + k_widen_i32 do, with a reduction over the workgroup in front of
// the two atomics (a wave each hitting the same two addresses made the request 13.5 us at 32 768 columns)
__global__ __launch_bounds__(256) void k_slab_export(const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                     const int32_t* __restrict__ count, int n, long long nnz,
                                                     const long long* __restrict__ d_nnz, int al, long long* __restrict__ out4,
                                                     long long* __restrict__ ext, int32_t* __restrict__ span,
                                                     long long* __restrict__ cnt64) {
  __shared__ int smin[4], smax[4];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const bool in = j < n;
  const int f = in ? first[j] : INT_MAX, l = in ? last[j] : -1;
  const bool has = in && l >= f;
  if (in) {
    ext[j] = (long long)(unsigned)f | ((long long)l << 32);
    span[j] = l >= f ? (l / al + 1) * al - f / al * al : 0;
    if (cnt64) cnt64[j] = count[j];
  }
  const int lo = wave_min_i32(has ? f : INT_MAX), hi = wave_max_i32(has ? l : -1);
  if (lane_id() == 0) { smin[threadIdx.x / WAVE] = lo; smax[threadIdx.x / WAVE] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int blo = min(min(smin[0], smin[1]), min(smin[2], smin[3])), bhi = max(max(smax[0], smax[1]), max(smax[2], smax[3]));
    if (bhi >= blo) {   // out4[0], out4[1] were preset to (INT_MAX, -1)
      atomicMin(&out4[0], (long long)blo);
      atomicMax(&out4[1], (long long)bhi);
    }
    if (blockIdx.x == 0) out4[2] = out4[3] = d_nnz ? d_nnz[0] : nnz;
  }
}
}  // namespace

void slab_export_async(const DevMat& X, int64_t* d_out4, const long long* d_nnz, int64_t* d_ext, int64_t* d_pre, int64_t* d_cnt64) {
  const int n = X.cols;
  const long long init[2] = {INT_MAX, -1};
  HIP_CHECK(hipMemcpyAsync(d_out4, init, sizeof(init), hipMemcpyHostToDevice, stream()));
  DevBuf<int32_t> span((size_t)n);
  hipLaunchKernelGGL(k_slab_export, dim3(std::max(1, cdiv(n, 256))), dim3(256), 0, stream(), X.slab->first.p, X.slab->last.p,
                     X.slab->count.p, n, (long long)X.nnz, d_nnz, std::max(1, X.slab->row_pad), reinterpret_cast<long long*>(d_out4),
                     reinterpret_cast<long long*>(d_ext), span.p, reinterpret_cast<long long*>(d_cnt64));
  scan_async<int32_t>(span.p, d_pre, (int64_t)n);
}

void slab_extents_async(const DevMat& X, int64_t* d_ext, int64_t* d_pre) {
  const int n = X.cols;
  DevBuf<int32_t> span((size_t)n);
  hipLaunchKernelGGL(k_slab_extents, dim3(cdiv(n, 256)), dim3(256), 0, stream(), X.slab->first.p, X.slab->last.p, n,
                     reinterpret_cast<long long*>(d_ext), span.p, std::max(1, X.slab->row_pad));
  scan_async<int32_t>(span.p, d_pre, (int64_t)n);
}

void slab_pack_runs_async(const DevMat& X, const int64_t* d_pre, int32_t ja, int32_t jb, double* dst) {
  if (jb <= ja) return;
  const SlabForm& f = *X.slab;
  hipLaunchKernelGGL(k_slab_pack_runs, dim3(cdiv((int64_t)(jb - ja) * WAVE, 256)), dim3(256), 0, stream(), f.first.p, f.last.p,
                     f.off.p, f.val.p, d_pre, ja, jb, dst, std::max(1, f.row_pad));
}

void slab_halo_layout_async(const int64_t* d_ext_all, const int64_t* d_pre_all, int pitch, int32_t dim, int P, int me,
                            int32_t ka, int32_t kb, const int32_t* d_ra, const int64_t* d_zoff, const double* d_recv,
                            const DevMat& X, int32_t* d_first, int32_t* d_last, unsigned long long* d_addr,
                            const int64_t* d_cnt_all, int32_t* d_count, const int32_t* h_ra, const int64_t* h_zoff) {
  if (kb <= ka) return;
  HaloSegs segs;
  std::memset(&segs, 0, sizeof(segs));
  if (h_ra && h_zoff && P <= 16) {   // (by value: d_ra / d_zoff are not read)
    for (int q = 0; q < P; ++q) { segs.ra[q] = h_ra[q]; segs.zoff[q] = h_zoff[q]; }
    d_ra = nullptr;
    d_zoff = nullptr;
  }
  hipLaunchKernelGGL(k_slab_halo_layout, dim3(cdiv(kb - ka, 256)), dim3(256), 0, stream(),
                     reinterpret_cast<const long long*>(d_ext_all), reinterpret_cast<const long long*>(d_pre_all), pitch, dim, P, me,
                     ka, kb, d_ra, d_zoff, segs, d_recv, X.slab->off.p, X.slab->val.p, d_first, d_last, d_addr,
                     reinterpret_cast<const long long*>(d_cnt_all), d_count, std::max(1, X.slab->row_pad));
}

void slab_counts_async(const DevMat& X, int64_t* d_cnt64) {
  hipLaunchKernelGGL(k_widen_i32, dim3(cdiv(X.cols, 256)), dim3(256), 0, stream(), X.slab->count.p,
                     reinterpret_cast<long long*>(d_cnt64), X.cols);
}

// ------------------------------------------------------------------ relabelled operands (label-ordered slab steps)
namespace {
struct RelabelCache {   // one entry: the order found for the pattern of D, D in that order
  const void* val = nullptr;
  unsigned long long serial = 0, epoch = 0;
  int64_t nnz = -1;
  int32_t cols = 0;
  bool usable = false;
  DevBuf<int32_t> newpos, lab;
  DevMat Dr;
  // the sparsity pattern the order was found for (an order-independent sum of per-entry hashes + dimensions): the next
  // operand with the SAME pattern and other values -- the next cycle of a self-consistent-field loop -- reuses the order
  // (and the verdict "no band in it") without searching again
  unsigned long long fingerprint = 0;
  bool searched = false;
};
__global__ __launch_bounds__(256) void k_pattern_fingerprint(Csc A, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  unsigned long long h = 0;
  for (int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE; j < A.cols; j += gridDim.x * blockDim.x / WAVE) {
    for (int64_t p = A.outer[j] + lane_id(); p < A.outer[j + 1]; p += WAVE) {
      unsigned long long x = ((unsigned long long)(unsigned)j << 32) | (unsigned)A.inner[p];
      x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;   // (murmur3 finaliser)
      h += x;
    }
  }
  h = (unsigned long long)wave_sum_i64((int64_t)h);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = h;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
unsigned long long pattern_fingerprint(const DevMat& D) {
  DevBuf<unsigned long long> acc(1);
  acc.zero();
  hipLaunchKernelGGL(k_pattern_fingerprint, dim3(1024), dim3(256), 0, stream(), view(D), acc.p);
  unsigned long long h = 0;
  ScalarFetch f;
  f.add(acc.p, 1, &h);
  f.run();
  return h ^ ((unsigned long long)D.nnz * 0x9e3779b97f4a7c15ull) ^ (unsigned long long)D.cols;
}
}  // namespace
unsigned long long pattern_fingerprint_of(const DevMat& A) { return pattern_fingerprint(A); }
int64_t column_span_sum(const DevMat& A) {
  const int n = A.cols;
  if (n == 0 || A.nnz == 0) return 0;
  DevBuf<int32_t> f((size_t)n), l((size_t)n), cnt((size_t)n), span((size_t)n);
  DevBuf<int64_t> pre((size_t)n + 1);
  hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(A), f.p, l.p, cnt.p);
  hipLaunchKernelGGL(k_span_of, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f.p, l.p, span.p, n);
  scan_async<int32_t>(span.p, pre.p, (int64_t)n);
  int64_t tot = 0;
  ScalarFetch ft;
  ft.add(pre.p + n, 1, &tot);
  ft.run();
  return tot;
}
namespace {
RelabelCache& relabel_cache() {
  static RelabelCache* c = new RelabelCache();
  return *c;
}
bool relabel_key_matches(const RelabelCache& c, const DevMat& D) {
  return c.val == D.val.p && c.serial == dev_alloc_serial(D.val.p) && c.serial != 0 && c.epoch == value_epoch() && c.nnz == D.nnz &&
         c.cols == D.cols;
}
__global__ void k_invert_perm(const int32_t* __restrict__ newpos, int n, int32_t* __restrict__ lab) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) lab[newpos[i]] = i;
}
}  // namespace

void relabel_giveup(const DevMat& D) {   // a step on the relabelled form was refused: do not enter it again for this operand
  RelabelCache& c = relabel_cache();
  if (relabel_key_matches(c, D)) c.usable = false;
}

namespace {
const int32_t* g_scope_labels = nullptr;
}
void set_scope_labels(const int32_t* lab) { g_scope_labels = lab; }
const int32_t* scope_labels() { return g_scope_labels; }

void drop_operand_caches() {   // the device memory kept between solves (expanded D, D in the recovered order)
  drop_thin_transposes();
  RelabelCache& c = relabel_cache();   // (the order itself -- 8 bytes per column -- and its pattern fingerprint stay)
  c.Dr = DevMat();
  c.val = nullptr;
  c.serial = 0;
  c.usable = false;
  drop_dot_operand();
  drop_pending_exchange();
}
long long& band_searches() {
  static long long n = 0;
  return n;
}

const DevMat* relabelled_operand(const DevMat& D) {
  RelabelCache& c = relabel_cache();
  return (relabel_key_matches(c, D) && c.usable) ? &c.Dr : nullptr;
}

bool relabel_enter(DevMat& X, const DevMat& D) {
  if (!options().label_order || !options().fused_update || !options().loose_iterates) return false;
  if (X.cplx || D.cplx || X.loose() || X.expanded() || D.loose() || D.expanded() || X.rows != X.cols || D.rows != X.rows ||
      D.cols != X.cols || X.nnz == 0 || D.nnz == 0 || X.cols < 64)
    return false;
  const int n = X.cols;
  RelabelCache& c = relabel_cache();
  if (relabel_key_matches(c, D) && !c.usable) return false;   // (tried for this operand: no band in it)
  {   // run-like as it stands?  Then there is nothing to recover (the slab kernels take it directly)
    DevBuf<int32_t> f((size_t)n), l((size_t)n), cnt((size_t)n), span((size_t)n);
    DevBuf<int64_t> pre((size_t)n + 1);
    DevBuf<unsigned long long> zc(1);
    zc.zero();
    hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(X), f.p, l.p, cnt.p);
    hipLaunchKernelGGL(k_span_of, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f.p, l.p, span.p, n);
    scan_async<int32_t>(span.p, pre.p, (int64_t)n);
    if (X.zero_free != 1)
      hipLaunchKernelGGL(k_count_zero_values, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(X), zc.p);
    int64_t tot = 0;
    unsigned long long hz = 0;
    ScalarFetch ft;
    ft.add(pre.p + n, 1, &tot);
    ft.add(zc.p, 1, &hz);
    ft.run();
    if ((double)tot <= 2.0 * (double)X.nnz) return false;
    if (hz != 0) return false;   // (stored zeros: the slab form cannot tell them from holes)
    X.zero_free = 1;
  }
  if (!relabel_key_matches(c, D)) {
    const unsigned long long fp = pattern_fingerprint(D);
    const bool same_pattern = c.searched && c.fingerprint == fp && c.cols == D.cols && c.nnz == D.nnz;
    const bool had_order = same_pattern && c.newpos.p != nullptr && (int64_t)c.newpos.n == (int64_t)n;
    c.val = D.val.p;
    c.serial = dev_alloc_serial(D.val.p);
    c.epoch = value_epoch();
    c.nnz = D.nnz;
    c.cols = D.cols;
    c.usable = false;
    c.Dr = DevMat();
    if (same_pattern && !had_order) return false;   // (this pattern was searched before: no band in it)
    if (!had_order) {
      c.fingerprint = fp;
      c.searched = true;
      c.newpos.release();
      int64_t bw = 0;
      DevBuf<int32_t> pos;
      band_searches() += 1;
      if (!find_band_order(D, pos, &bw)) return false;
      // worth it when the band holds the entries densely: rows per column of the band against entries per column
      if (bw > 700 || (double)(2 * bw + 1) > 3.0 * (double)D.nnz / (double)n) return false;
      c.newpos = std::move(pos);
      c.lab.alloc((size_t)n);
      hipLaunchKernelGGL(k_invert_perm, dim3(cdiv(n, 256)), dim3(256), 0, stream(), c.newpos.p, n, c.lab.p);
    }
    c.Dr = remap_general(D, c.newpos.p, c.newpos.p, n, 0, n, false);
    c.usable = true;
  }
  DevMat Xr = remap_general(X, c.newpos.p, c.newpos.p, n, 0, n, false);
  Xr.zero_free = 1;
  DevBuf<int32_t> lab((size_t)n);
  HIP_CHECK(hipMemcpyAsync(lab.p, c.lab.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
  if (!slab_from_csc(Xr, lab)) {
    c.usable = false;
    return false;
  }
  X = std::move(Xr);
  return true;
}

void row_range(const DevMat& A, int32_t* lo, int32_t* hi) {
  *lo = INT_MAX;
  *hi = -1;
  if (A.nnz == 0 || A.cols == 0) return;
  DevBuf<int> mm(2);
  int init[2] = {INT_MAX, -1};
  mm.upload(init, 2);
  hipLaunchKernelGGL(k_row_range, dim3(cdiv(A.cols, 256)), dim3(256), 0, stream(), view(A), mm.p);
  int h[2];
  mm.download(h, 2);
  *lo = h[0];
  *hi = h[1];
}

DevMat column_slice(const DevMat& A, int32_t c0, int32_t c1) {
  int64_t h[2] = {0, 0};
  HIP_CHECK(hipMemcpyAsync(&h[0], A.outer.p + c0, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
  HIP_CHECK(hipMemcpyAsync(&h[1], A.outer.p + c1, sizeof(int64_t), hipMemcpyDeviceToHost, stream()));
  sync_stream();
  DevMat R;
  R.alloc(A.rows, c1 - c0, A.cplx, h[1] - h[0]);
  hipLaunchKernelGGL(k_shift_outer, dim3(cdiv(c1 - c0 + 1, 256)), dim3(256), 0, stream(), A.outer.p + c0, R.outer.p, c1 - c0, -h[0]);
  if (R.nnz) {
    HIP_CHECK(hipMemcpyAsync(R.inner.p, A.inner.p + h[0], sizeof(int32_t) * (size_t)R.nnz, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(R.val.p, A.val.p + h[0] * (int64_t)A.wval(), sizeof(double) * (size_t)R.nnz * A.wval(), hipMemcpyDeviceToDevice, stream()));
  }
  return R;
}

namespace {
__global__ void k_mask_len(const int64_t* __restrict__ outer, int32_t* __restrict__ len, int cols, int col_offset, int block,
                           int slices, int slice) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  const bool keep = ((col_offset + j) / block) % slices == slice;
  len[j] = keep ? (int32_t)(outer[j + 1] - outer[j]) : 0;
}
template <typename T>
__global__ __launch_bounds__(256) void k_mask_copy(Csc A, const int64_t* __restrict__ dst_outer, int32_t* __restrict__ dst_inner,
                                                   T* __restrict__ dst_val) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int64_t d0 = dst_outer[j], n = dst_outer[j + 1] - d0, s0 = A.outer[j];
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  for (int64_t t = lane_id(); t < n; t += WAVE) {
    dst_inner[d0 + t] = A.inner[s0 + t];
    dst_val[d0 + t] = Av[s0 + t];
  }
}
}  // namespace

DevMat mask_columns(const DevMat& A, int32_t col_offset, int32_t block, int32_t slices, int32_t slice) {
  DevMat R;
  R.rows = A.rows;
  R.cols = A.cols;
  R.cplx = A.cplx;
  R.outer.alloc((size_t)A.cols + 1);
  if (A.cols == 0) {
    R.reset_empty(A.rows, 0, A.cplx);
    return R;
  }
  DevBuf<int32_t> len((size_t)A.cols);
  hipLaunchKernelGGL(k_mask_len, dim3(cdiv(A.cols, 256)), dim3(256), 0, stream(), A.outer.p, len.p, A.cols, col_offset, block,
                     slices, slice);
  scan_async<int32_t>(len.p, R.outer.p, (int64_t)A.cols);
  int64_t nnz = 0;
  {
    ScalarFetch f;
    f.add(R.outer.p + A.cols, 1, &nnz);
    f.run();
  }
  R.nnz = nnz;
  R.inner.alloc((size_t)nnz + kIndexSlack);
  R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_mask_copy<T>), dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), view(A), R.outer.p,
                       R.inner.p, reinterpret_cast<T*>(R.val.p));
  });
  return R;
}

DevMat concat_columns(const std::vector<const DevMat*>& parts) {
  if (parts.empty()) NTP_FATAL("concat_columns: no parts");
  int32_t cols = 0;
  int64_t nnz = 0;
  for (auto* p : parts) { cols += p->cols; nnz += p->nnz; }
  DevMat R;
  R.alloc(parts[0]->rows, cols, parts[0]->cplx, nnz);
  int32_t c = 0;
  int64_t z = 0;
  for (auto* p : parts) {
    hipLaunchKernelGGL(k_shift_outer, dim3(cdiv(p->cols + 1, 256)), dim3(256), 0, stream(), p->outer.p, R.outer.p + c, p->cols, z);
    if (p->nnz) {
      HIP_CHECK(hipMemcpyAsync(R.inner.p + z, p->inner.p, sizeof(int32_t) * (size_t)p->nnz, hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(R.val.p + z * (int64_t)R.wval(), p->val.p, sizeof(double) * (size_t)p->nnz * R.wval(), hipMemcpyDeviceToDevice, stream()));
    }
    c += p->cols;
    z += p->nnz;
  }
  return R;
}

DevMat from_triplets(const HostTriplets& t, int32_t rows, int32_t cols, int32_t col_offset) {
  // SortTripletList + ConstructMatrixFromTripletList (triplet_includes/SortTripletList.f90:20-67,
  // sparse_includes/ConstructMatrixFromTripletList.f90:17-27); host-side: the caller hands over
  // host memory (setup path)
  const size_t n = t.size();
  std::vector<size_t> order;
  order.reserve(n);
  for (size_t i = 0; i < n; ++i) {
    // a row outside the matrix would become an out-of-range LDS / register index in the kernels: refuse it here
    if (t.row[i] < 1 || t.row[i] > rows || t.col[i] < 1)
      NTP_FATAL("triplet " + std::to_string(i) + " (column " + std::to_string(t.col[i]) + ", row " + std::to_string(t.row[i]) +
                ") lies outside the " + std::to_string(rows) + "-row matrix");
    const int32_t c = t.col[i] - 1 - col_offset;
    if (c >= 0 && c < cols) order.push_back(i);
  }
  auto less = [&](size_t a, size_t b) {
    if (t.col[a] != t.col[b]) return t.col[a] < t.col[b];
    return t.row[a] < t.row[b];
  };
  if (!std::is_sorted(order.begin(), order.end(), less)) std::stable_sort(order.begin(), order.end(), less);
  const size_t m = order.size();
  const size_t w = t.cplx ? 2 : 1;
  std::vector<int64_t> outer((size_t)cols + 1, 0);
  std::vector<int32_t> inner(m);
  std::vector<double> val(m * w);
  for (size_t i = 0; i < m; ++i) {
    const size_t s = order[i];
    outer[(size_t)(t.col[s] - 1 - col_offset) + 1] += 1;
    inner[i] = t.row[s] - 1;
    for (size_t k = 0; k < w; ++k) val[i * w + k] = t.val[s * w + k];
  }
  for (int32_t j = 0; j < cols; ++j) outer[(size_t)j + 1] += outer[(size_t)j];
  DevMat R;
  R.alloc(rows, cols, t.cplx, (int64_t)m);
  R.outer.upload(outer.data(), outer.size());
  R.inner.upload(inner.data(), m);
  R.val.upload(val.data(), m * w);
  sync_stream();
  return R;
}

void to_triplets(const DevMat& A, int32_t col_offset, HostTriplets& out) {
  const size_t n = (size_t)A.nnz, w = A.wval();
  std::vector<int64_t> outer((size_t)A.cols + 1);
  out.cplx = A.cplx;
  out.col.resize(n);
  out.row.resize(n);
  out.val.resize(n * w);
  A.outer.download(outer.data(), outer.size());
  if (n) {
    A.inner.download(out.row.data(), n);
    A.val.download(out.val.data(), n * w);
  }
  for (int32_t j = 0; j < A.cols; ++j)
    for (int64_t p = outer[(size_t)j]; p < outer[(size_t)j + 1]; ++p) out.col[(size_t)p] = j + 1 + col_offset;
  for (size_t i = 0; i < n; ++i) out.row[i] += 1;
}


// ------------------------------------------------------------------------------------- slab algebra
// The vocabulary of the solver loops (product, B <- alpha A + beta B, copy, scale, dot, column norms) on matrices that
// STAY in slab form between the operations (SlabForm: dense column runs in aligned zero-padded slots): a product leaves
// its result where the tile kernel wrote it, the merges are passes over two runs per column, nothing is compacted to
// compressed columns and expanded again.  Real, unlabelled, square, one rank, FMA arithmetic (the MFMA tile kernel);
// every function refuses what it cannot take and leaves its operands as they were (the callers in psmatrix.cpp pack
// and take the general path).  A zero inside a run reads as "no entry" (DevMat::zero_free).
namespace {
// slot of the union of two runs per column (B may be absent)
__global__ void k_sa_span(const int32_t* __restrict__ fa, const int32_t* __restrict__ la, const int32_t* __restrict__ fb,
                          const int32_t* __restrict__ lb, int n, int al, int32_t* __restrict__ span) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int f = fa[j], l = la[j];
  if (fb) {
    const int f2 = fb[j], l2 = lb[j];
    if (l2 >= f2) {
      if (l < f) { f = f2; l = l2; }
      else { f = min(f, f2); l = max(l, l2); }
    }
  }
  span[j] = l >= f ? (l / al + 1) * al - f / al * al : 0;
}
// B <- alpha A + beta B by the AddSparseVectors rules (inc_decide; B's values scaled by beta first, as ScaleMatrix
// followed by IncrementMatrix), one wave per column, result in a fresh slot at base[j] (row r at base + r - a0, a0 =
// the multiple of al below the union's first row; pads and dropped rows zero).  HAVE_B false: a copy of A (alpha = 1).
// stat[0] |= 1 when a kept value is exactly zero (an unfiltered tail that underflowed: the slab form cannot hold it).
// (T = double2: complex runs of (re, im) pairs, offsets in elements, threshold on the modulus -- the complex sessions)
template <typename T, bool HAVE_B>
__global__ __launch_bounds__(256) void k_sa_axpby(int n, const int32_t* __restrict__ fa, const int32_t* __restrict__ la,
                                                  const int64_t* __restrict__ offa, const T* __restrict__ va,
                                                  const int32_t* __restrict__ fb, const int32_t* __restrict__ lb,
                                                  const int64_t* __restrict__ offb, const T* __restrict__ vb,
                                                  const int64_t* __restrict__ base, int al, double alpha, double beta, double thr,
                                                  T* __restrict__ out, int32_t* __restrict__ ofirst, int32_t* __restrict__ olast,
                                                  int32_t* __restrict__ ocount, int64_t* __restrict__ ooff,
                                                  unsigned long long* __restrict__ stat, int64_t bound) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int fA = fa[j], lA = la[j];
  int fB = INT_MAX, lB = -1;
  if (HAVE_B) { fB = fb[j]; lB = lb[j]; }
  const bool anyA = lA >= fA, anyB = lB >= fB;
  const int64_t slot = base[j];
  if (!anyA && !anyB) {
    if (lane == 0) { ofirst[j] = INT_MAX; olast[j] = -1; ocount[j] = 0; ooff[j] = slot; }
    return;
  }
  const int f = anyA ? (anyB ? min(fA, fB) : fA) : fB, l = anyA ? (anyB ? max(lA, lB) : lA) : lB;
  const int a0 = f / al * al, a1 = (l / al + 1) * al;
  // The union extent of two runs that lie far apart (an identity and a product with entries n / 2 rows from the
  // diagonal) is not bounded by the operands' slots: a column whose slot would end beyond the output buffer writes
  // nothing and flags the merge, which the host then refuses (the operands go back to compressed columns)
  if (slot + (int64_t)(a1 - a0) > bound) {
    if (lane == 0) { ofirst[j] = INT_MAX; olast[j] = -1; ocount[j] = 0; ooff[j] = slot; atomicOr(stat, 2ull); }
    return;
  }
  const int amax = anyA ? lA : -1, bmax = anyB ? lB : -1;
  const T* __restrict__ pa = anyA ? va + (offa[j] - fA) : va;
  const T* __restrict__ pb = (HAVE_B && anyB) ? vb + (offb[j] - fB) : va;
  T* __restrict__ dst = out + (slot - a0);
  int cnt = 0, kf = INT_MAX, kl = -1, zk = 0;
  for (int r = a0 + lane; r < a1; r += WAVE) {
    const T a = (anyA && r >= fA && r <= lA) ? pa[r] : Sc<T>::zero();
    const T b = (HAVE_B && anyB && r >= fB && r <= lB) ? pb[r] : Sc<T>::zero();
    const bool ha = !Sc<T>::is_zero(a), hb = !Sc<T>::is_zero(b);
    const T wa = Sc<T>::scale(alpha, a), bs = HAVE_B ? Sc<T>::scale(beta, b) : Sc<T>::zero();
    T o = Sc<T>::zero();
    bool keep = false;
    if (ha && hb) { o = Sc<T>::add(wa, bs); keep = Sc<T>::mag(o) > thr; }
    else if (ha) { o = wa; keep = (r > bmax) ? true : (Sc<T>::mag(wa) > thr); }
    else if (hb) { o = bs; keep = (r > amax) ? true : (Sc<T>::mag(bs) > thr); }
    dst[r] = keep ? o : Sc<T>::zero();
    zk |= (keep && Sc<T>::is_zero(o)) ? 1 : 0;
    cnt += keep ? 1 : 0;
    kf = min(kf, keep ? r : INT_MAX);
    kl = max(kl, keep ? r : -1);
  }
  cnt = (int)wave_sum_i64(cnt);
  kf = wave_min_i32(kf);
  kl = wave_max_i32(kl);
  if (__ballot(zk != 0) && lane == 0) atomicOr(stat, 1ull);
  if (lane == 0) {
    ofirst[j] = kf; olast[j] = kl; ocount[j] = cnt;
    ooff[j] = slot + (cnt ? kf - a0 : 0);
  }
}
// sum of a . b over the rows both columns hold, per column (x, 0) pairs for the deterministic two-level sum
__global__ __launch_bounds__(256) void k_sa_dot(int n, const int32_t* __restrict__ fa, const int32_t* __restrict__ la,
                                                const int64_t* __restrict__ offa, const double* __restrict__ va,
                                                const int32_t* __restrict__ fb, const int32_t* __restrict__ lb,
                                                const int64_t* __restrict__ offb, const double* __restrict__ vb,
                                                double* __restrict__ part) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int f = max(fa[j], fb[j]), l = min(la[j], lb[j]);
  double s = 0.0;
  if (l >= f) {
    const double* __restrict__ pa = va + (offa[j] - fa[j]);
    const double* __restrict__ pb = vb + (offb[j] - fb[j]);
    for (int r = f + lane; r <= l; r += WAVE) s = __dadd_rn(s, __dmul_rn(pa[r], pb[r]));
  }
  s = wave_sum_f64(s);
  if (lane == 0) { part[2 * (size_t)j] = s; part[2 * (size_t)j + 1] = 0.0; }
}
// mode 0: out0[j] = sum |v| of column j; mode 1: (diagonal - sum |off-diagonal|, diagonal + sum |off-diagonal|)
template <typename T>
__global__ __launch_bounds__(256) void k_sa_colstat(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                    const int64_t* __restrict__ off, const T* __restrict__ val, int col_offset,
                                                    int mode, double* __restrict__ out0, double* __restrict__ out1) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int f = first[j], l = last[j];
  double r = 0.0, d = 0.0;
  if (mode == 1) {   // (Gershgorin: the order-independent sum of k_colstat -- the zeros of a run add nothing)
    double rl = 0.0;
    if (l >= f) {
      const T* __restrict__ p = val + (off[j] - f);
      for (int i = f + lane; i <= l; i += WAVE) {
        const T v = p[i];
        if (i == j + col_offset) d = __dadd_rn(d, Sc<T>::re(v));   // (GershgorinBounds.f90: the real part of the diagonal)
        else dd_add(r, rl, Sc<T>::mag(v), 0.0);
      }
    }
    dd_wave_sum(r, rl);
    d = wave_sum_f64(d);
    if (lane == 0) { out0[j] = d - r; out1[j] = d + r; }
    return;
  }
  if (l >= f) {
    const T* __restrict__ p = val + (off[j] - f);
    for (int i = f + lane; i <= l; i += WAVE) r = __dadd_rn(r, Sc<T>::mag(p[i]));
  }
  r = wave_sum_f64(r);
  d = wave_sum_f64(d);
  if (lane == 0) {
    if (mode == 0) out0[j] = r;
    else { out0[j] = d - r; out1[j] = d + r; }
  }
}
template <typename T>
__global__ __launch_bounds__(256) void k_sa_scale(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                  const int64_t* __restrict__ off, T* __restrict__ val, double c) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int f = first[j], l = last[j];
  if (l < f) return;
  T* __restrict__ p = val + (off[j] - f);
  for (int i = f + lane_id(); i <= l; i += WAVE) p[i] = Sc<T>::scale(c, p[i]);
}
// sum of the entry counts of the columns (integers: any order gives the same total); out zeroed by the caller
__global__ __launch_bounds__(256) void k_sa_sum_i32(const int32_t* __restrict__ v, int n, long long* __restrict__ out) {
  __shared__ long long red[4];
  long long s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += v[i];
  s = wave_sum_i64(s);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(reinterpret_cast<unsigned long long*>(out), (unsigned long long)t);
  }
}
void sa_sum_counts(const int32_t* count, int n, DevBuf<long long>& tot) {
  tot.alloc(1);
  tot.zero();
  hipLaunchKernelGGL(k_sa_sum_i32, dim3(std::max(1, std::min(256, cdiv(n, 1024)))), dim3(256), 0, stream(), count, n, tot.p);
}
bool g_panels_ok = false;   // a slab session across ranks (psmatrix.cpp): operands are column panels, rows != columns
bool sa_operand(const DevMat& M) {
  return M.expanded() && !M.cplx && !M.slab->labelled() && (M.rows == M.cols || g_panels_ok);
}
}  // namespace
void slab_allow_panels(bool on) { g_panels_ok = on; }
bool slab_panels_ok() { return g_panels_ok; }
bool slab_plan_fits_tile(int max_kn, int max_w) { return max_w > 0 && spgemm_tile_fits(max_kn, max_w); }

bool slab_enter(DevMat& M) {
  if (M.expanded()) return sa_operand(M);
  if (M.blocked() || M.slab_hint < 0) return false;
  if (M.cplx || M.loose() || (M.rows != M.cols && !g_panels_ok) || M.nnz == 0 || (options().spgemm_fma != 0 && options().spgemm_fma != 1)) return false;
  const int n = M.cols;
  std::unique_ptr<SlabForm> f(new SlabForm());
  f->first.alloc((size_t)n); f->last.alloc((size_t)n); f->count.alloc((size_t)n); f->off.alloc((size_t)n + 1);
  DevBuf<int32_t> span((size_t)n);
  DevBuf<unsigned long long> zc(1);
  zc.zero();
  // (FMA arithmetic: aligned zero-padded slots for the MFMA tile kernel; unfused arithmetic: the runs packed back to back,
  // as the register-slab kernel reads them)
  const int al = options().spgemm_fma == 1 ? tile_expand_align() : 1;
  hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(M), f->first.p, f->last.p, f->count.p);
  hipLaunchKernelGGL(k_span_aligned, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f->first.p, f->last.p, span.p, n, al);
  scan_async<int32_t>(span.p, f->off.p, (int64_t)n);
  if (M.zero_free != 1)
    hipLaunchKernelGGL(k_count_zero_values, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(M), zc.p);
  int64_t tot = 0;
  unsigned long long hz = 0;
  {
    ScalarFetch ft;
    ft.add(f->off.p + n, 1, &tot);
    ft.add(zc.p, 1, &hz);
    ft.run();
  }
  const bool dbg = std::getenv("NTPOLY_AMD_DEBUG_SPGEMM") != nullptr;
  if (hz == 0) M.zero_free = 1;
  if ((double)tot > 2.0 * (double)M.nnz + 2.0 * al * (double)n) {   // (mostly holes: not run-like)
    if (dbg) std::fprintf(stderr, "[slab_enter] refused: slots %lld for %lld entries\n", (long long)tot, (long long)M.nnz);
    M.slab_hint = -1;
    return false;
  }
  f->val.alloc((size_t)tot + kIndexSlack);
  hipLaunchKernelGGL(k_aligned_offsets<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), f->first.p, f->last.p,
                     f->off.p, f->val.p, n, al);
  hipLaunchKernelGGL(k_slab_expand_a<double>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(M), f->first.p,
                     f->off.p, f->val.p);
  f->row_pad = al;
  f->slots = tot;
  DevMat R;
  R.rows = M.rows; R.cols = n; R.cplx = false; R.nnz = M.nnz; R.zero_free = hz == 0 ? 1 : 0;
  if (hz != 0) {   // (stored zeros: the slab form is a read-only view, the compressed columns stay -- SlabForm::origin)
    f->origin.reset(new DevMat(std::move(M)));
  }
  R.slab = std::move(f);
  M = std::move(R);
  return true;
}

// B <- alpha A + beta B (A == nullptr... see slab_clone); false: refused, B unchanged
static bool sa_axpby_impl(const DevMat* A, const DevMat& Bin, DevMat& Out, double alpha, double beta, double thr) {
  const DevMat& X = A ? *A : Bin;   // (clone: the one operand plays A)
  const bool cplx = X.cplx;         // (complex sessions: runs of (re, im) pairs, offsets in elements)
  const bool have_b = A != nullptr;
  const SlabForm& fa = *X.slab;
  const SlabForm* fb = have_b ? Bin.slab.get() : nullptr;
  const int n = X.cols;
  const int al = have_b ? std::max(fa.row_pad, fb->row_pad) : fa.row_pad;
  if (have_b && (fa.row_pad != fb->row_pad && (al % fa.row_pad != 0 || al % fb->row_pad != 0))) return false;
  const int64_t bound = fa.slots + (have_b ? fb->slots : 0) + 2LL * al * n;
  std::unique_ptr<SlabForm> fo(new SlabForm());
  fo->first.alloc((size_t)n); fo->last.alloc((size_t)n); fo->count.alloc((size_t)n); fo->off.alloc((size_t)n + 1);
  DevBuf<int32_t> span((size_t)n);
  DevBuf<int64_t> base((size_t)n + 1);
  DevBuf<unsigned long long> stat(2);
  stat.zero();
  hipLaunchKernelGGL(k_sa_span, dim3(cdiv(n, 256)), dim3(256), 0, stream(), fa.first.p, fa.last.p, have_b ? fb->first.p : nullptr,
                     have_b ? fb->last.p : nullptr, n, al, span.p);
  scan_async<int32_t>(span.p, base.p, (int64_t)n);
  fo->val.alloc(((size_t)bound + kIndexSlack) * (cplx ? 2 : 1));
  if (cplx) {
    const double2* va = reinterpret_cast<const double2*>(fa.val.p);
    const double2* vb = have_b ? reinterpret_cast<const double2*>(fb->val.p) : nullptr;
    double2* vo = reinterpret_cast<double2*>(fo->val.p);
    if (have_b)
      hipLaunchKernelGGL((k_sa_axpby<double2, true>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p,
                         va, fb->first.p, fb->last.p, fb->off.p, vb, base.p, al, alpha, beta, thr, vo, fo->first.p, fo->last.p, fo->count.p,
                         fo->off.p, stat.p, bound);
    else
      hipLaunchKernelGGL((k_sa_axpby<double2, false>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p,
                         va, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int64_t*)nullptr, (const double2*)nullptr, base.p, al, 1.0, 0.0,
                         0.0, vo, fo->first.p, fo->last.p, fo->count.p, fo->off.p, stat.p, bound);
  } else if (have_b)
    hipLaunchKernelGGL((k_sa_axpby<double, true>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p,
                       fa.val.p, fb->first.p, fb->last.p, fb->off.p, fb->val.p, base.p, al, alpha, beta, thr, fo->val.p, fo->first.p,
                       fo->last.p, fo->count.p, fo->off.p, stat.p, bound);
  else
    hipLaunchKernelGGL((k_sa_axpby<double, false>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p,
                       fa.val.p, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int64_t*)nullptr, (const double*)nullptr,
                       base.p, al, 1.0, 0.0, 0.0, fo->val.p, fo->first.p, fo->last.p, fo->count.p, fo->off.p, stat.p, bound);
  DevBuf<long long> tot;
  sa_sum_counts(fo->count.p, n, tot);
  int64_t nnz = 0, slots = 0;
  unsigned long long hs = 0;
  {
    ScalarFetch ft;
    ft.add(tot.p, 1, &nnz);
    ft.add(base.p + n, 1, &slots);
    ft.add(stat.p, 1, &hs);
    ft.run();
  }
  if (hs != 0) return false;
  fo->row_pad = al;
  fo->slots = slots;
  DevMat R;
  R.rows = X.rows; R.cols = n; R.cplx = cplx; R.nnz = nnz; R.zero_free = 1;
  R.slab = std::move(fo);
  Out = std::move(R);
  return true;
}

bool slab_axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold) {
  if (!sa_operand(A) || !sa_operand(B) || A.cols != B.cols || &A == &B) return false;
  if (alpha == 0.0 || beta == 0.0) return false;   // (ScaleMatrix by zero leaves stored zeros: not a slab)
  if (A.zero_free != 1 || B.zero_free != 1) return false;
  DevMat R;
  if (!sa_axpby_impl(&A, B, R, alpha, beta, threshold)) return false;
  value_epoch() += 1;
  B = std::move(R);
  return true;
}

// Out = alpha A + beta B, B left as it is (CopyMatrix(B, Out) followed by the merge above, without the copy)
bool slab_axpby_to(const DevMat& A, const DevMat& B, DevMat& Out, double alpha, double beta, double threshold) {
  if (!sa_operand(A) || !sa_operand(B) || A.cols != B.cols || &A == &B) return false;
  if (alpha == 0.0 || beta == 0.0 || A.zero_free != 1 || B.zero_free != 1) return false;
  DevMat R;
  if (!sa_axpby_impl(&A, B, R, alpha, beta, threshold)) return false;
  Out = std::move(R);
  return true;
}

bool slab_clone(const DevMat& A, DevMat& Out) {
  if (!sa_operand(A) || A.zero_free != 1) return false;
  return sa_axpby_impl(nullptr, A, Out, 1.0, 0.0, 0.0);
}

bool slab_scale(DevMat& A, double c) {
  if (!sa_operand(A) || c == 0.0 || A.slab->origin) return false;
  value_epoch() += 1;
  SlabForm& f = *A.slab;
  hipLaunchKernelGGL(k_sa_scale<double>, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.cols, f.first.p, f.last.p, f.off.p,
                     f.val.p, c);
  f.tiles.release();       // (the multiplier tiles are rebuilt from the runs when the matrix is next a right operand)
  f.tile_off.release();
  f.next_plan.reset();
  return true;
}

bool slab_dot(const DevMat& A, const DevMat& B, double out[2]) {
  if (!sa_operand(A) || !sa_operand(B) || A.cols != B.cols) return false;
  const int n = A.cols;
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  DevBuf<double> part((size_t)2 * n), res(2);
  hipLaunchKernelGGL(k_sa_dot, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p, fa.val.p,
                     fb.first.p, fb.last.p, fb.off.p, fb.val.p, part.p);
  reduce_sum2_async(part.p, n, res.p);
  unsigned long long h[2] = {0, 0};
  ScalarFetch ft;
  ft.add(res.p, 2, h);
  ft.run();
  std::memcpy(out, h, sizeof(h));
  return true;
}

bool slab_norm(const DevMat& A, double* out) {
  if (!sa_operand(A)) return false;
  const SlabForm& f = *A.slab;
  DevBuf<double> cs((size_t)A.cols);
  hipLaunchKernelGGL(k_sa_colstat<double>, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.cols, f.first.p, f.last.p, f.off.p,
                     f.val.p, 0, 0, cs.p, (double*)nullptr);
  *out = max_of(cs, (size_t)A.cols);
  return true;
}

bool slab_gershgorin(const DevMat& A, int32_t col_offset, double* mn, double* mx) {
  if (!sa_operand(A)) return false;
  const SlabForm& f = *A.slab;
  DevBuf<double> lo((size_t)A.cols), hi((size_t)A.cols), res(2);
  hipLaunchKernelGGL(k_sa_colstat<double>, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.cols, f.first.p, f.last.p, f.off.p,
                     f.val.p, col_offset, 1, lo.p, hi.p);
  launch_reduce_minmax(lo.p, hi.p, (int64_t)A.cols, res.p);
  double h[2];
  res.download(h, 2);
  *mn = h[0];
  *mx = h[1];
  return true;
}

namespace {
// multiplier tiles of the blocks of 16 columns from the columns' runs (the register-slab kernel reads its multipliers
// from them as scalars): tile of block b = rows kmin .. kmin + kn of its columns, row-major 16 wide, at tiles + boff[b]
// (an all-zero row pads odd k ranges)
__global__ __launch_bounds__(256) void k_sa_tiles(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                  const int64_t* __restrict__ off, const double* __restrict__ val,
                                                  const int32_t* __restrict__ blk_kmin, const int32_t* __restrict__ blk_kn,
                                                  const int64_t* __restrict__ boff, double* __restrict__ tiles, int nblocks) {
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int kmin = blk_kmin[b], kn = blk_kn[b], kn2 = (kn + 1) & ~1;
  const int c = threadIdx.x & 15, j = b * SLAB_J + c;
  const bool colv = j < n;
  const int f = colv ? first[j] : INT_MAX, l = colv ? last[j] : -1;
  const double* __restrict__ p = (colv && l >= f) ? val + (off[j] - f) : val;
  double* __restrict__ dst = tiles + boff[b];
  for (int idx = threadIdx.x; idx < kn2 * SLAB_J; idx += 256) {
    const int r = kmin + (idx >> 4);
    dst[idx] = (r >= f && r <= l) ? p[r] : 0.0;
  }
}
struct EmptyDot {   // "D" of the register-slab kernel's EPI 1 epilogue when only the pruned product is wanted: no entries
  DevBuf<int32_t> dmin, dmax;
  DevBuf<int64_t> doff;
  DevBuf<double> dexp;
  int n = 0;
};
const EmptyDot& empty_dot(int n) {
  static EmptyDot* e = new EmptyDot();
  if (e->n < n) {
    e->dmin.alloc((size_t)n); e->dmax.alloc((size_t)n); e->doff.alloc((size_t)n + 1); e->dexp.alloc(8);
    std::vector<int32_t> hmin((size_t)n, INT_MAX), hmax((size_t)n, -1);   // (once per dimension)
    e->dmin.upload(hmin.data(), (size_t)n);
    e->dmax.upload(hmax.data(), (size_t)n);
    sync_stream();
    e->doff.zero();
    e->dexp.zero();
    e->n = n;
  }
  return *e;
}

// Unfused arithmetic: C = alpha A B on the register-slab kernel (k_spgemm_slab with the epilogue that leaves the pruned
// product in slab form -- EPI 1, the "X * X" step of TRS2 -- against an empty D), operands and result in slab form with
// packed runs; the right operand's multiplier tiles are built from its runs when a fused kernel has not left them
bool slab_multiply_loop(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule) {
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  const int n = B.cols, snb = cdiv(n, SLAB_J);
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  auto give_up = [&]() {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  };
  SlabPlan P;
  // [flag 2 | product entries per block snb + 1 | (unused) snb | (dot, trace) per block 2 snb | plan statistics 24]
  DevBuf<int64_t> zwords((size_t)4 * snb + 4 + 24);
  zwords.zero();
  int64_t* fz_flag = zwords.p;
  int64_t* fz_pnnz = zwords.p + 2;
  double* fz_part = reinterpret_cast<double*>(zwords.p + 3 + 2 * (size_t)snb);
  unsigned long long* stats = reinterpret_cast<unsigned long long*>(zwords.p + 4 * (size_t)snb + 4);
  const bool have_tiles = fb.tiles.p != nullptr && (int64_t)fb.tile_off.n == (int64_t)snb + 1;
  DevBuf<int64_t> boff;
  launch_slab_plan(P, n, fb.first.p, fb.last.p, fa.first.p, fa.last.p, 0, stats, have_tiles ? nullptr : &boff);
  unsigned long long hs[3] = {0, 0, 0};
  int64_t btotal = 0;
  {
    ScalarFetch f;
    f.add(P.blk_toff.p + snb, 1, &P.total);
    f.add(stats + 16, 3, hs);
    if (!have_tiles) f.add(boff.p + snb, 1, &btotal);
    f.run();
  }
  P.max_w = (int)hs[0];
  P.max_kn = (int)hs[1];
  const int64_t max_w = P.max_w;
  if (max_w <= 0 || max_w > 8 * SLAB_SL * WAVE || (((int64_t)P.max_kn + 1) | 1) * SLAB_J * 8 > 128 * 1024) return give_up();
  if (!have_tiles) {
    SlabForm& mb = const_cast<SlabForm&>(fb);   // (a cache: the tiles are a function of the runs)
    mb.tiles.alloc((size_t)btotal + 16 * SLAB_J + kIndexSlack);
    hipLaunchKernelGGL(k_sa_tiles, dim3(xcd_grid(snb)), dim3(256), 0, stream(), n, fb.first.p, fb.last.p, fb.off.p, fb.val.p,
                       P.blk_kmin.p, P.blk_kn.p, boff.p, mb.tiles.p, snb);
    mb.tile_off = std::move(boff);
  }
  DevBuf<int64_t> tmpoff((size_t)n + 1);
  hipLaunchKernelGGL((k_slab_tmpoff<SLAB_J>), dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), n, P.blk_w.p, P.blk_toff.p, tmpoff.p);
  DevBuf<char> runs(((size_t)A.cols + 4) * sizeof(SlabRun));
  hipLaunchKernelGGL(k_slab_runs, dim3(cdiv(A.cols + 4, 256)), dim3(256), 0, stream(), fa.first.p, fa.last.p, fa.off.p,
                     reinterpret_cast<const char*>(fa.val.p), 8, reinterpret_cast<SlabRun*>(runs.p), A.cols);
  const size_t oslots = (size_t)P.total + kIndexSlack;
  std::unique_ptr<SlabForm> fo(new SlabForm());
  fo->first.alloc((size_t)n); fo->last.alloc((size_t)n); fo->count.alloc((size_t)n);
  fo->count.zero();
  fo->val.alloc(oslots);
  fo->tiles.alloc(oslots);
  const EmptyDot& ed = empty_dot(n);
  SlabFuseArgs fz;
  fz.dexp = ed.dexp.p; fz.doff = ed.doff.p; fz.dmin = ed.dmin.p; fz.dmax = ed.dmax.p;
  fz.ofirst = fo->first.p; fz.olast = fo->last.p; fz.tiles = fo->tiles.p;
  fz.part = fz_part; fz.pnnz = reinterpret_cast<long long*>(fz_pnnz); fz.flag = reinterpret_cast<int*>(fz_flag);
  fz.col_offset = 0;
  DevBuf<char> fz_args(sizeof(SlabFuseArgs));
  fz_args.upload(reinterpret_cast<const char*>(&fz), sizeof(SlabFuseArgs));
  const int dr = dense_rule ? 1 : 0;
  t_num.start();
  auto launch = [&](auto nw_tag, auto mode_tag) {
    constexpr int FNW = decltype(nw_tag)::value;
    hipLaunchKernelGGL((k_spgemm_slab<SLAB_J, SLAB_SL, FNW, decltype(mode_tag)::value, 1>), dim3(xcd_grid(snb)), dim3(FNW * WAVE), 0,
                       stream(), reinterpret_cast<const SlabRun*>(runs.p), fb.tiles.p, fb.tile_off.p, P.blk_kmin.p, P.blk_kn.p,
                       P.blk_lo.p, P.blk_w.p, P.blk_toff.p, (int32_t*)nullptr, fo->val.p, fo->count.p, alpha, threshold, dr, n, snb,
                       reinterpret_cast<const SlabFuseArgs*>(fz_args.p));
  };
  if (max_w > 6 * SLAB_SL * WAVE) launch(std::integral_constant<int, 8>{}, std::integral_constant<int, 0>{});
  else if (max_w > SLAB_NW * SLAB_SL * WAVE) launch(std::integral_constant<int, 6>{}, std::integral_constant<int, 0>{});
  else launch(std::integral_constant<int, SLAB_NW>{}, std::integral_constant<int, 8>{});
  t_num.stop();
  DevBuf<long long> tot;
  sa_sum_counts(fo->count.p, n, tot);
  int64_t nnz = 0, flagv = 0;
  {
    ScalarFetch f;
    f.add(tot.p, 1, &nnz);
    f.add(fz_flag, 1, &flagv);
    f.run();
  }
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  if ((int32_t)flagv != 0) return false;
  fo->off = std::move(tmpoff);
  fo->tile_off = std::move(P.blk_toff);
  fo->row_pad = 1;
  fo->slots = P.total;
  SpgemmStats st;
  st.nnz_a = A.nnz; st.nnz_b = B.nnz; st.nnz_c = nnz; st.slab = 1; st.tmp_entries = P.total;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.nnz_c += nnz;
  acc.alg_bytes += 12.0 * ((double)A.nnz + (double)B.nnz + (double)nnz) + 4.0 * ((double)A.cols + 2.0 * n + 3.0);
  DevMat R;
  R.rows = A.rows; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1;
  R.slab = std::move(fo);
  C = std::move(R);
  return true;
}
}  // namespace

// C = alpha A B with the threshold rule of the SpGEMM, operands and result in slab form: on the MFMA tile kernel in FMA
// arithmetic, on the register-slab kernel in unfused arithmetic
// Every reason slab_multiply has to decline a PANEL product (left halo, FMA arithmetic) once its plan is known -- ONE predicate,
// used by slab_multiply itself and by the caller before the ranks agree on the product's path (psmatrix.cpp panel_slab_multiply):
// a rank that says yes there cannot say no later, when the others are already inside the collectives that follow.
bool slab_multiply_takes_panel(const DevMat& A, const DevMat& B, int left_row_pad, int32_t ka, int32_t kb, const SlabPlan* plan) {
  if (options().spgemm_fma != 1 || kb <= ka) return false;
  if (!sa_operand(A) || !sa_operand(B) || options().spgemm_variant >= 0 || options().spgemm_force_bin > 0) return false;
  if (left_row_pad % 16 != 0) return false;
  const int snb = cdiv(B.cols, SLAB_J);
  if (!plan || plan->align != 16 * tile_rows() || (int64_t)plan->blk_lo.n != snb) return false;   // (a plan of its own would be a late decision)
  return plan->max_w > 0 && spgemm_tile_fits(plan->max_kn, plan->max_w);
}

bool slab_multiply(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule, const SlabHalo* left) {
  // left (a panel product, psmatrix.cpp): A and B are column panels of the distributed operands; the columns of the left
  // operand named by the rows of B's panel -- this rank's own and the halo -- are described by `left` (global column numbers)
  if (left && !slab_multiply_takes_panel(A, B, left->row_pad, left->ka, left->kb, left->plan)) return false;
  if (options().spgemm_fma == 0) {
    if (!sa_operand(A) || !sa_operand(B) || A.cols != B.rows || options().spgemm_variant >= 0 || options().spgemm_force_bin > 0 ||
        A.slab->row_pad != 1)
      return false;
    return slab_multiply_loop(A, B, C, alpha, threshold, dense_rule);
  }
  if (!sa_operand(A) || !sa_operand(B) || (!left && A.cols != B.rows) || options().spgemm_fma != 1 || options().spgemm_variant >= 0 ||
      options().spgemm_force_bin > 0) {
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM")) std::fprintf(stderr, "[slab_multiply] refused: operands / options\n");
    return false;
  }
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  const int a_pad = left ? left->row_pad : fa.row_pad;
  int trows = tile_rows();
  while (trows > 1 && a_pad % (16 * trows) != 0) trows >>= 1;
  if (a_pad % 16 != 0) return false;
  const int ka = left ? left->ka : 0, nka = left ? left->kb - left->ka : A.cols;
  const int32_t* afirst = left ? left->first - ka : fa.first.p;   // (global column numbers through biased pointers)
  const int32_t* alast = left ? left->last - ka : fa.last.p;
  const int n = B.cols, snb = cdiv(n, SLAB_J);
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  // ---- plan: windows and k ranges of the blocks; the sizes of the multiplier tiles when B has none yet (a panel product:
  // made by the caller from the gathered extents and read back with its exchange layout)
  SlabPlan own_plan;
  const int plan_align = 16 * tile_rows();
  const bool given_plan = left && left->plan && left->plan->align == plan_align && (int64_t)left->plan->blk_lo.n == snb;
  SlabPlan& P = given_plan ? *left->plan : own_plan;
  DevBuf<unsigned long long> stats(24);
  // (the multiplier tiles of B, when a fused step left them; otherwise the kernel reads the runs of B's columns)
  const bool have_tiles = fb.tiles.p != nullptr && (int64_t)fb.tile_off.n == (int64_t)snb + 1;
  if (!given_plan) {
    stats.zero();
    launch_slab_plan(P, n, fb.first.p, fb.last.p, afirst, alast, plan_align, stats.p);
    unsigned long long hs[3] = {0, 0, 0};
    ScalarFetch f;
    f.add(P.blk_toff.p + snb, 1, &P.total);
    f.add(stats.p + 16, 3, hs);
    f.run();
    P.max_w = (int)hs[0];
    P.max_kn = (int)hs[1];
  }
  auto give_up = [&]() {
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  };
  if (P.max_w <= 0 || !spgemm_tile_fits(P.max_kn, P.max_w)) {
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
      std::fprintf(stderr, "[slab_multiply] refused: window %d rows, k range %d\n", P.max_w, P.max_kn);
    return give_up();
  }
  const size_t oslots = (size_t)P.total + kIndexSlack;
  std::unique_ptr<SlabForm> fo(new SlabForm());
  fo->first.alloc((size_t)n); fo->last.alloc((size_t)n); fo->count.alloc((size_t)n); fo->off.alloc((size_t)n + 1);
  fo->count.zero();
  fo->val.alloc(oslots);
  // ---- a thin operand (an identity, the near-diagonal factor of a square-root loop): the gather kernels of spgemm_thin.hip
  // on the same plan and output slots -- a handful of products per entry instead of the whole k range of the block
  const int thin_mode = (options().thin_left == 0 || fa.labelled() || fb.labelled() || A.rows != A.cols || left) ? 0
                        : (A.nnz <= 8 * (int64_t)A.cols && B.nnz >= A.nnz)                                 ? 1
                        : (B.nnz <= 8 * (int64_t)n)                                                          ? 2
                                                                                                             : 0;
  if (thin_mode) {
    DevMat AT;
    if (thin_mode == 1) {
      DevMat Ap = packed_copy(A);
      AT = transpose(Ap);
    }
    DevBuf<int> tflag(2);
    tflag.zero();
    t_num.start();
    ThinSlabArgs ta;
    ta.blk_lo = P.blk_lo.p; ta.blk_w = P.blk_w.p; ta.blk_toff = P.blk_toff.p;
    ta.bfirst = fb.first.p; ta.blast = fb.last.p; ta.boff = fb.off.p; ta.bval = fb.val.p;
    ta.afirst = fa.first.p; ta.alast = fa.last.p; ta.aoff = fa.off.p; ta.aval = fa.val.p;
    if (thin_mode == 1) { ta.at_outer = AT.outer.p; ta.at_inner = AT.inner.p; ta.at_val = AT.val.p; }
    ta.out_val = fo->val.p; ta.count = fo->count.p; ta.ofirst = fo->first.p; ta.olast = fo->last.p; ta.ooff = fo->off.p;
    ta.alpha = alpha; ta.threshold = threshold; ta.dense_rule = dense_rule ? 1 : 0; ta.ncols = n; ta.nrows = A.rows; ta.flag = tflag.p;
    launch_thin_slab(ta, thin_mode == 1);
    t_num.stop();
    DevBuf<long long> tot;
    sa_sum_counts(fo->count.p, n, tot);
    int64_t nnz = 0, fl = 0;
    {
      ScalarFetch f;
      f.add(tot.p, 1, &nnz);
      f.add(tflag.p, 1, &fl);
      f.run();
    }
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
      std::fprintf(stderr, "[slab_multiply] thin %s: nnzA %lld nnzB %lld nnzC %lld slots %lld max_w %d max_kn %d flag %d\n", thin_mode == 1 ? "left" : "right",
                   (long long)A.nnz, (long long)B.nnz, (long long)nnz, (long long)P.total, P.max_w, P.max_kn, (int)fl);
    if ((int)(fl & 0xffffffffll) == 0) {
      t_all.stop();
      if (timing) {
        if (pending_timings().size() >= 4096) flush_spgemm_timers();
        pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
      }
      fo->row_pad = plan_align;
      fo->slots = P.total;
      SpgemmStats st;
      st.nnz_a = A.nnz; st.nnz_b = B.nnz; st.nnz_c = nnz; st.slab = 1; st.thin = 1; st.tmp_entries = P.total;
      last_spgemm_stats() = st;
      SpgemmAccum& acc = spgemm_accum();
      acc.calls += 1;
      acc.nnz_c += nnz;
      acc.alg_bytes += 12.0 * ((double)A.nnz + (double)B.nnz + (double)nnz) + 4.0 * ((double)A.cols + 2.0 * n + 3.0);
      DevMat R;
      R.rows = A.rows; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1;
      R.slab = std::move(fo);
      C = std::move(R);
      return true;
    }
    // (a column of the right operand lists more non-zeros than the kernel holds: the tile kernel below)
    fo->count.zero();
  }
  DevBuf<char> runs(((size_t)nka + 4) * sizeof(SlabRun));
  if (left)
    hipLaunchKernelGGL(k_slab_runs_addr, dim3(cdiv(nka + 4, 256)), dim3(256), 0, stream(), left->first, left->last, left->addr,
                       (unsigned long long)reinterpret_cast<uintptr_t>(fa.val.p), reinterpret_cast<SlabRun*>(runs.p), nka);
  else
    hipLaunchKernelGGL(k_slab_runs, dim3(cdiv(A.cols + 4, 256)), dim3(256), 0, stream(), fa.first.p, fa.last.p, fa.off.p,
                       reinterpret_cast<const char*>(fa.val.p), 8, reinterpret_cast<SlabRun*>(runs.p), A.cols);
  t_num.start();
  TileLaunch tl;
  tl.runs = reinterpret_cast<const SlabRun*>(runs.p) - ka;
  if (have_tiles) {
    tl.bblk = fb.tiles.p; tl.blk_boff = fb.tile_off.p;
  } else {
    tl.bblk = fb.val.p; tl.blk_boff = nullptr;
    tl.brun_first = fb.first.p; tl.brun_last = fb.last.p; tl.brun_off = fb.off.p; tl.brun_val = fb.val.p; tl.bbytes = fb.val.n * sizeof(double); tl.brun_pad = fb.row_pad;
  }
  tl.blk_kmin = P.blk_kmin.p; tl.blk_kn = P.blk_kn.p; tl.blk_lo = P.blk_lo.p;
  tl.blk_w = P.blk_w.p; tl.blk_toff = P.blk_toff.p; tl.out_val = fo->val.p; tl.count = fo->count.p;
  tl.ofirst = fo->first.p; tl.olast = fo->last.p; tl.ooff = fo->off.p; tl.otoff = nullptr;
  tl.alpha = alpha; tl.threshold = threshold; tl.dense_rule = (dense_rule ? 1 : 0) | 2; tl.ncols = n; tl.nblocks = snb;
  tl.max_kn = P.max_kn; tl.max_w = P.max_w; tl.epi = 0; tl.fz = nullptr; tl.rows = trows; tl.labelled = false;
  if (!left) { tl.abase = fa.val.p; tl.abytes = fa.val.n * sizeof(double); }   // (the runs of A in one buffer: 32-bit offsets)
  // the two-block geometry first (right operand by its runs); a pair of blocks that does not fit after all leaves a mark
  // that comes back with the entry count, and the product is repeated on k_spgemm_tile
  DevBuf<int> t2fail(2);
  bool used_tile2 = false;
  if (!have_tiles && !left && options().tile2 != 0 && !fa.no_tile2 && !fb.no_tile2) {
    t2fail.zero();
    used_tile2 = launch_spgemm_tile2(tl, t2fail.p);
  }
  if (!used_tile2) launch_spgemm_tile(tl);
  DevBuf<long long> tot;
  sa_sum_counts(fo->count.p, n, tot);
  int64_t nnz = 0, t2f = 0;
  {
    ScalarFetch f;
    f.add(tot.p, 1, &nnz);
    if (used_tile2) f.add(t2fail.p, 1, &t2f);
    if (left && left->on_fetch) left->on_fetch(f);
    f.run();
  }
  if (used_tile2 && (int)(t2f & 0xffffffffll) != 0) {
    tile2_counts()[1] += 1;
    A.slab->no_tile2 = true;     // (asked again with this operand: k_spgemm_tile at once)
    fo->count.zero();
    launch_spgemm_tile(tl);
    sa_sum_counts(fo->count.p, n, tot);
    ScalarFetch f;
    f.add(tot.p, 1, &nnz);
    f.run();
  } else if (used_tile2) {
    tile2_counts()[0] += 1;
  }
  t_num.stop();
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  fo->row_pad = 16 * trows;
  fo->slots = P.total;
  SpgemmStats st;
  st.nnz_a = A.nnz; st.nnz_b = B.nnz; st.nnz_c = nnz; st.slab = 1; st.tmp_entries = P.total;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.nnz_c += nnz;
  acc.alg_bytes += 12.0 * ((double)A.nnz + (double)B.nnz + (double)nnz) + 4.0 * ((double)A.cols + 2.0 * n + 3.0);
  DevMat R;
  R.rows = A.rows; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1;
  R.slab = std::move(fo);
  C = std::move(R);
  return true;
}


// ------------------------------------------------------------------ complex operands in slab form (a session that allows them:
// the SignFunction loop).  The same form -- a dense run of (re, im) pairs per column in a slot aligned to 16 rows -- produced and
// consumed by the complex MFMA tile kernel (spgemm_tile_c.hip): between two products of a loop nothing is expanded or packed.
namespace {
__global__ __launch_bounds__(256) void k_count_zero_values_c(Csc A, unsigned long long* __restrict__ out) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= A.cols) return;
  const int lane = lane_id();
  const double2* __restrict__ v = static_cast<const double2*>(A.val);
  int c = 0;
  for (int64_t p = A.outer[j] + lane, e = col_end(A, j); p < e; p += WAVE) c += (v[p].x == 0.0 && v[p].y == 0.0) ? 1 : 0;
  const unsigned long long m = __ballot(c != 0);
  if (m && lane == 0) atomicAdd(out, 1ull);
}
}  // namespace
bool sa_operand_c(const DevMat& M) {
  return M.expanded() && M.cplx && !M.slab->labelled() && !M.slab->origin && M.rows == M.cols && M.slab->row_pad % 16 == 0;
}
bool slab_enter_c(DevMat& M) {
  if (M.expanded()) return sa_operand_c(M);
  if (!M.cplx || M.blocked() || M.loose() || M.rows != M.cols || M.nnz == 0 || M.slab_hint < 0 || options().spgemm_fma != 1 ||
      options().complex_tile == 0)
    return false;
  const int n = M.cols, al = 16;
  std::unique_ptr<SlabForm> f(new SlabForm());
  f->first.alloc((size_t)n); f->last.alloc((size_t)n); f->count.alloc((size_t)n); f->off.alloc((size_t)n + 1);
  DevBuf<int32_t> span((size_t)n);
  DevBuf<unsigned long long> zc(1);
  zc.zero();
  hipLaunchKernelGGL(k_col_extent, dim3(cdiv(n, 256)), dim3(256), 0, stream(), view(M), f->first.p, f->last.p, f->count.p);
  hipLaunchKernelGGL(k_span_aligned, dim3(cdiv(n, 256)), dim3(256), 0, stream(), f->first.p, f->last.p, span.p, n, al);
  scan_async<int32_t>(span.p, f->off.p, (int64_t)n);
  if (M.zero_free != 1)
    hipLaunchKernelGGL(k_count_zero_values_c, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(M), zc.p);
  int64_t tot = 0;
  unsigned long long hz = 0;
  {
    ScalarFetch ft;
    ft.add(f->off.p + n, 1, &tot);
    ft.add(zc.p, 1, &hz);
    ft.run();
  }
  if (hz == 0) M.zero_free = 1;
  // (stored zeros would read as "no entry"; mostly holes: not run-like)
  if (hz != 0 || (double)tot > 2.0 * (double)M.nnz + 2.0 * al * (double)n) {
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
      std::fprintf(stderr, "[slab_enter_c] refused: slots %lld for %lld entries, columns with stored zeros %llu\n", (long long)tot, (long long)M.nnz, hz);
    M.slab_hint = -1;
    return false;
  }
  f->val.alloc(((size_t)tot + kIndexSlack) * 2);
  hipLaunchKernelGGL(k_aligned_offsets<double2>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), f->first.p, f->last.p,
                     f->off.p, reinterpret_cast<double2*>(f->val.p), n, al);
  hipLaunchKernelGGL(k_slab_expand_a<double2>, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(M), f->first.p,
                     f->off.p, reinterpret_cast<double2*>(f->val.p));
  f->row_pad = al;
  f->slots = tot;
  DevMat R;
  R.rows = M.rows; R.cols = n; R.cplx = true; R.nnz = M.nnz; R.zero_free = 1;
  R.slab = std::move(f);
  M = std::move(R);
  return true;
}

// C = alpha A B with the threshold rule of the SpGEMM, complex operands and result in slab form (the complex MFMA tile kernel:
// the tolerance mode of DESIGN.md section 4); false: not taken, C untouched
bool slab_multiply_c(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule) {
  if (!sa_operand_c(A) || !sa_operand_c(B) || A.cols != B.rows || options().spgemm_fma != 1 || options().complex_tile == 0 ||
      options().spgemm_variant >= 0 || options().spgemm_force_bin > 0) {
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM")) std::fprintf(stderr, "[slab_multiply_c] refused: operands / options\n");
    return false;
  }
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  const int n = B.cols, snb = cdiv(n, SLAB_CJ);
  const bool timing = options().time_kernels != 0;
  EventTimer t_all(timing), t_num(timing);
  t_all.start();
  DevBuf<int32_t> blk_lo(snb), blk_w(snb), blk_kmin(snb), blk_kn(snb);
  DevBuf<int64_t> bsz(snb), tsz(snb), blk_toff((size_t)snb + 1);
  DevBuf<unsigned long long> stats(24);
  stats.zero();
  hipLaunchKernelGGL((k_slab_plan<SLAB_CJ>), dim3(cdiv((int64_t)snb * WAVE, 256)), dim3(256), 0, stream(), n, fb.first.p, fb.last.p, fa.first.p,
                     fa.last.p, blk_lo.p, blk_w.p, blk_kmin.p, blk_kn.p, bsz.p, tsz.p, snb, 16);
  hipLaunchKernelGGL(k_slab_reduce, dim3(64), dim3(256), 0, stream(), blk_w.p, blk_kn.p, snb, (const int32_t*)nullptr, 0, stats.p);
  scan_async<int64_t>(tsz.p, blk_toff.p, (int64_t)snb);
  int64_t total = 0;
  unsigned long long hs[3] = {0, 0, 0};
  {
    ScalarFetch f;
    f.add(blk_toff.p + snb, 1, &total);
    f.add(stats.p + 16, 3, hs);
    f.run();
  }
  const int max_w = (int)hs[0], max_kn = (int)hs[1];
  if (max_w <= 0 || hs[0] > 16384 || !spgemm_tile_c_fits(max_kn, max_w)) {
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM")) std::fprintf(stderr, "[slab_multiply_c] refused: window %llu rows, k range %llu\n", hs[0], hs[1]);
    if (timing) {
      event_pool().push_back(t_all.a); event_pool().push_back(t_all.b);
      event_pool().push_back(t_num.a); event_pool().push_back(t_num.b);
    }
    return false;
  }
  DevBuf<char> runs(((size_t)A.cols + 4) * sizeof(SlabRun));
  hipLaunchKernelGGL(k_slab_runs, dim3(cdiv(A.cols + 4, 256)), dim3(256), 0, stream(), fa.first.p, fa.last.p, fa.off.p,
                     reinterpret_cast<const char*>(fa.val.p), 16, reinterpret_cast<SlabRun*>(runs.p), A.cols);
  std::unique_ptr<SlabForm> fo(new SlabForm());
  fo->first.alloc((size_t)n); fo->last.alloc((size_t)n); fo->count.alloc((size_t)n); fo->off.alloc((size_t)n + 1);
  fo->count.zero();
  fo->val.alloc(((size_t)total + kIndexSlack) * 2);
  t_num.start();
  TileLaunch tl;
  tl.runs = reinterpret_cast<const SlabRun*>(runs.p);
  tl.bblk = fb.val.p; tl.blk_boff = nullptr;
  tl.brun_first = fb.first.p; tl.brun_last = fb.last.p; tl.brun_off = fb.off.p; tl.brun_val = fb.val.p; tl.bbytes = fb.val.n * sizeof(double); tl.brun_pad = fb.row_pad;
  tl.blk_kmin = blk_kmin.p; tl.blk_kn = blk_kn.p; tl.blk_lo = blk_lo.p; tl.blk_w = blk_w.p; tl.blk_toff = blk_toff.p;
  tl.out_val = fo->val.p; tl.count = fo->count.p; tl.ofirst = fo->first.p; tl.olast = fo->last.p; tl.ooff = fo->off.p;
  tl.alpha = alpha; tl.threshold = threshold; tl.dense_rule = dense_rule ? 1 : 0; tl.ncols = n; tl.nblocks = snb;
  tl.max_kn = max_kn; tl.max_w = max_w; tl.epi = 0;
  launch_spgemm_tile_c(tl);
  t_num.stop();
  DevBuf<long long> tot;
  sa_sum_counts(fo->count.p, n, tot);
  int64_t nnz = 0;
  {
    ScalarFetch f;
    f.add(tot.p, 1, &nnz);
    f.run();
  }
  t_all.stop();
  if (timing) {
    if (pending_timings().size() >= 4096) flush_spgemm_timers();
    pending_timings().push_back(TimedCall{{t_all.a, t_all.b, t_num.a, t_num.b}});
  }
  fo->row_pad = 16;
  fo->slots = total;
  SpgemmStats st;
  st.nnz_a = A.nnz; st.nnz_b = B.nnz; st.nnz_c = nnz; st.slab = 1; st.tmp_entries = total;
  last_spgemm_stats() = st;
  SpgemmAccum& acc = spgemm_accum();
  acc.calls += 1;
  acc.nnz_c += nnz;
  acc.alg_bytes += 20.0 * ((double)A.nnz + (double)B.nnz + (double)nnz) + 4.0 * ((double)A.cols + 2.0 * n + 3.0);
  DevMat R;
  R.rows = A.rows; R.cols = n; R.cplx = true; R.nnz = nnz; R.zero_free = 1;
  R.slab = std::move(fo);
  C = std::move(R);
  return true;
}


// ---- the rest of the loops' vocabulary on complex matrices in slab form (complex sessions of the inverse, square-root and
// inverse-square-root loops: InverseSolversModule.F90:29-149, SquareRootSolversModule.F90:342-531): merge by the
// AddSparseVectors rules with the threshold on the modulus, copy, scaling by a real constant, the largest column sum of moduli
bool slab_axpby_c(const DevMat& A, DevMat& B, double alpha, double beta, double threshold) {
  if (!sa_operand_c(A) || !sa_operand_c(B) || A.cols != B.cols || &A == &B) return false;
  if (alpha == 0.0 || beta == 0.0 || A.zero_free != 1 || B.zero_free != 1) return false;
  DevMat R;
  if (!sa_axpby_impl(&A, B, R, alpha, beta, threshold)) return false;
  value_epoch() += 1;
  B = std::move(R);
  return true;
}
bool slab_axpby_to_c(const DevMat& A, const DevMat& B, DevMat& Out, double alpha, double beta, double threshold) {
  if (!sa_operand_c(A) || !sa_operand_c(B) || A.cols != B.cols || &A == &B) return false;
  if (alpha == 0.0 || beta == 0.0 || A.zero_free != 1 || B.zero_free != 1) return false;
  DevMat R;
  if (!sa_axpby_impl(&A, B, R, alpha, beta, threshold)) return false;
  Out = std::move(R);
  return true;
}
bool slab_clone_c(const DevMat& A, DevMat& Out) {
  if (!sa_operand_c(A) || A.zero_free != 1) return false;
  return sa_axpby_impl(nullptr, A, Out, 1.0, 0.0, 0.0);
}
bool slab_scale_c(DevMat& A, double c) {
  if (!sa_operand_c(A) || c == 0.0) return false;
  value_epoch() += 1;
  SlabForm& f = *A.slab;
  hipLaunchKernelGGL(k_sa_scale<double2>, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.cols, f.first.p, f.last.p, f.off.p,
                     reinterpret_cast<double2*>(f.val.p), c);
  return true;
}
bool slab_norm_c(const DevMat& A, double* out) {
  if (!sa_operand_c(A)) return false;
  const SlabForm& f = *A.slab;
  DevBuf<double> cs((size_t)A.cols);
  hipLaunchKernelGGL(k_sa_colstat<double2>, dim3(cdiv((int64_t)A.cols * WAVE, 256)), dim3(256), 0, stream(), A.cols, f.first.p, f.last.p, f.off.p,
                     reinterpret_cast<const double2*>(f.val.p), 0, 0, cs.p, (double*)nullptr);
  *out = max_of(cs, (size_t)A.cols);
  return true;
}

}  // namespace ntp
