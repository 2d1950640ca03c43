// Bandwidth-reducing order of a symmetric sparsity pattern (Cuthill-McKee by level-synchronous breadth-first search on
// the device), for operands that arrive relabelled: a symmetric relabelling only renames entries, so a matrix whose
// hidden structure is a band can be brought back into run form, multiplied by the register-slab kernels and have its
// results renamed back -- provided the two places where the reference's arithmetic depends on the LABELS follow the
// original ones (the order of the k steps of a product, the "beyond the other column's last row" test of the merge;
// kernels.hip, label-ordered slab steps).  This file only finds the order.
//
// Levels are expanded one kernel at a time (one wave per frontier vertex, neighbours claimed with a compare-and-swap on
// their level), every level is then ordered by the smallest position among a vertex's neighbours in the previous level
// (which makes the order of a relabelled band exact) and the positions are handed out.  The start vertex is
// pseudo-peripheral: a search from vertex 0's component, restarted from a vertex of its last level.  Unreached
// vertices (other components) start further searches.  A few thousand tiny launches for a band of 262 144 columns:
// a one-off cost per solve, outside the iteration.
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <climits>
#include <vector>

#include "common.hpp"
#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

__global__ void k_fill_i32v(int32_t* __restrict__ p, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// one wave per frontier vertex: unreached neighbours join the next level
__global__ __launch_bounds__(256) void k_bfs_expand(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                    const int32_t* __restrict__ frontier, int nf, int level,
                                                    int32_t* __restrict__ dist, int32_t* __restrict__ next,
                                                    int32_t* __restrict__ next_count) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (w >= nf) return;
  const int lane = lane_id();
  const int v = frontier[w];
  for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) {
    const int u = inner[p];
    if (dist[u] < 0 && atomicCAS(&dist[u], -1, level + 1) == -1) next[atomicAdd(next_count, 1)] = u;
  }
}

// key of a vertex of the new level: the smallest position among its neighbours of the previous level
__global__ __launch_bounds__(256) void k_bfs_keys(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                  const int32_t* __restrict__ level_list, int nl, int level,
                                                  const int32_t* __restrict__ dist, const int32_t* __restrict__ pos,
                                                  unsigned long long* __restrict__ keys) {
  const int w = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (w >= nl) return;
  const int lane = lane_id();
  const int v = level_list[w];
  int best = INT_MAX;
  for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) {
    const int u = inner[p];
    if (dist[u] == level - 1) best = min(best, pos[u]);
  }
  best = wave_min_i32(best);
  if (lane == 0) keys[w] = ((unsigned long long)(unsigned)best << 32) | (unsigned)v;   // ties: by vertex number
}

// sort of one level's keys inside one workgroup (levels of a band are a bandwidth wide); larger levels: by chunks of
// the grid (odd-even merge passes would be better; levels that wide mean there is no band to recover anyway)
__global__ __launch_bounds__(1024) void k_sort_small(unsigned long long* __restrict__ keys, int n) {
  extern __shared__ unsigned long long sk[];
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += blockDim.x) sk[i] = i < n ? keys[i] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const unsigned long long a = sk[i], b = sk[l];
          if ((a > b) == up) { sk[i] = b; sk[l] = a; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) keys[i] = sk[i];
}

__global__ void k_assign_pos(const unsigned long long* __restrict__ keys, int nl, int base, int32_t* __restrict__ pos,
                             int32_t* __restrict__ level_list) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl) return;
  const int v = (int)(unsigned)(keys[i] & 0xffffffffull);
  pos[v] = base + i;
  level_list[i] = v;   // the next expansion walks the level in its final order
}

// the vertex of a level with the fewest neighbours (ends of a band have about half the neighbours of its middle)
__global__ void k_min_degree(const int64_t* __restrict__ outer, const int32_t* __restrict__ list, int nl,
                             unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl) return;
  const int v = list[i];
  atomicMin(out, ((unsigned long long)(outer[v + 1] - outer[v]) << 32) | (unsigned)v);
}

// first vertex that no search has reached yet
__global__ void k_first_unreached(const int32_t* __restrict__ dist, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && dist[i] < 0) atomicMin(out, i);
}

// largest |new row - new column| over the entries
__global__ __launch_bounds__(256) void k_bandwidth(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                   int n, const int32_t* __restrict__ pos, int32_t* __restrict__ out) {
  const int v = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (v >= n) return;
  const int lane = lane_id();
  int w = 0;
  const int pv = pos[v];
  for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) w = max(w, abs(pos[inner[p]] - pv));
  w = wave_max_i32(w);
  if (lane == 0) atomicMax(out, w);
}

// refinement: key of a vertex = mean position of its neighbours and itself (fixed point).  The breadth-first order
// is right up to a shuffle inside the levels (all vertices of a level tie on their first parent); the mean over a
// vertex' neighbourhood averages that shuffle out, and sorting by it restores the order of a band in a few rounds
__global__ __launch_bounds__(256) void k_barycenter(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner, int n,
                                                    const int32_t* __restrict__ pos, unsigned long long* __restrict__ keys,
                                                    int32_t* __restrict__ ids) {
  const int v = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (v >= n) return;
  const int lane = lane_id();
  long long sum = 0;
  for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) sum += pos[inner[p]];
  sum = wave_sum_i64(sum);
  if (lane == 0) {
    const long long deg = outer[v + 1] - outer[v];
    const long long tot = sum + pos[v];
    keys[v] = (unsigned long long)((tot << 16) / (deg + 1));
    ids[v] = v;
  }
}
__global__ void k_pos_from_order(const int32_t* __restrict__ order, int n, int32_t* __restrict__ pos) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pos[order[i]] = i;
}

constexpr int kMaxLevelSorted = 4096;   // levels up to this size are ordered (48 KB of LDS); wider ones keep discovery order

// one search from `start` over the unreached part; positions from `base` on.  Returns the number of vertices reached;
// *last_vertex = a vertex of the last level
int bfs_from(const DevMat& A, int start, int base, DevBuf<int32_t>& dist, DevBuf<int32_t>& pos, DevBuf<int32_t>& cur,
             DevBuf<int32_t>& nxt, DevBuf<int32_t>& counter, DevBuf<unsigned long long>& keys, int* last_vertex, int* levels) {
  int reached = 0, level = 0, nf = 1;
  {
    const int32_t h[2] = {start, 0};
    HIP_CHECK(hipMemcpyAsync(cur.p, &h[0], sizeof(int32_t), hipMemcpyHostToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(dist.p + start, &h[1], sizeof(int32_t), hipMemcpyHostToDevice, stream()));
    const int32_t b = base;
    HIP_CHECK(hipMemcpyAsync(pos.p + start, &b, sizeof(int32_t), hipMemcpyHostToDevice, stream()));
    sync_stream();
  }
  reached = 1;
  *last_vertex = start;
  while (nf > 0) {
    counter.zero();
    hipLaunchKernelGGL(k_bfs_expand, dim3(cdiv((int64_t)nf * WAVE, 256)), dim3(256), 0, stream(), A.outer.p, A.inner.p, cur.p, nf,
                       level, dist.p, nxt.p, counter.p);
    int32_t nl = 0;
    {
      ScalarFetch f;   // (8-byte words: the counter buffer holds two ints)
      long long raw = 0;
      f.add(counter.p, 1, &raw);
      f.run();
      nl = (int32_t)(raw & 0xffffffffll);
    }
    if (nl == 0) break;
    hipLaunchKernelGGL(k_bfs_keys, dim3(cdiv((int64_t)nl * WAVE, 256)), dim3(256), 0, stream(), A.outer.p, A.inner.p, nxt.p, nl,
                       level + 1, dist.p, pos.p, keys.p);
    if (nl <= kMaxLevelSorted) {
      int m = 1;
      while (m < nl) m <<= 1;
      hipLaunchKernelGGL(k_sort_small, dim3(1), dim3(std::min(1024, std::max(64, m / 2))), (size_t)m * 8, stream(), keys.p, nl);
    }
    hipLaunchKernelGGL(k_assign_pos, dim3(cdiv(nl, 256)), dim3(256), 0, stream(), keys.p, nl, base + reached, pos.p, nxt.p);
    reached += nl;
    std::swap(cur.p, nxt.p);
    std::swap(cur.n, nxt.n);
    nf = nl;
    level += 1;
  }
  {   // a vertex of the last level: the one with the fewest neighbours
    DevBuf<unsigned long long> best(1);
    const unsigned long long init = ~0ull;
    best.upload(&init, 1);
    hipLaunchKernelGGL(k_min_degree, dim3(cdiv(std::max(nf, 1), 256)), dim3(256), 0, stream(), A.outer.p, cur.p, nf, best.p);
    unsigned long long raw = 0;
    ScalarFetch f;
    f.add(best.p, 1, &raw);
    f.run();
    *last_vertex = nf > 0 ? (int)(unsigned)(raw & 0xffffffffull) : start;
  }
  *levels = level;
  return reached;
}

}  // namespace

bool find_band_order(const DevMat& A, DevBuf<int32_t>& newpos, int64_t* bandwidth_out) {
  if (A.loose() || A.expanded() || A.rows != A.cols || A.cols < 2) return false;
  const int n = A.cols;
  DevBuf<int32_t> dist((size_t)n), pos((size_t)n), cur((size_t)n), nxt((size_t)n), counter(2);
  DevBuf<unsigned long long> keys((size_t)n);
  auto reset = [&]() {
    hipLaunchKernelGGL(k_fill_i32v, dim3(cdiv(n, 256)), dim3(256), 0, stream(), dist.p, (int64_t)n, -1);
    hipLaunchKernelGGL(k_fill_i32v, dim3(cdiv(n, 256)), dim3(256), 0, stream(), pos.p, (int64_t)n, -1);
  };
  int done = 0, start = 0, guard = 0;
  reset();
  while (done < n && guard++ < 64) {
    // pseudo-peripheral start inside this component: search, restart from the least connected vertex of the last
    // level, while the searches keep getting deeper (at most three times)
    int last = start, levels = 0, reached = 0;
    DevBuf<int32_t> dist_keep((size_t)n), pos_keep((size_t)n);
    HIP_CHECK(hipMemcpyAsync(dist_keep.p, dist.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(pos_keep.p, pos.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    int from = start, prev_levels = -1;
    for (int round = 0; round < 4; ++round) {
      reached = bfs_from(A, from, done, dist, pos, cur, nxt, counter, keys, &last, &levels);
      if (round == 3 || (round > 0 && levels <= prev_levels)) break;   // this search's order stands
      prev_levels = levels;
      from = last;
      HIP_CHECK(hipMemcpyAsync(dist.p, dist_keep.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(pos.p, pos_keep.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    }
    done += reached;
    if (done >= n) break;
    DevBuf<int32_t> first(2);
    const int32_t init[2] = {INT_MAX, 0};
    first.upload(init, 2);
    hipLaunchKernelGGL(k_first_unreached, dim3(cdiv(n, 256)), dim3(256), 0, stream(), dist.p, n, first.p);
    long long raw = 0;
    ScalarFetch f;
    f.add(first.p, 1, &raw);
    f.run();
    start = (int32_t)(raw & 0xffffffffll);
    if (start < 0 || start >= n) break;
  }
  if (done < n) return false;   // (too many components: not a band)
  auto bandwidth_of = [&](const int32_t* p) {
    DevBuf<int32_t> bw(2);
    bw.zero();
    hipLaunchKernelGGL(k_bandwidth, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), A.outer.p, A.inner.p, n, p, bw.p);
    long long raw = 0;
    ScalarFetch f;
    f.add(bw.p, 1, &raw);
    f.run();
    return (int64_t)(raw & 0xffffffffll);
  };
  int64_t best = bandwidth_of(pos.p);
  {   // barycenter rounds while they help
    DevBuf<unsigned long long> k2((size_t)n), k2s((size_t)n);
    DevBuf<int32_t> ids((size_t)n), order((size_t)n), trial((size_t)n);
    size_t tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, k2.p, k2s.p, ids.p, order.p, (size_t)n, 0, 64, stream());
    DevBuf<char> tmp(tmp_bytes);
    HIP_CHECK(hipMemcpyAsync(trial.p, pos.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    for (int round = 0; round < 8; ++round) {
      hipLaunchKernelGGL(k_barycenter, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), A.outer.p, A.inner.p, n, trial.p,
                         k2.p, ids.p);
      if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, k2.p, k2s.p, ids.p, order.p, (size_t)n, 0, 64, stream()) != hipSuccess) break;
      hipLaunchKernelGGL(k_pos_from_order, dim3(cdiv(n, 256)), dim3(256), 0, stream(), order.p, n, trial.p);
      const int64_t bw = bandwidth_of(trial.p);
      if (bw < best) {
        best = bw;
        HIP_CHECK(hipMemcpyAsync(pos.p, trial.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
      } else if (bw > best + best / 4) {
        break;
      }
    }
  }
  if (bandwidth_out) *bandwidth_out = best;
  newpos = std::move(pos);
  return true;
}

}  // namespace ntp
