// Bandwidth-reducing order of a symmetric sparsity pattern (Cuthill-McKee by level-synchronous breadth-first search on
// the device), for operands that arrive relabelled: a symmetric relabelling only renames entries, so a matrix whose
// hidden structure is a band can be brought back into run form, multiplied by the register-slab kernels and have its
// results renamed back -- provided the two places where the reference's arithmetic depends on the LABELS follow the
// original ones (the order of the k steps of a product, the "beyond the other column's last row" test of the merge;
// kernels.hip, label-ordered slab steps).  This file only finds the order.
//
// A search is ONE launch of a few co-resident workgroups (k_bfs_multi: a wave per frontier vertex, neighbours claimed with a
// compare-and-swap on their level, levels separated by a barrier over the grid); positions are handed out in discovery
// order.  The start vertex is pseudo-peripheral: the search is restarted from the least connected vertex of its last
// level while the searches keep getting deeper.  Unreached vertices (other components) start further searches.  The
// order inside the levels is then settled by barycenter rounds (key = mean position of a vertex' neighbourhood, one
// device sort per round), which make the order of a relabelled band exact.  A one-off cost per solve (N = 262 144,
// 201 per row: see profiles/README.md), outside the iteration.
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

__global__ void k_min_degree_all(const int64_t* __restrict__ outer, int n, unsigned long long* __restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long key = ~0ull;
  if (v < n) key = ((unsigned long long)(outer[v + 1] - outer[v]) << 32) | (unsigned)v;
  // (one atomic per wave)
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o, WAVE);
    key = other < key ? other : key;
  }
  if (lane_id() == 0 && key != ~0ull) atomicMin(out, key);
}

// first vertex that no search has reached yet
__global__ void k_first_unreached(const int32_t* __restrict__ dist, int n, int32_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && dist[i] < 0) atomicMin(out, i);
}

// largest |new row - new column| over the entries
__global__ __launch_bounds__(256) void k_bandwidth(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                   int n, const int32_t* __restrict__ pos, int32_t* __restrict__ out) {
  __shared__ int sw[4];
  const int lane = lane_id();
  int w = 0;
  for (int v = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE; v < n; v += gridDim.x * (blockDim.x / WAVE)) {
    const int pv = pos[v];
    for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) w = max(w, abs(pos[inner[p]] - pv));
  }
  w = wave_max_i32(w);
  if (lane == 0) sw[threadIdx.x / WAVE] = w;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(sw[0], sw[1]), max(sw[2], sw[3])));
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

// One whole breadth-first search in ONE launch of a few co-resident workgroups: levels are separated by a barrier over
// the grid (an atomic counter in memory), not by launches and host round trips -- a band of 262 144 columns has 2 600
// levels of a hundred vertices.  Inside a level the vertices are placed Cuthill-McKee fashion, by the position of their
// FIRST neighbour in the level before (key[], an atomicMin during the expansion), ties by vertex number: the same
// order from run to run, and one that already follows the band (the barycenter rounds afterwards only polish it;
// placing a level by vertex number alone -- tried -- costs them a fifth of the recovered bandwidth).  ctl: [0] barrier counter, [1..3] rotating level-size counters (level L appends to
// ctl[1 + L % 3]; the counter of level L + 1 is cleared during level L, two barriers after its last reader).
// out[0] = vertices reached, out[1] = levels, out[2] = the vertex of the last level with the fewest neighbours.
constexpr int kBfsBlocks = 8;
constexpr int kBfsLevelMax = 2048;   // vertices of a level ranked in LDS (24 KB)
__device__ inline void grid_barrier(unsigned* __restrict__ counter, unsigned nblocks, unsigned& epoch) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(counter, 1u);
    const unsigned want = nblocks * (epoch + 1);
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
    __threadfence();
  }
  __syncthreads();
  epoch += 1;
}
__global__ __launch_bounds__(1024) void k_bfs_multi(const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                    int start, int base, int32_t* __restrict__ dist, int32_t* __restrict__ pos,
                                                    int32_t* __restrict__ cur, int32_t* __restrict__ nxt, int32_t* __restrict__ key,
                                                    unsigned* __restrict__ ctl, unsigned long long* __restrict__ best,
                                                    long long* __restrict__ out) {
  const int lane = lane_id();
  const int gwave = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nwaves = gridDim.x * (blockDim.x / WAVE);
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
  unsigned epoch = 0;
  if (gtid == 0) {
    cur[0] = start;
    dist[start] = 0;
    pos[start] = base;
  }
  grid_barrier(ctl, gridDim.x, epoch);
  // every workgroup ranks the whole level in its own LDS (a hundred vertices: ten comparisons per thread) and so knows
  // the placed order of the frontier without reading what another workgroup wrote since the last barrier
  __shared__ unsigned long long lvl[kBfsLevelMax];
  __shared__ int srt[kBfsLevelMax];
  bool placed = false;   // the frontier's placed order is in srt (otherwise: arrival order, in cur)
  int reached = 1, level = 0, nf = 1;
  for (;;) {
    unsigned* cnt = ctl + 1 + level % 3;
    if (gtid == 0) __hip_atomic_store(ctl + 1 + (level + 1) % 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int fbase = base + reached - nf;   // position of the frontier's first vertex
    for (int w = gwave; w < nf; w += nwaves) {
      const int v = placed ? srt[w] : __hip_atomic_load(cur + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int pv = fbase + w;
      for (int64_t p = outer[v] + lane, e = outer[v + 1]; p < e; p += WAVE) {
        const int u = inner[p];
        const int d = __hip_atomic_load(dist + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d < 0 && atomicCAS(&dist[u], -1, level + 1) == -1)
          __hip_atomic_store(nxt + atomicAdd(cnt, 1u), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d < 0 || d == level + 1) atomicMin(&key[u], pv);   // (a vertex of the next level, whoever claimed it)
      }
    }
    grid_barrier(ctl, gridDim.x, epoch);
    const int nl = (int)__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nl == 0) break;
    // positions inside the level: by (first neighbour in the level before, vertex number), not by the order in which
    // the atomics above happened to arrive -- the recovered order (and with it the path a borderline operand takes) is
    // the same from run to run.  A level wider than the LDS copy (no band in sight) keeps its arrival order.
    placed = nl <= kBfsLevelMax;
    if (placed) {
      for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        const int v = __hip_atomic_load(nxt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int kv = __hip_atomic_load(key + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lvl[i] = ((unsigned long long)(unsigned)kv << 32) | (unsigned)v;
      }
      __syncthreads();
      for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        const unsigned long long me = lvl[i];
        int rank = 0;
        for (int q = 0; q < nl; ++q) rank += lvl[q] < me ? 1 : 0;
        const int v = (int)(unsigned)(me & 0xffffffffull);
        srt[rank] = v;
        if (i % (int)gridDim.x == (int)blockIdx.x) pos[v] = base + reached + rank;
      }
      __syncthreads();
    } else {
      for (int i = gtid; i < nl; i += nthreads)
        pos[__hip_atomic_load(nxt + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)] = base + reached + i;
    }
    reached += nl;
    level += 1;
    nf = nl;
    int32_t* t = cur; cur = nxt; nxt = t;
  }
  // the least connected vertex of the last level (cur holds it, nf entries)
  for (int i = gtid; i < nf; i += nthreads) {
    const int v = __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    atomicMin(best, ((unsigned long long)(outer[v + 1] - outer[v]) << 32) | (unsigned)v);
  }
  grid_barrier(ctl, gridDim.x, epoch);
  if (gtid == 0) {
    out[0] = reached;
    out[1] = level;
    out[2] = (long long)(unsigned)(__hip_atomic_load(best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull);
  }
}

int bfs_from(const DevMat& A, int start, int base, DevBuf<int32_t>& dist, DevBuf<int32_t>& pos, DevBuf<int32_t>& cur,
             DevBuf<int32_t>& nxt, int* last_vertex, int* levels) {
  DevBuf<long long> out(3);
  DevBuf<int32_t> key((size_t)A.cols);
  hipLaunchKernelGGL(k_fill_i32v, dim3(cdiv(A.cols, 256)), dim3(256), 0, stream(), key.p, (int64_t)A.cols, INT_MAX);
  DevBuf<unsigned> ctl(4);
  DevBuf<unsigned long long> best(1);
  ctl.zero();
  const unsigned long long init = ~0ull;
  best.upload(&init, 1);
  {
    // The grid barrier needs all workgroups resident at once: eight workgroups of 1024 threads, which any partition of the
    // device holds.  A PLAIN launch: hipLaunchCooperativeKernel would make the runtime check the residency, but one
    // cooperative launch puts the process under exclusive, time-sliced scheduling for the rest of its life -- two ranks
    // sharing a GPU then waited ~21 ms at every transport operation (a relabelled solve on two ranks 63.8 -> 2.5 ms per
    // iteration, profiles/README.md 116)
    const int64_t* a_outer = A.outer.p;
    const int32_t* a_inner = A.inner.p;
    int32_t *p_dist = dist.p, *p_pos = pos.p, *p_cur = cur.p, *p_nxt = nxt.p, *p_key = key.p;
    unsigned* p_ctl = ctl.p;
    unsigned long long* p_best = best.p;
    long long* p_out = out.p;
    hipLaunchKernelGGL(k_bfs_multi, dim3(kBfsBlocks), dim3(1024), 0, stream(), a_outer, a_inner, start, base, p_dist, p_pos, p_cur, p_nxt, p_key, p_ctl, p_best, p_out);
  }
  long long h[3] = {0, 0, 0};
  ScalarFetch f;
  f.add(out.p, 3, h);
  f.run();
  *levels = (int)h[1];
  *last_vertex = (int)h[2];
  return (int)h[0];
}

}  // namespace

namespace {
bool band_order_attempt(const DevMat& A, int max_rounds, DevBuf<int32_t>& newpos, int64_t* bandwidth_out) {
  const int n = A.cols;
  DevBuf<int32_t> dist((size_t)n), pos((size_t)n), cur((size_t)n), nxt((size_t)n);
  auto reset = [&]() {
    hipLaunchKernelGGL(k_fill_i32v, dim3(cdiv(n, 256)), dim3(256), 0, stream(), dist.p, (int64_t)n, -1);
    hipLaunchKernelGGL(k_fill_i32v, dim3(cdiv(n, 256)), dim3(256), 0, stream(), pos.p, (int64_t)n, -1);
  };
  int done = 0, start = 0, guard = 0;
  reset();
  {   // first start: the least connected vertex (an end of a band has about half the neighbours of its middle)
    DevBuf<unsigned long long> best(1);
    const unsigned long long init = ~0ull;
    best.upload(&init, 1);
    hipLaunchKernelGGL(k_min_degree_all, dim3(cdiv(n, 256)), dim3(256), 0, stream(), A.outer.p, n, best.p);
    unsigned long long raw = 0;
    ScalarFetch f;
    f.add(best.p, 1, &raw);
    f.run();
    start = (int)(unsigned)(raw & 0xffffffffull);
    if (start < 0 || start >= n) start = 0;
  }
  while (done < n && guard++ < 64) {
    // pseudo-peripheral start inside this component: search, restart from the least connected vertex of the last
    // level, while the searches keep getting deeper (at most three times)
    int last = start, levels = 0, reached = 0;
    DevBuf<int32_t> dist_keep((size_t)n), pos_keep((size_t)n);
    HIP_CHECK(hipMemcpyAsync(dist_keep.p, dist.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(pos_keep.p, pos.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice, stream()));
    int from = start, prev_levels = -1;
    for (int round = 0; round < max_rounds; ++round) {
      reached = bfs_from(A, from, done, dist, pos, cur, nxt, &last, &levels);
      if (round + 1 == max_rounds || (round > 0 && levels <= prev_levels)) break;   // this search's order stands
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
    hipLaunchKernelGGL(k_bandwidth, dim3(std::min(cdiv((int64_t)n * WAVE, 256), 2048)), dim3(256), 0, stream(), A.outer.p, A.inner.p, n, p, bw.p);
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
}  // namespace

bool find_band_order(const DevMat& A, DevBuf<int32_t>& newpos, int64_t* bandwidth_out) {
  if (A.loose() || A.expanded() || A.rows != A.cols || A.cols < 2) return false;
  // one search from the least connected vertex usually is the end of a band already: restarts from the far end
  // (pseudo-peripheral iteration) only when the band found is not tight around the entries
  int64_t bw = 0;
  if (!band_order_attempt(A, 1, newpos, &bw)) return false;
  const double per_col = (double)A.nnz / (double)A.cols;
  if ((double)(2 * bw + 1) > 1.25 * per_col + 2.0) {
    DevBuf<int32_t> pos2;
    int64_t bw2 = 0;
    if (band_order_attempt(A, 4, pos2, &bw2) && bw2 < bw) {
      newpos = std::move(pos2);
      bw = bw2;
    }
  }
  if (bandwidth_out) *bandwidth_out = bw;
  return true;
}

}  // namespace ntp
