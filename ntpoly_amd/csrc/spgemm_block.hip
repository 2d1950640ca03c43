// Block-sparse SpGEMM on the FP64 matrix cores (gfx950, v_mfma_f64_16x16x4_f64) for real square operands WITHOUT run
// structure -- the Hamiltonian of a 3-D system, a band hidden under a relabelling: the operands the run-based kernels
// (register-slab, MFMA tile) cannot take and that the LDS-hash kernels multiply one scalar product at a time.
//
// The reference's arithmetic (MultiplyBlock.f90:9-36, PruneList.f90:8-38) fixes, per entry C(i, j), the chain of
// multiply-adds over ascending k.  A symmetric relabelling of the index set only renames entries -- the reference's own
// load balancer applies a random one (LoadBalancerModule.F90:14-52) -- so the engine is free to pick the labelling in
// which it multiplies: here one that makes the matrix BLOCK sparse.
//
//  1. BlockOrder: the indices are clustered into blocks of <= 16 with similar neighbourhoods by heavy-edge matching on
//     the graph of |A| (five levels, ties broken towards indices that share their high bits, so a lattice in natural
//     order comes out as regular bricks), blocks into super-blocks of <= 4, super-blocks into a nested order.
//  2. BlockForm: the matrix as dense 16 x 16 tiles (zero = no entry), 4 x 4 tiles to a super-tile with a 16-bit mask.
//  3. Symbolic phase on super-tiles (a bitmap per super-column), numeric phase: one wave per candidate output
//     super-tile (64 x 64 entries = 16 accumulator tiles in registers), intersecting the super-row of A with the
//     super-column of B and walking the matches in ascending k:  P(16 x 16) += A(16 x 4) B(4 x 16) per instruction.
//     Every tile of A loaded feeds up to four instructions groups, every tile of B up to four.
//  4. Prune in the epilogue; kept tiles go to a pool; the result is turned back into compressed columns under the
//     caller's labels.
//
// Arithmetic: the matrix instruction is a chain of fma() in ascending k (tools/micro/mfma_f64_probe.hip), instructions,
// blocks and super-blocks follow in ascending POSITION, zeros are exact no-ops: C(i, j) is the FMA chain over ascending
// position -- bit for bit what the reference's FP-contracted build computes on the matrix relabelled by `pos`
// (tests/test_gpu_block.py runs the CPU restatement of the reference on exactly that matrix), and within roundoff of the chain over ascending
// labels (the tolerance contract of label-ordered operands).
#include "spgemm_block.hpp"

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce_by_key.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "device_util.hpp"

namespace ntp {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

// in-tile index of in-block position o (an involution): four lane groups of the matrix instruction hold words
// 4 g + q of a 16-word line, the instruction q consumes k = 4 q + g
__host__ __device__ inline int phys(int o) { return 4 * (o & 3) + (o >> 2); }
// word of in-tile (row_t, col_t) inside a tile's 256: column major with the 16-byte chunks of a column (two rows each)
// XOR-swizzled by col_t / 2 -- the image of a tile in LDS is then its image in memory (a linear copy), and the
// transposed read of the right-operand role (16 lanes = 16 columns, the same two chunks of each) is bank-conflict free
__host__ __device__ inline int tile_word(int row_t, int col_t) {
  return col_t * 16 + ((((row_t >> 1) ^ (col_t >> 1)) & 7) << 1) + (row_t & 1);
}

bool dbg() {
  static const bool d = std::getenv("NTPOLY_AMD_DEBUG_SPGEMM") != nullptr;
  return d;
}

// =====================================================================================================================
// 1. clustering
// =====================================================================================================================
// One level of heavy-edge matching on a graph in CSR form (off, nbr, w; level 0 = the matrix' own compressed columns
// with w = |value|).  A vertex picks its best free neighbour -- heaviest edge (quantised to 10 mantissa bits: sums
// that differ by roundoff tie), then the neighbour whose representative index shares the most high bits with its own
// (smallest xor: neighbours along the fastest-running coordinate of a lattice, 2 k with 2 k + 1), then the smallest
// representative -- among those with size(a) + size(b) <= cap; mutual picks are matched.  A few rounds per level.
__global__ __launch_bounds__(256) void k_match_pick(int nv, const int64_t* __restrict__ off, const int32_t* __restrict__ nbr,
                                                    const float* __restrict__ w, const int32_t* __restrict__ csize,
                                                    const int32_t* __restrict__ crep, const int32_t* __restrict__ mate, int cap,
                                                    int32_t* __restrict__ best) {
  const int a = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (a >= nv) return;
  const int lane = lane_id();
  if (mate[a] >= 0) {
    if (lane == 0) best[a] = -1;
    return;
  }
  const int sa = csize[a], ra = crep[a];
  unsigned long long k1 = 0, k2 = ~0ull;
  for (int64_t e = off[a] + lane; e < off[a + 1]; e += WAVE) {
    const int b = nbr[e];
    if (b == a || mate[b] >= 0 || sa + csize[b] > cap) continue;
    const unsigned wq = __float_as_uint(w[e]) >> 13;
    const unsigned x = (unsigned)(ra ^ crep[b]);
    const unsigned long long c1 = (1ull << 62) | ((unsigned long long)wq << 32) | (unsigned long long)(0xFFFFFFFFu - x);
    const unsigned long long c2 = ((unsigned long long)(unsigned)crep[b] << 32) | (unsigned)b;
    if (c1 > k1 || (c1 == k1 && c2 < k2)) { k1 = c1; k2 = c2; }
  }
  unsigned long long m1 = k1;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(m1, o, WAVE);
    m1 = t > m1 ? t : m1;
  }
  unsigned long long m2 = (k1 == m1) ? k2 : ~0ull;
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_xor(m2, o, WAVE);
    m2 = t < m2 ? t : m2;
  }
  if (lane == 0) best[a] = (m1 == 0) ? -1 : (int)(m2 & 0xFFFFFFFFull);
}
__global__ __launch_bounds__(256) void k_match_mutual(int nv, const int32_t* __restrict__ best, int32_t* __restrict__ mate) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nv) return;
  const int b = best[a];
  if (b >= 0 && best[b] == a) mate[a] = b;
}
__global__ __launch_bounds__(256) void k_match_roots(int nv, const int32_t* __restrict__ mate, int32_t* __restrict__ flag) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nv) return;
  flag[a] = (mate[a] < 0 || a < mate[a]) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_match_newid(int nv, const int32_t* __restrict__ mate, const int64_t* __restrict__ excl,
                                                     int32_t* __restrict__ newid, const int32_t* __restrict__ csize,
                                                     const int32_t* __restrict__ crep, int32_t* __restrict__ csize2,
                                                     int32_t* __restrict__ crep2) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= nv) return;
  const int root = (mate[a] < 0 || a < mate[a]) ? a : mate[a];
  const int id = (int)excl[root];
  newid[a] = id;
  atomicAdd(&csize2[id], csize[a]);
  atomicMin(&crep2[id], crep[a]);
}
// edges of the coarse graph: key = (new id of the source) << 32 | new id of the target; edges inside a cluster get a
// key behind every real one
__global__ __launch_bounds__(256) void k_coarse_keys(int nv, const int64_t* __restrict__ off, const int32_t* __restrict__ nbr,
                                                     const float* __restrict__ w, const int32_t* __restrict__ newid, int nv2,
                                                     unsigned long long* __restrict__ key, float* __restrict__ val) {
  const int a = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (a >= nv) return;
  const int lane = lane_id();
  const unsigned na = (unsigned)newid[a];
  for (int64_t e = off[a] + lane; e < off[a + 1]; e += WAVE) {
    const unsigned nb = (unsigned)newid[nbr[e]];
    key[e] = (na == nb) ? ((unsigned long long)(unsigned)nv2 << 32) : (((unsigned long long)na << 32) | nb);
    val[e] = w[e];
  }
}
// CSR of the coarse graph from its sorted unique keys (the group of in-cluster edges, if any, is the last one)
__global__ __launch_bounds__(256) void k_coarse_csr(int64_t nu, const unsigned long long* __restrict__ ukey, int nv2,
                                                    int64_t* __restrict__ off2, int32_t* __restrict__ nbr2) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > nu) return;
  // off2[v] = first edge whose source is >= v; thread i fills (source of edge i - 1, source of edge i]
  const int64_t prev = (i == 0) ? -1 : (int64_t)(ukey[i - 1] >> 32);
  int64_t cur = (i == nu) ? (int64_t)nv2 : (int64_t)(ukey[i] >> 32);
  if (cur > nv2) cur = nv2;
  for (int64_t v = std::max<int64_t>(prev + 1, 0); v <= cur && v <= nv2; ++v) off2[v] = i;
  if (i < nu) nbr2[i] = (int32_t)(ukey[i] & 0xFFFFFFFFull);
}
__global__ __launch_bounds__(256) void k_abs_f32(int64_t n, const double* __restrict__ v, float* __restrict__ w) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) w[i] = (float)fabs(v[i]);
}
__global__ __launch_bounds__(256) void k_fill_i32(int64_t n, int32_t* __restrict__ p, int32_t v, int32_t step) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v + (int32_t)i * step;
}

struct Graph {
  int32_t nv = 0;
  int64_t ne = 0;
  const int64_t* off = nullptr;
  const int32_t* nbr = nullptr;
  const float* w = nullptr;
  DevBuf<int64_t> off_own;
  DevBuf<int32_t> nbr_own;
  DevBuf<float> w_own;
};

inline int grid1(int64_t n) { return (int)std::max<int64_t>(1, (n + 255) / 256); }
inline int gridw(int64_t nwaves) { return (int)std::max<int64_t>(1, (nwaves * WAVE + 255) / 256); }

// caps of the matching levels: index clusters up to 16 (levels 0..5), then in units of blocks (2, 4, 4), then doubling
constexpr int kBlockLevels = 6, kSuperLevels = 3;
const int kBlockCaps[kBlockLevels] = {2, 4, 8, 16, 16, 16};
const int kSuperCaps[kSuperLevels] = {2, 4, 4};

unsigned long long block_pattern_fp(const DevMat& M);

std::shared_ptr<BlockOrder> build_block_order(const DevMat& M) {
  const int32_t n = M.cols;
  std::shared_ptr<BlockOrder> bo(new BlockOrder());
  bo->n = n;
  bo->built_from_nnz = M.nnz;
  bo->seed_fp = block_pattern_fp(M);
  Graph g;
  g.nv = n;
  g.ne = M.nnz;
  g.off = M.outer.p;
  g.nbr = M.inner.p;
  g.w_own.alloc((size_t)std::max<int64_t>(1, M.nnz));
  hipLaunchKernelGGL(k_abs_f32, dim3(grid1(M.nnz)), dim3(256), 0, stream(), M.nnz, M.val.p, g.w_own.p);
  g.w = g.w_own.p;
  DevBuf<int32_t> csize((size_t)n), crep((size_t)n);
  hipLaunchKernelGGL(k_fill_i32, dim3(grid1(n)), dim3(256), 0, stream(), (int64_t)n, csize.p, 1, 0);
  hipLaunchKernelGGL(k_fill_i32, dim3(grid1(n)), dim3(256), 0, stream(), (int64_t)n, crep.p, 0, 1);
  std::vector<std::vector<int32_t>> maps;   // maps[l][cluster of state l] = cluster of state l + 1
  int k16 = -1, k64 = -1;                   // states whose clusters are the blocks / the super-blocks
  int level = 0;
  while (true) {
    int cap;
    if (level < kBlockLevels) cap = kBlockCaps[level];
    else if (level < kBlockLevels + kSuperLevels) cap = kSuperCaps[level - kBlockLevels];
    else cap = 4 << std::min(24, level - kBlockLevels - kSuperLevels + 1);
    if (level == kBlockLevels) {   // (from here on a cluster's size counts blocks)
      k16 = level;
      hipLaunchKernelGGL(k_fill_i32, dim3(grid1(g.nv)), dim3(256), 0, stream(), (int64_t)g.nv, csize.p, 1, 0);
    }
    if (level == kBlockLevels + kSuperLevels) k64 = level;
    if (level >= kBlockLevels + kSuperLevels && (g.nv <= 1 || g.ne == 0 || level > 60)) break;
    const int nv = g.nv;
    DevBuf<int32_t> mate((size_t)nv), best((size_t)nv), flag((size_t)nv), newid((size_t)nv);
    DevBuf<int64_t> excl((size_t)nv + 1);
    HIP_CHECK(hipMemsetAsync(mate.p, 0xFF, sizeof(int32_t) * (size_t)nv, stream()));
    if (g.ne > 0) {
      for (int round = 0; round < 6; ++round) {
        hipLaunchKernelGGL(k_match_pick, dim3(gridw(nv)), dim3(256), 0, stream(), nv, g.off, g.nbr, g.w, csize.p, crep.p, mate.p, cap, best.p);
        hipLaunchKernelGGL(k_match_mutual, dim3(grid1(nv)), dim3(256), 0, stream(), nv, best.p, mate.p);
      }
    }
    hipLaunchKernelGGL(k_match_roots, dim3(grid1(nv)), dim3(256), 0, stream(), nv, mate.p, flag.p);
    scan_i32_async(flag.p, excl.p, (int64_t)nv);
    int64_t nv2_64 = 0;
    {
      ScalarFetch f;
      f.add(excl.p + nv, 1, &nv2_64);
      f.run();
    }
    const int nv2 = (int)nv2_64;
    DevBuf<int32_t> csize2((size_t)nv2), crep2((size_t)nv2);
    csize2.zero();
    HIP_CHECK(hipMemsetAsync(crep2.p, 0x7F, sizeof(int32_t) * (size_t)nv2, stream()));
    hipLaunchKernelGGL(k_match_newid, dim3(grid1(nv)), dim3(256), 0, stream(), nv, mate.p, excl.p, newid.p, csize.p, crep.p, csize2.p, crep2.p);
    maps.emplace_back((size_t)nv);
    HIP_CHECK(hipMemcpyAsync(maps.back().data(), newid.p, sizeof(int32_t) * (size_t)nv, hipMemcpyDeviceToHost, stream()));
    // coarse graph
    Graph g2;
    g2.nv = nv2;
    if (g.ne > 0 && nv2 > 1) {
      DevBuf<unsigned long long> key((size_t)g.ne), key_s((size_t)g.ne), ukey((size_t)g.ne);
      DevBuf<float> val((size_t)g.ne), val_s((size_t)g.ne), uval((size_t)g.ne);
      DevBuf<unsigned long long> ucount(1);
      hipLaunchKernelGGL(k_coarse_keys, dim3(gridw(nv)), dim3(256), 0, stream(), nv, g.off, g.nbr, g.w, newid.p, nv2, key.p, val.p);
      int bits = 1;
      while ((1ll << bits) <= (long long)nv2) ++bits;
      size_t tb = 0;
      HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, key.p, key_s.p, val.p, val_s.p, (size_t)g.ne, 0, 32 + bits, stream()));
      {
        DevBuf<char> tmp(tb);
        HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, key.p, key_s.p, val.p, val_s.p, (size_t)g.ne, 0, 32 + bits, stream()));
      }
      tb = 0;
      HIP_CHECK(rocprim::reduce_by_key(nullptr, tb, key_s.p, val_s.p, (size_t)g.ne, ukey.p, uval.p, ucount.p, rocprim::plus<float>(),
                                       rocprim::equal_to<unsigned long long>(), stream()));
      {
        DevBuf<char> tmp(tb);
        HIP_CHECK(rocprim::reduce_by_key(tmp.p, tb, key_s.p, val_s.p, (size_t)g.ne, ukey.p, uval.p, ucount.p, rocprim::plus<float>(),
                                         rocprim::equal_to<unsigned long long>(), stream()));
      }
      unsigned long long nu = 0, lastkey = 0;
      {
        ScalarFetch f;
        f.add(ucount.p, 1, &nu);
        f.run();
      }
      if (nu > 0) {
        ScalarFetch f;
        f.add(ukey.p + (nu - 1), 1, &lastkey);
        f.run();
        if ((lastkey >> 32) >= (unsigned long long)nv2) nu -= 1;   // (the in-cluster edges)
      }
      g2.ne = (int64_t)nu;
      g2.off_own.alloc((size_t)nv2 + 1);
      g2.nbr_own.alloc((size_t)std::max<unsigned long long>(1, nu));
      g2.w_own.alloc((size_t)std::max<unsigned long long>(1, nu));
      hipLaunchKernelGGL(k_coarse_csr, dim3(grid1((int64_t)nu + 1)), dim3(256), 0, stream(), (int64_t)nu, ukey.p, nv2, g2.off_own.p, g2.nbr_own.p);
      if (nu) HIP_CHECK(hipMemcpyAsync(g2.w_own.p, uval.p, sizeof(float) * (size_t)nu, hipMemcpyDeviceToDevice, stream()));
      sync_stream();   // (the temporaries of this level go out of scope; the allocator is stream ordered, the host vectors are not)
      g2.off = g2.off_own.p;
      g2.nbr = g2.nbr_own.p;
      g2.w = g2.w_own.p;
    } else {
      sync_stream();
      g2.ne = 0;
    }
    if (dbg()) std::fprintf(stderr, "[block order] level %d cap %d: %d -> %d clusters, %lld -> %lld edges\n", level, cap, nv, nv2, (long long)g.ne, (long long)g2.ne);
    const bool stalled = nv2 == nv;
    g = std::move(g2);
    csize = std::move(csize2);
    crep = std::move(crep2);
    ++level;
    if (stalled && level > kBlockLevels + kSuperLevels) break;
  }
  if (k64 < 0) k64 = kBlockLevels + kSuperLevels;
  // ---- positions on the host: depth-first order of the cluster tree (children in ascending id)
  const int T = (int)maps.size();   // states 0 .. T
  std::vector<int32_t> seq;         // clusters of the current state in order
  {
    const size_t top = T > 0 ? (size_t)(*std::max_element(maps[T - 1].begin(), maps[T - 1].end())) + 1 : (size_t)n;
    seq.resize(top);
    for (size_t i = 0; i < top; ++i) seq[i] = (int32_t)i;
  }
  for (int l = T - 1; l >= 0; --l) {
    const std::vector<int32_t>& mp = maps[(size_t)l];
    const size_t np = seq.size();
    std::vector<int64_t> cnt(np + 1, 0);
    for (int32_t p : mp) cnt[(size_t)p + 1] += 1;
    for (size_t i = 0; i < np; ++i) cnt[i + 1] += cnt[i];
    std::vector<int32_t> child(mp.size());
    {
      std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
      for (size_t c = 0; c < mp.size(); ++c) child[(size_t)fill[(size_t)mp[c]]++] = (int32_t)c;   // ascending c inside a parent
    }
    std::vector<int32_t> next;
    next.reserve(mp.size());
    for (int32_t p : seq)
      for (int64_t e = cnt[(size_t)p]; e < cnt[(size_t)p + 1]; ++e) next.push_back(child[(size_t)e]);
    seq.swap(next);
  }
  // ancestors at the block and super-block states
  std::vector<int32_t> a16((size_t)n), a64((size_t)n);
  for (int32_t v = 0; v < n; ++v) {
    int32_t c = v;
    for (int l = 0; l < std::min(k16, T); ++l) c = maps[(size_t)l][(size_t)c];
    a16[(size_t)v] = c;
    for (int l = std::min(k16, T); l < std::min(k64, T); ++l) c = maps[(size_t)l][(size_t)c];
    a64[(size_t)v] = c;
  }
  std::vector<int32_t> pos((size_t)n);
  int32_t S = -1, bslot = 0, vslot = 0, cur16 = -1, cur64 = -1;
  for (int32_t i = 0; i < n; ++i) {
    const int32_t v = seq[(size_t)i];
    if (a64[(size_t)v] != cur64) { S += 1; bslot = 0; vslot = 0; cur64 = a64[(size_t)v]; cur16 = a16[(size_t)v]; }
    else if (a16[(size_t)v] != cur16) { bslot += 1; vslot = 0; cur16 = a16[(size_t)v]; }
    if (bslot > 3 || vslot > 15) NTP_FATAL("internal: block order: a cluster exceeds its capacity");
    pos[(size_t)v] = 64 * S + 16 * bslot + vslot;
    vslot += 1;
  }
  bo->ns = S + 1;
  std::vector<int32_t> lab((size_t)64 * (size_t)bo->ns, -1);
  for (int32_t v = 0; v < n; ++v) lab[(size_t)pos[(size_t)v]] = v;
  bo->pos.alloc((size_t)n);
  bo->lab.alloc(lab.size());
  bo->pos.upload(pos.data(), (size_t)n);
  bo->lab.upload(lab.data(), lab.size());
  sync_stream();
  static unsigned long long serial = 0;
  bo->serial = ++serial;
  if (dbg()) std::fprintf(stderr, "[block order] n %d: %d super-blocks (%d positions, %.1f %% padding), %d levels\n", n, bo->ns, 64 * bo->ns,
                          100.0 * (64.0 * bo->ns - n) / std::max(1.0, 64.0 * bo->ns), T);
  return bo;
}

// =====================================================================================================================
// 2. compressed columns <-> block form
// =====================================================================================================================
// A workgroup per super-column J (the <= 64 columns whose positions are 64 J .. 64 J + 63).  Pass 1 ORs, per super-row I,
// the 16-bit tile mask of the super-tile (I, J) in LDS (two masks to a word) and counts super-tiles and tiles; pass 2
// (after the scans) rebuilds the masks, lists the super-tiles in ascending I, zero-fills the column's tiles and
// scatters the values.  Stored zeros are not entries of the block form (a zero factor contributes exact zeros).
__device__ inline void bs_mark_column_masks(const Csc& M, const int32_t* __restrict__ pos, const int32_t* __restrict__ lab, int J,
                                            unsigned* __restrict__ m2) {
  const int wave = threadIdx.x / WAVE, lane = lane_id(), nw = blockDim.x / WAVE;
  for (int c = wave; c < 64; c += nw) {
    const int j = lab[64 * J + c];
    if (j < 0) continue;
    const double* __restrict__ v = static_cast<const double*>(M.val);
    for (int64_t e = M.outer[j] + lane; e < M.outer[j + 1]; e += WAVE) {
      if (v[e] == 0.0) continue;
      const int pr = pos[M.inner[e]];
      const int I = pr >> 6, bit = 4 * (c >> 4) + ((pr >> 4) & 3);
      atomicOr(&m2[I >> 1], 1u << (bit + 16 * (I & 1)));
    }
  }
}
__global__ __launch_bounds__(256) void k_bs_count(Csc M, const int32_t* __restrict__ pos, const int32_t* __restrict__ lab, int ns,
                                                  int32_t* __restrict__ cnt_st, int32_t* __restrict__ cnt_tile) {
  extern __shared__ unsigned m2[];
  __shared__ int red[2];
  const int J = blockIdx.x;
  for (int i = threadIdx.x; i < (ns + 1) / 2; i += blockDim.x) m2[i] = 0;
  if (threadIdx.x < 2) red[threadIdx.x] = 0;
  __syncthreads();
  bs_mark_column_masks(M, pos, lab, J, m2);
  __syncthreads();
  int st = 0, tl = 0;
  for (int i = threadIdx.x; i < (ns + 1) / 2; i += blockDim.x) {
    const unsigned w = m2[i];
    st += ((w & 0xFFFFu) != 0) + ((w >> 16) != 0);
    tl += __popc(w);
  }
  st = (int)wave_sum_i64(st);
  tl = (int)wave_sum_i64(tl);
  if (lane_id() == 0) { atomicAdd(&red[0], st); atomicAdd(&red[1], tl); }
  __syncthreads();
  if (threadIdx.x == 0) { cnt_st[J] = red[0]; cnt_tile[J] = red[1]; }
}
__global__ __launch_bounds__(256) void k_bs_fill(Csc M, const int32_t* __restrict__ pos, const int32_t* __restrict__ lab, int ns,
                                                 const int64_t* __restrict__ soff, const int64_t* __restrict__ tbase,
                                                 int32_t* __restrict__ srow, int32_t* __restrict__ smask, int64_t* __restrict__ sbase,
                                                 double* __restrict__ tiles) {
  extern __shared__ unsigned m2[];          // (ns + 1) / 2 words of masks
  __shared__ int wsum_st[8], wsum_tl[8];
  const int J = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid / WAVE, nw = blockDim.x / WAVE;
  const int nwords = (ns + 1) / 2;
  for (int i = tid; i < nwords; i += blockDim.x) m2[i] = 0;
  __syncthreads();
  bs_mark_column_masks(M, pos, lab, J, m2);
  __syncthreads();
  // ordered enumeration: thread t owns the words [t * per, (t + 1) * per)
  const int per = (nwords + blockDim.x - 1) / blockDim.x;
  const int w0 = min(nwords, tid * per), w1 = min(nwords, w0 + per);
  int st = 0, tl = 0;
  for (int i = w0; i < w1; ++i) {
    const unsigned w = m2[i];
    st += ((w & 0xFFFFu) != 0) + ((w >> 16) != 0);
    tl += __popc(w);
  }
  // exclusive scan over the workgroup (wave scan + wave totals)
  int xs = st, xt = tl;
  for (int o = 1; o < WAVE; o <<= 1) {
    const int a = __shfl_up(xs, o, WAVE), b = __shfl_up(xt, o, WAVE);
    if (lane >= o) { xs += a; xt += b; }
  }
  if (lane == WAVE - 1) { wsum_st[wave] = xs; wsum_tl[wave] = xt; }
  __syncthreads();
  int bs = 0, bt = 0;
  for (int k = 0; k < wave; ++k) { bs += wsum_st[k]; bt += wsum_tl[k]; }
  int ks = bs + xs - st, kt = bt + xt - tl;   // exclusive
  const int64_t s0 = soff[J], t0 = tbase[J];
  for (int i = w0; i < w1; ++i) {
    const unsigned w = m2[i];
    for (int h = 0; h < 2; ++h) {
      const unsigned mk = (w >> (16 * h)) & 0xFFFFu;
      const int I = 2 * i + h;
      if (mk && I < ns) {
        srow[s0 + ks] = I;
        smask[s0 + ks] = (int32_t)mk;
        sbase[s0 + ks] = t0 + kt;
        ks += 1;
        kt += __popc(mk);
      }
    }
  }
  (void)nw;
  // zero-fill the tiles of this super-column
  const int64_t t1 = tbase[J + 1];
  v2d* __restrict__ z = reinterpret_cast<v2d*>(tiles + t0 * 256);
  const int64_t nz2 = (t1 - t0) * 128;
  const v2d zero = {0.0, 0.0};
  for (int64_t i = tid; i < nz2; i += blockDim.x) z[i] = zero;
  __threadfence();
  __syncthreads();
  // scatter
  const int64_t s1 = soff[J + 1];
  for (int c = wave; c < 64; c += nw) {
    const int j = lab[64 * J + c];
    if (j < 0) continue;
    const double* __restrict__ v = static_cast<const double*>(M.val);
    for (int64_t e = M.outer[j] + lane; e < M.outer[j + 1]; e += WAVE) {
      const double x = v[e];
      if (x == 0.0) continue;
      const int pr = pos[M.inner[e]];
      const int I = pr >> 6, bit = 4 * (c >> 4) + ((pr >> 4) & 3);
      int64_t lo = s0, hi = s1;   // (srow of this super-column was written above by this workgroup)
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (srow[mid] < I) lo = mid + 1; else hi = mid;
      }
      const unsigned mk = (unsigned)smask[lo];
      const int64_t slot = sbase[lo] + __popc(mk & ((1u << bit) - 1u));
      tiles[slot * 256 + tile_word(phys(pr & 15), phys(c & 15))] = x;
    }
  }
}
// super-tiles by super-row: keys (I << 32 | J) of every super-tile, sorted -> roff / rcol / ridx
__global__ __launch_bounds__(256) void k_bs_row_keys(int ns, const int64_t* __restrict__ soff, const int32_t* __restrict__ srow,
                                                     unsigned long long* __restrict__ key, int32_t* __restrict__ idx) {
  const int J = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (J >= ns) return;
  for (int64_t t = soff[J] + lane_id(); t < soff[J + 1]; t += WAVE) {
    key[t] = ((unsigned long long)(unsigned)srow[t] << 32) | (unsigned)J;
    idx[t] = (int32_t)t;
  }
}
__global__ __launch_bounds__(256) void k_bs_row_csr(int64_t nst, const unsigned long long* __restrict__ key, int ns,
                                                    int64_t* __restrict__ roff, int32_t* __restrict__ rcol) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i > nst) return;
  const int64_t prev = (i == 0) ? -1 : (int64_t)(key[i - 1] >> 32);
  const int64_t cur = (i == nst) ? (int64_t)ns : (int64_t)(key[i] >> 32);
  for (int64_t v = prev + 1; v <= cur; ++v) roff[v] = i;
  if (i < nst) rcol[i] = (int32_t)(key[i] & 0xFFFFFFFFull);
}

size_t bs_count_lds(int ns) { return (size_t)((ns + 1) / 2) * 4; }
size_t bs_fill_lds(int ns) { return (size_t)((ns + 1) / 2) * 4; }
constexpr int kMaxSuperBlocks = 60000;   // (LDS of the conversion kernels: 2 bytes per super-row -- dimensions up to 3.8 M)

// compressed columns -> block form; false (nothing built) when the tiles would be emptier than min_fill
bool to_block(const DevMat& M, const std::shared_ptr<BlockOrder>& bo, BlockForm& F, double min_fill, double* fill_out) {
  const int ns = bo->ns;
  DevBuf<int32_t> cnt_st((size_t)ns), cnt_tile((size_t)ns);
  F.soff.alloc((size_t)ns + 1);
  DevBuf<int64_t> tbase((size_t)ns + 1);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_count), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_fill), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    attr_done = true;
  }
  const Csc Mv = view(M);
  hipLaunchKernelGGL(k_bs_count, dim3(ns), dim3(256), bs_count_lds(ns), stream(), Mv, bo->pos.p, bo->lab.p, ns, cnt_st.p, cnt_tile.p);
  scan_i32_async(cnt_st.p, F.soff.p, (int64_t)ns);
  scan_i32_async(cnt_tile.p, tbase.p, (int64_t)ns);
  int64_t nst = 0, nt = 0;
  {
    ScalarFetch f;
    f.add(F.soff.p + ns, 1, &nst);
    f.add(tbase.p + ns, 1, &nt);
    f.run();
  }
  const double fill = nt > 0 ? (double)M.nnz / (256.0 * (double)nt) : 0.0;
  if (fill_out) *fill_out = fill;
  if (nt == 0 || fill < min_fill) return false;
  F.order = bo;
  F.ns = ns;
  F.nst = nst;
  F.ntiles = nt;
  F.nnz = M.nnz;
  F.srow.alloc((size_t)nst);
  F.smask.alloc((size_t)nst);
  F.sbase.alloc((size_t)nst);
  F.tiles.alloc((size_t)nt * 256 + 512);
  hipLaunchKernelGGL(k_bs_fill, dim3(ns), dim3(256), bs_fill_lds(ns), stream(), Mv, bo->pos.p, bo->lab.p, ns, F.soff.p, tbase.p, F.srow.p,
                     F.smask.p, F.sbase.p, F.tiles.p);
  F.have_rows = false;
  return true;
}

void build_rows(BlockForm& F) {
  if (F.have_rows) return;
  const int64_t nst = F.nst;
  DevBuf<unsigned long long> key((size_t)nst), key_s((size_t)nst);
  DevBuf<int32_t> idx((size_t)nst);
  F.ridx.alloc((size_t)nst);
  F.rcol.alloc((size_t)nst);
  F.roff.alloc((size_t)F.ns + 1);
  hipLaunchKernelGGL(k_bs_row_keys, dim3(gridw(F.ns)), dim3(256), 0, stream(), F.ns, F.soff.p, F.srow.p, key.p, idx.p);
  size_t tb = 0;
  HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, key.p, key_s.p, idx.p, F.ridx.p, (size_t)nst, 0, 64, stream()));
  DevBuf<char> tmp(tb);
  HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, key.p, key_s.p, idx.p, F.ridx.p, (size_t)nst, 0, 64, stream()));
  hipLaunchKernelGGL(k_bs_row_csr, dim3(grid1(nst + 1)), dim3(256), 0, stream(), nst, key_s.p, F.ns, F.roff.p, F.rcol.p);
  F.have_rows = true;
}

// =====================================================================================================================
// 3. symbolic phase: candidate output super-tiles
// =====================================================================================================================
// super-column J of C can hold the super-rows of A's super-columns K named by B's super-column J: a bitmap of ns bits
template <bool FILL>
__global__ __launch_bounds__(256) void k_bs_symbolic(int ns, const int64_t* __restrict__ soffA, const int32_t* __restrict__ srowA,
                                                     const int64_t* __restrict__ soffB, const int32_t* __restrict__ srowB,
                                                     int32_t* __restrict__ ccount, const int64_t* __restrict__ coff,
                                                     int32_t* __restrict__ ci, int32_t* __restrict__ cj) {
  extern __shared__ unsigned bm[];
  __shared__ int wsum[8];
  __shared__ int total;
  const int J = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid / WAVE, nw = blockDim.x / WAVE;
  const int nwords = (ns + 31) / 32;
  for (int i = tid; i < nwords; i += blockDim.x) bm[i] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int64_t t = soffB[J] + wave; t < soffB[J + 1]; t += nw) {
    const int K = srowB[t];
    for (int64_t u = soffA[K] + lane; u < soffA[K + 1]; u += WAVE) {
      const int I = srowA[u];
      atomicOr(&bm[I >> 5], 1u << (I & 31));
    }
  }
  __syncthreads();
  const int per = (nwords + blockDim.x - 1) / blockDim.x;
  const int w0 = min(nwords, tid * per), w1 = min(nwords, w0 + per);
  int c = 0;
  for (int i = w0; i < w1; ++i) c += __popc(bm[i]);
  if (!FILL) {
    c = (int)wave_sum_i64(c);
    if (lane == 0) atomicAdd(&total, c);
    __syncthreads();
    if (tid == 0) ccount[J] = total;
    return;
  }
  int x = c;
  for (int o = 1; o < WAVE; o <<= 1) {
    const int a = __shfl_up(x, o, WAVE);
    if (lane >= o) x += a;
  }
  if (lane == WAVE - 1) wsum[wave] = x;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < wave; ++k) base += wsum[k];
  int64_t k = coff[J] + base + x - c;
  for (int i = w0; i < w1; ++i) {
    unsigned w = bm[i];
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      ci[k] = 32 * i + b;
      cj[k] = J;
      ++k;
    }
  }
}

// slice masks of the tiles (BlockForm::quads): a wave per super-tile, lane l looks at the words 4 l .. 4 l + 3 of every tile
// (statistics, optional: with ccountA = entries per column position of the LEFT operand the same pass counts the
// intermediate products of the multiply, sum over the entries (k, j) of this matrix of ccountA[k])
__global__ __launch_bounds__(256) void k_bs_quads(int64_t nst, const int32_t* __restrict__ smask, const int64_t* __restrict__ sbase,
                                                  const double* __restrict__ tiles, unsigned long long* __restrict__ quads,
                                                  const int32_t* __restrict__ srow, const int32_t* __restrict__ ccountA,
                                                  unsigned long long* __restrict__ prod_out) {
  const int64_t s = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  if (s >= nst) return;
  const int lane = lane_id();
  const int w0 = 4 * lane, col_t = w0 >> 4, ch = (w0 & 15) >> 1;
  const int cq = phys(col_t) >> 2;                       // column slice of this lane's words
  int rq[4], rpos[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    rpos[e] = phys(((((ch + (e >> 1)) ^ (col_t >> 1)) & 7) << 1) | (e & 1));   // in-block position of the row of word 4 l + e
    rq[e] = rpos[e] >> 2;
  }
  const int rbase = ccountA ? 64 * srow[s] : 0;
  long long prods = 0;
  const unsigned mk = (unsigned)smask[s];
  const double* __restrict__ base = tiles + sbase[s] * 256;
  unsigned long long colq = 0, rowq = 0;
  int rank = 0;
  for (int t = 0; t < 16; ++t) {
    if ((mk & (1u << t)) == 0) continue;
    const v4d v = *reinterpret_cast<const v4d*>(base + rank * 256 + w0);
    rank += 1;
    unsigned c4 = 0, r4 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] != 0.0) {
        c4 |= 1u << cq;
        r4 |= 1u << rq[e];
        if (ccountA) prods += ccountA[rbase + 16 * (t & 3) + rpos[e]];
      }
    for (int o = 32; o > 0; o >>= 1) { c4 |= __shfl_xor(c4, o, WAVE); r4 |= __shfl_xor(r4, o, WAVE); }
    colq |= (unsigned long long)c4 << (4 * t);
    rowq |= (unsigned long long)r4 << (4 * t);
  }
  if (lane == 0) { quads[2 * s] = colq; quads[2 * s + 1] = rowq; }
  if (ccountA) {
    prods = wave_sum_i64(prods);
    if (lane == 0 && prods) atomicAdd(&prod_out[s & 63], (unsigned long long)prods);
  }
}
// ccountA / prod_out (optional, statistics): see k_bs_quads; true when the products were counted by this pass
bool block_quads(BlockForm& F, const int32_t* ccountA = nullptr, unsigned long long* prod_out = nullptr) {
  if (F.have_quads) return false;
  F.quads.alloc((size_t)2 * std::max<int64_t>(1, F.nst));
  if (F.nst > 0)
    hipLaunchKernelGGL(k_bs_quads, dim3(gridw(F.nst)), dim3(256), 0, stream(), F.nst, F.smask.p, F.sbase.p, F.tiles.p, F.quads.p,
                       F.srow.p, ccountA, prod_out);
  F.have_quads = true;
  return ccountA != nullptr;
}

// =====================================================================================================================
// 4. numeric phase
// =====================================================================================================================
struct BsArgs {
  // left operand by super-rows, right operand by super-columns
  const int64_t* roffA; const int32_t* rcolA; const int32_t* ridxA;
  const int32_t* smaskA; const int64_t* sbaseA; const double* tilesA; const unsigned long long* quadsA;
  const int64_t* soffB; const int32_t* srowB; const int32_t* smaskB; const int64_t* sbaseB; const double* tilesB; const unsigned long long* quadsB;
  // candidates and results
  int64_t ncand;
  const int32_t *ci, *cj;
  const int32_t* order;  // [ncand] workgroup w computes candidate order[w]: Z-order over (I, J), so that the workgroups resident on an
                         // XCD at the same time share super-rows of A and super-columns of B in its L2
  int32_t* cmask;        // [ncand] kept tiles of the candidate (0: nothing kept)
  int64_t* cbase;        // [ncand] first tile slot in the pool
  int32_t* ccnt;         // [ncand] kept entries
  double* pool;          // tile pool of the result
  int64_t pool_tiles;    // its capacity
  unsigned long long* counters;   // [0] tiles handed out, [1] overflow flag, [2] matrix instructions issued
  double alpha, threshold;
  int dense_rule;
  int nwg;
  int ablate;            // experiment build (-DNTP_ABLATIONS) only: 1 no A loads, 2 no B loads, 3 no matrix instructions (WRONG results)
  // the matches of every candidate -- the K with super-tiles on both sides, ascending, as (index of A's super-tile, index of B's)
  // -- found by k_bs_match before the numeric kernel (option block_match; nullptr: the numeric kernel searches itself)
  const int64_t* moff;
  const int32_t* mcnt;
  const int2* mlist;
};

// The intersection of super-row I of A and super-column J of B for every candidate, ONCE and in a kernel of its own: a wave per
// candidate with a handful of registers (eight and more waves per SIMD cover the seven dependent loads of a binary search)
// instead of inside the numeric kernel, where three fat waves per SIMD wait for them and both waves of a candidate repeat
// them.  Matches are written in ascending K (the order of the FMA chain).
__global__ __launch_bounds__(256) void k_bs_match_bound(int64_t ncand, const int32_t* __restrict__ ci, const int32_t* __restrict__ cj,
                                                        const int64_t* __restrict__ roffA, const int64_t* __restrict__ soffB,
                                                        int32_t* __restrict__ bound) {
  const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (c >= ncand) return;
  const int I = ci[c], J = cj[c];
  bound[c] = (int32_t)min(roffA[I + 1] - roffA[I], soffB[J + 1] - soffB[J]);
}
__global__ __launch_bounds__(256) void k_bs_match(int64_t ncand, const int32_t* __restrict__ ci, const int32_t* __restrict__ cj,
                                                  const int64_t* __restrict__ roffA, const int32_t* __restrict__ rcolA,
                                                  const int32_t* __restrict__ ridxA, const int64_t* __restrict__ soffB,
                                                  const int32_t* __restrict__ srowB, const int64_t* __restrict__ moff,
                                                  int2* __restrict__ mlist, int32_t* __restrict__ mcnt) {
  const int64_t cand = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  if (cand >= ncand) return;
  const int lane = lane_id();
  const int I = ci[cand], J = cj[cand];
  const int64_t ra0 = roffA[I], ra1 = roffA[I + 1], cb0 = soffB[J], cb1 = soffB[J + 1];
  int2* __restrict__ out = mlist + moff[cand];
  int cnt = 0;
  for (int64_t base = ra0; base < ra1; base += WAVE) {
    const int64_t e = base + lane;
    const bool in = e < ra1;
    const int K = in ? rcolA[e] : INT_MAX;
    int64_t lo = cb0, hi = cb1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (srowB[mid] < K) lo = mid + 1; else hi = mid;
    }
    const bool found = in && lo < cb1 && srowB[lo] == K;
    const unsigned long long m = __ballot(found);
    if (found) out[cnt + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(ridxA[e], (int)lo);
    cnt += (int)__popcll(m);
  }
  if (lane == 0) mcnt[cand] = cnt;
}

__device__ inline v4d bs_zero4() { const v4d z = {0.0, 0.0, 0.0, 0.0}; return z; }

// One WORKGROUP of two waves per candidate super-tile (I, J); wave h owns the row blocks 2 h and 2 h + 1: acc[x][b] = tile
// (row block 2 h + x, column block b), 8 accumulator tiles = 64 VGPRs; no LDS in the loop, no barriers before the
// epilogue.  The super-row of A and the super-column of B are intersected 64 entries of A's list at a time (a lane per
// entry, binary search in B's list; both waves do the same walk); the matches are walked in ascending K.  Per match and
// per block kb of K with tiles on both sides: the tiles B(kb, b) that exist are loaded (lane (g, n) reads the rows
// 4 g .. 4 g + 3 of column n: two swizzled 16-byte chunks of the column's line), then the wave's tiles A(2 h + x, kb) (lane
// (g, m) reads in-tile row phys(m) of the columns 4 g + q: four full lines per instruction), and every pair issues its
// matrix instructions in ascending q -- those whose 4-wide slice of either tile is empty are skipped (BlockForm::quads).
// History (profiles/README.md, round 4): a wave per row block with the B tiles through the L1 or staged in LDS (44 L1
// line accesses per matrix instruction; a memory latency per match behind every barrier), one wave per candidate with 16
// accumulator tiles (216 VGPRs, two waves per SIMD: 29.5 ms on the 64^3 iterate), two waves per candidate (135 VGPRs,
// three waves per SIMD: 23.5 ms).
// (a workgroup is ONE candidate: two waves, each owning two of the four row blocks -- 8 accumulator tiles = 64 VGPRs instead
// of 128, so that three to four waves fit a SIMD and cover each other's load round trips; the two waves do the same walk
// and finish together.  A workgroup of four DIFFERENT candidates kept its finished waves' slots until the last was done:
// measured occupancy 1.1 waves per SIMD of the possible 2, profiles/r04_pmc_block_v6_wg4.txt)
// HV = waves per candidate: 2 (each wave owns NX = 2 row blocks; operands with sparse tiles, where the waves' load round
// trips pace the kernel) or 1 (one wave owns all four: every B tile is read once; operands with dense tiles -- a relabelled
// band at fill 0.7: 4.7 ms against 6.7 with two waves)
// UNF (unfused arithmetic, option spgemm_fma = 0: every product rounded, then added -- the reference's default x86-64 build):
// the same walk with the tile products on the vector units.  Lane (g, n) owns the in-tile rows 4 g .. 4 g + 3 of in-tile
// column n of every accumulator tile; per pair of tiles and per slice of four k positions that has entries on both sides it
// adds round(A(row, k) B(k, n)) for the slice's positions k in ASCENDING POSITION -- with the matches, blocks and slices
// walked in ascending position as well every entry is the reference's sum in ascending k of the relabelled matrix (zeros
// inside a tile are exact no-ops).  The column of the B tile sits in 16 registers (one load per k, the lanes of a column
// share their lines), the four rows of the A tile's column k are two 16-byte loads that the 16 lanes of a group share.
// LST: the candidates' matches come from k_bs_match (option block_match) -- a compile-time switch: as a run-time branch in the
// walk it cost the kernel 30 % (19.8 -> 25.7 ms on the 64^3 iterate, same registers, 15 more waits)
template <int HV, bool UNF = false, bool LST = false>
__global__ __launch_bounds__(64 * HV) __attribute__((amdgpu_waves_per_eu(UNF ? 3 : 1, 8))) void k_bs_numeric(const BsArgs a) {
  constexpr int NX = 4 / HV;
  __shared__ unsigned wmask[HV];
  __shared__ int wcount[HV];
  __shared__ long long slot_base;
  const int wg = xcd_block(a.nwg);
  if (wg < 0) return;
  const int lane = lane_id(), h = uni_i32(threadIdx.x / WAVE);     // h: row blocks NX h .. NX h + NX - 1
  const int64_t cand = uni_i32(a.order[wg]);
  const int I = uni_i32(a.ci[cand]), J = uni_i32(a.cj[cand]);
  v4d acc[NX][4];
#pragma unroll
  for (int x = 0; x < NX; ++x)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[x][b] = bs_zero4();
  const int64_t ra0 = uni_i64(a.roffA[I]), ra1 = uni_i64(a.roffA[I + 1]);
  const int64_t cb0 = uni_i64(a.soffB[J]), cb1 = uni_i64(a.soffB[J + 1]);
  const int g = lane >> 4, m = lane & 15;
  int aoffq[4];                                  // A role: word of (row phys(m), column 4 g + q)
#pragma unroll
  for (int q = 0; q < 4; ++q) aoffq[q] = tile_word(phys(m), 4 * g + q);
  const int blo = m * 16 + (((2 * g) ^ (m >> 1)) & 7) * 2;      // B role: rows 4 g, 4 g + 1 of column m; rows 4 g + 2, 4 g + 3 in the
  const int bhi = m * 16 + (((2 * g + 1) ^ (m >> 1)) & 7) * 2;  // neighbouring chunk
  unsigned nprod = 0;
  constexpr bool listed = LST;
  const int64_t m0 = listed ? uni_i64(a.moff[cand]) : 0;
  const int64_t w0 = listed ? 0 : ra0, w1 = listed ? (int64_t)uni_i32(a.mcnt[cand]) : ra1;
  for (int64_t base = w0; base < w1; base += WAVE) {
    const int64_t e = base + lane;
    const bool in = e < w1;
    int ia, ib;
    bool found;
    if constexpr (listed) {   // (the matches were found by k_bs_match: 64 of them per load)
      const int2 pr = in ? a.mlist[m0 + e] : make_int2(0, 0);
      ia = pr.x;
      ib = pr.y - (int)cb0;
      found = in;
    } else {
      const int K = in ? a.rcolA[e] : INT_MAX;
      ia = in ? a.ridxA[e] : 0;
      int64_t lo = cb0, hi = cb1;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a.srowB[mid] < K) lo = mid + 1; else hi = mid;
      }
      found = in && lo < cb1 && a.srowB[lo] == K;
      ib = (int)(lo - cb0);
    }
    unsigned long long match = __ballot(found);
    while (match) {
      const int l = __ffsll((long long)match) - 1;
      match &= match - 1;
      const int ia_s = readlane_i32(ia, l);
      const int64_t ib_s = cb0 + readlane_i32(ib, l);
      const unsigned mA = (unsigned)uni_i32(a.smaskA[ia_s]), mB = (unsigned)uni_i32(a.smaskB[ib_s]);
      if (((mA >> (NX * h)) & (NX == 2 ? 0x3333u : 0xFFFFu)) == 0) continue;            // no tile of A in this wave's row blocks
      const double* __restrict__ tA = a.tilesA + uni_i64(a.sbaseA[ia_s]) * 256;
      const double* __restrict__ tB = a.tilesB + uni_i64(a.sbaseB[ib_s]) * 256;
      const unsigned long long cqA = (unsigned long long)uni_i64((int64_t)a.quadsA[2 * (int64_t)ia_s]);       // column slices of A's tiles
      const unsigned long long rqB = (unsigned long long)uni_i64((int64_t)a.quadsB[2 * ib_s + 1]);            // row slices of B's tiles
#pragma unroll 1
      for (int kb = 0; kb < 4; ++kb) {
        const unsigned colA = (mA >> (4 * kb + NX * h)) & ((1u << NX) - 1u);     // bit x: tile A(NX h + x, kb)
        const unsigned rowB = (mB >> kb) & 0x1111u;              // bit 4 b: tile B(kb, b)
        if (colA == 0 || rowB == 0) continue;
        if constexpr (UNF) {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            if (!(rowB & (1u << (4 * b)))) continue;
            const double* __restrict__ pB = tB + __popc(mB & ((1u << (4 * b + kb)) - 1u)) * 256;
            const unsigned rb4 = (unsigned)(rqB >> (4 * (4 * b + kb))) & 15u;     // row slices of B(kb, b) with entries
            double bcol[16];      // bcol[k] = B(position k of the block, column n)
#pragma unroll
            for (int k = 0; k < 16; ++k) bcol[k] = pB[tile_word(phys(k), m)];
#pragma unroll
            for (int x = 0; x < NX; ++x) {
              if (!(colA & (1u << x))) continue;
              const unsigned m4 = rb4 & ((unsigned)(cqA >> (4 * (4 * kb + NX * h + x))) & 15u);
              if (m4 == 0u) continue;
              const double* __restrict__ pA = tA + __popc(mA & ((1u << (4 * kb + NX * h + x)) - 1u)) * 256;
#pragma unroll
              for (int qq = 0; qq < 4; ++qq) {
                if (!(m4 & (1u << qq))) continue;     // (slice qq = positions 4 qq .. 4 qq + 3: empty on one side)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                  const int k = 4 * qq + kk, c = phys(k);
                  const v2d a01 = *reinterpret_cast<const v2d*>(pA + c * 16 + ((((2 * g) ^ (c >> 1)) & 7) << 1));
                  const v2d a23 = *reinterpret_cast<const v2d*>(pA + c * 16 + ((((2 * g + 1) ^ (c >> 1)) & 7) << 1));
                  const double bv = bcol[k];
                  acc[x][b][0] = __dadd_rn(acc[x][b][0], __dmul_rn(a01[0], bv));
                  acc[x][b][1] = __dadd_rn(acc[x][b][1], __dmul_rn(a01[1], bv));
                  acc[x][b][2] = __dadd_rn(acc[x][b][2], __dmul_rn(a23[0], bv));
                  acc[x][b][3] = __dadd_rn(acc[x][b][3], __dmul_rn(a23[1], bv));
                }
                nprod += 1;
              }
            }
          }
          continue;
        }
        v2d b01[4], b23[4];   // (only the fragments of existing tiles are loaded -- and read)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (rowB & (1u << (4 * b))) {
            const double* __restrict__ pB = tB + __popc(mB & ((1u << (4 * b + kb)) - 1u)) * 256;
            b01[b] = *reinterpret_cast<const v2d*>(pB + blo);
            b23[b] = *reinterpret_cast<const v2d*>(pB + bhi);
          }
        }
        double af[NX][4];
#pragma unroll
        for (int x = 0; x < NX; ++x) {
          if (colA & (1u << x)) {
            const double* __restrict__ pA = tA + __popc(mA & ((1u << (4 * kb + NX * h + x)) - 1u)) * 256;
#pragma unroll
            for (int q = 0; q < 4; ++q) af[x][q] = pA[aoffq[q]];
          }
        }
#pragma unroll
        for (int x = 0; x < NX; ++x) {
          if (colA & (1u << x)) {
            const unsigned ca4 = (unsigned)(cqA >> (4 * (4 * kb + NX * h + x))) & 15u;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              if (rowB & (1u << (4 * b))) {
                // instruction q = column slice q of the A tile times row slice q of the B tile: skipped when either is empty
                const unsigned m4 = ca4 & ((unsigned)(rqB >> (4 * (4 * b + kb))) & 15u);
                if (m4 & 1u) acc[x][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][0], b01[b][0], acc[x][b], 0, 0, 0);
                if (m4 & 2u) acc[x][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][1], b01[b][1], acc[x][b], 0, 0, 0);
                if (m4 & 4u) acc[x][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][2], b23[b][0], acc[x][b], 0, 0, 0);
                if (m4 & 8u) acc[x][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[x][3], b23[b][1], acc[x][b], 0, 0, 0);
                nprod += __popc(m4);
              }
            }
          }
        }
      }
    }
  }
  // ---- epilogue: prune (PruneList.f90:22: strict >; the dense branch tests before the scaling), kept tiles to the pool
  const double alpha = a.alpha, thr = a.threshold;
  const bool dense = (a.dense_rule & 1) != 0;
  unsigned mine = 0;     // bit 4 b + NX h + x
  int cnt = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int x = 0; x < NX; ++x) {
      bool any = false;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = acc[x][b][r];
        const double sv = __dmul_rn(alpha, v);
        const bool keep = dense ? (fabs(v) > thr) : (fabs(sv) > thr);
        acc[x][b][r] = keep ? sv : 0.0;
        any |= keep;
        cnt += keep ? 1 : 0;
      }
      if (__ballot(any) != 0ull) mine |= 1u << (4 * b + NX * h + x);
    }
  cnt = (int)wave_sum_i64(cnt);
  if (lane == 0) { wmask[h] = mine; wcount[h] = cnt; }
  __syncthreads();
  unsigned mall = 0;
  int call = 0;
#pragma unroll
  for (int w = 0; w < HV; ++w) { mall |= wmask[w]; call += wcount[w]; }
  const unsigned mC = (unsigned)uni_i32((int)mall);
  const int nt = __popc(mC);
  if (threadIdx.x == 0) {
    long long sl = 0;
    if (nt) sl = (long long)atomicAdd(&a.counters[0], (unsigned long long)nt);
    slot_base = sl;
  }
  __syncthreads();
  const int64_t slot0 = uni_i64(slot_base);
  const bool ok = slot0 + nt <= a.pool_tiles;
  if (nt && ok) {
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int x = 0; x < NX; ++x) {
        const int bit = 4 * b + NX * h + x;
        if (mine & (1u << bit)) {
          // lane (g, n) holds the in-tile rows 4 g + r of column n: the chunks 2 g and 2 g + 1 of that column (swizzled)
          double* __restrict__ pt = a.pool + (slot0 + __popc(mC & ((1u << bit) - 1u))) * 256;
          v2d lo2, hi2;
          lo2[0] = acc[x][b][0]; lo2[1] = acc[x][b][1]; hi2[0] = acc[x][b][2]; hi2[1] = acc[x][b][3];
          *reinterpret_cast<v2d*>(pt + blo) = lo2;
          *reinterpret_cast<v2d*>(pt + bhi) = hi2;
        }
      }
  }
  if (threadIdx.x == 0) {
    if (nt && !ok) atomicOr(&a.counters[1], 1ull);
    a.cmask[cand] = ok ? (int32_t)mC : 0;
    a.cbase[cand] = slot0;
    a.ccnt[cand] = ok ? call : 0;
  }
  if (lane == 0 && nprod) atomicAdd(&a.counters[2], (unsigned long long)nprod);
}

// Z-order key of a candidate: the bits of its super-row and super-column interleaved
__device__ inline unsigned bs_spread16(unsigned v) {
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}
__global__ __launch_bounds__(256) void k_bs_zorder(int64_t ncand, const int32_t* __restrict__ ci, const int32_t* __restrict__ cj,
                                                   unsigned* __restrict__ key, int32_t* __restrict__ idx, int mode) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ncand) return;
  key[i] = mode == 0 ? (unsigned)i : mode == 2 ? (((unsigned)ci[i] << 16) | (unsigned)cj[i]) : (bs_spread16((unsigned)ci[i]) | (bs_spread16((unsigned)cj[i]) << 1));
  idx[i] = (int32_t)i;
}

// candidates that kept something -> the super-tiles of C (the candidates are ordered by super-column, then super-row)
__global__ __launch_bounds__(256) void k_bs_flag(int64_t ncand, const int32_t* __restrict__ cmask, int32_t* __restrict__ flag) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < ncand) flag[i] = cmask[i] != 0 ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_bs_compact(int64_t ncand, const int32_t* __restrict__ cmask, const int64_t* __restrict__ cbase,
                                                    const int32_t* __restrict__ ci, const int64_t* __restrict__ excl,
                                                    int32_t* __restrict__ srow, int32_t* __restrict__ smask, int64_t* __restrict__ sbase) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= ncand || cmask[i] == 0) return;
  const int64_t k = excl[i];
  srow[k] = ci[i];
  smask[k] = cmask[i];
  sbase[k] = cbase[i];
}
__global__ __launch_bounds__(256) void k_bs_soff(int ns, const int64_t* __restrict__ coff, const int64_t* __restrict__ excl,
                                                 int64_t* __restrict__ soff) {
  const int J = blockIdx.x * blockDim.x + threadIdx.x;
  if (J <= ns) soff[J] = excl[coff[J]];
}

// =====================================================================================================================
// 5. block form -> compressed columns under the caller's labels
// =====================================================================================================================
// A wave per column position: lane (a, i) looks at row i of tile (a, cb) of every super-tile of the super-column.
// FILL = false counts the entries (cnt[label]); FILL = true writes (row label, value) in position order at outer[label]
// (the rows are sorted by label afterwards).
template <bool FILL>
__global__ __launch_bounds__(256) void k_bs_unblock(int ns, const int64_t* __restrict__ soff, const int32_t* __restrict__ srow,
                                                    const int32_t* __restrict__ smask, const int64_t* __restrict__ sbase,
                                                    const double* __restrict__ tiles, const int32_t* __restrict__ lab,
                                                    int32_t* __restrict__ cnt, const int64_t* __restrict__ outer,
                                                    int32_t* __restrict__ inner, double* __restrict__ val) {
  const int pc = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (pc >= 64 * ns) return;
  const int j = lab[pc];
  if (j < 0) return;
  const int lane = lane_id(), J = pc >> 6, cb = (pc >> 4) & 3, a = lane >> 4, i = lane & 15;
  const int bit = 4 * cb + a;
  const int coloff = tile_word(i, phys(pc & 15));   // in-tile row i holds in-block position phys(i)
  int64_t w = FILL ? outer[j] : 0;
  int c = 0;
  for (int64_t t = soff[J]; t < soff[J + 1]; ++t) {
    const unsigned mk = (unsigned)smask[t];
    if (((mk >> (4 * cb)) & 15u) == 0) continue;
    double v = 0.0;
    if (mk & (1u << bit)) v = tiles[(sbase[t] + __popc(mk & ((1u << bit) - 1u))) * 256 + coloff];
    const bool nz = v != 0.0;
    if (FILL) {
      const unsigned long long bal = __ballot(nz);
      if (nz) {
        const int64_t at = w + __popcll(bal & lanemask_lt());
        inner[at] = lab[64 * srow[t] + 16 * a + phys(i)];
        val[at] = v;
      }
      w += __popcll(bal);
    } else {
      c += nz ? 1 : 0;
    }
  }
  if (!FILL) {
    c = (int)wave_sum_i64(c);
    if (lane == 0) cnt[j] = c;
  }
}
// statistics: intermediate products of A B = sum over the entries B(k, j) of the length of A(:, k)
__global__ __launch_bounds__(256) void k_bs_products(Csc A, Csc B, unsigned long long* __restrict__ out) {
  const int j = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (j >= B.cols) return;
  long long s = 0;
  for (int64_t e = B.outer[j] + lane_id(); e < B.outer[j + 1]; e += WAVE) {
    const int k = B.inner[e];
    s += A.outer[k + 1] - A.outer[k];
  }
  s = wave_sum_i64(s);
  if (lane_id() == 0 && s) atomicAdd(&out[(j >> 2) & 63], (unsigned long long)s);   // (64 partial sums: no single hot address)
}
__global__ __launch_bounds__(256) void k_bs_sum_i32(int64_t n, const int32_t* __restrict__ x, unsigned long long* __restrict__ out) {
  long long s = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += x[i];
  s = wave_sum_i64(s);
  if (lane_id() == 0 && s) atomicAdd(out, (unsigned long long)s);
}

void from_block(const BlockForm& F, int64_t nnz, DevMat& C) {
  const BlockOrder& bo = *F.order;
  const int n = bo.n, ns = F.ns;
  DevMat R;
  R.rows = n; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1;
  R.outer.alloc((size_t)n + 1);
  DevBuf<int32_t> cnt((size_t)n);
  cnt.zero();
  hipLaunchKernelGGL((k_bs_unblock<false>), dim3(gridw((int64_t)64 * ns)), dim3(256), 0, stream(), ns, F.soff.p, F.srow.p, F.smask.p, F.sbase.p,
                     F.tiles.p, bo.lab.p, cnt.p, (const int64_t*)nullptr, (int32_t*)nullptr, (double*)nullptr);
  scan_i32_async(cnt.p, R.outer.p, (int64_t)n);
  R.inner.alloc((size_t)nnz + kIndexSlack);
  R.val.alloc((size_t)nnz + kIndexSlack);
  if (nnz > 0) {
    DevBuf<int32_t> tin((size_t)nnz);
    DevBuf<double> tval((size_t)nnz);
    hipLaunchKernelGGL((k_bs_unblock<true>), dim3(gridw((int64_t)64 * ns)), dim3(256), 0, stream(), ns, F.soff.p, F.srow.p, F.smask.p,
                       F.sbase.p, F.tiles.p, bo.lab.p, (int32_t*)nullptr, R.outer.p, tin.p, tval.p);
    int bits = 1;
    while ((1ll << bits) < (long long)n) ++bits;
    size_t tb = 0;
    HIP_CHECK(rocprim::segmented_radix_sort_pairs(nullptr, tb, tin.p, R.inner.p, tval.p, R.val.p, (unsigned)nnz, (unsigned)n, R.outer.p,
                                                  R.outer.p + 1, 0, bits, stream()));
    DevBuf<char> tmp(tb);
    HIP_CHECK(rocprim::segmented_radix_sort_pairs(tmp.p, tb, tin.p, R.inner.p, tval.p, R.val.p, (unsigned)nnz, (unsigned)n, R.outer.p,
                                                  R.outer.p + 1, 0, bits, stream()));
  }
  C = std::move(R);
}

// =====================================================================================================================
// 6. the TRS2 update in block form
// =====================================================================================================================
// statistics: intermediate products of A B with both operands in block form = sum over the entries B(k, j) of the
// entries of column k of A (ccountA by position)
__global__ __launch_bounds__(256) void k_bs_products_blk(int ns, const int64_t* __restrict__ soff, const int32_t* __restrict__ srow,
                                                         const int32_t* __restrict__ smask, const int64_t* __restrict__ sbase,
                                                         const double* __restrict__ tiles, const int32_t* __restrict__ ccountA,
                                                         unsigned long long* __restrict__ out) {
  const int pc = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (pc >= 64 * ns) return;
  const int lane = lane_id(), J = pc >> 6, cb = (pc >> 4) & 3, a = lane >> 4, i = lane & 15;
  const int bit = 4 * cb + a;
  const int coloff = tile_word(i, phys(pc & 15));
  long long s = 0;
  for (int64_t t = soff[J]; t < soff[J + 1]; ++t) {
    const unsigned mk = (unsigned)smask[t];
    if ((mk & (1u << bit)) == 0) continue;
    const double v = tiles[(sbase[t] + __popc(mk & ((1u << bit) - 1u))) * 256 + coloff];
    if (v != 0.0) s += ccountA[64 * srow[t] + 16 * a + phys(i)];
  }
  s = wave_sum_i64(s);
  if (lane == 0 && s) atomicAdd(&out[(pc >> 2) & 63], (unsigned long long)s);
}
// per column position: entries and largest row label.  A wave per SUPER-TILE: every tile is read once, whole (lane l takes the words 4 l .. 4 l + 3, all in
// in-tile column l / 4), the four lanes of a column are reduced and leave their part with two atomics per column position
// (a wave per column position walks the super-column with 64 scattered words per step: three times slower)
__global__ __launch_bounds__(256) void k_bs_colstat_st(int64_t nst, int ns, const int64_t* __restrict__ soff, const int32_t* __restrict__ srow,
                                                       const int32_t* __restrict__ smask, const int64_t* __restrict__ sbase,
                                                       const double* __restrict__ tiles, const int32_t* __restrict__ lab,
                                                       int32_t* __restrict__ ccount, int32_t* __restrict__ plast) {
  const int64_t s = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE;
  if (s >= nst) return;
  const int lane = lane_id();
  int lo = 0, hi = ns;          // super-column of super-tile s: the last J with soff[J] <= s
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (soff[mid] <= s) lo = mid; else hi = mid;
  }
  const int J = lo, I = srow[s];
  const int w0 = 4 * lane, col_t = w0 >> 4, ch = (w0 & 15) >> 1;
  int rpos[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) rpos[e] = phys(((((ch + (e >> 1)) ^ (col_t >> 1)) & 7) << 1) | (e & 1));
  const unsigned mk = (unsigned)smask[s];
  const double* __restrict__ base = tiles + sbase[s] * 256;
  int cnt[4] = {0, 0, 0, 0}, mx[4] = {-1, -1, -1, -1};
  int rank = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if ((mk & (1u << t)) == 0) continue;
    const v4d v = *reinterpret_cast<const v4d*>(base + rank * 256 + w0);
    rank += 1;
    const int rb = t & 3, cb = t >> 2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (v[e] != 0.0) {
        cnt[cb] += 1;
        mx[cb] = max(mx[cb], lab[64 * I + 16 * rb + rpos[e]]);
      }
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    int c = cnt[cb], m2 = mx[cb];
    c += __shfl_xor(c, 1, WAVE); m2 = max(m2, __shfl_xor(m2, 1, WAVE));
    c += __shfl_xor(c, 2, WAVE); m2 = max(m2, __shfl_xor(m2, 2, WAVE));
    if ((lane & 3) == 0 && c > 0) {
      const int pc = 64 * J + 16 * cb + phys(col_t);
      atomicAdd(&ccount[pc], c);
      atomicMax(&plast[pc], m2);
    }
  }
}
void block_colstat(BlockForm& F) {
  if (F.have_stat) return;
  const int ns = F.ns;
  F.ccount.alloc((size_t)64 * ns);
  F.plast.alloc((size_t)64 * ns);
  F.ccount.zero();
  HIP_CHECK(hipMemsetAsync(F.plast.p, 0xFF, sizeof(int32_t) * (size_t)64 * ns, stream()));
  if (F.nst > 0)
    hipLaunchKernelGGL(k_bs_colstat_st, dim3(gridw(F.nst)), dim3(256), 0, stream(), F.nst, ns, F.soff.p, F.srow.p, F.smask.p, F.sbase.p,
                       F.tiles.p, F.order->lab.p, F.ccount.p, F.plast.p);
  F.have_stat = true;
}

// union of the super-tiles of two matrices, super-column by super-column (a bitmap per column, as the symbolic phase)
template <bool FILL>
__global__ __launch_bounds__(256) void k_bs_union(int ns, const int64_t* __restrict__ soffA, const int32_t* __restrict__ srowA,
                                                  const int64_t* __restrict__ soffB, const int32_t* __restrict__ srowB,
                                                  int32_t* __restrict__ ccount, const int64_t* __restrict__ coff,
                                                  int32_t* __restrict__ ci, int32_t* __restrict__ cj) {
  extern __shared__ unsigned bm[];
  __shared__ int wsum[8];
  __shared__ int total;
  const int J = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = tid / WAVE;
  const int nwords = (ns + 31) / 32;
  for (int i = tid; i < nwords; i += blockDim.x) bm[i] = 0;
  if (tid == 0) total = 0;
  __syncthreads();
  for (int64_t t = soffA[J] + tid; t < soffA[J + 1]; t += blockDim.x) { const int I = srowA[t]; atomicOr(&bm[I >> 5], 1u << (I & 31)); }
  for (int64_t t = soffB[J] + tid; t < soffB[J + 1]; t += blockDim.x) { const int I = srowB[t]; atomicOr(&bm[I >> 5], 1u << (I & 31)); }
  __syncthreads();
  const int per = (nwords + blockDim.x - 1) / blockDim.x;
  const int w0 = min(nwords, tid * per), w1 = min(nwords, w0 + per);
  int c = 0;
  for (int i = w0; i < w1; ++i) c += __popc(bm[i]);
  if (!FILL) {
    c = (int)wave_sum_i64(c);
    if (lane == 0) atomicAdd(&total, c);
    __syncthreads();
    if (tid == 0) ccount[J] = total;
    return;
  }
  int x = c;
  for (int o = 1; o < WAVE; o <<= 1) {
    const int a = __shfl_up(x, o, WAVE);
    if (lane >= o) x += a;
  }
  if (lane == WAVE - 1) wsum[wave] = x;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < wave; ++k) base += wsum[k];
  int64_t k = coff[J] + base + x - c;
  for (int i = w0; i < w1; ++i) {
    unsigned w = bm[i];
    while (w) {
      const int b = __ffs(w) - 1;
      w &= w - 1;
      ci[k] = 32 * i + b;
      cj[k] = J;
      ++k;
    }
  }
}

struct BsMergeArgs {
  // P = the product (role A of the merge, scaled by am), X = the iterate (role B, scaled by bm), D = the dot operand
  const int64_t* soffP; const int32_t* srowP; const int32_t* smaskP; const int64_t* sbaseP; const double* tilesP; const int32_t* plastP;
  const int64_t* soffX; const int32_t* srowX; const int32_t* smaskX; const int64_t* sbaseX; const double* tilesX; const int32_t* plastX;
  const int64_t* soffD; const int32_t* srowD; const int32_t* smaskD; const int64_t* sbaseD; const double* tilesD;
  const int32_t* lab;
  int64_t ncand;
  const int32_t *ci, *cj;
  int32_t* cmask; int64_t* cbase; int32_t* ccnt;
  double* pdot;          // [2 ncand]: (dot, trace) of the candidate
  double* pool; int64_t pool_tiles;
  unsigned long long* counters;   // [0] tiles handed out, [1] overflow / kept-zero flags
  int32_t* ccount_out; int32_t* plast_out;   // per column position of the result (atomics)
  double am, bm, thr;
};
__device__ inline int64_t bs_find(const int32_t* __restrict__ srow, int64_t lo, int64_t hi, int I) {
  const int64_t end = hi;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (srow[mid] < I) lo = mid + 1; else hi = mid;
  }
  return (lo < end && srow[lo] == I) ? lo : -1;
}
// One wave per super-tile of the union (MODE 2) or of the product (MODE 1).  Lane l owns the words 4 l .. 4 l + 3 of every
// tile (32 contiguous bytes).  MODE 2: result = am P + bm X element by element with the AddSparseVectors rules
// (sparse_includes/AddSparseVectors.f90:21-70; inc_decide of kernels.hip): both present -> kept if |sum| > threshold; one
// present -> kept if |value| > threshold, or unfiltered when its row LABEL lies beyond the other column's last label;
// kept tiles go to the pool, the (dot with D, trace, entries) of the super-tile to the candidate's slots, entries and last
// label per column to the result's column statistics.  MODE 1: the result is P itself: dot, trace only.
template <int MODE>
__global__ __launch_bounds__(64) void k_bs_merge(const BsMergeArgs a) {
  const int64_t cand = blockIdx.x;
  if (cand >= a.ncand) return;
  const int lane = lane_id();
  const int I = uni_i32(a.ci[cand]), J = uni_i32(a.cj[cand]);
  const int64_t tp = bs_find(a.srowP, a.soffP[J], a.soffP[J + 1], I);
  const int64_t tx = MODE == 2 ? bs_find(a.srowX, a.soffX[J], a.soffX[J + 1], I) : -1;
  const int64_t td = bs_find(a.srowD, a.soffD[J], a.soffD[J + 1], I);
  const unsigned mP = tp >= 0 ? (unsigned)a.smaskP[tp] : 0u, mX = tx >= 0 ? (unsigned)a.smaskX[tx] : 0u, mD = td >= 0 ? (unsigned)a.smaskD[td] : 0u;
  const double* __restrict__ bP = tp >= 0 ? a.tilesP + a.sbaseP[tp] * 256 : a.tilesP;
  const double* __restrict__ bX = tx >= 0 ? a.tilesX + a.sbaseX[tx] * 256 : a.tilesP;
  const double* __restrict__ bD = td >= 0 ? a.tilesD + a.sbaseD[td] * 256 : a.tilesP;
  // this lane's four words of a tile: words 4 l + e -> in-tile (row, column) -> in-block positions
  const int w0 = 4 * lane, col_t = w0 >> 4, ch = (w0 & 15) >> 1;   // (two chunks: ch, ch + 1; e = 0, 1 in the first, 2, 3 in the second)
  int rowpos[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int row_t = ((((ch + (e >> 1)) ^ (col_t >> 1)) & 7) << 1) | (e & 1);
    rowpos[e] = phys(row_t);
  }
  const int colpos = phys(col_t);
  double dsum = 0.0, tsum = 0.0;
  v4d outv[16];
  unsigned mC = 0;
  int cnt = 0;
  unsigned flags = 0;
  const unsigned mU = MODE == 2 ? (mP | mX) : mP;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    if ((mU & (1u << t)) == 0) continue;
    const int rb = t & 3, cb = t >> 2;
    v4d p4 = bs_zero4(), x4 = bs_zero4(), d4 = bs_zero4();
    if (mP & (1u << t)) p4 = *reinterpret_cast<const v4d*>(bP + __popc(mP & ((1u << t) - 1u)) * 256 + w0);
    if (MODE == 2 && (mX & (1u << t))) x4 = *reinterpret_cast<const v4d*>(bX + __popc(mX & ((1u << t) - 1u)) * 256 + w0);
    if (mD & (1u << t)) d4 = *reinterpret_cast<const v4d*>(bD + __popc(mD & ((1u << t) - 1u)) * 256 + w0);
    const int pc = 64 * J + 16 * cb + colpos;
    int lastP = -1, lastX = -1;
    if (MODE == 2) { lastP = a.plastP[pc]; lastX = a.plastX[pc]; }
    v4d o4 = bs_zero4();
    bool any = false;
    int kc = 0, kl = -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      double o;
      bool keep;
      if (MODE == 2) {
        const double p = p4[e], x = x4[e];
        const bool ha = p != 0.0, hb = x != 0.0;
        const double wa = __dmul_rn(a.am, p), bs = __dmul_rn(a.bm, x);
        o = 0.0;
        keep = false;
        const int rl = (ha || hb) ? a.lab[64 * I + 16 * rb + rowpos[e]] : -1;
        if (ha && hb) { o = __dadd_rn(wa, bs); keep = fabs(o) > a.thr; }
        else if (ha) { o = wa; keep = (rl > lastX) ? true : (fabs(wa) > a.thr); }
        else if (hb) { o = bs; keep = (rl > lastP) ? true : (fabs(bs) > a.thr); }
        if (keep && o == 0.0) flags |= 2u;
        if (keep) kl = max(kl, rl);
      } else {
        o = p4[e];
        keep = o != 0.0;
      }
      o4[e] = keep ? o : 0.0;
      any |= keep;
      kc += keep ? 1 : 0;
      if (keep) {
        dsum = __dadd_rn(dsum, __dmul_rn(o, d4[e]));
        if (I == J && rb == cb && rowpos[e] == colpos) tsum = __dadd_rn(tsum, o);
      }
    }
    cnt += kc;
    if (MODE == 2) {
      outv[t] = o4;
      if (__ballot(any) != 0ull) mC |= 1u << t;
      // column statistics of the result: the 4 words of a lane lie in one column (16 lanes x 4 words... 4 lanes per column)
      if (kc) { atomicAdd(&a.ccount_out[pc], kc); atomicMax(&a.plast_out[pc], kl); }
    }
  }
  cnt = (int)wave_sum_i64(cnt);
  dsum = wave_sum_f64(dsum);
  tsum = wave_sum_f64(tsum);
  if (MODE == 2) {
    const int nt = __popc(mC);
    int64_t slot0 = 0;
    bool ok = true;
    if (nt) {
      unsigned long long s = 0;
      if (lane == 0) s = atomicAdd(&a.counters[0], (unsigned long long)nt);
      slot0 = uni_i64((int64_t)s);
      if (slot0 + nt > a.pool_tiles) { ok = false; flags |= 1u; }
    }
    if (nt && ok) {
      int rank = 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (mC & (1u << t)) {
          *reinterpret_cast<v4d*>(a.pool + (slot0 + rank) * 256 + w0) = outv[t];
          rank += 1;
        }
      }
    }
    if (__ballot(flags != 0) != 0ull) {
      unsigned f = flags;
      for (int o = 32; o > 0; o >>= 1) f |= __shfl_xor(f, o, WAVE);
      if (lane == 0) atomicOr(&a.counters[1], (unsigned long long)f);
    }
    if (lane == 0) {
      a.cmask[cand] = ok ? (int32_t)mC : 0;
      a.cbase[cand] = slot0;
      a.ccnt[cand] = ok ? cnt : 0;
    }
  }
  if (lane == 0) { a.pdot[2 * cand] = dsum; a.pdot[2 * cand + 1] = tsum; }
}
// (dot, trace) pairs summed in a fixed shape: 256 partial sums, then one block
__global__ __launch_bounds__(256) void k_bs_sum_pairs(int64_t n, const double* __restrict__ x, double* __restrict__ part) {
  __shared__ double sh[2][4];
  double s0 = 0.0, s1 = 0.0;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t b0 = blockIdx.x * per, b1 = min(n, b0 + per);
  for (int64_t i = b0 + threadIdx.x; i < b1; i += blockDim.x) { s0 = __dadd_rn(s0, x[2 * i]); s1 = __dadd_rn(s1, x[2 * i + 1]); }
  s0 = wave_sum_f64(s0);
  s1 = wave_sum_f64(s1);
  if (lane_id() == 0) { sh[0][threadIdx.x / WAVE] = s0; sh[1][threadIdx.x / WAVE] = s1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = __dadd_rn(__dadd_rn(sh[0][0], sh[0][1]), __dadd_rn(sh[0][2], sh[0][3]));
    part[2 * blockIdx.x + 1] = __dadd_rn(__dadd_rn(sh[1][0], sh[1][1]), __dadd_rn(sh[1][2], sh[1][3]));
  }
}
__global__ __launch_bounds__(256) void k_bs_expand_j(int ns, const int64_t* __restrict__ soff, int32_t* __restrict__ cj) {
  const int J = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (J >= ns) return;
  for (int64_t t = soff[J] + lane_id(); t < soff[J + 1]; t += WAVE) cj[t] = J;
}

__global__ __launch_bounds__(256) void k_bs_scale(int64_t nwords, double* __restrict__ tiles, double c) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < nwords) tiles[i] = __dmul_rn(c, tiles[i]);
}
// max over the columns of the sum of |v| (non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(256) void k_bs_colabs_max(int ns, const int64_t* __restrict__ soff, const int32_t* __restrict__ smask,
                                                       const int64_t* __restrict__ sbase, const double* __restrict__ tiles,
                                                       unsigned long long* __restrict__ out) {
  const int pc = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE);
  if (pc >= 64 * ns) return;
  const int lane = lane_id(), J = pc >> 6, cb = (pc >> 4) & 3, a = lane >> 4, i = lane & 15;
  const int bit = 4 * cb + a;
  const int coloff = tile_word(i, phys(pc & 15));
  double s = 0.0;
  for (int64_t t = soff[J]; t < soff[J + 1]; ++t) {
    const unsigned mk = (unsigned)smask[t];
    if ((mk & (1u << bit)) == 0) continue;
    s = __dadd_rn(s, fabs(tiles[(sbase[t] + __popc(mk & ((1u << bit) - 1u))) * 256 + coloff]));
  }
  s = wave_sum_f64(s);
  if (lane == 0) atomicMax(out, (unsigned long long)__double_as_longlong(s));
}

// order-independent fingerprint of a sparsity pattern (sum of per-entry hashes + dimensions): what a block order is keyed on
__global__ __launch_bounds__(256) void k_bs_pattern_fp(int32_t cols, const int64_t* __restrict__ outer, const int32_t* __restrict__ inner,
                                                        unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  unsigned long long h = 0;
  for (int j = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / WAVE); j < cols; j += (int)(gridDim.x * (int64_t)blockDim.x / WAVE)) {
    for (int64_t q = outer[j] + lane_id(); q < outer[j + 1]; q += WAVE) {
      unsigned long long x = ((unsigned long long)(unsigned)j << 32) | (unsigned)inner[q];
      x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;   // (murmur3 finaliser)
      h += x;
    }
  }
  h = (unsigned long long)wave_sum_i64((int64_t)h);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = h;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}
unsigned long long block_pattern_fp(const DevMat& M) {   // (compressed columns)
  DevBuf<unsigned long long> acc(1);
  acc.zero();
  hipLaunchKernelGGL(k_bs_pattern_fp, dim3(1024), dim3(256), 0, stream(), M.cols, M.outer.p, M.inner.p, acc.p);
  unsigned long long h = 0;
  HIP_CHECK(hipMemcpyAsync(&h, acc.p, sizeof(h), hipMemcpyDeviceToHost, stream()));
  sync_stream();
  return h ^ ((unsigned long long)M.nnz * 0x9e3779b97f4a7c15ull) ^ (unsigned long long)M.cols;
}

// =====================================================================================================================
// caches
// =====================================================================================================================
struct CachedForm {   // the block form of a matrix in compressed columns, valid while the matrix is what it was
  const void* val = nullptr;
  unsigned long long serial = 0, epoch = 0, order_serial = 0;
  int64_t nnz = -1;
  int32_t cols = 0;
  std::shared_ptr<BlockForm> form;
  unsigned long long used = 0;
};
struct BlockCache {
  CachedForm forms[4];
  unsigned long long clock = 0;
  std::shared_ptr<BlockOrder> order;     // the order of the dimension being multiplied (one of `kept`)
  std::shared_ptr<BlockOrder> kept[4];   // the orders of the last four dimensions: a dimension's order is made once and kept
  unsigned long long kept_used[4] = {0, 0, 0, 0};
  int32_t refused_n = -1;                // a dimension whose matrices have no blocks (remembered with the entry count it was tried on)
  int64_t refused_nnz = 0;
  int64_t pool_hint = 0;                 // tiles the last result of this dimension needed
  int32_t pool_hint_n = -1;
};
BlockCache& cache() {
  static BlockCache* c = new BlockCache();
  return *c;
}
// makes the most recently used kept order of dimension n the current one (false: there is none)
bool select_order(BlockCache& bc, int32_t n) {
  if (bc.order && bc.order->n == n) return true;
  int best = -1;
  for (int i = 0; i < 4; ++i)
    if (bc.kept[i] && bc.kept[i]->n == n && (best < 0 || bc.kept_used[i] > bc.kept_used[best])) best = i;
  if (best < 0) return false;
  bc.order = bc.kept[best];
  bc.kept_used[best] = ++bc.clock;
  return true;
}
// the kept order of dimension n made from the pattern with fingerprint fp, made current (false: there is none)
bool select_order_fp(BlockCache& bc, int32_t n, unsigned long long fp) {
  for (int i = 0; i < 4; ++i)
    if (bc.kept[i] && bc.kept[i]->n == n && bc.kept[i]->seed_fp == fp) {
      bc.order = bc.kept[i];
      bc.kept_used[i] = ++bc.clock;
      return true;
    }
  return false;
}
// keeps an order (four slots: one per (dimension, seed pattern), least recently used replaced) and makes it current
void install_order(BlockCache& bc, const std::shared_ptr<BlockOrder>& o) {
  int slot = 0;
  for (int i = 0; i < 4; ++i) {
    if (!bc.kept[i] || (bc.kept[i]->n == o->n && bc.kept[i]->seed_fp == o->seed_fp)) { slot = i; break; }
    if (bc.kept_used[i] < bc.kept_used[slot]) slot = i;
  }
  bc.kept[slot] = o;
  bc.kept_used[slot] = ++bc.clock;
  bc.order = o;
}
// a pattern tiles "as well as the order's own" when its fill reaches this fraction of the fill the order's seed has in it
constexpr double kSeedFillFraction = 0.75;
constexpr double kMinFill = 0.08;

}  // namespace

void drop_block_caches() {
  for (CachedForm& f : cache().forms) f = CachedForm();
  cache().order.reset();
  for (auto& k : cache().kept) k.reset();
  cache().refused_n = -1;
  cache().pool_hint_n = -1;
}

bool block_order_for(const DevMat& M, std::vector<int32_t>& pos_host) {
  if (M.cplx || M.rows != M.cols || M.loose() || M.expanded()) return false;
  BlockCache& c = cache();
  if (!select_order(c, M.cols)) install_order(c, build_block_order(M));
  pos_host.resize((size_t)M.cols);
  HIP_CHECK(hipMemcpyAsync(pos_host.data(), c.order->pos.p, sizeof(int32_t) * (size_t)M.cols, hipMemcpyDeviceToHost, stream()));
  sync_stream();
  return true;
}

bool block_order_of_pattern(const DevMat& M, std::vector<int32_t>& pos_host, int32_t* ns_out) {
  if (M.cplx || M.rows != M.cols || M.loose() || M.expanded() || M.blocked() || M.nnz < 8LL * M.cols) return false;
  BlockCache& c = cache();
  const unsigned long long fp = block_pattern_fp(M);
  if (!select_order_fp(c, M.cols, fp)) install_order(c, build_block_order(M));
  if (c.order->ns > kMaxSuperBlocks) return false;
  pos_host.resize((size_t)M.cols);
  HIP_CHECK(hipMemcpyAsync(pos_host.data(), c.order->pos.p, sizeof(int32_t) * (size_t)M.cols, hipMemcpyDeviceToHost, stream()));
  sync_stream();
  if (ns_out) *ns_out = c.order->ns;
  return true;
}

void install_block_positions(int32_t n, int32_t ns, const std::vector<int32_t>& pos) {
  std::shared_ptr<BlockOrder> bo(new BlockOrder());
  bo->n = n;
  bo->ns = ns;
  std::vector<int32_t> lab((size_t)64 * (size_t)ns, -1);
  for (int32_t v = 0; v < n; ++v) lab[(size_t)pos[(size_t)v]] = v;
  bo->pos.alloc((size_t)n);
  bo->lab.alloc(lab.size());
  bo->pos.upload(pos.data(), (size_t)n);
  bo->lab.upload(lab.data(), lab.size());
  sync_stream();
  static unsigned long long serial = 1ull << 40;   // (apart from the serials of the orders build_block_order makes)
  bo->serial = ++serial;
  unsigned long long h = 0xcbf29ce484222325ull;      // (no pattern's fingerprint: the order is the caller's -- keyed on its positions,
  for (int32_t v = 0; v < n; ++v) h = (h ^ (unsigned long long)(unsigned)pos[(size_t)v]) * 0x100000001b3ull;   // so that the same order finds its slot again)
  bo->seed_fp = h | 1ull;
  for (CachedForm& f : cache().forms) f = CachedForm();
  install_order(cache(), bo);
}

namespace {
void block_colstat(BlockForm& F);
// C = alpha A B pruned, block form in, block form out (symbolic phase, numeric phase, the result's super-tiles)
void block_product(BlockCache& bc, BlockForm& FA, BlockForm& FB, double alpha, double threshold, bool dense_rule, BlockForm& FC,
                   unsigned long long* nnz_out, unsigned long long hc[4], int64_t* ncand_out, hipEvent_t ev_begin, hipEvent_t ev_end,
                   unsigned long long* nprod_out, const DevMat* Acsc, const DevMat* Bcsc) {
  const int32_t n = bc.order->n;
  build_rows(FA);
  const int ns = bc.order->ns;
  // (statistics with operands in block form: the products are counted by the pass that makes B's slice masks, from the
  // entries per column of A -- no pass of their own)
  DevBuf<unsigned long long> prod64(64);
  bool prod_fused = false;
  if (nprod_out && !(Acsc && Bcsc)) {
    prod64.zero();
    block_colstat(FA);
    if (&FA == &FB) prod_fused = block_quads(FA, FA.ccount.p, prod64.p);
    else { block_quads(FA); prod_fused = block_quads(FB, FA.ccount.p, prod64.p); }
  }
  block_quads(FA);
  block_quads(FB);
  // ---- symbolic
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_symbolic<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_symbolic<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_done = true;
  }
  const size_t sym_lds = (size_t)((ns + 31) / 32) * 4;
  DevBuf<int32_t> ccount((size_t)ns);
  DevBuf<int64_t> coff((size_t)ns + 1);
  hipLaunchKernelGGL((k_bs_symbolic<false>), dim3(ns), dim3(256), sym_lds, stream(), ns, FA.soff.p, FA.srow.p, FB.soff.p, FB.srow.p, ccount.p,
                     (const int64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  scan_i32_async(ccount.p, coff.p, (int64_t)ns);
  int64_t ncand = 0;
  {
    ScalarFetch f;
    f.add(coff.p + ns, 1, &ncand);
    f.run();
  }
  *ncand_out = ncand;
  if (ncand == 0) return;
  DevBuf<int32_t> ci((size_t)ncand), cj((size_t)ncand), cmask((size_t)ncand), ccnt((size_t)ncand);
  DevBuf<int64_t> cbase((size_t)ncand);
  hipLaunchKernelGGL((k_bs_symbolic<true>), dim3(ns), dim3(256), sym_lds, stream(), ns, FA.soff.p, FA.srow.p, FB.soff.p, FB.srow.p,
                     (int32_t*)nullptr, coff.p, ci.p, cj.p);
  // processing order of the candidates
  DevBuf<int32_t> order((size_t)ncand);
  {
    DevBuf<unsigned> zk((size_t)ncand), zk_s((size_t)ncand);
    DevBuf<int32_t> zi((size_t)ncand);
    static const int zmode = std::getenv("NTPOLY_AMD_BS_ORDER") ? std::atoi(std::getenv("NTPOLY_AMD_BS_ORDER")) : 1;
    hipLaunchKernelGGL(k_bs_zorder, dim3(grid1(ncand)), dim3(256), 0, stream(), ncand, ci.p, cj.p, zk.p, zi.p, zmode);
    size_t tb = 0;
    HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tb, zk.p, zk_s.p, zi.p, order.p, (size_t)ncand, 0, 32, stream()));
    DevBuf<char> tmp(tb);
    HIP_CHECK(rocprim::radix_sort_pairs(tmp.p, tb, zk.p, zk_s.p, zi.p, order.p, (size_t)ncand, 0, 32, stream()));
  }
  // ---- the matches of every candidate (option block_match): slots by the upper bound min(entries of A's super-row, entries of
  // B's super-column), filled in ascending K by a wave per candidate
  DevBuf<int64_t> moff;
  DevBuf<int32_t> mcnt;
  DevBuf<int2> mlist;
  if (options().block_match != 0) {
    DevBuf<int32_t> bound((size_t)ncand);
    moff.alloc((size_t)ncand + 1);
    mcnt.alloc((size_t)ncand);
    hipLaunchKernelGGL(k_bs_match_bound, dim3(grid1(ncand)), dim3(256), 0, stream(), ncand, ci.p, cj.p, FA.roff.p, FB.soff.p, bound.p);
    scan_i32_async(bound.p, moff.p, ncand);
    int64_t mtotal = 0;
    {
      ScalarFetch f;
      f.add(moff.p + ncand, 1, &mtotal);
      f.run();
    }
    mlist.alloc((size_t)mtotal + 64);
    hipLaunchKernelGGL(k_bs_match, dim3(gridw(ncand)), dim3(256), 0, stream(), ncand, ci.p, cj.p, FA.roff.p, FA.rcol.p, FA.ridx.p, FB.soff.p,
                       FB.srow.p, moff.p, mlist.p, mcnt.p);
  }
  // ---- numeric (the pool is sized from the last product of this dimension; an overflow is repeated with the exact size)
  FC = BlockForm();
  FC.order = bc.order;
  FC.ns = ns;
  const bool dense_tiles = (double)FA.nnz > 0.5 * 256.0 * (double)FA.ntiles && (double)FB.nnz > 0.5 * 256.0 * (double)FB.ntiles;
  int64_t pool = std::max<int64_t>(1024, std::max(FA.ntiles, FB.ntiles) * 2);
  if (bc.pool_hint_n == n) pool = std::max(pool, bc.pool_hint + bc.pool_hint / 4);
  pool = std::min<int64_t>(pool, ncand * 16);
  DevBuf<unsigned long long> counters(4);
  for (int attempt = 0; attempt < 2; ++attempt) {
    FC.tiles.alloc((size_t)pool * 256 + 512);
    counters.zero();
    BsArgs a;
    a.roffA = FA.roff.p; a.rcolA = FA.rcol.p; a.ridxA = FA.ridx.p; a.smaskA = FA.smask.p; a.sbaseA = FA.sbase.p; a.tilesA = FA.tiles.p;
    a.soffB = FB.soff.p; a.srowB = FB.srow.p; a.smaskB = FB.smask.p; a.sbaseB = FB.sbase.p; a.tilesB = FB.tiles.p;
    a.quadsA = FA.quads.p; a.quadsB = FB.quads.p;
    a.ncand = ncand; a.ci = ci.p; a.cj = cj.p; a.order = order.p; a.cmask = cmask.p; a.cbase = cbase.p; a.ccnt = ccnt.p;
    a.pool = FC.tiles.p; a.pool_tiles = pool; a.counters = counters.p;
    a.alpha = alpha; a.threshold = threshold; a.dense_rule = dense_rule ? 1 : 0;
    a.nwg = (int)ncand;
    a.moff = moff.p; a.mcnt = mcnt.p; a.mlist = mlist.p;
    a.ablate = 0;
#ifdef NTP_ABLATIONS
    if (const char* v = std::getenv("NTPOLY_AMD_BS_ABLATE")) a.ablate = std::atoi(v);
#endif
    if (ev_begin) HIP_CHECK(hipEventRecord(ev_begin, stream()));
    // (dense tiles: one wave per candidate; sparse tiles: two)
    const bool lst = a.mlist != nullptr;
    if (options().spgemm_fma == 0) {
      if (lst) hipLaunchKernelGGL((k_bs_numeric<2, true, true>), dim3(xcd_grid(a.nwg)), dim3(128), 0, stream(), a);
      else hipLaunchKernelGGL((k_bs_numeric<2, true>), dim3(xcd_grid(a.nwg)), dim3(128), 0, stream(), a);
    } else if (dense_tiles) {
      if (lst) hipLaunchKernelGGL((k_bs_numeric<1, false, true>), dim3(xcd_grid(a.nwg)), dim3(64), 0, stream(), a);
      else hipLaunchKernelGGL((k_bs_numeric<1>), dim3(xcd_grid(a.nwg)), dim3(64), 0, stream(), a);
    } else {
      if (lst) hipLaunchKernelGGL((k_bs_numeric<2, false, true>), dim3(xcd_grid(a.nwg)), dim3(128), 0, stream(), a);
      else hipLaunchKernelGGL((k_bs_numeric<2>), dim3(xcd_grid(a.nwg)), dim3(128), 0, stream(), a);
    }
    if (ev_end) HIP_CHECK(hipEventRecord(ev_end, stream()));
    {
      ScalarFetch f;
      f.add(counters.p, 4, hc);
      f.run();
    }
    if (hc[1] == 0) break;
    if (attempt == 1) NTP_FATAL("internal: block SpGEMM: the tile pool overflowed twice");
    pool = (int64_t)hc[0];
  }
  bc.pool_hint = (int64_t)hc[0];
  bc.pool_hint_n = n;
  // ---- the result's super-tiles and entry count
  DevBuf<int32_t> flag((size_t)ncand);
  DevBuf<int64_t> excl((size_t)ncand + 1);
  DevBuf<unsigned long long> tot(65);   // [0] entries of the result, [1 .. 64] partial product counts
  tot.zero();
  if (nprod_out) {
    if (Acsc && Bcsc) hipLaunchKernelGGL(k_bs_products, dim3(gridw(n)), dim3(256), 0, stream(), view(*Acsc), view(*Bcsc), tot.p + 1);
    else if (prod_fused) {
      HIP_CHECK(hipMemcpyAsync(tot.p + 1, prod64.p, 64 * sizeof(unsigned long long), hipMemcpyDeviceToDevice, stream()));
    } else {   // (B's masks existed already: entries per column of A, then the sum over the entries of B)
      block_colstat(FA);
      hipLaunchKernelGGL(k_bs_products_blk, dim3(gridw((int64_t)64 * ns)), dim3(256), 0, stream(), ns, FB.soff.p, FB.srow.p, FB.smask.p, FB.sbase.p,
                         FB.tiles.p, FA.ccount.p, tot.p + 1);
    }
  }
  hipLaunchKernelGGL(k_bs_flag, dim3(grid1(ncand)), dim3(256), 0, stream(), ncand, cmask.p, flag.p);
  scan_i32_async(flag.p, excl.p, ncand);
  hipLaunchKernelGGL(k_bs_sum_i32, dim3(std::min(1024, grid1(ncand))), dim3(256), 0, stream(), ncand, ccnt.p, tot.p);
  int64_t nstC = 0;
  unsigned long long nnzC = 0, nprod = 0;
  {
    unsigned long long t2[65];
    ScalarFetch f;
    f.add(excl.p + ncand, 1, &nstC);
    f.add(tot.p, 65, t2);
    f.run();
    nnzC = t2[0];
    for (int i = 1; i < 65; ++i) nprod += t2[i];
  }
  if (nnzC >= (1ull << 32)) NTP_FATAL("block SpGEMM: the product holds 2^32 entries or more");
  *nnz_out = nnzC;
  if (nprod_out) *nprod_out = nprod;
  FC.nst = nstC;
  FC.ntiles = (int64_t)hc[0];
  FC.nnz = (int64_t)nnzC;
  FC.soff.alloc((size_t)ns + 1);
  FC.srow.alloc((size_t)std::max<int64_t>(1, nstC));
  FC.smask.alloc((size_t)std::max<int64_t>(1, nstC));
  FC.sbase.alloc((size_t)std::max<int64_t>(1, nstC));
  hipLaunchKernelGGL(k_bs_compact, dim3(grid1(ncand)), dim3(256), 0, stream(), ncand, cmask.p, cbase.p, ci.p, excl.p, FC.srow.p, FC.smask.p, FC.sbase.p);
  hipLaunchKernelGGL(k_bs_soff, dim3(grid1(ns + 1)), dim3(256), 0, stream(), ns, coff.p, excl.p, FC.soff.p);
}

}  // namespace

DevMat block_unpack(const DevMat& M) {
  DevMat R;
  from_block(*M.blk, M.nnz, R);
  R.block_hint = 1;   // (its next product goes to the block path first)
  return R;
}

namespace {
// the block form of an operand: its own (DevMat::blk, made in the current order), the cached one, or a fresh conversion
std::shared_ptr<BlockForm> operand_form(const DevMat& M, BlockCache& bc, double min_fill, double* fill, bool* converted) {
  *converted = false;
  if (M.blocked()) {
    if (M.blk->order.get() == bc.order.get()) {
      *fill = M.blk->ntiles > 0 ? (double)M.nnz / (256.0 * (double)M.blk->ntiles) : 0.0;
      return M.blk;
    }
    // (made in an order that has been replaced since: through compressed columns)
    DevMat P = block_unpack(M);
    std::shared_ptr<BlockForm> F(new BlockForm());
    *converted = true;
    if (!to_block(P, bc.order, *F, min_fill, fill)) return nullptr;
    sync_stream();   // (P is released on return)
    return F;
  }
  const unsigned long long ser = dev_alloc_serial(M.val.p), ep = matrix_value_epoch();
  for (CachedForm& f : bc.forms)
    if (f.form && f.val == M.val.p && f.serial == ser && ser != 0 && f.epoch == ep && f.nnz == M.nnz && f.cols == M.cols &&
        f.order_serial == bc.order->serial) {
      f.used = ++bc.clock;
      *fill = f.form->ntiles > 0 ? (double)M.nnz / (256.0 * (double)f.form->ntiles) : 0.0;
      return f.form;
    }
  std::shared_ptr<BlockForm> F(new BlockForm());
  *converted = true;
  if (!to_block(M, bc.order, *F, min_fill, fill)) return nullptr;
  CachedForm* slot = &bc.forms[0];
  for (CachedForm& f : bc.forms)
    if (!f.form) { slot = &f; break; } else if (f.used < slot->used) slot = &f;
  slot->val = M.val.p; slot->serial = ser; slot->epoch = ep; slot->order_serial = bc.order->serial; slot->nnz = M.nnz; slot->cols = M.cols;
  slot->form = F;
  slot->used = ++bc.clock;
  return F;
}
}  // namespace

bool block_trs2_step(DevMat& X, int mode, double threshold, bool dense_rule, const DevMat& D, double out[4], BlockInfo* info,
                     hipEvent_t ev_begin, hipEvent_t ev_end) {
  if (info) *info = BlockInfo();
  if (X.cplx || D.cplx || X.rows != X.cols || D.rows != X.rows || D.cols != X.cols || X.nnz == 0 || D.nnz == 0) return false;
  if (X.loose() || X.expanded() || D.loose() || D.expanded() || D.blocked()) return false;
  if (!block_arithmetic_ok() || options().block_path == 0 || options().spgemm_variant >= 0 || options().spgemm_force_bin > 0) return false;
  BlockCache& bc = cache();
  const int32_t n = X.cols;
  if (!select_order(bc, n)) return false;   // (only dimensions the block path has multiplied before)
  const int ns = bc.order->ns;
  std::shared_ptr<BlockForm> pFX;
  double fx = 0, fd = 0;
  if (X.blocked()) {
    if (X.blk->order.get() != bc.order.get()) return false;
    pFX = X.blk;
    fx = pFX->ntiles > 0 ? (double)X.nnz / (256.0 * (double)pFX->ntiles) : 0.0;
  } else {
    pFX.reset(new BlockForm());   // (the iterate changes every step: not cached)
    if (!to_block(X, bc.order, *pFX, options().block_path == 2 ? 0.0 : kMinFill, &fx)) return false;
  }
  bool conv = false;
  std::shared_ptr<BlockForm> pFD = operand_form(D, bc, 0.0, &fd, &conv);
  if (!pFD) return false;
  BlockForm& FX = *pFX;
  BlockForm& FD = *pFD;
  BlockForm FP;
  unsigned long long nnzP = 0, nprod = 0, hc[4] = {0, 0, 0, 0};
  int64_t ncand = 0;
  const bool count_products = info != nullptr && options().time_kernels != 0;
  block_product(bc, FX, FX, 1.0, threshold, dense_rule, FP, &nnzP, hc, &ncand, ev_begin, ev_end, count_products ? &nprod : nullptr,
                X.blocked() ? nullptr : &X, X.blocked() ? nullptr : &X);
  if (ncand == 0 || nnzP == 0) return false;
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_union<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_union<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_done = true;
  }
  DevBuf<double> part(512), res(2);
  double hres[2] = {0, 0};
  BsMergeArgs a;
  a.soffP = FP.soff.p; a.srowP = FP.srow.p; a.smaskP = FP.smask.p; a.sbaseP = FP.sbase.p; a.tilesP = FP.tiles.p; a.plastP = nullptr;
  a.soffX = FX.soff.p; a.srowX = FX.srow.p; a.smaskX = FX.smask.p; a.sbaseX = FX.sbase.p; a.tilesX = FX.tiles.p; a.plastX = nullptr;
  a.soffD = FD.soff.p; a.srowD = FD.srow.p; a.smaskD = FD.smask.p; a.sbaseD = FD.sbase.p; a.tilesD = FD.tiles.p;
  a.lab = bc.order->lab.p;
  a.cmask = nullptr; a.cbase = nullptr; a.ccnt = nullptr; a.pool = nullptr; a.pool_tiles = 0; a.counters = nullptr;
  a.ccount_out = nullptr; a.plast_out = nullptr;
  a.am = -1.0; a.bm = 2.0; a.thr = threshold;
  DevMat R;
  R.rows = n; R.cols = n; R.cplx = false; R.zero_free = 1;
  if (mode == 1) {
    // X <- P: dot and trace over the product's own super-tiles
    const int64_t nc = FP.nst;
    DevBuf<int32_t> cj((size_t)nc);
    DevBuf<double> pdot((size_t)2 * nc);
    hipLaunchKernelGGL(k_bs_expand_j, dim3(gridw(ns)), dim3(256), 0, stream(), ns, FP.soff.p, cj.p);
    a.ncand = nc; a.ci = FP.srow.p; a.cj = cj.p; a.pdot = pdot.p;
    hipLaunchKernelGGL((k_bs_merge<1>), dim3((unsigned)nc), dim3(64), 0, stream(), a);
    hipLaunchKernelGGL(k_bs_sum_pairs, dim3(256), dim3(256), 0, stream(), nc, pdot.p, part.p);
    hipLaunchKernelGGL(k_bs_sum_pairs, dim3(1), dim3(256), 0, stream(), (int64_t)256, part.p, res.p);
    {
      ScalarFetch f;
      f.add(res.p, 2, hres);
      f.run();
    }
    R.nnz = (int64_t)nnzP;
    R.blk.reset(new BlockForm(std::move(FP)));
  } else {
    block_colstat(FX);
    block_colstat(FP);
    a.plastP = FP.plast.p;
    a.plastX = FX.plast.p;
    const size_t lds = (size_t)((ns + 31) / 32) * 4;
    DevBuf<int32_t> ucount((size_t)ns);
    DevBuf<int64_t> uoff((size_t)ns + 1);
    hipLaunchKernelGGL((k_bs_union<false>), dim3(ns), dim3(256), lds, stream(), ns, FX.soff.p, FX.srow.p, FP.soff.p, FP.srow.p, ucount.p,
                       (const int64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
    scan_i32_async(ucount.p, uoff.p, (int64_t)ns);
    int64_t nc = 0;
    {
      ScalarFetch f;
      f.add(uoff.p + ns, 1, &nc);
      f.run();
    }
    DevBuf<int32_t> ci((size_t)nc), cj((size_t)nc), cmask((size_t)nc), ccnt((size_t)nc), flag((size_t)nc);
    DevBuf<int64_t> cbase((size_t)nc), excl((size_t)nc + 1);
    DevBuf<double> pdot((size_t)2 * nc);
    hipLaunchKernelGGL((k_bs_union<true>), dim3(ns), dim3(256), lds, stream(), ns, FX.soff.p, FX.srow.p, FP.soff.p, FP.srow.p, (int32_t*)nullptr,
                       uoff.p, ci.p, cj.p);
    BlockForm FN;
    FN.order = bc.order;
    FN.ns = ns;
    const int64_t pool = FX.ntiles + FP.ntiles;   // (the union holds no more tiles than its operands together)
    FN.tiles.alloc((size_t)pool * 256 + 512);
    FN.ccount.alloc((size_t)64 * ns);
    FN.plast.alloc((size_t)64 * ns);
    FN.ccount.zero();
    HIP_CHECK(hipMemsetAsync(FN.plast.p, 0xFF, sizeof(int32_t) * (size_t)64 * ns, stream()));
    DevBuf<unsigned long long> counters(4), tot(1);
    counters.zero();
    tot.zero();
    a.ncand = nc; a.ci = ci.p; a.cj = cj.p; a.cmask = cmask.p; a.cbase = cbase.p; a.ccnt = ccnt.p; a.pdot = pdot.p;
    a.pool = FN.tiles.p; a.pool_tiles = pool; a.counters = counters.p; a.ccount_out = FN.ccount.p; a.plast_out = FN.plast.p;
    hipLaunchKernelGGL((k_bs_merge<2>), dim3((unsigned)nc), dim3(64), 0, stream(), a);
    hipLaunchKernelGGL(k_bs_sum_pairs, dim3(256), dim3(256), 0, stream(), nc, pdot.p, part.p);
    hipLaunchKernelGGL(k_bs_sum_pairs, dim3(1), dim3(256), 0, stream(), (int64_t)256, part.p, res.p);
    hipLaunchKernelGGL(k_bs_flag, dim3(grid1(nc)), dim3(256), 0, stream(), nc, cmask.p, flag.p);
    scan_i32_async(flag.p, excl.p, nc);
    hipLaunchKernelGGL(k_bs_sum_i32, dim3(std::min(1024, grid1(nc))), dim3(256), 0, stream(), nc, ccnt.p, tot.p);
    unsigned long long hcnt[4] = {0, 0, 0, 0}, nnzN = 0;
    int64_t nstN = 0;
    {
      ScalarFetch f;
      f.add(res.p, 2, hres);
      f.add(counters.p, 4, hcnt);
      f.add(excl.p + nc, 1, &nstN);
      f.add(tot.p, 1, &nnzN);
      f.run();
    }
    if (hcnt[1] != 0) {   // (a kept value that is exactly zero: the block form cannot hold it -- compressed columns decide)
      if (dbg()) std::fprintf(stderr, "[block trs2] merge refused (flags %llu)\n", hcnt[1]);
      return false;
    }
    FN.nst = nstN;
    FN.ntiles = (int64_t)hcnt[0];
    FN.nnz = (int64_t)nnzN;
    FN.have_stat = true;
    FN.soff.alloc((size_t)ns + 1);
    FN.srow.alloc((size_t)std::max<int64_t>(1, nstN));
    FN.smask.alloc((size_t)std::max<int64_t>(1, nstN));
    FN.sbase.alloc((size_t)std::max<int64_t>(1, nstN));
    hipLaunchKernelGGL(k_bs_compact, dim3(grid1(nc)), dim3(256), 0, stream(), nc, cmask.p, cbase.p, ci.p, excl.p, FN.srow.p, FN.smask.p, FN.sbase.p);
    hipLaunchKernelGGL(k_bs_soff, dim3(grid1(ns + 1)), dim3(256), 0, stream(), ns, uoff.p, excl.p, FN.soff.p);
    sync_stream();   // (the candidate arrays are released on return)
    R.nnz = (int64_t)nnzN;
    R.blk.reset(new BlockForm(std::move(FN)));
  }
  X = std::move(R);
  out[0] = hres[0];
  out[1] = 0.0;
  out[2] = hres[1];
  out[3] = 0.0;
  if (info) {
    info->used = 1;
    info->fill_a = fx; info->fill_b = fx;
    info->tiles_a = FX.ntiles; info->tiles_b = FX.ntiles; info->tiles_c = (int64_t)hc[0];
    info->cand = ncand;
    info->tile_products = (int64_t)((hc[2] + 3) / 4);   // (matrix instructions issued / 4: slices without entries are skipped)
    info->nnz_c = (int64_t)nnzP;
    info->products = (int64_t)nprod;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// block algebra
namespace {
bool algebra_ok(const DevMat& M) {
  return !M.cplx && M.rows == M.cols && !M.loose() && !M.expanded() && block_arithmetic_ok() && options().block_path != 0 &&
         options().spgemm_variant < 0 && options().spgemm_force_bin <= 0;
}
// the form of an operand of the algebra in the order of its dimension (nullptr: none / other order)
std::shared_ptr<BlockForm> algebra_form(const DevMat& M, BlockCache& bc) {
  if (!algebra_ok(M) || M.nnz == 0 || !select_order(bc, M.cols)) return nullptr;
  if (bc.order->ns > kMaxSuperBlocks) return nullptr;
  double fill = 0;
  bool conv = false;
  if (M.blocked() && M.blk->order.get() != bc.order.get()) return nullptr;
  return operand_form(M, bc, 0.0, &fill, &conv);
}
DevMat blocked_matrix(int32_t n, int64_t nnz, BlockForm&& F) {
  DevMat R;
  R.rows = n; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1; R.block_hint = 1;
  R.blk.reset(new BlockForm(std::move(F)));
  return R;
}
void set_dummy_d(BsMergeArgs& a, const int64_t* zero_soff) {
  a.soffD = zero_soff; a.srowD = nullptr; a.smaskD = nullptr; a.sbaseD = nullptr; a.tilesD = nullptr;
}
}  // namespace

bool block_axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold) {
  if (&A == &B || alpha == 0.0 || beta == 0.0 || A.cols != B.cols) return false;
  if (!A.blocked() && !B.blocked()) return false;
  BlockCache& bc = cache();
  std::shared_ptr<BlockForm> pA = algebra_form(A, bc), pB = algebra_form(B, bc);
  if (!pA || !pB) return false;
  BlockForm &FA = *pA, &FB = *pB;
  const int32_t n = B.cols;
  const int ns = bc.order->ns;
  block_colstat(FA);
  block_colstat(FB);
  static bool attr_done = false;
  if (!attr_done) {
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_union<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bs_union<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_done = true;
  }
  const size_t lds = (size_t)((ns + 31) / 32) * 4;
  DevBuf<int32_t> ucount((size_t)ns);
  DevBuf<int64_t> uoff((size_t)ns + 1), zsoff((size_t)ns + 1);
  zsoff.zero();
  hipLaunchKernelGGL((k_bs_union<false>), dim3(ns), dim3(256), lds, stream(), ns, FB.soff.p, FB.srow.p, FA.soff.p, FA.srow.p, ucount.p,
                     (const int64_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr);
  scan_i32_async(ucount.p, uoff.p, (int64_t)ns);
  int64_t nc = 0;
  {
    ScalarFetch f;
    f.add(uoff.p + ns, 1, &nc);
    f.run();
  }
  if (nc == 0) return false;
  DevBuf<int32_t> ci((size_t)nc), cj((size_t)nc), cmask((size_t)nc), ccnt((size_t)nc), flag((size_t)nc);
  DevBuf<int64_t> cbase((size_t)nc), excl((size_t)nc + 1);
  DevBuf<double> pdot((size_t)2 * nc);
  hipLaunchKernelGGL((k_bs_union<true>), dim3(ns), dim3(256), lds, stream(), ns, FB.soff.p, FB.srow.p, FA.soff.p, FA.srow.p, (int32_t*)nullptr,
                     uoff.p, ci.p, cj.p);
  BlockForm FN;
  FN.order = bc.order;
  FN.ns = ns;
  const int64_t pool = FA.ntiles + FB.ntiles;
  FN.tiles.alloc((size_t)pool * 256 + 512);
  FN.ccount.alloc((size_t)64 * ns);
  FN.plast.alloc((size_t)64 * ns);
  FN.ccount.zero();
  HIP_CHECK(hipMemsetAsync(FN.plast.p, 0xFF, sizeof(int32_t) * (size_t)64 * ns, stream()));
  DevBuf<unsigned long long> counters(4), tot(1);
  counters.zero();
  tot.zero();
  BsMergeArgs a;
  // (role P = A scaled by alpha, role X = B scaled by beta: IncrementMatrix(A, B, alpha) after ScaleMatrix(B, beta))
  a.soffP = FA.soff.p; a.srowP = FA.srow.p; a.smaskP = FA.smask.p; a.sbaseP = FA.sbase.p; a.tilesP = FA.tiles.p; a.plastP = FA.plast.p;
  a.soffX = FB.soff.p; a.srowX = FB.srow.p; a.smaskX = FB.smask.p; a.sbaseX = FB.sbase.p; a.tilesX = FB.tiles.p; a.plastX = FB.plast.p;
  set_dummy_d(a, zsoff.p);
  a.lab = bc.order->lab.p;
  a.ncand = nc; a.ci = ci.p; a.cj = cj.p; a.cmask = cmask.p; a.cbase = cbase.p; a.ccnt = ccnt.p; a.pdot = pdot.p;
  a.pool = FN.tiles.p; a.pool_tiles = pool; a.counters = counters.p; a.ccount_out = FN.ccount.p; a.plast_out = FN.plast.p;
  a.am = alpha; a.bm = beta; a.thr = threshold;
  hipLaunchKernelGGL((k_bs_merge<2>), dim3((unsigned)nc), dim3(64), 0, stream(), a);
  hipLaunchKernelGGL(k_bs_flag, dim3(grid1(nc)), dim3(256), 0, stream(), nc, cmask.p, flag.p);
  scan_i32_async(flag.p, excl.p, nc);
  hipLaunchKernelGGL(k_bs_sum_i32, dim3(std::min(1024, grid1(nc))), dim3(256), 0, stream(), nc, ccnt.p, tot.p);
  unsigned long long hcnt[4] = {0, 0, 0, 0}, nnzN = 0;
  int64_t nstN = 0;
  {
    ScalarFetch f;
    f.add(counters.p, 4, hcnt);
    f.add(excl.p + nc, 1, &nstN);
    f.add(tot.p, 1, &nnzN);
    f.run();
  }
  if (hcnt[1] != 0) return false;   // (a kept value that is exactly zero: compressed columns decide)
  FN.nst = nstN;
  FN.ntiles = (int64_t)hcnt[0];
  FN.nnz = (int64_t)nnzN;
  FN.have_stat = true;
  FN.soff.alloc((size_t)ns + 1);
  FN.srow.alloc((size_t)std::max<int64_t>(1, nstN));
  FN.smask.alloc((size_t)std::max<int64_t>(1, nstN));
  FN.sbase.alloc((size_t)std::max<int64_t>(1, nstN));
  hipLaunchKernelGGL(k_bs_compact, dim3(grid1(nc)), dim3(256), 0, stream(), nc, cmask.p, cbase.p, ci.p, excl.p, FN.srow.p, FN.smask.p, FN.sbase.p);
  hipLaunchKernelGGL(k_bs_soff, dim3(grid1(ns + 1)), dim3(256), 0, stream(), ns, uoff.p, excl.p, FN.soff.p);
  sync_stream();   // (the candidate arrays are released on return)
  B = blocked_matrix(n, (int64_t)nnzN, std::move(FN));
  return true;
}

bool block_scale(DevMat& A, double c) {
  if (!A.blocked() || !algebra_ok(A) || c == 0.0) return false;
  // (the tiles may be shared with a copy made by block_clone's cheap path: they are not -- clones are deep)
  BlockForm& F = *A.blk;
  const int64_t nw = F.ntiles * 256;
  if (nw > 0) hipLaunchKernelGGL(k_bs_scale, dim3(grid1(nw)), dim3(256), 0, stream(), nw, F.tiles.p, c);
  return true;
}

bool block_clone(const DevMat& A, DevMat& Out) {
  if (!A.blocked() || !algebra_ok(A)) return false;
  const BlockForm& F = *A.blk;
  BlockForm G;
  G.order = F.order; G.ns = F.ns; G.nst = F.nst; G.ntiles = F.ntiles; G.nnz = F.nnz;
  auto dup = [](auto& dst, const auto& src, size_t count) {
    using T = std::remove_reference_t<decltype(*src.p)>;
    dst.alloc(std::max<size_t>(1, count));
    if (count) HIP_CHECK(hipMemcpyAsync(dst.p, src.p, sizeof(T) * count, hipMemcpyDeviceToDevice, stream()));
  };
  dup(G.soff, F.soff, (size_t)F.ns + 1);
  dup(G.srow, F.srow, (size_t)F.nst);
  dup(G.smask, F.smask, (size_t)F.nst);
  dup(G.sbase, F.sbase, (size_t)F.nst);
  // (the tiles of the copy are packed: slot s of the copy = slot s of the original's pool, whose used part is contiguous)
  G.tiles.alloc((size_t)F.ntiles * 256 + 512);
  if (F.ntiles) HIP_CHECK(hipMemcpyAsync(G.tiles.p, F.tiles.p, sizeof(double) * (size_t)F.ntiles * 256, hipMemcpyDeviceToDevice, stream()));
  if (F.have_stat) {
    dup(G.ccount, F.ccount, (size_t)64 * F.ns);
    dup(G.plast, F.plast, (size_t)64 * F.ns);
    G.have_stat = true;
  }
  if (F.have_quads) {
    dup(G.quads, F.quads, (size_t)2 * std::max<int64_t>(1, F.nst));
    G.have_quads = true;
  }
  Out = blocked_matrix(A.cols, A.nnz, std::move(G));
  return true;
}

bool block_dot_trace(const DevMat& A, const DevMat& B, double* dot, double* trace_a) {
  if (A.cols != B.cols || (!A.blocked() && !B.blocked())) return false;
  BlockCache& bc = cache();
  std::shared_ptr<BlockForm> pA = algebra_form(A, bc), pB = algebra_form(B, bc);
  if (!pA || !pB) return false;
  BlockForm &FA = *pA, &FB = *pB;
  const int ns = bc.order->ns;
  const int64_t nc = FA.nst;
  if (nc == 0) { if (dot) *dot = 0.0; if (trace_a) *trace_a = 0.0; return true; }
  DevBuf<int32_t> cj((size_t)nc);
  DevBuf<double> pdot((size_t)2 * nc), part(512), res(2);
  hipLaunchKernelGGL(k_bs_expand_j, dim3(gridw(ns)), dim3(256), 0, stream(), ns, FA.soff.p, cj.p);
  BsMergeArgs a;
  a.soffP = FA.soff.p; a.srowP = FA.srow.p; a.smaskP = FA.smask.p; a.sbaseP = FA.sbase.p; a.tilesP = FA.tiles.p; a.plastP = nullptr;
  a.soffX = nullptr; a.srowX = nullptr; a.smaskX = nullptr; a.sbaseX = nullptr; a.tilesX = nullptr; a.plastX = nullptr;
  a.soffD = FB.soff.p; a.srowD = FB.srow.p; a.smaskD = FB.smask.p; a.sbaseD = FB.sbase.p; a.tilesD = FB.tiles.p;
  a.lab = bc.order->lab.p;
  a.ncand = nc; a.ci = FA.srow.p; a.cj = cj.p; a.cmask = nullptr; a.cbase = nullptr; a.ccnt = nullptr; a.pdot = pdot.p;
  a.pool = nullptr; a.pool_tiles = 0; a.counters = nullptr; a.ccount_out = nullptr; a.plast_out = nullptr;
  a.am = 1.0; a.bm = 0.0; a.thr = 0.0;
  hipLaunchKernelGGL((k_bs_merge<1>), dim3((unsigned)nc), dim3(64), 0, stream(), a);
  hipLaunchKernelGGL(k_bs_sum_pairs, dim3(256), dim3(256), 0, stream(), nc, pdot.p, part.p);
  hipLaunchKernelGGL(k_bs_sum_pairs, dim3(1), dim3(256), 0, stream(), (int64_t)256, part.p, res.p);
  double h[2] = {0, 0};
  {
    ScalarFetch f;
    f.add(res.p, 2, h);
    f.run();
  }
  if (dot) *dot = h[0];
  if (trace_a) *trace_a = h[1];
  return true;
}

bool block_norm(const DevMat& A, double* out) {
  if (!A.blocked() || !algebra_ok(A)) return false;
  const BlockForm& F = *A.blk;
  DevBuf<unsigned long long> mx(1);
  mx.zero();
  hipLaunchKernelGGL(k_bs_colabs_max, dim3(gridw((int64_t)64 * F.ns)), dim3(256), 0, stream(), F.ns, F.soff.p, F.smask.p, F.sbase.p, F.tiles.p, mx.p);
  unsigned long long h = 0;
  {
    ScalarFetch f;
    f.add(mx.p, 1, &h);
    f.run();
  }
  std::memcpy(out, &h, sizeof(double));
  return true;
}

bool spgemm_block(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule, BlockInfo* info,
                  hipEvent_t ev_begin, hipEvent_t ev_end, bool keep_blocked) {
  if (info) *info = BlockInfo();
  if (A.cplx || B.cplx || A.rows != A.cols || B.rows != B.cols || A.cols != B.rows) return false;
  if (A.loose() || A.expanded() || B.loose() || B.expanded()) return false;
  const int32_t n = A.cols;
  if (n < 256 || A.nnz == 0 || B.nnz == 0) return false;
  BlockCache& bc = cache();
  const int force = options().block_path;
  const bool any_blocked = A.blocked() || B.blocked();
  if (!any_blocked && force != 2 && bc.refused_n == n && (double)A.nnz <= 1.5 * (double)bc.refused_nnz && (double)A.nnz >= 0.5 * (double)bc.refused_nnz) return false;
  const double min_fill = force == 2 ? 0.0 : kMinFill;
  // The order of a dimension is made ONCE, from the first operand that is dense enough to say something about the index
  // set (8 entries per column), and kept: the same product gives the same bits whenever it is computed.  An operand that
  // does not tile well in it (fill below kMinFill) goes to the LDS-hash kernels.
  // Which order: an operand already in block form brings its own (a loop's iterates stay in the order they were started
  // in).  Operands in compressed columns take the most recently used order of the dimension -- as long as the left
  // operand tiles in it about as well as the pattern the order was made from; a pattern that does not (another graph
  // on the same index set: fill 0.28 in a foreign clustering against 0.41 in its own) gets an order of ITS OWN, kept
  // beside the first under (dimension, pattern fingerprint).
  if (A.blocked()) bc.order = A.blk->order;
  else if (B.blocked()) bc.order = B.blk->order;
  else if (!select_order(bc, n)) {
    if (A.nnz < 8LL * n) return false;
    install_order(bc, build_block_order(A));
  }
  if (bc.order->ns > kMaxSuperBlocks) return false;
  double fa = 0, fb = 0;
  bool conv = false;
  std::shared_ptr<BlockForm> pFA = operand_form(A, bc, min_fill, &fa, &conv);
  if (!any_blocked && A.nnz >= 8LL * n) {
    if (bc.order->seed_fill <= 0.0 && bc.order->built_from_nnz == A.nnz && pFA) bc.order->seed_fill = fa;   // (the seed itself, first time)
    if (fa < kSeedFillFraction * bc.order->seed_fill || !pFA) {
      const unsigned long long fp = block_pattern_fp(A);
      if (fp != bc.order->seed_fp) {
        const std::shared_ptr<BlockOrder> before = bc.order;
        double f2 = 0;
        std::shared_ptr<BlockForm> p2;
        if (!select_order_fp(bc, n, fp)) install_order(bc, build_block_order(A));
        if (bc.order->ns <= kMaxSuperBlocks) p2 = operand_form(A, bc, min_fill, &f2, &conv);
        if (p2 && bc.order->seed_fill <= 0.0) bc.order->seed_fill = f2;
        if (p2 && f2 > fa) {   // its own order serves it better
          if (dbg()) std::fprintf(stderr, "[block path] pattern %016llx: own order (fill %.3f against %.3f in the order of %016llx)\n", fp, f2, fa, before->seed_fp);
          pFA = p2;
          fa = f2;
        } else {               // (no better: stay with the order in use)
          bc.order = before;
        }
      }
    }
  }
  if (!pFA) {
    bc.refused_n = n;
    bc.refused_nnz = A.nnz;
    if (dbg()) std::fprintf(stderr, "[block path] refused: fill %.3f of A (n %d, %lld entries)\n", fa, n, (long long)A.nnz);
    return false;
  }
  const bool same = &A == &B;
  std::shared_ptr<BlockForm> pFB = pFA;
  if (!same) {
    pFB = operand_form(B, bc, min_fill, &fb, &conv);
    if (!pFB) {
      if (dbg()) std::fprintf(stderr, "[block path] refused: fill %.3f of B\n", fb);
      return false;
    }
  } else {
    fb = fa;
  }
  BlockForm& FA = *pFA;
  BlockForm* FB = pFB.get();
  BlockForm FC;
  unsigned long long nnzC = 0, nprod = 0, hc[4] = {0, 0, 0, 0};
  int64_t ncand = 0;
  const bool count_products = info != nullptr && options().time_kernels != 0;
  block_product(bc, FA, *FB, alpha, threshold, dense_rule, FC, &nnzC, hc, &ncand, ev_begin, ev_end, count_products ? &nprod : nullptr,
                (!A.blocked() && !B.blocked()) ? &A : nullptr, (!A.blocked() && !B.blocked()) ? &B : nullptr);
  if (ncand == 0) {
    C.reset_empty(n, n, false);
    if (info) { info->used = 1; info->fill_a = fa; info->fill_b = fb; }
    return true;
  }
  if (keep_blocked) {
    DevMat R;
    R.rows = n; R.cols = n; R.cplx = false; R.nnz = (int64_t)nnzC; R.zero_free = 1;
    R.blk.reset(new BlockForm(std::move(FC)));
    C = std::move(R);
  } else {
    from_block(FC, (int64_t)nnzC, C);
  }
  if (info) {
    info->used = 1;
    info->fill_a = fa; info->fill_b = fb;
    info->tiles_a = FA.ntiles; info->tiles_b = FB->ntiles; info->tiles_c = (int64_t)hc[0];
    info->cand = ncand;
    info->tile_products = (int64_t)((hc[2] + 3) / 4);   // (matrix instructions issued / 4: slices without entries are skipped)
    info->nnz_c = (int64_t)nnzC;
    info->products = (int64_t)nprod;
  }
  if (dbg())
    std::fprintf(stderr, "[block path] n %d: fill A %.3f B %.3f, tiles %lld x %lld -> %lld (of %lld candidates x 16), %lld tile products, %lld entries\n", n, fa,
                 fb, (long long)FA.ntiles, (long long)FB->ntiles, (long long)hc[0], (long long)ncand, (long long)hc[2], (long long)nnzC);
  return true;
}

}  // namespace ntp
