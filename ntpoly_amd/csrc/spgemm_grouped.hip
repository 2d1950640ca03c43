// Grouped LDS-hash SpGEMM: the numeric kernel for operands WITHOUT run structure -- matrices under a load-balancing
// permutation (LoadBalancerModule.F90:14-52), 3-D Hamiltonians, anything whose columns scatter over the whole row
// range -- where the register-slab kernels do not apply and the one-column-per-wave LDS hash re-fetches every column
// of A once per output column (12 B per product from L2 / Infinity Cache) and pays an LDS compare-and-swap plus a
// read-modify-write per product.
//
// Idea: output columns with similar patterns share almost all of their work.  A workgroup owns a GROUP of G output
// columns (16 real / 8 complex) whose B columns are similar (adjacent columns of a locally ordered matrix, or columns
// brought together by a min-hash signature sort when the ordering hides the similarity).  Walking the UNION of the
// group's B rows k in ascending order (the reference's accumulation order for every output entry,
// sparse_includes/MultiplyBlock.f90:9-36), each column A(:, k) is fetched ONCE per group, its rows are translated to
// dense slot numbers through one LDS hash table shared by the G columns (first touch allocates the next slot), and its
// values are scattered into a slot-indexed vector x in LDS.  From there the kernel is the register-slab kernel again:
// lane l of chunk c owns slot 64 c + l, the partial sums acc[chunk][column] live in VGPRs, x is read back with one
// conflict-free ds_read_b64 per chunk and the G multipliers B(k, column) arrive as SGPRs from a per-group tile
// (zero where a column lacks row k: x + 0 * b = x exactly, as in the slab kernels).  Per product this costs 1/G of a
// hash probe and of an LDS write instead of a CAS and a read-modify-write, and the A column is read once per group.
// The epilogue sorts the (row, slot) pairs of the group once (bitonic, in LDS), gathers every column's sums in row
// order through LDS, prunes (sparse_includes/PruneList.f90:22, strict >) and compacts with ballot + popcount prefix.
// Results are bit-identical to the per-column kernels and to the reference: every C(i, j) sees the same products in the
// same ascending-k order with the same unfused multiply and add.
//
// Groups whose row union outgrows a table class are retried with the next class (512 / 1024 / 1536 slots) and are
// finally handed back to the per-column LDS hash kernel (kernels.hip), so any operand is accepted.
#include "spgemm_grouped.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "device_util.hpp"

namespace ntp {

namespace {

constexpr int GH_KCAP = 4096;   // largest union of B rows per group the tile builder takes
constexpr int GH_KH = 8192;     // its hash set
constexpr int GH_MAXLEN = 1024; // longest column of A a group walks in one step (entries per thread = GH_MAXLEN / threads)

template <typename T>
struct GhG { static constexpr int value = Sc<T>::cplx ? 8 : 16; };

// one step of a group: column k of A (entry range) -- fetched with one scalar load
struct alignas(16) GhRec {
  int64_t start;
  int32_t len;
  int32_t k;
};

__device__ inline unsigned gh_hash(unsigned k) { return k * 2654435761u; }

// ------------------------------------------------------------------ column order
__global__ void k_gh_iota(int32_t* __restrict__ cols, int n, int npad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) cols[i] = i < n ? i : -1;
}

// min-hash signature of every column of B (two hash functions, 32 bits each): columns with similar row sets get
// equal signatures with probability = their Jaccard similarity, so a sort by signature brings them together.
__global__ __launch_bounds__(256) void k_gh_signature(Csc B, unsigned long long* __restrict__ sig,
                                                      int32_t* __restrict__ ids) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= B.cols) return;
  const int lane = lane_id();
  unsigned m1 = 0xffffffffu, m2 = 0xffffffffu;
  for (int64_t p = B.outer[j] + lane; p < B.outer[j + 1]; p += WAVE) {
    const unsigned k = (unsigned)B.inner[p];
    unsigned a = k * 0x9E3779B1u;
    a ^= a >> 15;
    a *= 0x85EBCA6Bu;
    a ^= a >> 13;
    unsigned b = (k ^ 0x5bd1e995u) * 0xC2B2AE35u;
    b ^= b >> 16;
    b *= 0x27D4EB2Fu;
    b ^= b >> 15;
    m1 = min(m1, a);
    m2 = min(m2, b);
  }
  for (int o = 32; o > 0; o >>= 1) {
    m1 = min(m1, (unsigned)__shfl_xor((int)m1, o, WAVE));
    m2 = min(m2, (unsigned)__shfl_xor((int)m2, o, WAVE));
  }
  if (lane == 0) {
    sig[j] = ((unsigned long long)m1 << 32) | m2;
    ids[j] = j;
  }
}

__global__ void k_gh_pad(int32_t* __restrict__ cols, int n, int npad) {
  const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
  if (i < npad) cols[i] = -1;
}

// ---- clusters and the order inside them
// A cluster = a run of equal first signatures in the sorted order (columns that share their min-hash row).  Inside a
// cluster the columns are ordered along a line: ref1 = the cluster's first column, ref2 = the column that shares the
// fewest rows with ref1 (an end of the cluster), and the sort key is the number of rows shared with ref2, descending.
// For a banded matrix under a relabelling this recovers the hidden order inside every cluster exactly (the overlap of
// two columns falls by one per position of distance); in general it is a one-dimensional embedding by Jaccard distance.
__global__ void k_gh_cluster_flags(const unsigned long long* __restrict__ sig_sorted, int32_t* __restrict__ flag, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  flag[p] = (p == 0 || (sig_sorted[p] >> 32) != (sig_sorted[p - 1] >> 32)) ? 1 : 0;
}
// cid[p] = cluster of position p; cstart[c] = first position of cluster c (cstart[nc] = n is written by the caller's fill)
__global__ void k_gh_cluster_ids(const int32_t* __restrict__ flag, const int64_t* __restrict__ excl, int32_t* __restrict__ cid,
                                 int32_t* __restrict__ cstart, unsigned long long* __restrict__ best, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n) return;
  if (p == n) {
    cstart[excl[n]] = n;
    return;
  }
  const int c = (int)excl[p] + flag[p] - 1;
  cid[p] = c;
  if (flag[p]) {
    cstart[c] = p;
    best[c] = ~0ull;
  }
}
// ov[p] = rows shared by the column at position p and its cluster's reference column (ref_pos == nullptr: the cluster's
// first column; else the position stored in the low half of ref_pos[cluster]); optionally tracks the per-cluster minimum
__global__ __launch_bounds__(256) void k_gh_overlap(Csc B, const int32_t* __restrict__ order, const int32_t* __restrict__ cid,
                                                    const int32_t* __restrict__ cstart,
                                                    const unsigned long long* __restrict__ ref_pos, int32_t* __restrict__ ov,
                                                    unsigned long long* __restrict__ best, int n) {
  const int p = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (p >= n) return;
  const int lane = lane_id();
  const int c = cid[p];
  const int rp = ref_pos ? (int)(ref_pos[c] & 0xffffffffu) : cstart[c];
  const int j = order[p], r = order[rp];
  const int64_t rs = B.outer[r], re = B.outer[r + 1];
  int cnt = 0;
  for (int64_t q = B.outer[j] + lane; q < B.outer[j + 1]; q += WAVE) {
    const int k = B.inner[q];
    int64_t lo = rs, hi = re;   // first entry of the reference column that is >= k
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (B.inner[mid] < k) lo = mid + 1;
      else hi = mid;
    }
    cnt += (lo < re && B.inner[lo] == k) ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, WAVE);
  if (lane == 0) {
    ov[p] = cnt;
    if (best) atomicMin(&best[c], ((unsigned long long)(unsigned)cnt << 32) | (unsigned)p);
  }
}
__global__ void k_gh_order_keys(const int32_t* __restrict__ cid, const int32_t* __restrict__ ov,
                                unsigned long long* __restrict__ key, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) key[p] = ((unsigned long long)(unsigned)cid[p] << 32) | (unsigned)(0x7fffffff - ov[p]);
}
// groups never straddle clusters: a cluster of m columns is cut into ceil(m / G) groups of nearly equal size
__global__ void k_gh_cluster_groups(const int32_t* __restrict__ cstart, int32_t* __restrict__ ng, int nc, int G) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nc) ng[c] = (cstart[c + 1] - cstart[c] + G - 1) / G;
}
__global__ void k_gh_fill_groups(const int32_t* __restrict__ order, const int32_t* __restrict__ cid,
                                 const int32_t* __restrict__ cstart, const int64_t* __restrict__ goff,
                                 int32_t* __restrict__ cols, int n, int G) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int c = cid[p];
  const int m = cstart[c + 1] - cstart[c], r = p - cstart[c];
  const int ng = (m + G - 1) / G, base = m / ng, rem = m % ng;   // `rem` groups hold base + 1 columns
  int b, i;
  if (r < rem * (base + 1)) {
    b = r / (base + 1);
    i = r % (base + 1);
  } else {
    const int r2 = r - rem * (base + 1);
    b = rem + r2 / base;
    i = r2 % base;
  }
  cols[((int64_t)goff[c] + b) * G + i] = order[p];
}
__global__ void k_gh_fill_i32(int32_t* __restrict__ a, int64_t n, int32_t v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = v;
}

// ------------------------------------------------------------------ union of the B rows of a group, multiplier tile
// FILL = false: grp_kn[g] = |union| (-1 when it exceeds GH_KCAP), grp_maxlen[g] = longest A column over the union.
// FILL = true : recs[off + t] = step t (k ascending), tile[(off + t) * G + c] = B(k_t, column c of the group) or 0.
template <typename T, bool FILL>
__global__ __launch_bounds__(256) void k_gh_union(Csc A, Csc B, const int32_t* __restrict__ cols,
                                                  int32_t* __restrict__ grp_kn, int32_t* __restrict__ grp_maxlen,
                                                  const int64_t* __restrict__ grp_off, GhRec* __restrict__ recs,
                                                  T* __restrict__ tiles, int ngroups) {
  constexpr int G = GhG<T>::value;
  __shared__ int hk[GH_KH];
  __shared__ int uk[FILL ? GH_KCAP : 1];
  __shared__ int ctl[4];
  const int gi = xcd_block(ngroups);
  if (gi < 0) return;
  if (FILL && grp_kn[gi] <= 0) return;
  const int tid = threadIdx.x, wave = tid / WAVE, lane = lane_id();
  for (int s = tid; s < GH_KH; s += 256) hk[s] = -1;
  if (tid < 4) ctl[tid] = 0;
  __syncthreads();
  for (int c = wave; c < G; c += 4) {
    const int col = cols[gi * G + c];
    if (col < 0) continue;
    const int64_t s = B.outer[col], e = B.outer[col + 1];
    for (int64_t p0 = s; p0 < e; p0 += WAVE) {
      if (__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) > GH_KCAP) break;
      const int64_t p = p0 + lane;
      bool fresh = false;
      const int k = p < e ? B.inner[p] : -1;
      // (a row of B whose column of A is empty is no step: nothing to multiply -- row strips of A, kernels.hip
      // spgemm_striped, leave most columns of a strip empty)
      if (k >= 0 && A.outer[k + 1] > A.outer[k]) {
        unsigned h = (gh_hash((unsigned)k) >> 19) & (GH_KH - 1);
        for (;;) {
          const int old = atomicCAS(&hk[h], -1, k);
          if (old == -1) { fresh = true; break; }
          if (old == k) break;
          h = (h + 1) & (GH_KH - 1);
        }
      }
      const int nf = __popcll(__ballot(fresh));
      if (lane == 0 && nf) atomicAdd(&ctl[0], nf);
    }
  }
  __syncthreads();
  const int kn = ctl[0];
  if (!FILL) {
    if (kn > GH_KCAP) {
      if (tid == 0) {
        grp_kn[gi] = -1;
        grp_maxlen[gi] = 0;
      }
      return;
    }
    int mx = 0;
    for (int s = tid; s < GH_KH; s += 256) {
      const int k = hk[s];
      if (k >= 0) mx = max(mx, (int)(A.outer[k + 1] - A.outer[k]));
    }
    mx = wave_max_i32(mx);
    if (lane == 0) atomicMax(&ctl[1], mx);
    __syncthreads();
    if (tid == 0) {
      grp_kn[gi] = kn;
      grp_maxlen[gi] = ctl[1];
    }
    return;
  } else {
    // unique rows -> uk (any order), then sorted
    for (int s0 = 0; s0 < GH_KH; s0 += 256) {
      const int k = hk[s0 + tid];
      const bool occ = k >= 0;
      const unsigned long long m = __ballot(occ);
      int base = 0;
      if (lane == 0 && m) base = atomicAdd(&ctl[2], __popcll(m));
      base = __shfl(base, 0, WAVE);
      if (occ) uk[base + __popcll(m & lanemask_lt())] = k;
    }
    int p2 = 1;
    while (p2 < kn) p2 <<= 1;
    __syncthreads();
    for (int s = kn + tid; s < p2; s += 256) uk[s] = INT_MAX;
    __syncthreads();
    for (int kk = 2; kk <= p2; kk <<= 1) {
      for (int jj = kk >> 1; jj > 0; jj >>= 1) {
        for (int t = tid; t < p2; t += 256) {
          const int ixj = t ^ jj;
          if (ixj > t) {
            const int x = uk[t], y = uk[ixj];
            const bool up = (t & kk) == 0;
            if ((x > y) == up) {
              uk[t] = y;
              uk[ixj] = x;
            }
          }
        }
        __syncthreads();
      }
    }
    const int64_t off = grp_off[gi];
    for (int t = tid; t < kn; t += 256) {
      const int k = uk[t];
      GhRec r;
      r.start = A.outer[k];
      r.len = (int32_t)(A.outer[k + 1] - r.start);
      r.k = k;
      recs[off + t] = r;
    }
    T* __restrict__ tile = tiles + off * G;
    for (int i = tid; i < kn * G; i += 256) tile[i] = Sc<T>::zero();
    __threadfence_block();
    __syncthreads();
    const T* __restrict__ Bv = static_cast<const T*>(B.val);
    for (int c = wave; c < G; c += 4) {
      const int col = cols[gi * G + c];
      if (col < 0) continue;
      const int64_t s = B.outer[col], e = B.outer[col + 1];
      for (int64_t p = s + lane; p < e; p += WAVE) {
        const int k = B.inner[p];
        if (A.outer[k + 1] <= A.outer[k]) continue;   // (no step: see the union above)
        int lo = 0, hi = kn - 1;  // position of k in the sorted union
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (uk[mid] < k) lo = mid + 1;
          else hi = mid;
        }
        tile[(int64_t)lo * G + c] = Bv[p];
      }
    }
  }
}

// The tile builder proper (the FILL = true branch above is kept for unions beyond 1024 rows): hash set sized for the
// union, positions looked up through the same table instead of a binary search.  KH buckets, unions of at most KH / 2.
template <typename T, int KH>
__global__ __launch_bounds__(256) void k_gh_fill(Csc A, Csc B, const int32_t* __restrict__ cols,
                                                 const int32_t* __restrict__ grp_kn, const int64_t* __restrict__ grp_off,
                                                 GhRec* __restrict__ recs, T* __restrict__ tiles, int ngroups,
                                                 unsigned long long* __restrict__ nproducts) {
  constexpr int G = GhG<T>::value, KMAX = KH / 2;
  __shared__ int hk[KH];      // row ids
  __shared__ int hp[KH];      // position of the bucket's row in the sorted union
  __shared__ unsigned long long uk[KMAX];   // (row << 32 | bucket), sorted by row
  __shared__ int ctl[2];
  const int gi = xcd_block(ngroups);
  if (gi < 0) return;
  const int kn = grp_kn[gi];
  if (kn <= 0) return;
  const int tid = threadIdx.x, wave = tid / WAVE, lane = lane_id();
  for (int s = tid; s < KH; s += 256) hk[s] = -1;
  if (tid < 2) ctl[tid] = 0;
  __syncthreads();
  constexpr int SH = KH == 2048 ? 21 : 19;
  for (int c = wave; c < G; c += 4) {
    const int col = cols[gi * G + c];
    if (col < 0) continue;
    const int64_t s = B.outer[col], e = B.outer[col + 1];
    for (int64_t p = s + lane; p < e; p += WAVE) {
      const int k = B.inner[p];
      if (A.outer[k + 1] <= A.outer[k]) continue;   // (an empty column of A is no step, as in k_gh_union)
      unsigned h = gh_hash((unsigned)k) >> SH;
      for (;;) {
        const int old = atomicCAS(&hk[h], -1, k);
        if (old == -1 || old == k) break;
        h = (h + 1) & (KH - 1);
      }
    }
  }
  __syncthreads();
  for (int s0 = 0; s0 < KH; s0 += 256) {
    const int k = hk[s0 + tid];
    const bool occ = k >= 0;
    const unsigned long long m = __ballot(occ);
    int base = 0;
    if (lane == 0 && m) base = atomicAdd(&ctl[0], __popcll(m));
    base = __shfl(base, 0, WAVE);
    if (occ) uk[base + __popcll(m & lanemask_lt())] = ((unsigned long long)(unsigned)k << 32) | (unsigned)(s0 + tid);
  }
  int p2 = 64;
  while (p2 < kn) p2 <<= 1;
  __syncthreads();
  for (int s = kn + tid; s < p2; s += 256) uk[s] = ~0ull;
  __syncthreads();
  for (int kk = 2; kk <= p2; kk <<= 1) {
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < p2; t += 256) {
        const int ixj = t ^ jj;
        if (ixj > t) {
          const unsigned long long x = uk[t], y = uk[ixj];
          const bool up = (t & kk) == 0;
          if ((x > y) == up) {
            uk[t] = y;
            uk[ixj] = x;
          }
        }
      }
      __syncthreads();
    }
  }
  const int64_t off = grp_off[gi];
  for (int t = tid; t < kn; t += 256) {
    const unsigned long long e = uk[t];
    const int k = (int)(e >> 32);
    hp[(int)(e & 0xffffffffu)] = t;
    GhRec r;
    r.start = A.outer[k];
    r.len = (int32_t)(A.outer[k + 1] - r.start);
    r.k = k;
    recs[off + t] = r;
  }
  T* __restrict__ tile = tiles + off * G;
  for (int i = tid; i < kn * G; i += 256) tile[i] = Sc<T>::zero();
  __threadfence_block();
  __syncthreads();
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  long long np = 0;   // intermediate products of this group's columns (statistics)
  for (int c = wave; c < G; c += 4) {
    const int col = cols[gi * G + c];
    if (col < 0) continue;
    const int64_t s = B.outer[col], e = B.outer[col + 1];
    for (int64_t p = s + lane; p < e; p += WAVE) {
      const int k = B.inner[p];
      if (A.outer[k + 1] <= A.outer[k]) continue;
      unsigned h = gh_hash((unsigned)k) >> SH;
      while (hk[h] != k) h = (h + 1) & (KH - 1);
      tile[(int64_t)hp[h] * G + c] = Bv[p];
      np += A.outer[k + 1] - A.outer[k];
    }
  }
  np = wave_sum_i64(np);
  if (lane == 0 && np) atomicAdd(nproducts, (unsigned long long)np);
}

// groups whose union is empty need no numeric work; unusable ones (kn < 0) stay "to do" and end in the fallback
__global__ void k_gh_init_state(const int32_t* __restrict__ grp_kn, int32_t* __restrict__ kn_pos,
                                uint8_t* __restrict__ state, unsigned long long* __restrict__ nbad, int ngroups) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const int kn = grp_kn[g];
  kn_pos[g] = kn > 0 ? kn : 0;
  state[g] = kn == 0 ? 1 : 0;
  if (kn < 0) atomicAdd(nbad, 1ull);
  if (kn > 0) atomicMax(nbad + 1, (unsigned long long)kn);
}

// ------------------------------------------------------------------ numeric kernel
// One workgroup (NW waves) per group.  The walk over the union of the B rows is cut into PHASES of KB = NW / WPC
// consecutive steps: WPC waves share one column of A (a wave takes every WPC-th chunk of 64 entries), so the KB columns
// of a phase are hashed and scattered concurrently and ONE barrier, one round of load latencies and one round of hash
// probes serve KB steps.  Software pipeline, iteration i:
//   L(i+2)  request the first PF chunks of this wave's share of its column of phase i+2 (registers)
//   S(i+1)  hash the rows of phase i+1 (requested one iteration ago) to slots, scatter the values into x[set(i+1)]
//   F(i)    products of phase i: for its steps in ascending k, every owned chunk of slots reads x[set(i)], multiplies
//           with the G multipliers (SGPRs: one scalar load of the tile row, requested one step ahead) into the register
//           sums and writes zeros back (the reader owns the slot, so the two sets need no other cleaning)
//   barrier
template <int CAP, int ELEM>
struct GhTable {
  // entries of the table: four per slot (load <= 1/4), in buckets of two read together -- with linear probing over
  // single entries at load 1/2 the slowest of the 64 lanes of a probe needed 4-6 rounds of LDS latency, and the
  // lookups were 45 % of the kernel (profiles/README.md item 28).  (Complex values at the largest class: LDS allows
  // load 3/8 only.)
  static constexpr int TH = CAP <= 512 ? 2048 : CAP <= 1024 ? 4096 : (ELEM > 8 ? 4096 : 8192);
  static constexpr int SHIFT = TH == 2048 ? 22 : TH == 4096 ? 21 : 20;   // top bits of the multiplicative hash -> bucket (TH / 2 of them)
};

// MF (real operands, FMA arithmetic): the products of a phase on the FP64 matrix cores.  The four steps of a phase are the k
// dimension of ONE v_mfma_f64_16x16x4_f64 per tile of 16 slots: A operand = x[step][slot] straight from the slot-indexed LDS
// copies (lane l: slot 16 tile + l % 16 of step l / 16), B operand = the phase's four multiplier rows (lane l: row l / 16,
// column l % 16 of the group's tile, one coalesced load per phase), accumulator = the sums of 16 slots x 16 columns.  The matrix
// instruction adds its four products in ascending k with one rounding each -- the chain of fma() the vector path computes
// (tools/micro/mfma_f64_probe.hip), so the results are bit for bit the same; a slot a column does not touch contributes
// fma(0, b, acc) = acc.  A lane then owns (4 slots x 1 column) per tile instead of (1 slot x 16 columns) per chunk; the
// epilogue is the same prune / rank-bitmap / compaction re-indexed.
template <typename T, int NW, int SL, int WPC, bool MF = false>
__global__ __launch_bounds__(NW* WAVE) __attribute__((amdgpu_waves_per_eu((SL == 1 && !Sc<T>::cplx) ? 6 : 2))) void k_spgemm_ghash(
    Csc A, const int32_t* __restrict__ cols, const int32_t* __restrict__ grp_kn, const int32_t* __restrict__ grp_maxlen,
    const int64_t* __restrict__ grp_off, const GhRec* __restrict__ recs, const T* __restrict__ tiles,
    const int64_t* __restrict__ tmpoff, int32_t* __restrict__ out_inner, T* __restrict__ out_val,
    int32_t* __restrict__ count, uint8_t* __restrict__ grp_state, unsigned long long* __restrict__ stats, double alpha,
    double threshold, int dense_rule, int ngroups, int ablate) {
  constexpr int G = GhG<T>::value;
  constexpr int NT = NW * WAVE, CAP = NT * SL, TH = GhTable<CAP, (int)sizeof(T)>::TH, SHIFT = GhTable<CAP, (int)sizeof(T)>::SHIFT;
  constexpr int KB = NW / WPC;   // steps per phase
  constexpr int PF = (MF && WPC == 2) ? 3 : 2;   // chunks of a wave's share requested a phase ahead (two waves per column: 3 x 128 entries)
  constexpr unsigned long long EMPTY = ~0ull;
  static_assert(NW % WPC == 0 && KB >= 1 && 2 * KB * CAP >= 2 * CAP, "geometry");
  __shared__ __attribute__((aligned(16))) unsigned long long htab[TH];   // (row << 32 | slot); the epilogue sorts (row, slot) pairs in the same memory
  static_assert(!MF || (!Sc<T>::cplx && KB == 4 && CAP % (16 * NW) == 0), "matrix-core products: real operands, four steps per phase");
  constexpr int XP = MF ? CAP + 16 : CAP;   // (MF: rows 16 slots apart in the banks -- the four steps a lane group reads do not collide)
  constexpr int NTILE = CAP / (16 * NW);    // MF: tiles of 16 slots per wave (tile u of wave w = slots 16 (u NW + w) ..)
  __shared__ T xbuf[2][KB][XP];             // slot-indexed copies of the A columns of two consecutive phases
  __shared__ int slot_row[CAP];
  __shared__ int ctl[4];                    // [0] slots handed out, [1], [2] overflow seen while scattering an even / odd
                                            // phase (read after the barrier that ends that scatter, rewritten two
                                            // barriers later: every thread reads the same value), [3] valid rows (epilogue)
  const int gi = xcd_block(ngroups);
  if (gi < 0) return;
  if (grp_state[gi] != 0) return;
  const int kn = grp_kn[gi];
  if (kn <= 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  if (grp_maxlen[gi] > GH_MAXLEN) {  // a column of A too long for the scatter of one phase: the group goes to the fallback
    if (tid == 0) atomicAdd(&stats[0], 1ull);
    return;
  }
  const int nph = (kn + KB - 1) / KB;
  for (int s = tid; s < TH; s += NT) htab[s] = EMPTY;
  for (int s = tid; s < 2 * KB * XP; s += NT) (&xbuf[0][0][0])[s] = Sc<T>::zero();
  for (int s = tid; s < CAP; s += NT) slot_row[s] = -1;
  if (tid < 4) ctl[tid] = 0;
  __syncthreads();

  const int32_t* __restrict__ Ai = A.inner;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const int64_t off = grp_off[gi];
  const GhRec* __restrict__ rec = recs + off;
  const T* __restrict__ tile = tiles + off * G;
  const int my_kb = wave / WPC, my_part = wave % WPC;   // this wave's column of a phase and its share of it

  T acc[MF ? 1 : SL][MF ? 1 : G];
  typedef double mf_v4d __attribute__((ext_vector_type(4)));
  [[maybe_unused]] mf_v4d macc[MF ? NTILE : 1];   // MF: element v of tile u = slot 16 (u NW + wave) + 4 v + lane / 16, column lane % 16
  if constexpr (MF) {
#pragma unroll
    for (int u = 0; u < NTILE; ++u) macc[u] = mf_v4d{0.0, 0.0, 0.0, 0.0};
  } else {
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[s][g] = Sc<T>::zero();
  }

  struct Fetch {
    int idx[PF];
    T val[PF];
    int64_t start;   // this wave's column of the phase (wave-uniform), kept so that the scatter need not load it again
    int len;
  };
  // L: first PF chunks of this wave's share of column (ph, my_kb).  The step record comes in through the vector path
  // (requested one phase earlier as `rnext`, lane 0) -- a scalar load here would be a cold miss on the critical path.
  int64_t rn_start = 0;   // record of this wave's column of the NEXT phase to be loaded (lane 0 holds it)
  int rn_len = 0;
  auto request_rec = [&](int ph) {
    const int t = ph * KB + my_kb;
    rn_start = 0;
    rn_len = 0;
    if (t < kn && lane == 0) {
      rn_start = rec[t].start;
      rn_len = rec[t].len;
    }
  };
  long long warm = 0;   // keeps the cache warm-up loads alive (never stored)
  auto load_phase = [&](int ph, Fetch& f) {
    const int64_t start = readlane_i64(rn_start, 0);
    const int len = readlane_i32(rn_len, 0);
    request_rec(ph + 1);
    // the multiplier rows of that phase are read with scalar loads later: pull them into L2 now
    if (wave == 0 && lane < KB * G * (int)sizeof(T) / 8) {
      const int64_t w = (int64_t)ph * KB * G * (int)sizeof(T) / 8 + lane;
      if (w < (int64_t)kn * G * (int)sizeof(T) / 8) warm += reinterpret_cast<const long long*>(tile)[w];
    }
    f.start = start;
    f.len = len;
#pragma unroll
    for (int c = 0; c < PF; ++c) {
      const int q = (c * WPC + my_part) * WAVE + lane;
      f.idx[c] = -1;
      f.val[c] = Sc<T>::zero();
      if (q < len) {
        f.idx[c] = Ai[start + q];
        f.val[c] = Av[start + q];
      }
    }
  };
  // row -> slot (first touch allocates the next one); -1 when the table class is exhausted.  Insertion is two-step so
  // that concurrent first touches of one row (the KB columns of a phase overlap heavily) never waste a slot: the bucket
  // is claimed with the slot field PENDING, the winner then draws the slot number and publishes it; whoever meets a
  // pending entry of its own row re-reads the bucket (the winner finishes inside the same loop iteration, so lanes of
  // one wave cannot wait on each other forever).
  constexpr unsigned PENDING = 0x7fffffffu, NOSLOT = 0x7ffffffeu;
  auto slot_of = [&](int i, int par) -> int {
    if (ablate & 16) return (int)(gh_hash((unsigned)i) >> SHIFT) & (CAP - 1);   // (timing experiment: no table, wrong results)
    constexpr unsigned NBUCK = TH / 2;
    unsigned b = gh_hash((unsigned)i) >> SHIFT;
    for (;;) {
      const unsigned long long e0 = __hip_atomic_load(&htab[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned long long e1 = __hip_atomic_load(&htab[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const bool m0 = (int)(e0 >> 32) == i, m1 = (int)(e1 >> 32) == i;
      if (m0 || m1) {
        const unsigned sl = (unsigned)((m0 ? e0 : e1) & 0xffffffffu);
        if (sl == PENDING) continue;
        return sl == NOSLOT ? -1 : (int)sl;
      }
      if (e0 == EMPTY || e1 == EMPTY) {   // not in the table: claim the first free entry of the bucket
        const unsigned at = 2 * b + (e0 == EMPTY ? 0u : 1u);
        const unsigned long long claim = ((unsigned long long)(unsigned)i << 32) | PENDING;
        const unsigned long long old = atomicCAS(&htab[at], EMPTY, claim);
        if (old == EMPTY) {
          const int mine = atomicAdd(&ctl[0], 1);
          const bool ok = mine < CAP;
          if (ok) slot_row[mine] = i;
          else ctl[1 + par] = 1;
          __hip_atomic_store(&htab[at], ((unsigned long long)(unsigned)i << 32) | (ok ? (unsigned)mine : NOSLOT), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_WORKGROUP);
          return ok ? mine : -1;
        }
        continue;   // somebody else took that entry (the same row, or another one): look at the bucket again
      }
      b = (b + 1) & (NBUCK - 1);
    }
  };
  // S: values of column (ph, my_kb) into xbuf[set][my_kb]
  auto scatter_phase = [&](int ph, int set, const Fetch& f) {
    const int par = ph & 1;
#pragma unroll
    for (int c = 0; c < PF; ++c) {
      const int i = f.idx[c];
      if (i >= 0) {
        const int slot = slot_of(i, par);
        if (slot >= 0) xbuf[set][my_kb][slot] = f.val[c];
      }
    }
    // the rest of a long column: loaded here
    const int64_t start = uni_i64(f.start);
    const int len = uni_i32(f.len);
    for (int q0 = (PF * WPC + my_part) * WAVE; q0 < len; q0 += WPC * WAVE) {
      const int q = q0 + lane;
      if (q < len) {
        const int slot = slot_of(Ai[start + q], par);
        if (slot >= 0) xbuf[set][my_kb][slot] = Av[start + q];
      }
    }
  };

  Fetch fc, fn;
  request_rec(0);
  load_phase(0, fc);
  load_phase(1, fn);
  scatter_phase(0, 0, fc);
  __syncthreads();
  bool overflow = ctl[1] != 0;   // (phase 0 is even)
  // (ablate & 8: in-kernel stamps of wave 0 -- cycles spent in the load / scatter / product / barrier parts, summed into stats[8..12])
  unsigned long long st_load = 0, st_scat = 0, st_fp = 0, st_bar = 0;
  for (int ph = 0; ph < nph && !overflow; ++ph) {
    const int set = ph & 1;
    unsigned long long c0s = 0, c1s = 0, c2s = 0, c3s = 0;
    if (ablate & 8) c0s = __builtin_amdgcn_s_memtime();
    fc = fn;
    load_phase(ph + 2, fn);
    if (ablate & 8) c1s = __builtin_amdgcn_s_memtime();
    if (ph + 1 < nph && !(ablate & 2)) scatter_phase(ph + 1, set ^ 1, fc);
    if (ablate & 8) c2s = __builtin_amdgcn_s_memtime();
    // products of phase ph: chunks that hold slots handed out before the last barrier
    const int nsl = uni_i32(min(__hip_atomic_load(&ctl[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), CAP));
    const int t0 = ph * KB;
    const int nstep = min(KB, kn - t0);
    if constexpr (MF) {
      // this lane's multiplier: row t0 + lane / 16, column lane % 16 of the tile (rows beyond the union: zero, whatever the
      // padding of the tile holds)
      const int q = lane >> 4, jj = lane & 15;
      double bv = 0.0;
      if constexpr (!Sc<T>::cplx) bv = (q < nstep) ? tile[(int64_t)(t0 + q) * G + jj] : 0.0;
      // (tried and measured slower by 8 %: all of the wave's tiles read up front, the two entries of a bucket in one 16-byte read)
#pragma unroll
      for (int u = 0; u < NTILE; ++u) {
        const int s0 = 16 * (u * NW + wave);
        if (s0 < nsl) {
          if constexpr (!Sc<T>::cplx) {
            const double xv = xbuf[set][q][s0 + jj];
            if (__ballot(xv != 0.0) != 0ull) {
              xbuf[set][q][s0 + jj] = 0.0;
              macc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, bv, macc[u], 0, 0, 0);
            }
          }
        }
      }
    } else {
    // the multipliers (wave-uniform: scalar registers) are fetched two rows per round trip of the scalar cache
    for (int kb = 0; kb < nstep; kb += 2) {
      const T* __restrict__ brow = tile + (int64_t)(t0 + kb) * G;
      T b0[G], b1[G];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        b0[g] = brow[g];
        b1[g] = brow[G + g];   // (the tile is padded by 8 rows: the row after the last one exists)
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        if (kb + half < nstep) {
#pragma unroll
          for (int s = 0; s < SL; ++s) {
            const int c0 = (wave + NW * s) * WAVE;
            if ((c0 < nsl && !(ablate & 1)) || (ablate & 3) == 2) {   // (ablate: timing experiments, wrong results)
              const T xv = xbuf[set][kb + half][c0 + lane];
              if (__ballot(!Sc<T>::is_zero(xv)) != 0ull || (ablate & 3) == 2) {
                xbuf[set][kb + half][c0 + lane] = Sc<T>::zero();
#pragma unroll
                for (int g = 0; g < G; ++g) acc[s][g] = Sc<T>::fmadd(xv, half ? b1[g] : b0[g], acc[s][g], (dense_rule & 2) != 0);
              }
            }
          }
        }
      }
    }
    }
    if (ablate & 8) c3s = __builtin_amdgcn_s_memtime();
    __syncthreads();
    overflow = ctl[1 + ((ph + 1) & 1)] != 0;
    if (ablate & 8) {
      st_load += c1s - c0s;
      st_scat += c2s - c1s;
      st_fp += c3s - c2s;
      st_bar += __builtin_amdgcn_s_memtime() - c3s;
    }
  }
  const unsigned long long epi0 = (ablate & 8) ? __builtin_amdgcn_s_memtime() : 0ull;
  if ((ablate & 8) && tid == 0) {
    atomicAdd(&stats[8], st_load);
    atomicAdd(&stats[9], st_scat);
    atomicAdd(&stats[10], st_fp);
    atomicAdd(&stats[11], st_bar);
    atomicAdd(&stats[12], (unsigned long long)nph);
  }
  if (warm == 0x7fffffffffffffffll && kn < 0) stats[7] = (unsigned long long)warm;   // (never true: keeps the warm-up loads)
  if (overflow) {  // the row union outgrew this table class: the group stays "to do" for the next one
    if (tid == 0) atomicAdd(&stats[0], 1ull);
    return;
  }

  if (ablate & 4) {
    if (tid == 0) grp_state[gi] = 1;
    return;
  }
  // ---- epilogue.  The (row, slot) pairs of the group are sorted once (bitonic in LDS; stages whose partners lie in
  // the same 64 elements are private to a wave and need no workgroup barrier), which gives every slot its rank in row
  // order.  The sums never leave their registers: for every column the owner of a slot sets a bit at the slot's RANK in
  // a per-column bitmap when the entry survives the prune rule (PruneList.f90:22), a prefix over the bitmap words turns
  // a rank into the output position, and the owner writes row and value there -- the column comes out sorted.
  const int nsl = min(ctl[0], CAP);
  int p2 = 64;
  while (p2 < nsl) p2 <<= 1;
  unsigned long long* skey = htab;
  constexpr int NWORD = CAP / 64;
  int* rank_s = reinterpret_cast<int*>(&xbuf[0][0][0]);                                  // [CAP]
  unsigned long long* bm = reinterpret_cast<unsigned long long*>(rank_s + CAP);          // [G][NWORD]
  int* pre = reinterpret_cast<int*>(bm + G * NWORD);                                     // [G][NWORD]
  long long* colbase = reinterpret_cast<long long*>(pre + G * NWORD);                    // [G]
  static_assert(sizeof(int) * CAP + (sizeof(unsigned long long) + sizeof(int)) * G * NWORD + sizeof(long long) * G <=
                    sizeof(T) * 2 * KB * CAP, "epilogue arrays fit the x buffers");
  __syncthreads();
  for (int s = tid; s < p2; s += NT) {
    const int row = s < nsl ? slot_row[s] : -1;
    skey[s] = row >= 0 ? (((unsigned long long)(unsigned)row << 32) | (unsigned)s) : EMPTY;
  }
  for (int s = tid; s < G * NWORD; s += NT) bm[s] = 0ull;
  if (tid < G) {
    const int col = cols[gi * G + tid];
    colbase[tid] = col >= 0 ? tmpoff[col] : 0;
  }
  __syncthreads();
  for (int kk = 2; kk <= p2; kk <<= 1) {
    if ((kk >> 1) > 32) __syncthreads();   // the wave-private stages of the previous round are read across waves now
    for (int jj = kk >> 1; jj > 0; jj >>= 1) {
      for (int t = tid; t < p2; t += NT) {
        const int ixj = t ^ jj;
        if (ixj > t) {
          const unsigned long long x = skey[t], y = skey[ixj];
          const bool up = (t & kk) == 0;
          if ((x > y) == up) {
            skey[t] = y;
            skey[ixj] = x;
          }
        }
      }
      if (jj > 32) __syncthreads();                                   // partners in other waves' elements
      else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");     // same wave: its LDS writes before its next reads
    }
  }
  __syncthreads();
  for (int r = tid; r < nsl; r += NT) {
    const unsigned long long e = skey[r];
    if (e != EMPTY) rank_s[(int)(e & 0xffffffffu)] = r;
  }
  if (tid == 0) atomicMax(&stats[1], (unsigned long long)nsl);
  __syncthreads();
  unsigned long long keepbits = 0;   // bit s * G + g (MF: bit 4 u + v)
  int myrank[SL], myrow[SL];
  if constexpr (MF) {
    // (a lane's sums: slots 16 (u NW + wave) + 4 v + lane / 16 of column lane % 16)
    const int q = lane >> 4, g = lane & 15;
#pragma unroll
    for (int u = 0; u < NTILE; ++u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int slot = 16 * (u * NW + wave) + 4 * v + q;
        const int row = slot < nsl ? slot_row[slot] : -1;
        if (row >= 0) {
          const double val = macc[u][v];
          if (fabs((dense_rule & 1) ? val : __dmul_rn(alpha, val)) > threshold) {
            const int rk = rank_s[slot];
            keepbits |= 1ull << (4 * u + v);
            atomicOr(&bm[g * NWORD + (rk >> 6)], 1ull << (rk & 63));
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int s = 0; s < SL; ++s) {
    const int slot = (wave + NW * s) * WAVE + lane;
    myrow[s] = slot < nsl ? slot_row[slot] : -1;
    myrank[s] = myrow[s] >= 0 ? rank_s[slot] : 0;
    if (myrow[s] >= 0) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const T v = acc[s][g];
        if (Sc<T>::mag((dense_rule & 1) ? v : Sc<T>::scale(alpha, v)) > threshold) {
          keepbits |= 1ull << (s * G + g);
          atomicOr(&bm[g * NWORD + (myrank[s] >> 6)], 1ull << (myrank[s] & 63));
        }
      }
    }
  }
  }
  __syncthreads();
  for (int i = tid; i < G * NWORD; i += NT) {
    const int g = i / NWORD, w = i % NWORD;
    int run = 0;
    for (int w2 = 0; w2 < w; ++w2) run += __popcll(bm[g * NWORD + w2]);
    pre[i] = run;
    if (w == NWORD - 1) {
      const int col = cols[gi * G + g];
      if (col >= 0) count[col] = run + __popcll(bm[i]);
    }
  }
  __syncthreads();
  if constexpr (MF) {
    const int q = lane >> 4, g = lane & 15;
#pragma unroll
    for (int u = 0; u < NTILE; ++u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if ((keepbits >> (4 * u + v)) & 1ull) {
          const int slot = 16 * (u * NW + wave) + 4 * v + q;
          const int rk = rank_s[slot];
          const int w = rk >> 6, bit = rk & 63;
          const int64_t pos = colbase[g] + pre[g * NWORD + w] + __popcll(bm[g * NWORD + w] & ((1ull << bit) - 1ull));
          out_inner[pos] = slot_row[slot];
          if constexpr (!Sc<T>::cplx) out_val[pos] = __dmul_rn(alpha, macc[u][v]);
        }
      }
    }
  } else {
#pragma unroll
  for (int s = 0; s < SL; ++s) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if ((keepbits >> (s * G + g)) & 1ull) {
        const int w = myrank[s] >> 6, bit = myrank[s] & 63;
        const int64_t pos = colbase[g] + pre[g * NWORD + w] + __popcll(bm[g * NWORD + w] & ((1ull << bit) - 1ull));
        out_inner[pos] = myrow[s];
        out_val[pos] = Sc<T>::scale(alpha, acc[s][g]);
      }
    }
  }
  }
  if ((ablate & 8) && tid == 0) {
    atomicAdd(&stats[13], __builtin_amdgcn_s_memtime() - epi0);
    atomicAdd(&stats[14], 1ull);
  }
  if (tid == 0) grp_state[gi] = 1;
}

// columns of finished groups are DONE, the others go to the per-column LDS hash
template <int G>
__global__ void k_gh_finish(const int32_t* __restrict__ cols, const uint8_t* __restrict__ grp_state,
                            uint8_t* __restrict__ bin_arr, int32_t* __restrict__ count,
                            unsigned long long* __restrict__ stats, int ngroups) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ngroups * G) return;
  const int col = cols[i];
  if (col < 0) return;
  const bool done = grp_state[i / G] == 1;
  bin_arr[col] = done ? GH_BIN_DONE : GH_BIN_HASH;
  if (!done) {
    count[col] = 0;
    atomicAdd(&stats[2], 1ull);
    if (i % G == 0) atomicAdd(&stats[3], 1ull);
  }
}

template <typename T, int NW, int SL, int WPC, bool MF = false>
void launch_ghash(const DevMat& A, int ngroups, const int32_t* cols, const int32_t* grp_kn, const int32_t* grp_maxlen,
                  const int64_t* grp_off, const GhRec* recs, const double* tiles, const int64_t* tmpoff, int32_t* tmp_inner,
                  double* tmp_val, int32_t* count, uint8_t* state, unsigned long long* stats, double alpha, double thr, int dr) {
  const int sv = options().spgemm_variant;
#ifdef NTP_ABLATIONS
  const int ablate = (sv >= 511 && sv <= 541) ? sv - 510 : 0;   // bits: 1 no products, 2 no hashing / scatter, 4 no epilogue
#else
  // (the wrong-result experiment bits exist only in the experiment build, -DNTP_ABLATIONS; 518 = in-kernel stamps stays)
  const int ablate = (sv == 518) ? 8 : 0;
#endif
  hipLaunchKernelGGL((k_spgemm_ghash<T, NW, SL, WPC, MF>), dim3(xcd_grid(ngroups)), dim3(NW * WAVE), 0, stream(), view(A), cols, grp_kn,
                     grp_maxlen, grp_off, recs, reinterpret_cast<const T*>(tiles), tmpoff, tmp_inner,
                     reinterpret_cast<T*>(tmp_val), count, state, stats, alpha, thr, dr, ngroups, ablate);
}

}  // namespace

bool spgemm_grouped(const DevMat& A, const DevMat& B, const int64_t* tmpoff, int32_t* tmp_inner, double* tmp_val,
                    int32_t* count, uint8_t* bin_arr, double alpha, double threshold, int dense_rule, int mode,
                    GroupedInfo* info, hipEvent_t numeric_begin) {
  const bool force = mode == 1;
  const int n = B.cols;
  const int G = A.cplx ? 8 : 16;
  int ngroups = cdiv(n, G);
  GroupedInfo gi;
  DevBuf<int32_t> cols, grp_kn, grp_maxlen, kn_pos;
  DevBuf<int64_t> grp_off;
  DevBuf<uint8_t> state;
  DevBuf<unsigned long long> stats(16);

  auto count_pass = [&](const int32_t* c, int32_t* kn, int32_t* ml, int ng) {
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_gh_union<T, false>), dim3(xcd_grid(ng)), dim3(256), 0, stream(), view(A), view(B), c, kn, ml,
                         (const int64_t*)nullptr, (GhRec*)nullptr, (T*)nullptr, ng);
    });
  };
  // steps of the numeric kernel = sum of the groups' unions; *cost: the same with unusable groups (union beyond the
  // tile builder's capacity) counted at that capacity, so that they do not make an ordering look good
  int64_t max_kn = 0;   // largest union of the order counted last
  auto total_of = [&](const int32_t* kn, int32_t* pos, int64_t* off, uint8_t* st, int ng, int64_t* cost) -> int64_t {
    stats.zero();
    hipLaunchKernelGGL(k_gh_init_state, dim3(cdiv(ng, 256)), dim3(256), 0, stream(), kn, pos, st, stats.p, ng);
    scan_i32_async(pos, off, (int64_t)ng);
    int64_t total = 0;
    unsigned long long nb[2] = {0, 0};
    ScalarFetch f;
    f.add(off + ng, 1, &total);
    f.add(stats.p, 2, nb);
    f.run();
    *cost = total + (int64_t)nb[0] * GH_KCAP;
    max_kn = (int64_t)nb[1];
    return total;
  };

  // The column order of the last multiply of this dimension is kept: purification iterates keep their similarity
  // structure from one multiply to the next, so the clustering is only redone when the kept order stops paying
  // (its union ratio has grown by more than 15 % since it was made).
  struct OrderCache {
    int n = -1, ngroups = 0, minhash = 0;
    double ratio0 = 0;
    DevBuf<int32_t> cols;
  };
  static OrderCache* cache[2] = {new OrderCache(), new OrderCache()};   // real / complex (leaked on purpose, like the context)
  OrderCache& oc = *cache[A.cplx ? 1 : 0];
  const double ideal = std::max(1.0, (double)B.nnz / (double)G);
  int64_t cost = 0, total = 0;
  double ratio = 0;
  bool reuse = false;
  if (oc.n == n && oc.ngroups > 0) {
    ngroups = oc.ngroups;
    grp_kn.alloc((size_t)ngroups); grp_maxlen.alloc((size_t)ngroups); kn_pos.alloc((size_t)ngroups);
    grp_off.alloc((size_t)ngroups + 1); state.alloc((size_t)ngroups);
    count_pass(oc.cols.p, grp_kn.p, grp_maxlen.p, ngroups);
    total = total_of(grp_kn.p, kn_pos.p, grp_off.p, state.p, ngroups, &cost);
    ratio = (double)cost / ideal;
    reuse = ratio <= std::max(1.3, 1.15 * oc.ratio0);
    gi.minhash = oc.minhash;
  }
  if (mode == 2 && !(reuse && oc.minhash)) {
    if (info) *info = gi;
    return false;
  }
  if (!reuse) {
    // natural order first: adjacent columns of a locally ordered matrix are similar
    ngroups = cdiv(n, G);
    cols.alloc((size_t)ngroups * G);
    grp_kn.alloc((size_t)ngroups); grp_maxlen.alloc((size_t)ngroups); kn_pos.alloc((size_t)ngroups);
    grp_off.alloc((size_t)ngroups + 1); state.alloc((size_t)ngroups);
    gi.minhash = 0;
    hipLaunchKernelGGL(k_gh_iota, dim3(cdiv(ngroups * G, 256)), dim3(256), 0, stream(), cols.p, n, ngroups * G);
    count_pass(cols.p, grp_kn.p, grp_maxlen.p, ngroups);
    total = total_of(grp_kn.p, kn_pos.p, grp_off.p, state.p, ngroups, &cost);
    ratio = (double)cost / ideal;
    const int64_t max_kn_natural = max_kn;
    if (ratio > 1.5) {
      // cluster the columns by min-hash signature, order every cluster along a line, cut it into groups, count again
      DevBuf<unsigned long long> sig((size_t)n), sig_sorted((size_t)n), best((size_t)n + 1);
      DevBuf<int32_t> ids((size_t)n), order((size_t)n), order2((size_t)n), flag((size_t)n), cid((size_t)n), cstart((size_t)n + 1),
          ov((size_t)n), ng_c((size_t)n);
      DevBuf<int64_t> excl((size_t)n + 1), goff((size_t)n + 1);
      hipLaunchKernelGGL(k_gh_signature, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(B), sig.p, ids.p);
      size_t tmp_bytes = 0;
      (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, sig.p, sig_sorted.p, ids.p, order.p, (size_t)n, 0, 64, stream());
      DevBuf<char> tmp(tmp_bytes + 16);
      if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, sig.p, sig_sorted.p, ids.p, order.p, (size_t)n, 0, 64, stream()) != hipSuccess)
        NTP_FATAL("radix sort failed");
      hipLaunchKernelGGL(k_gh_cluster_flags, dim3(cdiv(n, 256)), dim3(256), 0, stream(), sig_sorted.p, flag.p, n);
      scan_i32_async(flag.p, excl.p, (int64_t)n);
      hipLaunchKernelGGL(k_gh_cluster_ids, dim3(cdiv(n + 1, 256)), dim3(256), 0, stream(), flag.p, excl.p, cid.p, cstart.p, best.p, n);
      const int nb = cdiv((int64_t)n * WAVE, 256);
      hipLaunchKernelGGL(k_gh_overlap, dim3(nb), dim3(256), 0, stream(), view(B), order.p, cid.p, cstart.p,
                         (const unsigned long long*)nullptr, ov.p, best.p, n);
      hipLaunchKernelGGL(k_gh_overlap, dim3(nb), dim3(256), 0, stream(), view(B), order.p, cid.p, cstart.p, best.p, ov.p,
                         (unsigned long long*)nullptr, n);
      hipLaunchKernelGGL(k_gh_order_keys, dim3(cdiv(n, 256)), dim3(256), 0, stream(), cid.p, ov.p, sig.p, n);
      if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, sig.p, sig_sorted.p, order.p, order2.p, (size_t)n, 0, 64, stream()) != hipSuccess)
        NTP_FATAL("radix sort failed");
      int64_t nc = 0;
      {
        ScalarFetch f;
        f.add(excl.p + n, 1, &nc);
        f.run();
      }
      hipLaunchKernelGGL(k_gh_cluster_groups, dim3(cdiv(nc, 256)), dim3(256), 0, stream(), cstart.p, ng_c.p, (int)nc, G);
      scan_i32_async(ng_c.p, goff.p, nc);
      int64_t ng2 = 0;
      {
        ScalarFetch f;
        f.add(goff.p + nc, 1, &ng2);
        f.run();
      }
      DevBuf<int32_t> cols2((size_t)ng2 * G), grp_kn2((size_t)ng2), grp_maxlen2((size_t)ng2), kn_pos2((size_t)ng2);
      DevBuf<int64_t> grp_off2((size_t)ng2 + 1);
      DevBuf<uint8_t> state2((size_t)ng2);
      hipLaunchKernelGGL(k_gh_fill_i32, dim3(cdiv(ng2 * G, 256)), dim3(256), 0, stream(), cols2.p, ng2 * G, -1);
      hipLaunchKernelGGL(k_gh_fill_groups, dim3(cdiv(n, 256)), dim3(256), 0, stream(), order2.p, cid.p, cstart.p, goff.p, cols2.p, n, G);
      count_pass(cols2.p, grp_kn2.p, grp_maxlen2.p, (int)ng2);
      int64_t cost2 = 0;
      const int64_t total2 = total_of(grp_kn2.p, kn_pos2.p, grp_off2.p, state2.p, (int)ng2, &cost2);
      if (cost2 < cost) {
        cols = std::move(cols2);
        grp_kn = std::move(grp_kn2);
        grp_maxlen = std::move(grp_maxlen2);
        grp_off = std::move(grp_off2);
        state = std::move(state2);
        ngroups = (int)ng2;
        total = total2;
        cost = cost2;
        ratio = (double)cost / ideal;
        gi.minhash = 1;
      } else {
        max_kn = max_kn_natural;
      }
    }
    oc.n = n;
    oc.ngroups = ngroups;
    oc.minhash = gi.minhash;
    oc.ratio0 = ratio;
    oc.cols = std::move(cols);
  }
  const int32_t* colp = oc.cols.p;
  gi.groups = ngroups;
  gi.union_ratio = ratio;
  gi.tile_rows = total;
  if (ratio > 6.0 && !force) {
    if (info) *info = gi;
    return false;
  }
  const int npad = ngroups * G;

  // (a quarter of headroom: the unions grow from one purification step to the next, and a block of the caching
  // allocator that fits the next multiply as well saves a hipMalloc of hundreds of MB inside the solver loop)
  const size_t tile_rows_cap = (size_t)total + (size_t)total / 4 + 8;
  DevBuf<GhRec> recs(tile_rows_cap);
  DevBuf<double> tiles(tile_rows_cap * (size_t)G * A.wval());
  DevBuf<unsigned long long> prod(1);
  prod.zero();
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    if (max_kn <= 1024)
      hipLaunchKernelGGL((k_gh_fill<T, 2048>), dim3(xcd_grid(ngroups)), dim3(256), 0, stream(), view(A), view(B), colp, grp_kn.p,
                         grp_off.p, recs.p, reinterpret_cast<T*>(tiles.p), ngroups, prod.p);
    else
      hipLaunchKernelGGL((k_gh_fill<T, 8192>), dim3(xcd_grid(ngroups)), dim3(256), 0, stream(), view(A), view(B), colp, grp_kn.p,
                         grp_off.p, recs.p, reinterpret_cast<T*>(tiles.p), ngroups, prod.p);
  });

  if (numeric_begin) HIP_CHECK(hipEventRecord(numeric_begin, stream()));
  static int hint[2] = {0, 0};  // table class that took most groups last time (real / complex)
  int& start = hint[A.cplx ? 1 : 0];
  int64_t todo = ngroups;
  const int first = start;
  for (int level = first; level < 3; ++level) {
    stats.zero();
    // (real operands in FMA arithmetic: the products on the matrix cores, four steps per phase -- option ghash_mfma)
    const bool mf = !A.cplx && (dense_rule & 2) != 0 && options().ghash_mfma != 0;
    if (mf) {
      if (level == 0)
        launch_ghash<double, 8, 1, 2, true>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                                            count, state.p, stats.p, alpha, threshold, dense_rule);
      else if (level == 1)   // (1024 slots: SIXTEEN waves -- four per column of a phase, four tiles of 16 slots each; the two sets of four
                             // slot-indexed columns and the table fill the LDS of a CU, so the waves of one workgroup are its occupancy)
        launch_ghash<double, 16, 1, 4, true>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                                             count, state.p, stats.p, alpha, threshold, dense_rule);
      else   // (the largest class: four slot-indexed columns of 1536 per set and the table do not fit the LDS together -- vector units)
        launch_ghash<double, 8, 3, 4>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                                      count, state.p, stats.p, alpha, threshold, dense_rule);
    } else
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      if (level == 0)
        launch_ghash<T, 8, 1, 2>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
      else if (level == 1)
        launch_ghash<T, 8, 2, 4>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
      else
        launch_ghash<T, 8, 3, 4>(A, ngroups, colp, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
    });
    unsigned long long h[16] = {0};
    ScalarFetch f;
    f.add(stats.p, 16, h);
    f.run();
    if (options().spgemm_variant == 518 && h[12])
      std::fprintf(stderr, "[ghash stamps, level %d] cycles per phase (wave 0): load %.0f scatter %.0f products %.0f barrier %.0f (phases %llu); "
                   "per group: loop %.0f epilogue %.0f (groups %llu)\n",
                   level, (double)h[8] / h[12], (double)h[9] / h[12], (double)h[10] / h[12], (double)h[11] / h[12], h[12],
                   (double)(h[8] + h[9] + h[10] + h[11]) / std::max(1ull, h[14]), (double)h[13] / std::max(1ull, h[14]), h[14]);
    gi.level = level;
    const int64_t left = (int64_t)h[0];
    if (level == first) {
      // next multiply: start one class up when most groups overflowed, one class down when every row union would
      // have fitted the smaller table with room to spare
      if (left * 2 > todo && level < 2) start = level + 1;
      else if (left == 0 && level > 0 && (int64_t)h[1] * 10 <= (int64_t)(level == 1 ? 512 : 1024) * 8) start = level - 1;
    }
    todo = left;
    if (left == 0) break;
  }
  stats.zero();
  if (A.cplx)
    hipLaunchKernelGGL(k_gh_finish<8>, dim3(cdiv(npad, 256)), dim3(256), 0, stream(), colp, state.p, bin_arr, count, stats.p, ngroups);
  else
    hipLaunchKernelGGL(k_gh_finish<16>, dim3(cdiv(npad, 256)), dim3(256), 0, stream(), colp, state.p, bin_arr, count, stats.p, ngroups);
  unsigned long long h[4] = {0, 0, 0, 0}, hprod = 0;
  {
    ScalarFetch f;
    f.add(stats.p, 4, h);
    f.add(prod.p, 1, &hprod);
    f.run();
  }
  gi.products = (int64_t)hprod;
  gi.failed_cols = (int64_t)h[2];
  gi.failed_groups = (int64_t)h[3];
  if (info) *info = gi;
  return true;
}

}  // namespace ntp
