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
// Results are bit-identical to the per-column kernels and to the oracle: every C(i, j) sees the same products in the
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
      if (*(volatile int*)&ctl[0] > GH_KCAP) break;
      const int64_t p = p0 + lane;
      bool fresh = false;
      if (p < e) {
        const int k = B.inner[p];
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

// groups whose union is empty need no numeric work; unusable ones (kn < 0) stay "to do" and end in the fallback
__global__ void k_gh_init_state(const int32_t* __restrict__ grp_kn, int32_t* __restrict__ kn_pos,
                                uint8_t* __restrict__ state, unsigned long long* __restrict__ nbad, int ngroups) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ngroups) return;
  const int kn = grp_kn[g];
  kn_pos[g] = kn > 0 ? kn : 0;
  state[g] = kn == 0 ? 1 : 0;
  if (kn < 0) atomicAdd(nbad, 1ull);
}

// ------------------------------------------------------------------ numeric kernel
// One workgroup per group.  The walk over the union of the B rows is cut into PHASES of up to KB consecutive steps
// (KB = as many columns of A as fit GH_MAXLEN entries, at most 4): one barrier, one round of load latencies and one
// round of hash probes serve KB steps.  Software pipeline, iteration i:
//   L(i+2)  request the entries of the A columns of phase i+2 and its KB rows of multipliers (registers)
//   S(i+1)  hash the rows of phase i+1 (requested one iteration ago) to slots, scatter the values into x[set(i+1)],
//           put the multipliers into LDS
//   F(i)    products of phase i: for its steps in ascending k, every owned chunk of slots reads x[set(i)], multiplies
//           with the G multipliers (LDS broadcast reads) into the register sums and writes zeros back (the reader owns
//           the slot, so the two sets need no other cleaning)
//   barrier
template <int CAP>
struct GhTable {
  static constexpr int TH = CAP <= 512 ? 1024 : CAP <= 1024 ? 2048 : 4096;    // buckets (load <= 1/2 ... 3/8)
  static constexpr int SHIFT = CAP <= 512 ? 22 : CAP <= 1024 ? 21 : 20;        // top bits of the multiplicative hash
};

template <typename T, int NW, int SL>
__global__ __launch_bounds__(NW* WAVE) __attribute__((amdgpu_waves_per_eu((SL <= 2 ? 4 : 2)))) void k_spgemm_ghash(
    Csc A, const int32_t* __restrict__ cols, const int32_t* __restrict__ grp_kn, const int32_t* __restrict__ grp_maxlen,
    const int64_t* __restrict__ grp_off, const GhRec* __restrict__ recs, const T* __restrict__ tiles,
    const int64_t* __restrict__ tmpoff, int32_t* __restrict__ out_inner, T* __restrict__ out_val,
    int32_t* __restrict__ count, uint8_t* __restrict__ grp_state, unsigned long long* __restrict__ stats, double alpha,
    double threshold, int dense_rule, int ngroups) {
  constexpr int G = GhG<T>::value;
  constexpr int NT = NW * WAVE, CAP = NT * SL, TH = GhTable<CAP>::TH, SHIFT = GhTable<CAP>::SHIFT;
  constexpr int EF = GH_MAXLEN / NT;   // entries per thread and phase
  // steps per phase the two x sets may hold: 32 KB of LDS for them (48 KB for the largest table class)
  constexpr int XB = (CAP <= 1024 ? 32768 : 49152) / (2 * CAP * (int)sizeof(T));
  constexpr int KBX = XB >= 4 ? 4 : XB >= 2 ? 2 : 1;
  constexpr unsigned long long EMPTY = ~0ull;
  static_assert(EF >= 1 && EF * NT == GH_MAXLEN, "threads per workgroup must divide GH_MAXLEN");
  __shared__ unsigned long long htab[TH];   // (row << 32 | slot); the epilogue sorts (row, slot) pairs in the same memory
  __shared__ T xbuf[2][KBX][CAP];           // slot-indexed copies of the A columns of two consecutive phases
  __shared__ T mult[2][KBX][G];             // their rows of multipliers
  __shared__ int slot_row[CAP];
  __shared__ int ctl[4];                    // [0] slots handed out, [1], [2] overflow seen while scattering an even / odd
                                            // phase (read after the barrier that ends that scatter, rewritten two
                                            // barriers later: every thread reads the same value), [3] valid rows (epilogue)
  __shared__ int cnt_s[SL * NW][2];
  const int gi = xcd_block(ngroups);
  if (gi < 0) return;
  if (grp_state[gi] != 0) return;
  const int kn = grp_kn[gi];
  if (kn <= 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  const int maxlen = grp_maxlen[gi];
  if (maxlen > GH_MAXLEN) {  // a column of A longer than one phase scatters: the group goes to the fallback
    if (tid == 0) atomicAdd(&stats[0], 1ull);
    return;
  }
  const int kb_n = min(KBX, GH_MAXLEN / max(1, maxlen));   // steps per phase: their columns hold <= GH_MAXLEN entries together
  const int nph = (kn + kb_n - 1) / kb_n;
  for (int s = tid; s < TH; s += NT) htab[s] = EMPTY;
  for (int s = tid; s < 2 * KBX * CAP; s += NT) (&xbuf[0][0][0])[s] = Sc<T>::zero();
  for (int s = tid; s < CAP; s += NT) slot_row[s] = -1;
  if (tid < 4) ctl[tid] = 0;
  __syncthreads();

  const int32_t* __restrict__ Ai = A.inner;
  const T* __restrict__ Av = static_cast<const T*>(A.val);
  const int64_t off = grp_off[gi];
  const GhRec* __restrict__ rec = recs + off;
  const T* __restrict__ tile = tiles + off * G;

  T acc[SL][G];
#pragma unroll
  for (int s = 0; s < SL; ++s)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[s][g] = Sc<T>::zero();

  // registers of one phase in flight: this thread's entries (row, value, step within the phase) and its multiplier
  struct Fetch {
    int idx[EF];
    T val[EF];
    int kbs;    // 2 bits per entry
    T m;
  };
  // L: request phase `ph`
  auto load_phase = [&](int ph, Fetch& f) {
    const int t0 = ph * kb_n;
    int64_t st[4];
    int pre[5];
    pre[0] = 0;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int t = min(t0 + kb, kn - 1);
      const bool on = kb < kb_n && t0 + kb < kn;
      st[kb] = rec[t].start;
      pre[kb + 1] = pre[kb] + (on ? rec[t].len : 0);
    }
    f.kbs = 0;
#pragma unroll
    for (int e = 0; e < EF; ++e) {
      const int q = e * NT + tid;
      f.idx[e] = -1;
      f.val[e] = Sc<T>::zero();
      if (e * NT >= pre[4]) continue;
      if (q < pre[4]) {
        const int kb = (q >= pre[1] ? 1 : 0) + (q >= pre[2] ? 1 : 0) + (q >= pre[3] ? 1 : 0);
        const int64_t base = kb == 0 ? st[0] : kb == 1 ? st[1] - pre[1] : kb == 2 ? st[2] - pre[2] : st[3] - pre[3];
        f.idx[e] = Ai[base + q];
        f.val[e] = Av[base + q];
        f.kbs |= kb << (2 * e);
      }
    }
    const int nrow = min(kb_n, kn - t0);
    f.m = Sc<T>::zero();
    if (tid < nrow * G) f.m = tile[(int64_t)t0 * G + tid];
  };
  // S: rows -> slots (first touch allocates), values into xbuf[set], multipliers into mult[set]
  auto scatter_phase = [&](int ph, int set, const Fetch& f) {
#pragma unroll
    for (int e = 0; e < EF; ++e) {
      const int i = f.idx[e];
      if (i >= 0) {
        unsigned h = gh_hash((unsigned)i) >> SHIFT;
        int mine = -1, slot = -1;
        for (;;) {
          const unsigned long long cur = *(volatile unsigned long long*)&htab[h];
          if ((int)(cur >> 32) == i) { slot = (int)(cur & 0xffffffffu); break; }
          if (cur == EMPTY) {
            if (mine < 0) {
              mine = atomicAdd(&ctl[0], 1);
              if (mine >= CAP) { ctl[1 + (ph & 1)] = 1; break; }
            }
            const unsigned long long want = ((unsigned long long)(unsigned)i << 32) | (unsigned)mine;
            const unsigned long long old = atomicCAS(&htab[h], EMPTY, want);
            if (old == EMPTY) { slot_row[mine] = i; slot = mine; break; }
            if ((int)(old >> 32) == i) { slot = (int)(old & 0xffffffffu); break; }  // same row, inserted meanwhile: `mine` stays a hole
          }
          h = (h + 1) & (TH - 1);
        }
        if (slot >= 0) xbuf[set][(f.kbs >> (2 * e)) & 3][slot] = f.val[e];
      }
    }
    if (tid < KBX * G) (&mult[set][0][0])[tid] = f.m;
  };

  Fetch fc, fn;
  load_phase(0, fc);
  if (nph > 1) load_phase(1, fn);
  scatter_phase(0, 0, fc);
  __syncthreads();
  bool overflow = ctl[1] != 0;   // (phase 0 is even)
  for (int ph = 0; ph < nph && !overflow; ++ph) {
    const int set = ph & 1;
    fc = fn;
    if (ph + 2 < nph) load_phase(ph + 2, fn);
    if (ph + 1 < nph) scatter_phase(ph + 1, set ^ 1, fc);
    // products of phase ph: chunks that hold slots handed out before the last barrier
    const int nsl = uni_i32(min(*(volatile int*)&ctl[0], CAP));
    const int nstep = min(kb_n, kn - ph * kb_n);
    for (int kb = 0; kb < nstep; ++kb) {
      T xv[SL];
      bool act[SL];
#pragma unroll
      for (int s = 0; s < SL; ++s) {
        const int c0 = (wave + NW * s) * WAVE;
        act[s] = c0 < nsl;
        xv[s] = Sc<T>::zero();
        if (act[s]) {
          xv[s] = xbuf[set][kb][c0 + lane];
          act[s] = __ballot(!Sc<T>::is_zero(xv[s])) != 0ull;
          if (act[s]) xbuf[set][kb][c0 + lane] = Sc<T>::zero();
        }
      }
#pragma unroll
      for (int g0 = 0; g0 < G; g0 += 4) {
        T m[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) m[g] = mult[set][kb][g0 + g];
#pragma unroll
        for (int s = 0; s < SL; ++s) {
          if (act[s]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[s][g0 + g] = Sc<T>::add(acc[s][g0 + g], Sc<T>::mul(xv[s], m[g]));
          }
        }
      }
    }
    __syncthreads();
    overflow = ctl[1 + ((ph + 1) & 1)] != 0;
  }
  if (overflow) {  // the row union outgrew this table class: the group stays "to do" for the next one
    if (tid == 0) atomicAdd(&stats[0], 1ull);
    return;
  }

  // ---- epilogue: rows in ascending order, then column by column through LDS
  const int nsl = min(ctl[0], CAP);
  int p2 = 64;
  while (p2 < nsl) p2 <<= 1;
  unsigned long long* skey = htab;
  T* colbuf = &xbuf[0][0][0];   // two columns of CAP sums each (xbuf holds >= 2 * CAP elements)
  __syncthreads();
  for (int s = tid; s < p2; s += NT) {
    const int row = s < nsl ? slot_row[s] : -1;
    const bool ok = row >= 0;
    skey[s] = ok ? (((unsigned long long)(unsigned)row << 32) | (unsigned)s) : EMPTY;
    const int nv = __popcll(__ballot(ok));
    if (lane == 0 && nv) atomicAdd(&ctl[3], nv);
  }
  __syncthreads();
  for (int kk = 2; kk <= p2; kk <<= 1) {
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
      __syncthreads();
    }
  }
  const int nvalid = ctl[3];
  if (tid == 0) atomicMax(&stats[1], (unsigned long long)nsl);
#pragma unroll
  for (int g0 = 0; g0 < G; g0 += 2) {
#pragma unroll
    for (int s = 0; s < SL; ++s) {
      const int c0 = (wave + NW * s) * WAVE;
      colbuf[c0 + lane] = acc[s][g0];
      colbuf[CAP + c0 + lane] = acc[s][g0 + 1];
    }
    __syncthreads();
    unsigned keepbits = 0;
#pragma unroll
    for (int rd = 0; rd < SL; ++rd) {
      const int r = rd * NT + tid;
      const bool valid = r < nvalid;
      const int slot = valid ? (int)(skey[r] & 0xffffffffu) : 0;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const T v = valid ? colbuf[cb * CAP + slot] : Sc<T>::zero();
        const bool keep = valid && (Sc<T>::mag(dense_rule ? v : Sc<T>::scale(alpha, v)) > threshold);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) cnt_s[rd * NW + wave][cb] = __popcll(m);
        keepbits |= keep ? (1u << (rd * 2 + cb)) : 0u;
      }
    }
    __syncthreads();
    if (tid < 2) {
      int run = 0;
      for (int seg = 0; seg < SL * NW; ++seg) {
        const int c = cnt_s[seg][tid];
        cnt_s[seg][tid] = run;
        run += c;
      }
      const int col = cols[gi * G + g0 + tid];
      if (col >= 0) count[col] = run;
    }
    __syncthreads();
#pragma unroll
    for (int rd = 0; rd < SL; ++rd) {
      const int r = rd * NT + tid;
      const unsigned long long key = r < nvalid ? skey[r] : 0ull;
      const int slot = (int)(key & 0xffffffffu);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const bool keep = (keepbits >> (rd * 2 + cb)) & 1u;
        const unsigned long long m = __ballot(keep);
        if (keep) {
          const int col = cols[gi * G + g0 + cb];
          const int64_t pos = tmpoff[col] + cnt_s[rd * NW + wave][cb] + __popcll(m & lanemask_lt());
          out_inner[pos] = (int)(key >> 32);
          out_val[pos] = Sc<T>::scale(alpha, colbuf[cb * CAP + slot]);
        }
      }
    }
    __syncthreads();
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

template <typename T, int NW, int SL>
void launch_ghash(const DevMat& A, int ngroups, const int32_t* cols, const int32_t* grp_kn, const int32_t* grp_maxlen,
                  const int64_t* grp_off, const GhRec* recs, const double* tiles, const int64_t* tmpoff, int32_t* tmp_inner,
                  double* tmp_val, int32_t* count, uint8_t* state, unsigned long long* stats, double alpha, double thr, int dr) {
  hipLaunchKernelGGL((k_spgemm_ghash<T, NW, SL>), dim3(xcd_grid(ngroups)), dim3(NW * WAVE), 0, stream(), view(A), cols, grp_kn,
                     grp_maxlen, grp_off, recs, reinterpret_cast<const T*>(tiles), tmpoff, tmp_inner,
                     reinterpret_cast<T*>(tmp_val), count, state, stats, alpha, thr, dr, ngroups);
}

}  // namespace

bool spgemm_grouped(const DevMat& A, const DevMat& B, const int64_t* tmpoff, int32_t* tmp_inner, double* tmp_val,
                    int32_t* count, uint8_t* bin_arr, double alpha, double threshold, int dense_rule, bool force,
                    GroupedInfo* info) {
  const int n = B.cols;
  const int G = A.cplx ? 8 : 16;
  const int ngroups = cdiv(n, G), npad = ngroups * G;
  GroupedInfo gi;
  gi.groups = ngroups;
  DevBuf<int32_t> cols((size_t)npad), cols2, grp_kn((size_t)ngroups), grp_maxlen((size_t)ngroups), kn_pos((size_t)ngroups);
  DevBuf<int32_t> grp_kn2, grp_maxlen2;
  DevBuf<int64_t> grp_off((size_t)ngroups + 1);
  DevBuf<uint8_t> state((size_t)ngroups);
  DevBuf<unsigned long long> stats(8);

  auto count_pass = [&](const int32_t* c, int32_t* kn, int32_t* ml) {
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      hipLaunchKernelGGL((k_gh_union<T, false>), dim3(xcd_grid(ngroups)), dim3(256), 0, stream(), view(A), view(B), c, kn, ml,
                         (const int64_t*)nullptr, (GhRec*)nullptr, (T*)nullptr, ngroups);
    });
  };
  // steps of the numeric kernel = sum of the groups' unions; *cost: the same with unusable groups (union beyond the
  // tile builder's capacity) counted at that capacity, so that they do not make an ordering look good
  auto total_of = [&](const int32_t* kn, int64_t* cost) -> int64_t {
    stats.zero();
    hipLaunchKernelGGL(k_gh_init_state, dim3(cdiv(ngroups, 256)), dim3(256), 0, stream(), kn, kn_pos.p, state.p, stats.p, ngroups);
    scan_i32_async(kn_pos.p, grp_off.p, (int64_t)ngroups);
    int64_t total = 0;
    unsigned long long nbad = 0;
    ScalarFetch f;
    f.add(grp_off.p + ngroups, 1, &total);
    f.add(stats.p, 1, &nbad);
    f.run();
    *cost = total + (int64_t)nbad * GH_KCAP;
    return total;
  };

  // natural order first: adjacent columns of a locally ordered matrix are similar
  hipLaunchKernelGGL(k_gh_iota, dim3(cdiv(npad, 256)), dim3(256), 0, stream(), cols.p, n, npad);
  count_pass(cols.p, grp_kn.p, grp_maxlen.p);
  int64_t cost = 0;
  int64_t total = total_of(grp_kn.p, &cost);
  const double ideal = std::max(1.0, (double)B.nnz / (double)G);
  double ratio = (double)cost / ideal;
  if (ratio > 1.5) {
    // cluster the columns by min-hash signature and count again
    DevBuf<unsigned long long> sig((size_t)n), sig_sorted((size_t)n);
    DevBuf<int32_t> ids((size_t)n);
    cols2.alloc((size_t)npad);
    grp_kn2.alloc((size_t)ngroups);
    grp_maxlen2.alloc((size_t)ngroups);
    hipLaunchKernelGGL(k_gh_signature, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), view(B), sig.p, ids.p);
    size_t tmp_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp_bytes, sig.p, sig_sorted.p, ids.p, cols2.p, (size_t)n, 0, 64, stream());
    DevBuf<char> tmp(tmp_bytes + 16);
    if (rocprim::radix_sort_pairs(tmp.p, tmp_bytes, sig.p, sig_sorted.p, ids.p, cols2.p, (size_t)n, 0, 64, stream()) != hipSuccess)
      NTP_FATAL("radix sort failed");
    if (npad > n) hipLaunchKernelGGL(k_gh_pad, dim3(cdiv(npad - n, 256)), dim3(256), 0, stream(), cols2.p, n, npad);
    count_pass(cols2.p, grp_kn2.p, grp_maxlen2.p);
    int64_t cost2 = 0;
    const int64_t total2 = total_of(grp_kn2.p, &cost2);
    if (cost2 < cost) {
      std::swap(cols, cols2);
      std::swap(grp_kn, grp_kn2);
      std::swap(grp_maxlen, grp_maxlen2);
      total = total2;
      cost = cost2;
      ratio = (double)cost / ideal;
      gi.minhash = 1;
    } else {
      total = total_of(grp_kn.p, &cost);  // offsets and states of the natural order again
    }
  }
  gi.union_ratio = ratio;
  gi.tile_rows = total;
  if (ratio > 6.0 && !force) {
    if (info) *info = gi;
    return false;
  }

  DevBuf<GhRec> recs((size_t)total + 8);
  DevBuf<double> tiles(((size_t)total + 8) * (size_t)G * A.wval());
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_gh_union<T, true>), dim3(xcd_grid(ngroups)), dim3(256), 0, stream(), view(A), view(B), cols.p, grp_kn.p,
                       grp_maxlen.p, grp_off.p, recs.p, reinterpret_cast<T*>(tiles.p), ngroups);
  });

  static int hint[2] = {0, 0};  // table class that took most groups last time (real / complex)
  int& start = hint[A.cplx ? 1 : 0];
  int64_t todo = ngroups;
  const int first = start;
  for (int level = first; level < 3; ++level) {
    stats.zero();
    dispatch_type(A.cplx, [&](auto tag) {
      using T = decltype(tag);
      if (level == 0)
        launch_ghash<T, 4, 2>(A, ngroups, cols.p, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
      else if (level == 1)
        launch_ghash<T, 8, 2>(A, ngroups, cols.p, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
      else
        launch_ghash<T, 8, 3>(A, ngroups, cols.p, grp_kn.p, grp_maxlen.p, grp_off.p, recs.p, tiles.p, tmpoff, tmp_inner, tmp_val,
                              count, state.p, stats.p, alpha, threshold, dense_rule);
    });
    unsigned long long h[2] = {0, 0};
    ScalarFetch f;
    f.add(stats.p, 2, h);
    f.run();
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
    hipLaunchKernelGGL(k_gh_finish<8>, dim3(cdiv(npad, 256)), dim3(256), 0, stream(), cols.p, state.p, bin_arr, count, stats.p, ngroups);
  else
    hipLaunchKernelGGL(k_gh_finish<16>, dim3(cdiv(npad, 256)), dim3(256), 0, stream(), cols.p, state.p, bin_arr, count, stats.p, ngroups);
  unsigned long long h[4] = {0, 0, 0, 0};
  {
    ScalarFetch f;
    f.add(stats.p, 4, h);
    f.run();
  }
  gi.failed_cols = (int64_t)h[2];
  gi.failed_groups = (int64_t)h[3];
  if (info) *info = gi;
  return true;
}

}  // namespace ntp
