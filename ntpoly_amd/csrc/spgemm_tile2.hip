// SpGEMM numeric phase on the FP64 matrix cores, second geometry: TWO blocks of 16 output columns per workgroup, the
// multiplier rows streamed through LDS in chunks of 32 k.  Same operands, same plan, same results (bit for bit) as
// k_spgemm_tile (spgemm_tile.hip; MultiplyBlock.f90:9-36 + PruneList.f90:8-38 and, fused, the TRS2 update of
// DensityMatrixSolversModule.F90:380-413 through AddSparseVectors.f90:21-70).
//
// Why: k_spgemm_tile reads every 16 R x 4 fragment of A once per block of 16 columns -- 1 KB from the L2 per two matrix
// instructions, 10-15 GB per launch of the headline, most of what an XCD's L2 delivers -- and spends as many vector
// instructions on a fragment's address as on anything else.  Here a fragment feeds both column blocks of the pair: half
// the L2 traffic and half the address arithmetic per matrix instruction.  Thirty-two columns of multipliers over the
// whole k range do not fit the LDS beside a second workgroup, so the k range is cut into CHUNKS of 32 k (8 groups of 4):
// the workgroup walks the chunks in step (one barrier each), the chunk behind the barrier is multiplied while the next one
// is fetched into the other half of a double buffer -- the multiplier tile is never waited for after the first chunk.
//
//   workgroup = 16 waves, pair of column blocks (2p, 2p + 1), union row window [LO, HI) cut into SLABS of 32 rows
//   wave v owns slabs v and v + 16 (a banded operand reaches slab s from the k groups around it: the two are never in
//   progress together -- checked, see `fail`), 32 rows x 32 columns of partial sums = 4 tiles = 32 VGPRs
//   per k group and slab: one run record (LDS), ONE 16-byte run load, two multiplier reads (LDS), FOUR matrix instructions
//   a slab whose k groups are done runs its epilogue (prune, fused update, energy, trace, result runs) at once -- under the
//   matrix instructions of the waves that are still multiplying -- and the wave moves on to its second slab
//
// Arithmetic: as k_spgemm_tile -- v_mfma_f64_16x16x4_f64 is a chain of fma() over ascending k, groups and chunks follow in
// ascending k, zero padding is exact -- so every C(i, j) is the FMA chain of the reference's FP-contracted build.
#include "spgemm_tile.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline v2d ld2(unsigned long long addr) { return *reinterpret_cast<const v2d __attribute__((address_space(1)))*>(addr); }
__device__ inline v2d ld2(const double* p) { return ld2(reinterpret_cast<unsigned long long>(p)); }
__device__ inline void st2(double* p, const v2d& v) {
  *reinterpret_cast<v2d __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(p)) = v;
}

constexpr int T2_NW = 16;            // waves per workgroup
constexpr int T2_NT = T2_NW * WAVE;  // threads
constexpr int T2_CG = 8;             // k groups per chunk
constexpr int T2_CK = 4 * T2_CG;     // k per chunk
constexpr int T2_KP = 34;            // doubles between two columns of a chunk in LDS: lane (jj, q) reads word 34 jj + q -- conflict-free
constexpr int T2_ROWS = 32;          // rows of a slab (two rows per lane of the A fragment)
constexpr int T2_MAXS = 2 * T2_NW;   // slabs of a window
constexpr int T2_DEFER = 64;         // deferred elements per column block (more: the step is refused, as k_spgemm_tile does)

struct alignas(16) T2Rec {           // run of column k: rz = address of (hypothetical) row 0; a lane's rows ra, ra + 1 touch it iff
  unsigned long long rz;             // (unsigned)(ra - first) <= span
  int32_t first;
  uint32_t span;
};
struct alignas(16) T2Defer {
  int32_t r, jj, prow, pad;
  double o, d;
};

struct Tile2Args {
  const SlabRun* runs;
  const int32_t *blk_kmin, *blk_kn, *blk_lo, *blk_w;
  const int64_t* blk_toff;
  double* out_val;
  int32_t* count;
  int32_t *ofirst, *olast;
  int64_t* ooff;
  int64_t* otoff;
  double alpha, threshold;
  int dense_rule, ncols, nblocks, npairs;
  int kcap;                 // k the LDS records are sized for (a multiple of 32, >= the largest union k range + 4)
  SlabFuseArgs fzv;
  const int32_t *brun_first, *brun_last;
  const int64_t* brun_off;
  const double* brun_val;
  const double* zero;
  int* fail;                // set when a pair's geometry does not fit: nothing of the launch may be used
  int ablate;               // timing experiments (WRONG results): NTPOLY_AMD_T2_ABLATE bits: 1 no epilogues, 2 no matrix instructions, 4 every run load from the zero page, 8 no chunk barriers
};

// per column block of the pair (LDS)
struct T2Group {
  unsigned colmask[T2_MAXS];
  int col_cnt[16], col_first[16], col_last[16], col_pmax[16], col_pad[16];
  double red[2 * T2_NW];
  int misc[4];                        // [0] deferred elements, [1] product entries, [2..3] products (64 bit)
  T2Defer dlist[T2_DEFER];
  double dsums[2 * T2_DEFER];
};

__host__ __device__ inline size_t tile2_lds_bytes(int kcap) {
  return (size_t)2 * 32 * T2_KP * 8 + (size_t)(kcap + 8) * sizeof(T2Rec) + (size_t)2 * (kcap / 4 + 2) * 4 + 16 + 2 * sizeof(T2Group) + 64;
}

template <int EPI>
__global__ __launch_bounds__(T2_NT) void k_spgemm_tile2(const Tile2Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = xcd_block(a.npairs);
  if (p < 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  // ---- the two column blocks
  int lo_[2], w_[2], kmin_[2], kn_[2];
  int64_t tb_[2];
  bool act_[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int b = 2 * p + c;
    const bool have = b < a.nblocks;
    const int bc = have ? b : a.nblocks - 1;
    lo_[c] = a.blk_lo[bc]; w_[c] = a.blk_w[bc]; kmin_[c] = a.blk_kmin[bc]; kn_[c] = have ? a.blk_kn[bc] : 0;
    tb_[c] = a.blk_toff[bc];
    act_[c] = have && kn_[c] > 0;
    if (have && kn_[c] == 0) {   // no product entries in these columns
      const int j = b * SLAB_J + tid;
      if (tid < SLAB_J && j < a.ncols) {
        a.ofirst[j] = INT_MAX;
        a.olast[j] = -1;
        a.count[j] = 0;
        a.ooff[j] = tb_[c] + (int64_t)tid * w_[c];
        if constexpr (EPI == 2) {
          if (a.fzv.xmax[j] >= a.fzv.xmin[j]) atomicOr(a.fzv.flag, 1);
        }
      }
      if (EPI != 0 && tid == 0) a.otoff[b] = tb_[c];
    }
  }
  if (p == 0 && tid == 0) {   // (the end markers of the result's offset arrays)
    a.ooff[a.ncols] = a.blk_toff[a.nblocks];
    if (EPI != 0 && a.otoff) a.otoff[a.nblocks] = a.blk_toff[a.nblocks];
  }
  if (!act_[0] && !act_[1]) return;
  const int LO = min(act_[0] ? lo_[0] : INT_MAX, act_[1] ? lo_[1] : INT_MAX);
  const int HI = max(act_[0] ? lo_[0] + w_[0] : -1, act_[1] ? lo_[1] + w_[1] : -1);
  const int KMIN = min(act_[0] ? kmin_[0] : INT_MAX, act_[1] ? kmin_[1] : INT_MAX);
  const int KEND = max(act_[0] ? kmin_[0] + kn_[0] : -1, act_[1] ? kmin_[1] + kn_[1] : -1);
  const int S = (HI - LO) / T2_ROWS;
  const int KN = KEND - KMIN, KG = (KN + 3) >> 2, NCH = (KG + T2_CG - 1) / T2_CG;
  if (S > T2_MAXS || NCH * T2_CK + 4 > a.kcap || ((HI - LO) % T2_ROWS) != 0) {
    if (tid == 0) atomicOr(a.fail, 1);
    return;
  }
  // ---- LDS
  double* Bs = reinterpret_cast<double*>(smem);                                  // [2][32 columns][T2_KP]
  T2Rec* recs = reinterpret_cast<T2Rec*>(Bs + 2 * 32 * T2_KP);                   // [kcap + 8]
  int* grmin = reinterpret_cast<int*>(recs + a.kcap + 8);                        // [kcap / 4 + 2]
  int* grmax = grmin + (a.kcap / 4 + 2);
  T2Group* grp = reinterpret_cast<T2Group*>((reinterpret_cast<uintptr_t>(grmax + (a.kcap / 4 + 2)) + 15) & ~(uintptr_t)15);
  int* ovl = reinterpret_cast<int*>(grp + 2);

  // ---- block prologue: run records of the k range (a thread each), row range of every k group; chunk 0 of the multipliers
  {
    const uint4* __restrict__ rp = reinterpret_cast<const uint4*>(a.runs + KMIN);
    const int K4 = NCH * T2_CK;
    for (int i0 = 0; i0 < K4 + 8; i0 += T2_NT) {
      const int i = i0 + tid;
      const int ic = min(i, KN - 1);
      const uint4 r0 = rp[2 * ic], r1 = rp[2 * ic + 1];      // (addr_lo, addr_hi, nbytes, flags), (first8, first, span62, pad)
      const int rows = i < KN ? (int)(r0.z >> 3) : 0;
      const int first = (int)r1.y;
      T2Rec rec;
      rec.rz = 0;
      rec.first = INT_MAX;
      rec.span = 0u;
      int rmin = INT_MAX, rmax = -1;
      if (rows > 0) {
        const unsigned long long addr = (unsigned long long)r0.x | ((unsigned long long)r0.y << 32);
        rec.rz = addr - (unsigned long long)((long long)first * 8);
        rec.first = first - 1;
        rec.span = (uint32_t)rows;
        rmin = first;
        rmax = first + rows - 1;
      }
      rmin = min(rmin, __builtin_amdgcn_mov_dpp(rmin, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
      rmax = max(rmax, __builtin_amdgcn_mov_dpp(rmax, 0xb1, 0xf, 0xf, false));
      rmin = min(rmin, __builtin_amdgcn_mov_dpp(rmin, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
      rmax = max(rmax, __builtin_amdgcn_mov_dpp(rmax, 0x4e, 0xf, 0xf, false));
      if (i < K4 + 8) {
        recs[i] = rec;
        if ((i & 3) == 0) {
          grmin[i >> 2] = rmin;
          grmax[i >> 2] = rmax;
        }
      }
    }
  }
  // multipliers: thread (column bcol = tid / 32 of the pair, k offset bk = tid % 32 of the chunk) -- 32 consecutive k of
  // a column's run are 256 contiguous bytes
  const int bcol = tid >> 5, bk = tid & 31;
  int bf = INT_MAX, bl = -1;
  const double* bp = a.zero;
  {
    const int c = bcol >> 4;
    const int j = (2 * p + c) * SLAB_J + (bcol & 15);
    if (act_[c] && j < a.ncols) {
      bf = a.brun_first[j];
      bl = a.brun_last[j];
      if (bl >= bf) bp = a.brun_val + (a.brun_off[j] - bf);
    }
  }
  auto bfetch = [&](int ch) -> double {
    const int k = KMIN + ch * T2_CK + bk;
    return (k >= bf && k <= bl) ? bp[k] : 0.0;
  };
  double* const bslot = Bs + bcol * T2_KP + bk;          // + buffer * 32 * T2_KP
  [[maybe_unused]] long long nprod = 0;
  [[maybe_unused]] const int32_t* __restrict__ in_count = nullptr;
  if constexpr (EPI != 0) in_count = a.fzv.prod ? a.fzv.in_count : nullptr;
  auto count_products = [&](double v, int ch) {
    if constexpr (EPI != 0) {
      if (a.fzv.prod && v != 0.0) nprod += in_count ? (long long)in_count[KMIN + ch * T2_CK + bk] : 1ll;
    }
  };
  {
    const double v0 = bfetch(0);
    count_products(v0, 0);
    bslot[0] = v0;
  }
  for (int i = tid; i < 2 * (int)(sizeof(T2Group) / 4); i += T2_NT) reinterpret_cast<int*>(grp)[i] = 0;
  if (tid == 0) ovl[0] = 0;
  __syncthreads();
  if (tid < 32) {
    T2Group& G = grp[tid >> 4];
    G.col_first[tid & 15] = INT_MAX;
    G.col_last[tid & 15] = -1;
    G.col_pmax[tid & 15] = -1;
  }

  // ---- this wave's slabs and the k groups that reach them (a ballot over the groups' row ranges)
  int sg0[2], sg1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int s = wave + T2_NW * t;
    const int r0 = LO + T2_ROWS * s;
    int g0 = INT_MAX, g1 = -1;
    if (s < S) {
      for (int c = 0; c < KG; c += WAVE) {
        const int gq = min(c + lane, KG);
        const bool hit = gq < KG && grmin[gq] <= r0 + T2_ROWS - 1 && grmax[gq] >= r0;
        const unsigned long long m = __ballot(hit);
        if (m) {
          if (g0 == INT_MAX) g0 = c + (int)__builtin_ctzll(m);
          g1 = c + 63 - (int)__builtin_clzll(m);
        }
      }
    }
    sg0[t] = g0;
    sg1[t] = g1;
  }
  // the second slab must begin in a later chunk than the first one ends in
  if (sg1[0] >= 0 && sg1[1] >= 0 && (sg0[1] / T2_CG) <= (sg1[0] / T2_CG)) {
    if (lane == 0) ovl[0] = 1;
  }
  __syncthreads();
  if (ovl[0]) {
    if (tid == 0) atomicOr(a.fail, 1);
    return;
  }

  // ---- per-lane constants
  const int jj = lane & 15, q = lane >> 4;
  const double* const zp = a.zero;
  const unsigned long long zaddr = reinterpret_cast<unsigned long long>(zp);
  const double alpha = a.alpha, thr = a.threshold;
  const bool dense_rule = (a.dense_rule & 1) != 0;
  const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(recs) + q;       // record of group g: rq[4 g]
  const double* const bq = Bs + jj * T2_KP + q;                                  // column block c, group u of the chunk: bq[c * 16 * KP + 4 u]

  // per column block: this lane's column
  [[maybe_unused]] double dsum[2] = {0.0, 0.0}, tsum[2] = {0.0, 0.0};
  int pn[2] = {0, 0};

  // ---- epilogue of one slab for one column block: lane holds rows r0 + 2 (4 v + q) + m (v = 0..3, m = 0..1) of column jj
  auto epilogue = [&](const int c, const int r0, const v4d& acc0, const v4d& acc1) {
    if (!act_[c] || r0 < lo_[c] || r0 >= lo_[c] + w_[c]) return;
    T2Group& G = grp[c];
    const int lo = lo_[c], w = w_[c];
    const int64_t tbase = tb_[c];
    const int t = (r0 - lo) / T2_ROWS;
    int j = (2 * p + c) * SLAB_J + jj;
    // (what follows is per column and per slab: it must not be computed ahead of the chunk loop and carried through it in
    // registers -- the loop needs them for fragments and partial sums)
    asm volatile("" : "+v"(j));
    const bool colv = j < a.ncols;
    const int jc = min(j, a.ncols - 1);
    double* const orun = a.out_val + (tbase + (int64_t)jj * w - lo);
    [[maybe_unused]] double* otile = nullptr;
    [[maybe_unused]] int xf = INT_MAX, xlrow = -1, xpl = -1, df = INT_MAX, dl = -1;
    [[maybe_unused]] const double *xrz = zp, *drz = zp;
    [[maybe_unused]] double am = 0, bm = 0, thr_m = 0;
    [[maybe_unused]] int diag = -1;
    if constexpr (EPI != 0) {
      if (a.fzv.tiles) otile = a.fzv.tiles + (tbase - (int64_t)lo * SLAB_J + jj);
      const int d0 = a.fzv.dmin[jc], d1 = a.fzv.dmax[jc];
      if (colv && d1 >= d0) {
        df = d0;
        dl = d1;
        drz = a.fzv.dexp + (a.fzv.doff[jc] - d0);
      }
      diag = j + a.fzv.col_offset;
      if constexpr (EPI == 2) {
        am = a.fzv.am; bm = a.fzv.bm; thr_m = a.fzv.thr_m;
        const int x0 = a.fzv.xmin[jc], x1 = a.fzv.xmax[jc];
        if (colv && x1 >= x0) {
          xf = x0;
          xlrow = x1;
          xrz = a.fzv.xexp + (a.fzv.xoff[jc] - x0);
          xpl = x1;
        }
      }
    }
    {  // slabs in which nothing can be kept are done here (an entry of X in the slab: at least its run reaches it)
      bool live = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const double vv = m ? acc1[v] : acc0[v];
          live |= dense_rule ? (fabs(vv) > thr) : (fabs(__dmul_rn(alpha, vv)) > thr);
        }
      }
      if constexpr (EPI == 2) live |= (r0 + T2_ROWS - 1 >= xf) & (r0 <= xlrow);
      if (__ballot(live) == 0ull) {
        if (lane == 0) G.colmask[t] = 0u;
        return;
      }
    }
    v2d res[4];
    unsigned long long anykeep = 0;
    int c_l = 0, f_l = INT_MAX, l_l = -1, pm_l = -1;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {   // (two halves: the values of X and D of four rows at a time -- registers)
      [[maybe_unused]] v2d xv[2], dv[2];
      if constexpr (EPI != 0) {
#pragma unroll
        for (int vv2 = 0; vv2 < 2; ++vv2) {
          const int rb = r0 + 2 * (4 * (2 * hh + vv2) + q);
          if constexpr (EPI == 2) xv[vv2] = ld2(((rb + 1 >= xf) & (rb <= xlrow)) ? xrz + rb : zp);
          dv[vv2] = ld2(((rb + 1 >= df) & (rb <= dl)) ? drz + rb : zp);
        }
      }
#pragma unroll
      for (int vv2 = 0; vv2 < 2; ++vv2) {
        const int v = 2 * hh + vv2;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int r = r0 + 2 * (4 * v + q) + m;
          const double vv = m ? acc1[v] : acc0[v];
          const double sv = __dmul_rn(alpha, vv);
          const bool ha = dense_rule ? (fabs(vv) > thr) : (fabs(sv) > thr);
          bool keep;
          double o;
          [[maybe_unused]] double dval = 0.0;
          if constexpr (EPI != 0) dval = dv[vv2][m];
          if constexpr (EPI != 2) {
            keep = ha;
            o = sv;
          } else {
            const double bv = xv[vv2][m];
            const bool hb = bv != 0.0;
            const double bs = __dmul_rn(bm, bv);
            const double wa = __dmul_rn(am, sv);
            const double both = __dadd_rn(wa, bs);
            o = ha ? (hb ? both : wa) : bs;
            const bool big = fabs(o) > thr_m;
            if (ha) {
              keep = (!hb && r > xpl) || big;
            } else {
              keep = hb && big;
              if (hb && !big) {
                const int slot = atomicAdd(&G.misc[0], 1);
                if (slot < T2_DEFER) {
                  *reinterpret_cast<int4*>(&G.dlist[slot]) = make_int4(r, jj, r, 0);
                  G.dlist[slot].o = o;
                  G.dlist[slot].d = dval;
                }
              }
            }
          }
          pn[c] += (int)__popcll(__ballot(ha));
          anykeep |= __ballot(keep);
          if constexpr (EPI != 0) {
            dsum[c] = __dadd_rn(dsum[c], __dmul_rn(keep ? o : 0.0, keep ? dval : 0.0));
            tsum[c] = __dadd_rn(tsum[c], (keep && r == diag) ? o : 0.0);
            pm_l = max(pm_l, ha ? r : -1);
          }
          c_l += keep ? 1 : 0;
          f_l = min(f_l, keep ? r : INT_MAX);
          l_l = max(l_l, keep ? r : -1);
          res[v][m] = keep ? o : 0.0;
          __builtin_amdgcn_sched_barrier(0);   // (one element after the other: interleaved they need twice the registers)
        }
      }
    }
    const unsigned cm = (unsigned)((anykeep | (anykeep >> 16) | (anykeep >> 32) | (anykeep >> 48)) & 0xffffull);
    if (c_l) {
      atomicAdd(&G.col_cnt[jj], c_l);
      atomicMin(&G.col_first[jj], f_l);
      atomicMax(&G.col_last[jj], l_l);
    }
    if constexpr (EPI == 2) {
      if (pm_l >= 0) atomicMax(&G.col_pmax[jj], pm_l);
    }
    if ((cm >> jj) & 1u) {
#pragma unroll
      for (int v = 0; v < 4; ++v) st2(orun + (r0 + 2 * (4 * v + q)), res[v]);
    }
    if constexpr (EPI != 0) {
      if (cm && otile) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int m = 0; m < 2; ++m) otile[(int64_t)(r0 + 2 * (4 * v + q) + m) * SLAB_J] = res[v][m];
        }
      }
    }
    if (lane == 0) G.colmask[t] = cm;
  };

  // ---- the chunks.  Every wave passes every chunk's barrier; between them it multiplies the chunk into the slab it has in
  // progress (if the slab's k groups reach into the chunk).  Its two slabs follow one another: [first chunk, last chunk] of
  // the second lies behind that of the first (checked above).
  constexpr int RING = 4;   // run loads in flight per wave (k groups ahead)
  const int KGP = NCH * T2_CG;   // a k group behind the last one: its records are empty
  int ch = 0;
  double bnext = 0.0;
  auto chunk_open = [&](int chn) {    // chunk chn is in buffer chn & 1 behind this barrier; nobody reads the other buffer any more
    if (chn > 0 && !(a.ablate & 8)) __syncthreads();
    if (chn + 1 < NCH) bnext = bfetch(chn + 1);   // in flight under this chunk's matrix instructions
  };
  auto chunk_close = [&](int chn) {
    if (chn + 1 < NCH) {
      count_products(bnext, chn + 1);
      bslot[((chn + 1) & 1) * 32 * T2_KP] = bnext;
    }
  };
  for (int t = 0; t < 2; ++t) {
    const int s = wave + T2_NW * t;
    if (s >= S) break;
    const int g0 = t ? sg0[1] : sg0[0], g1 = t ? sg1[1] : sg1[0];
    const int r0 = LO + T2_ROWS * s;
    v4d acc00 = v4d{0.0, 0.0, 0.0, 0.0}, acc10 = acc00, acc01 = acc00, acc11 = acc00;   // [row parity][column block]
    if (g1 >= 0) {
      const int c0 = g0 / T2_CG, c1 = g1 / T2_CG;
      const int rl = r0 + 2 * jj;                  // A fragment: rows rl, rl + 1 of column 4 g + q
      const unsigned long long r8 = (unsigned long long)((long long)rl * 8);
      auto run_load = [&](const uint4 raw) -> v2d {
        const unsigned long long rz = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
        const bool ok = ((unsigned)(rl - (int)raw.z) <= raw.w) && !(a.ablate & 4);
        return ld2(ok ? rz + r8 : zaddr);
      };
      // the fragments of the slab's first groups are requested before the chunks in front of it are waited through
      v2d ring[RING];
#pragma unroll
      for (int u = 0; u < RING; ++u) ring[u] = run_load(rq[4 * min(c0 * T2_CG + u, KGP)]);
      uint4 raw = rq[4 * min(c0 * T2_CG + RING, KGP)];
      for (; ch < c0; ++ch) {
        chunk_open(ch);
        chunk_close(ch);
      }
      for (; ch <= c1; ++ch) {
        chunk_open(ch);
        const int gb = ch * T2_CG;
        const double* const bb = bq + (ch & 1) * 32 * T2_KP;
        double b0 = bb[0], b1 = bb[16 * T2_KP];
#pragma unroll
        for (int u = 0; u < T2_CG; ++u) {
          const int g = gb + u;
          // (the order is pinned: record read one slot ahead | matrix instructions | refill of the slot they have read)
          const uint4 raw_n = rq[4 * min(g + RING + 1, KGP)];
          double n0 = 0.0, n1 = 0.0;
          if (u + 1 < T2_CG) {
            n0 = bb[4 * (u + 1)];
            n1 = bb[16 * T2_KP + 4 * (u + 1)];
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g >= g0 && g <= g1 && !(a.ablate & 2)) {
            acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u % RING][0], b0, acc00, 0, 0, 0);
            acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u % RING][1], b0, acc10, 0, 0, 0);
            acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u % RING][0], b1, acc01, 0, 0, 0);
            acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u % RING][1], b1, acc11, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          ring[u % RING] = run_load(raw);
          raw = raw_n;
          b0 = n0;
          b1 = n1;
          __builtin_amdgcn_sched_barrier(0);
        }
        chunk_close(ch);
      }
    }
    // the slab is done (no k group reaches it: zeros -- entries of X alone, if any): its epilogue, under the matrix
    // instructions of the waves that are still multiplying
    if (a.ablate & 1) {
      if (acc00[0] + acc10[1] + acc01[2] + acc11[3] == 1.2345e300) grp[0].colmask[0] = 1u;
      continue;
    }
    epilogue(0, r0, acc00, acc10);
    epilogue(1, r0, acc01, acc11);
  }
  for (; ch < NCH; ++ch) {
    chunk_open(ch);
    chunk_close(ch);
  }

  // ---- the pair's two column blocks
  if constexpr (EPI != 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const double x = wave_sum_f64(dsum[c]), y = wave_sum_f64(tsum[c]);
      if (lane == 0) {
        grp[c].red[2 * wave] = x;
        grp[c].red[2 * wave + 1] = y;
      }
    }
    if (a.fzv.prod) {   // (threads 0..511 hold column block 0, the others column block 1)
      const long long ps = wave_sum_i64(nprod);
      if (lane == 0 && ps) atomicAdd(reinterpret_cast<unsigned long long*>(grp[wave >> 3].misc + 2), (unsigned long long)ps);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (lane == 0 && pn[c]) atomicAdd(&grp[c].misc[1], pn[c]);
  __syncthreads();
  // from here on: waves 0..7 finish column block 0, waves 8..15 column block 1 (a half whose block is not active only keeps
  // the barriers company)
  const int c = wave >> 3, htid = tid & 511;
  const bool mine = act_[c];
  T2Group& G = grp[c];
  const int b = 2 * p + c, lo = lo_[c], w = w_[c];
  const int64_t tbase = tb_[c];
  const int T = mine ? w / T2_ROWS : 0, rend = lo + w;
  auto bsync = [&]() { __syncthreads(); };
  if constexpr (EPI == 2) {
    const int nd = mine ? G.misc[0] : 0;
    if (mine && htid < SLAB_J) {   // every stored row of X(:, j) must be a row of this block's window
      const int jt = min(b * SLAB_J + htid, a.ncols - 1);
      const int x0 = a.fzv.xmin[jt], x1 = a.fzv.xmax[jt];
      if (b * SLAB_J + htid < a.ncols && x1 >= x0 && (x0 < lo || x1 >= lo + w)) atomicOr(a.fzv.flag, 1);
    }
    if (nd > T2_DEFER) {
      if (htid == 0) atomicOr(a.fzv.flag, 1);
    } else {
      for (int i = htid; i < nd; i += 512) {
        const int4 e = *reinterpret_cast<const int4*>(&G.dlist[i]);   // (r, jj, prow, pad)
        const bool kept = e.z > G.col_pmax[e.y];
        if (kept) {
          atomicAdd(&G.col_cnt[e.y], 1);
          atomicMin(&G.col_first[e.y], e.x);
          atomicMax(&G.col_last[e.y], e.x);
        }
        G.dlist[i].pad = kept ? 1 : 0;
      }
    }
    bsync();
    if (nd <= T2_DEFER && htid < nd) {
      const T2Defer e = G.dlist[htid];
      if (e.pad) {
        int rank = 0;
        for (int m2 = 0; m2 < nd; ++m2) {
          const T2Defer f = G.dlist[m2];
          rank += (f.pad && (f.r < e.r || (f.r == e.r && f.jj < e.jj))) ? 1 : 0;
        }
        G.dsums[rank] = __dmul_rn(e.o, e.d);
        G.dsums[T2_DEFER + rank] = (e.r == b * SLAB_J + e.jj + a.fzv.col_offset) ? e.o : 0.0;
      }
    }
    bsync();
  }
  if (mine && htid < SLAB_J) {
    const int jt = b * SLAB_J + htid;
    const int cf = G.col_first[htid], cl = G.col_last[htid];
    if (jt < a.ncols) {
      a.count[jt] = G.col_cnt[htid];
      a.ofirst[jt] = cf;
      a.olast[jt] = cl;
      a.ooff[jt] = tbase + (int64_t)htid * w + (cl >= cf ? cf - lo : 0);
    }
  }
  int tk0 = INT_MAX, tk1 = -1;
#pragma unroll
  for (int cc = 0; cc < SLAB_J; ++cc) {
    tk0 = min(tk0, G.col_first[cc]);
    tk1 = max(tk1, G.col_last[cc]);
  }
  if constexpr (EPI != 0) {
    if (mine && htid == 0) {
      a.otoff[b] = tbase + (tk1 >= tk0 ? (int64_t)(tk0 - lo) * SLAB_J : 0);
      a.fzv.pnnz[b] = G.misc[1];
      if (a.fzv.prod) a.fzv.prod[b] = *reinterpret_cast<long long*>(G.misc + 2);
    }
    if (mine && htid == 64) {
      double x = 0.0, y = 0.0;
      for (int qq = 0; qq < T2_NW; ++qq) {
        x = __dadd_rn(x, G.red[2 * qq]);
        y = __dadd_rn(y, G.red[2 * qq + 1]);
      }
      if constexpr (EPI == 2) {
        const int nd = min(G.misc[0], T2_DEFER);
        int nk = 0;
        for (int m2 = 0; m2 < nd; ++m2) nk += G.dlist[m2].pad;
        for (int m2 = 0; m2 < nk; ++m2) {
          x = __dadd_rn(x, G.dsums[m2]);
          y = __dadd_rn(y, G.dsums[T2_DEFER + m2]);
        }
      }
      a.fzv.part[2 * b] = x;
      a.fzv.part[2 * b + 1] = y;
    }
  }
  // holes: a slab strictly inside a column's run that was skipped above holds zeros
  for (int pp = htid; pp < T * SLAB_J; pp += 512) {
    const int t = pp >> 4, cc = pp & 15;
    const unsigned cmk = G.colmask[t];
    const int cf = G.col_first[cc], cl = G.col_last[cc];
    const int r0h = lo + T2_ROWS * t;
    if (cl >= cf && r0h + T2_ROWS - 1 >= cf && r0h <= cl && !((cmk >> cc) & 1u)) {
      double* dst = a.out_val + (tbase + (int64_t)cc * w - lo);
      for (int r = r0h; r < min(r0h + T2_ROWS, rend); ++r) dst[r] = 0.0;
    }
    if constexpr (EPI != 0) {
      if (a.fzv.tiles && tk1 >= tk0 && r0h + T2_ROWS - 1 >= tk0 && r0h <= tk1 && cmk == 0u) {
        double* dst = a.fzv.tiles + (tbase - (int64_t)lo * SLAB_J + cc);
        for (int r = r0h; r < min(r0h + T2_ROWS, rend); ++r) dst[(int64_t)r * SLAB_J] = 0.0;
      }
    }
  }
  if constexpr (EPI == 2) {
    const int nd = mine ? min(G.misc[0], T2_DEFER) : 0;
    bsync();   // (the zeros above first)
    for (int i = htid; i < nd; i += 512) {
      const T2Defer e = G.dlist[i];
      if (!e.pad) continue;
      a.out_val[tbase + (int64_t)e.jj * w + (e.r - lo)] = e.o;
      if (a.fzv.tiles) a.fzv.tiles[tbase + (int64_t)(e.r - lo) * SLAB_J + e.jj] = e.o;
    }
  }
}

}  // namespace

// true: launched (the caller reads *fail back with its totals: non-zero = a pair did not fit, nothing of the launch counts)
bool launch_spgemm_tile2(const TileLaunch& L, int* fail) {
  if (L.rows != 2 || L.labelled || L.brun_val == nullptr || L.nblocks <= 0) return false;
  // union of two neighbouring blocks: at most the larger one plus what the neighbour adds; checked per pair in the kernel
  if (L.max_w > T2_MAXS * T2_ROWS || L.max_kn <= 0) return false;
  Tile2Args a;
  a.runs = static_cast<const SlabRun*>(L.runs);
  a.blk_kmin = L.blk_kmin; a.blk_kn = L.blk_kn; a.blk_lo = L.blk_lo; a.blk_w = L.blk_w; a.blk_toff = L.blk_toff;
  a.out_val = L.out_val; a.count = L.count; a.ofirst = L.ofirst; a.olast = L.olast; a.ooff = L.ooff; a.otoff = L.otoff;
  a.alpha = L.alpha; a.threshold = L.threshold; a.dense_rule = L.dense_rule; a.ncols = L.ncols; a.nblocks = L.nblocks;
  a.npairs = (L.nblocks + 1) / 2;
  a.kcap = ((L.max_kn + 32 + 4 + T2_CK - 1) / T2_CK + 1) * T2_CK;
  if (tile2_lds_bytes(a.kcap) > 64 * 1024) return false;
  if (L.fz) a.fzv = *static_cast<const SlabFuseArgs*>(L.fz);
  a.brun_first = L.brun_first; a.brun_last = L.brun_last; a.brun_off = L.brun_off; a.brun_val = L.brun_val;
  static DevBuf<double>* zeros = nullptr;
  if (!zeros) {
    zeros = new DevBuf<double>(8);
    zeros->zero();
  }
  a.zero = zeros->p;
  a.fail = fail;
  const char* abl = std::getenv("NTPOLY_AMD_T2_ABLATE");   // (timing experiments, read at every launch)
  a.ablate = abl ? std::atoi(abl) : 0;
  const size_t lds = tile2_lds_bytes(a.kcap);
  if (L.epi == 0) hipLaunchKernelGGL((k_spgemm_tile2<0>), dim3(xcd_grid(a.npairs)), dim3(T2_NT), lds, stream(), a);
  else if (L.epi == 1) hipLaunchKernelGGL((k_spgemm_tile2<1>), dim3(xcd_grid(a.npairs)), dim3(T2_NT), lds, stream(), a);
  else hipLaunchKernelGGL((k_spgemm_tile2<2>), dim3(xcd_grid(a.npairs)), dim3(T2_NT), lds, stream(), a);
  return true;
}

}  // namespace ntp
