// SpGEMM numeric phase on the FP64 matrix cores, second geometry: TWO blocks of 16 output columns per workgroup.  Same
// operands, same plan, same results (bit for bit) as k_spgemm_tile (spgemm_tile.hip; MultiplyBlock.f90:9-36 +
// PruneList.f90:8-38 and, fused, the TRS2 update of DensityMatrixSolversModule.F90:380-413 through
// AddSparseVectors.f90:21-70).
//
// Why: k_spgemm_tile reads every 32 x 4 fragment of A once per block of 16 columns -- 1 KB from the L2 per two matrix
// instructions, 10-15 GB per launch of the headline -- and spends as many vector instructions on a fragment's address as on
// anything else.  Here a fragment feeds both column blocks of a pair: half the L2 traffic and half the address arithmetic
// per matrix instruction.
//
//   workgroup = 16 waves = one per CU (the multipliers of 32 columns over the pair's k range fill most of the LDS),
//   pair of column blocks (2p, 2p + 1), union row window [LO, HI) cut into SLABS of 32 rows; wave v takes slabs v, v + 16, ...
//   one after the other, each: 32 rows x 32 columns of partial sums = 4 tiles = 32 VGPRs; per k group ONE run record (LDS),
//   ONE 16-byte run load, two multiplier reads (LDS), FOUR matrix instructions; when the slab's k groups are done, its
//   epilogue (prune, fused update, energy, trace, result runs) -- under the matrix instructions of the other waves.
//   After the prologue the waves do not meet again until the pair's column statistics are written.
//
// (A first version streamed the multipliers through LDS in chunks of 32 k behind workgroup barriers: every slab epilogue and
// every wave's memory stalls sat on the critical path of all sixteen waves -- 2.28 ms per headline step against 1.54;
// profiles/README.md.)
//
// Arithmetic: as k_spgemm_tile -- v_mfma_f64_16x16x4_f64 is a chain of fma() over ascending k, the groups follow in
// ascending k, zero padding is exact -- so every C(i, j) is the FMA chain of the reference's FP-contracted build.
#include "../spgemm_tile.hpp"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../device_util.hpp"
#include "../kernels.hpp"

namespace ntp {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ inline v2d ld2(unsigned long long addr) { return *reinterpret_cast<const v2d __attribute__((address_space(1)))*>(addr); }
__device__ inline v2d ld2(const double* p) { return ld2(reinterpret_cast<unsigned long long>(p)); }
__device__ inline void st2(double* p, const v2d& v) {
  *reinterpret_cast<v2d __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(p)) = v;
}

constexpr int T2_MAXNW = 16;         // waves per workgroup: a template parameter, 12 or 16
constexpr int T2_ROWS = 32;          // rows of a slab (two rows per lane of the A fragment)
constexpr int T2_MAXS = 64;          // slabs of a window (2048 rows)
// multipliers in LDS: column jc of the pair at Bs[jc * kp + (k - KMIN)], kp = 2 (mod 32) doubles: lane (jj, q) of a
// fragment reads word kp jj + q + 4 g -- the 32 lanes of a ds_read_b64 group fall into 64 different banks
__host__ __device__ inline int tile2_kp(int kcap) { return ((kcap + 31) / 32) * 32 + 2; }
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
  int kcap;                 // k the LDS is sized for (a multiple of 32, >= the largest union k range + 8)
  SlabFuseArgs fzv;
  const int32_t *brun_first, *brun_last;
  const int64_t* brun_off;
  const double* brun_val;
  const double* zero;
  int* fail;                // set when a pair does not fit (its k range or window beyond the LDS): nothing of the launch may be used
  long long* stamps;        // diagnostics (NTPOLY_AMD_T2_STAMPS): [sampled pair][wave][8] s_memtime stamps, or nullptr
  int order;                // slab order of a wave (experiments): 0 = s, s + NW, ...; 1 = centre first
  int ablate;               // timing experiments (WRONG results): NTPOLY_AMD_T2_ABLATE = 1: no epilogues
};

// per column block of the pair (LDS)
struct alignas(16) T2Col {            // what the fused epilogue needs of column j of X and D: extents and the address of (hypothetical) row 0
  int32_t xf, xl, df, dl;             // (an empty column: first > last)
  unsigned long long xrz, drz;
};
struct T2Group {
  T2Col col[16];
  unsigned colmask[T2_MAXS];
  int col_cnt[16], col_first[16], col_last[16], col_pmax[16], col_pad[16];
  double red[2 * T2_MAXNW];
  int misc[4];                        // [0] deferred elements, [1] product entries, [2..3] products (64 bit)
  T2Defer dlist[T2_DEFER];
  double dsums[2 * T2_DEFER];
};

__host__ __device__ inline size_t tile2_lds_bytes(int kcap) {
  return (size_t)32 * tile2_kp(kcap) * 8 + (size_t)(kcap + 8) * sizeof(T2Rec) + (size_t)2 * (kcap / 4 + 2) * 4 + 16 + 2 * sizeof(T2Group) + 64;
}

#define T2STAMP(i) do { if (a.stamps && lane == 0 && (p & 63) == 0 && (p >> 6) < 128) a.stamps[(((p >> 6) * T2_MAXNW) + wave) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

template <int EPI, int T2_RING, int T2_NW>   // T2_RING: run loads in flight per wave (k groups ahead); T2_NW: waves per workgroup
__global__ __launch_bounds__(T2_NW* WAVE) void k_spgemm_tile2(const Tile2Args a) {
  constexpr int T2_NT = T2_NW * WAVE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int p = xcd_block(a.npairs);
  if (p < 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  T2STAMP(0);
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
  const int KN = KEND - KMIN, KG = (KN + 3) >> 2, K4 = KG * 4;
  if (S > T2_MAXS || K4 + 8 > a.kcap || ((HI - LO) % T2_ROWS) != 0) {
    if (tid == 0) atomicOr(a.fail, 1);
    return;
  }
  // ---- LDS
  const int KP = tile2_kp(a.kcap);
  double* Bs = reinterpret_cast<double*>(smem);                                  // [32 columns][KP]
  T2Rec* recs = reinterpret_cast<T2Rec*>(Bs + 32 * KP);                          // [kcap + 8]
  int* grmin = reinterpret_cast<int*>(recs + a.kcap + 8);                        // [kcap / 4 + 2]
  int* grmax = grmin + (a.kcap / 4 + 2);
  T2Group* grp = reinterpret_cast<T2Group*>((reinterpret_cast<uintptr_t>(grmax + (a.kcap / 4 + 2)) + 15) & ~(uintptr_t)15);

  // ---- block prologue: run records of the k range (a thread each), row range of every k group; the multipliers
  // (the extents of this thread's multiplier column first: its values are a second round trip behind them)
  constexpr int LPC = T2_NT / 32;   // threads per multiplier column: consecutive k of a column's run to consecutive threads
  const int bcol = tid / LPC, bk = tid % LPC;
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
  {
    const uint4* __restrict__ rp = reinterpret_cast<const uint4*>(a.runs + KMIN);
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
  [[maybe_unused]] long long nprod_wave = 0;   // (lane 0 of a wave: the products of the multipliers it staged)
  // multipliers: thread (column bcol = tid / 32 of the pair, k offsets tid % 32 + 32 i) -- 32 consecutive k of a column's
  // run are 256 contiguous bytes; all of a thread's loads are in flight together
  {
    double* const bdst = Bs + bcol * KP + bk;
    [[maybe_unused]] long long nprod = 0;
    constexpr int BU = 12;   // (the headline's k range in one round trip)
    for (int k0 = 0; k0 < K4; k0 += LPC * BU) {
      double v[BU];
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        const int k = KMIN + k0 + LPC * u + bk;
        v[u] = (k >= bf && k <= bl) ? bp[k] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < BU; ++u) {
        if (k0 + LPC * u + bk < K4) bdst[k0 + LPC * u] = v[u];
        if constexpr (EPI != 0) {
          if (a.fzv.prod && v[u] != 0.0) nprod += a.fzv.in_count ? (long long)a.fzv.in_count[KMIN + k0 + LPC * u + bk] : 1ll;
        }
      }
    }
    if constexpr (EPI != 0) {
      if (a.fzv.prod) {   // (the first half of the waves holds column block 0, the second half column block 1)
        const long long ps = wave_sum_i64(nprod);
        if (lane == 0 && ps) nprod_wave = ps;
      }
    }
  }
  for (int i = tid; i < 2 * (int)(sizeof(T2Group) / 4); i += T2_NT) reinterpret_cast<int*>(grp)[i] = 0;
  __syncthreads();
  if (tid < 32) {
    T2Group& G = grp[tid >> 4];
    G.col_first[tid & 15] = INT_MAX;
    G.col_last[tid & 15] = -1;
    G.col_pmax[tid & 15] = -1;
    if constexpr (EPI != 0) {
      const int c = tid >> 4, j = (2 * p + c) * SLAB_J + (tid & 15);
      T2Col cc;
      cc.xf = INT_MAX; cc.xl = -1; cc.df = INT_MAX; cc.dl = -1;
      cc.xrz = cc.drz = reinterpret_cast<unsigned long long>(a.zero);
      if (act_[c] && j < a.ncols) {
        const int d0 = a.fzv.dmin[j], d1 = a.fzv.dmax[j];
        if (d1 >= d0) {
          cc.df = d0;
          cc.dl = d1;
          cc.drz = reinterpret_cast<unsigned long long>(a.fzv.dexp + (a.fzv.doff[j] - d0));
        }
        if constexpr (EPI == 2) {
          const int x0 = a.fzv.xmin[j], x1 = a.fzv.xmax[j];
          if (x1 >= x0) {
            cc.xf = x0;
            cc.xl = x1;
            cc.xrz = reinterpret_cast<unsigned long long>(a.fzv.xexp + (a.fzv.xoff[j] - x0));
          }
        }
      }
      G.col[tid & 15] = cc;
    }
  }

  __syncthreads();

  T2STAMP(1);
  // ---- per-lane constants
  const int jj = lane & 15, q = lane >> 4;
  const double* const zp = a.zero;
  const unsigned long long zaddr = reinterpret_cast<unsigned long long>(zp);
  const double alpha = a.alpha, thr = a.threshold;
  const bool dense_rule = (a.dense_rule & 1) != 0;
  const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(recs) + q;       // record of group g: rq[4 g]
  const double* const bq = Bs + jj * KP + q;                                     // column block c, group g: bq[c * 16 * KP + 4 g]

  // per column block: this lane's column
  [[maybe_unused]] double dsum[2] = {0.0, 0.0}, tsum[2] = {0.0, 0.0};
  int pn[2] = {0, 0};

  // ---- epilogue of one slab for one column block (lane holds rows r0 + 2 (4 v + q) + m, v = 0..3, m = 0..1, of column jj),
  // in stages so that what it reads of X and D is in flight under the arithmetic in front of it: ep_open (the column's
  // constants from LDS), ep_load (half h: rows of v = 2 h, 2 h + 1), ep_live, ep_half (the elements of half h), ep_close
  struct EpCol {
    bool in;                       // the slab lies in this column block's window
    int t;                         // its slab number there
  };
  struct EpVals { v2d x[2], d[2]; };
  struct EpState {
    v2d res[4];
    unsigned long long anykeep;
    int c_l, f_l, l_l, pm_l;
  };
  auto ep_open = [&](const int c, const int r0) -> EpCol {
    EpCol e;
    e.in = act_[c] && r0 >= lo_[c] && r0 < lo_[c] + w_[c];
    e.t = (r0 - lo_[c]) / T2_ROWS;
    return e;
  };
  // (the column's extents and addresses are read from LDS where they are needed: kept in registers across the stages they
  // cost what the partial sums of the other column block need)
  auto ep_load = [&](const int c, const int r0, const int hh) -> EpVals {
    EpVals v;
    [[maybe_unused]] T2Col cc;
    if constexpr (EPI != 0) {
      const volatile int4* cp = reinterpret_cast<const volatile int4*>(&grp[c].col[jj]);
      const int4 c0 = const_cast<const int4&>(cp[0]), c1 = const_cast<const int4&>(cp[1]);   // (read here, every time)
      cc.xf = c0.x; cc.xl = c0.y; cc.df = c0.z; cc.dl = c0.w;
      cc.xrz = (unsigned long long)(unsigned)c1.x | ((unsigned long long)(unsigned)c1.y << 32);
      cc.drz = (unsigned long long)(unsigned)c1.z | ((unsigned long long)(unsigned)c1.w << 32);
    }
#pragma unroll
    for (int vv2 = 0; vv2 < 2; ++vv2) {
      v.x[vv2] = v2d{0.0, 0.0};
      v.d[vv2] = v2d{0.0, 0.0};
      if constexpr (EPI != 0) {
        const int rb = r0 + 2 * (4 * (2 * hh + vv2) + q);
        if constexpr (EPI == 2) v.x[vv2] = ld2(((rb + 1 >= cc.xf) & (rb <= cc.xl)) ? cc.xrz + 8ull * (unsigned long long)(long long)rb : zaddr);
        v.d[vv2] = ld2(((rb + 1 >= cc.df) & (rb <= cc.dl)) ? cc.drz + 8ull * (unsigned long long)(long long)rb : zaddr);
      }
    }
    return v;
  };
  // slabs in which nothing can be kept (an entry of X in the slab: at least its run reaches it)
  auto ep_live = [&](const int c, const EpCol& e, const int r0, const v4d& acc0, const v4d& acc1) -> bool {
    if (!e.in) return false;
    bool live = false;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const double vv = m ? acc1[v] : acc0[v];
        live |= dense_rule ? (fabs(vv) > thr) : (fabs(__dmul_rn(alpha, vv)) > thr);
      }
    }
    if constexpr (EPI == 2) {
      const int xf = grp[c].col[jj].xf, xl = grp[c].col[jj].xl;
      live |= (r0 + T2_ROWS - 1 >= xf) & (r0 <= xl);
    }
    const bool any = __ballot(live) != 0ull;
    if (!any && lane == 0) grp[c].colmask[e.t] = 0u;
    return any;
  };
  auto ep_begin = [&]() -> EpState {
    EpState st;
    st.anykeep = 0;
    st.c_l = 0; st.f_l = INT_MAX; st.l_l = -1; st.pm_l = -1;
    return st;
  };
  auto ep_half = [&](const int c, const EpCol& e, const int r0, const int hh, const EpVals& vals, const v4d& acc0, const v4d& acc1, EpState& st) {
    T2Group& G = grp[c];
    [[maybe_unused]] const double am = a.fzv.am, bm = a.fzv.bm, thr_m = a.fzv.thr_m;
    [[maybe_unused]] int xlast = -1;
    if constexpr (EPI == 2) xlast = grp[c].col[jj].xl;
    [[maybe_unused]] const int diag = (2 * p + c) * SLAB_J + jj + (EPI != 0 ? a.fzv.col_offset : 0);
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
        if constexpr (EPI != 2) {
          keep = ha;
          o = sv;
        } else {
          const double bv = vals.x[vv2][m];
          const bool hb = bv != 0.0;
          const double bs = __dmul_rn(bm, bv);
          const double wa = __dmul_rn(am, sv);
          const double both = __dadd_rn(wa, bs);
          o = ha ? (hb ? both : wa) : bs;
          const bool big = fabs(o) > thr_m;
          // AddSparseVectors (inc_decide): both present -> threshold on the sum; one present -> threshold unless it lies
          // beyond the other column's last entry.  "Beyond the product column's last kept entry" is not known yet for an
          // element of X alone that fails the threshold: decided when the block is done (dlist)
          if (ha) {
            keep = (!hb && r > xlast) || big;
          } else {
            keep = hb && big;
            if (hb && !big) {
              const int slot = atomicAdd(&G.misc[0], 1);
              if (slot < T2_DEFER) {   // (rare: the value of D beside it is fetched here, not carried through the common path)
                *reinterpret_cast<int4*>(&G.dlist[slot]) = make_int4(r, jj, r, 0);
                G.dlist[slot].o = o;
                const T2Col& cc = G.col[jj];
                G.dlist[slot].d = (r >= cc.df && r <= cc.dl) ? reinterpret_cast<const double*>(cc.drz)[r] : 0.0;
              }
            }
          }
        }
        pn[c] += (int)__popcll(__ballot(ha));
        st.anykeep |= __ballot(keep);
        if constexpr (EPI != 0) st.pm_l = max(st.pm_l, ha ? r : -1);
        st.c_l += keep ? 1 : 0;
        st.f_l = min(st.f_l, keep ? r : INT_MAX);
        st.l_l = max(st.l_l, keep ? r : -1);
        st.res[v][m] = keep ? o : 0.0;
        __builtin_amdgcn_sched_barrier(0);   // (one element after the other: interleaved they need twice the registers)
      }
    }
    // energy and trace terms of the half, in the same element order, from what was kept (a dropped element is a zero: its
    // term +-0 leaves the sums as they are -- they start at +0 and never become -0): the values of D are not live above
    if constexpr (EPI != 0) {
#pragma unroll
      for (int vv2 = 0; vv2 < 2; ++vv2) {
        const int v = 2 * hh + vv2;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int r = r0 + 2 * (4 * v + q) + m;
          const double o = st.res[v][m];
          dsum[c] = __dadd_rn(dsum[c], __dmul_rn(o, vals.d[vv2][m]));
          tsum[c] = __dadd_rn(tsum[c], r == diag ? o : 0.0);
        }
      }
    }
  };
  auto ep_close = [&](const int c, const EpCol& e, const int r0, const EpState& st) {
    T2Group& G = grp[c];
    const unsigned cm = (unsigned)((st.anykeep | (st.anykeep >> 16) | (st.anykeep >> 32) | (st.anykeep >> 48)) & 0xffffull);
    if (st.c_l) {
      atomicAdd(&G.col_cnt[jj], st.c_l);
      atomicMin(&G.col_first[jj], st.f_l);
      atomicMax(&G.col_last[jj], st.l_l);
    }
    if constexpr (EPI == 2) {
      if (st.pm_l >= 0) atomicMax(&G.col_pmax[jj], st.pm_l);
    }
    const int lo = lo_[c], w = w_[c];
    const int64_t tbase = tb_[c];
    if ((cm >> jj) & 1u) {   // the column has an entry in this slab: its 32 rows are written (zeros = holes)
      double* const orun = a.out_val + (tbase + (int64_t)jj * w - lo);
#pragma unroll
      for (int v = 0; v < 4; ++v) st2(orun + (r0 + 2 * (4 * v + q)), st.res[v]);
    }
    if constexpr (EPI != 0) {
      if (cm && a.fzv.tiles) {
        double* const otile = a.fzv.tiles + (tbase - (int64_t)lo * SLAB_J + jj);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int m = 0; m < 2; ++m) otile[(int64_t)(r0 + 2 * (4 * v + q) + m) * SLAB_J] = st.res[v][m];
        }
      }
    }
    if (lane == 0) G.colmask[e.t] = cm;
  };

  // ---- the slabs of this wave, one after the other
  int sidx = 2;
  const int mid = (S - 1) >> 1;
  for (int si = wave; si < S; si += T2_NW) {
    // centre first: the slabs in the middle of the window have the longest k ranges -- they start together, one per wave,
    // and the short slabs of the window's edges (and their short epilogues) fill the end of the pair's life
    const int s = a.order == 0 ? si : (si & 1) ? mid + ((si + 1) >> 1) : mid - (si >> 1);
    const int r0 = LO + T2_ROWS * s;
    int g0 = INT_MAX, g1 = -1;   // the k groups that can reach the slab: a ballot over the groups' row ranges
    for (int c = 0; c < KG; c += WAVE) {
      const int gq = min(c + lane, KG);
      const bool hit = gq < KG && grmin[gq] <= r0 + T2_ROWS - 1 && grmax[gq] >= r0;
      const unsigned long long m = __ballot(hit);
      if (m) {
        if (g0 == INT_MAX) g0 = c + (int)__builtin_ctzll(m);
        g1 = c + 63 - (int)__builtin_clzll(m);
      }
    }
    v4d acc00 = v4d{0.0, 0.0, 0.0, 0.0}, acc10 = acc00, acc01 = acc00, acc11 = acc00;   // [row parity][column block]
    EpCol e0;
    EpVals va;
    va.x[0] = va.x[1] = va.d[0] = va.d[1] = v2d{0.0, 0.0};
    if (g1 >= g0) {
      // (the stores of the slab before must have left the counter the run loads are counted on: vector loads and stores
      // share it on this target and return out of order with respect to each other -- with a store pending the compiler
      // can only wait for ALL loads in front of every group of matrix instructions, not for the oldest)
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
      // Software pipeline over the k groups g0 .. g1 (records are padded behind the last group -- the look-ahead is clamped
      // to it -- and a group beyond g1 has no row in this slab): the record of group g + RING + 1 is read from LDS while
      // the run load of group g + RING is issued from the record read one step earlier, the multiplier rows of g + 1
      // are read, and group g -- fragment landed RING steps ago -- is multiplied into the four tiles.
      const int rl = r0 + 2 * jj;                  // A fragment: rows rl, rl + 1 of column 4 g + q
      const unsigned long long r8 = (unsigned long long)((long long)rl * 8);
      auto run_load = [&](const uint4 raw) -> v2d {
        const unsigned long long rz = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
        const bool ok = (unsigned)(rl - (int)raw.z) <= raw.w;
        return ld2(ok ? rz + r8 : zaddr);
      };
      v2d ring[T2_RING];
#pragma unroll
      for (int u = 0; u < T2_RING; ++u) ring[u] = run_load(rq[4 * min(g0 + u, KG)]);
      uint4 raw = rq[4 * min(g0 + T2_RING, KG)];
      const double* const b1q = bq + 16 * KP;
      double b0 = bq[4 * g0], b1 = b1q[4 * g0];
      int g = g0;
      for (; g + T2_RING - 1 <= g1; g += T2_RING) {
#pragma unroll
        for (int u = 0; u < T2_RING; ++u) {
          // (the order is pinned: record read one slot ahead | matrix instructions | refill of the slot they have read)
          const uint4 raw_n = rq[4 * min(g + u + T2_RING + 1, KG)];
          const int gn = min(g + u + 1, KG - 1);
          const double n0 = bq[4 * gn], n1 = b1q[4 * gn];
          __builtin_amdgcn_sched_barrier(0);
          acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][0], b0, acc00, 0, 0, 0);
          acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][1], b0, acc10, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][0], b1, acc01, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][1], b1, acc11, 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          ring[u] = run_load(raw);
          raw = raw_n;
          b0 = n0;
          b1 = n1;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // what the epilogue of column block 0 reads first is requested here, in front of the last groups
      e0 = ep_open(0, r0);
      if (e0.in) va = ep_load(0, r0, 0);
#pragma unroll
      for (int u = 0; u < T2_RING - 1; ++u) {
        if (g + u <= g1) {
          const double t0 = bq[4 * (g + u)], t1 = b1q[4 * (g + u)];
          acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][0], t0, acc00, 0, 0, 0);
          acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][1], t0, acc10, 0, 0, 0);
          acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][0], t1, acc01, 0, 0, 0);
          acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][1], t1, acc11, 0, 0, 0);
        }
      }
    } else {   // (no k group reaches the slab: zeros -- entries of X alone, if any)
      e0 = ep_open(0, r0);
      if (e0.in) va = ep_load(0, r0, 0);
    }
    T2STAMP(sidx); ++sidx;
    // the slab is done: its epilogue, under the matrix instructions of the waves that are still multiplying
    if (a.ablate & 1) {
      if (acc00[0] + acc10[1] + acc01[2] + acc11[3] + va.d[0][0] == 1.2345e300) grp[0].colmask[0] = 1u;
      continue;
    }
    {
      const bool live0 = ep_live(0, e0, r0, acc00, acc10);
      EpVals vb;
      if (live0) {
        vb = ep_load(0, r0, 1);
        EpState st = ep_begin();
        ep_half(0, e0, r0, 0, va, acc00, acc10, st);
        const EpCol e1 = ep_open(1, r0);
        if (e1.in) va = ep_load(1, r0, 0);
        ep_half(0, e0, r0, 1, vb, acc00, acc10, st);
        ep_close(0, e0, r0, st);
        if (ep_live(1, e1, r0, acc01, acc11)) {
          vb = ep_load(1, r0, 1);
          EpState s1 = ep_begin();
          ep_half(1, e1, r0, 0, va, acc01, acc11, s1);
          ep_half(1, e1, r0, 1, vb, acc01, acc11, s1);
          ep_close(1, e1, r0, s1);
        }
      } else {
        const EpCol e1 = ep_open(1, r0);
        if (e1.in) va = ep_load(1, r0, 0);
        if (ep_live(1, e1, r0, acc01, acc11)) {
          vb = ep_load(1, r0, 1);
          EpState s1 = ep_begin();
          ep_half(1, e1, r0, 0, va, acc01, acc11, s1);
          ep_half(1, e1, r0, 1, vb, acc01, acc11, s1);
          ep_close(1, e1, r0, s1);
        }
      }
    }
    T2STAMP(sidx); ++sidx;
  }
  T2STAMP(6);

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
    if (a.fzv.prod && lane == 0 && nprod_wave)
      atomicAdd(reinterpret_cast<unsigned long long*>(grp[wave / (T2_NW / 2)].misc + 2), (unsigned long long)nprod_wave);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
    if (lane == 0 && pn[c]) atomicAdd(&grp[c].misc[1], pn[c]);
  __syncthreads();
  // from here on: the first half of the waves finishes column block 0, the second half column block 1 (a half whose block
  // is not active only keeps the barriers company)
  constexpr int HT = T2_NT / 2;
  const int c = wave / (T2_NW / 2), htid = tid - c * HT;
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
      for (int i = htid; i < nd; i += HT) {
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
  for (int pp = htid; pp < T * SLAB_J; pp += HT) {
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
    for (int i = htid; i < nd; i += HT) {
      const T2Defer e = G.dlist[i];
      if (!e.pad) continue;
      a.out_val[tbase + (int64_t)e.jj * w + (e.r - lo)] = e.o;
      if (a.fzv.tiles) a.fzv.tiles[tbase + (int64_t)(e.r - lo) * SLAB_J + e.jj] = e.o;
    }
  }
  T2STAMP(7);
}

}  // namespace

// true: launched (the caller reads *fail back with its totals: non-zero = a pair did not fit, nothing of the launch counts)
bool launch_spgemm_tile2(const TileLaunch& L, int* fail) {
  if (L.rows != 2 || L.labelled || L.brun_val == nullptr || L.nblocks <= 0) return false;
  // (the union of two neighbouring blocks: checked per pair in the kernel)
  if (L.max_w + 32 > T2_MAXS * T2_ROWS || L.max_kn <= 0) return false;
  Tile2Args a;
  a.runs = static_cast<const SlabRun*>(L.runs);
  a.blk_kmin = L.blk_kmin; a.blk_kn = L.blk_kn; a.blk_lo = L.blk_lo; a.blk_w = L.blk_w; a.blk_toff = L.blk_toff;
  a.out_val = L.out_val; a.count = L.count; a.ofirst = L.ofirst; a.olast = L.olast; a.ooff = L.ooff; a.otoff = L.otoff;
  a.alpha = L.alpha; a.threshold = L.threshold; a.dense_rule = L.dense_rule; a.ncols = L.ncols; a.nblocks = L.nblocks;
  a.npairs = (L.nblocks + 1) / 2;
  a.kcap = ((L.max_kn + 16 + 12 + 31) / 32) * 32;    // (a neighbour adds its 16 columns to the k range of a banded operand)
  if (tile2_lds_bytes(a.kcap) > 156 * 1024) return false;
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
  static DevBuf<long long>* stamps = nullptr;
  const char* stf = std::getenv("NTPOLY_AMD_T2_STAMPS");
  if (stf && !stamps) stamps = new DevBuf<long long>(128 * T2_MAXNW * 8);
  if (stf) stamps->zero();
  a.stamps = stf ? stamps->p : nullptr;
  const char* rg = std::getenv("NTPOLY_AMD_T2_RING");   // (experiments)
  const int ring = rg ? std::atoi(rg) : 4;
  const char* od = std::getenv("NTPOLY_AMD_T2_ORDER");
  a.order = od ? std::atoi(od) : 1;
  const char* wv = std::getenv("NTPOLY_AMD_T2_WAVES");   // (experiments)
  const int nw = wv ? std::atoi(wv) : 12;
  auto go = [&](auto epi_tag, auto ring_tag, auto nw_tag) {
    constexpr int E = decltype(epi_tag)::value, RG = decltype(ring_tag)::value, NWV = decltype(nw_tag)::value;
    static bool raised = false;   // (one per instantiation)
    if (!raised) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spgemm_tile2<E, RG, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
      raised = true;
    }
    hipLaunchKernelGGL((k_spgemm_tile2<E, RG, NWV>), dim3(xcd_grid(a.npairs)), dim3(NWV * WAVE), lds, stream(), a);
  };
  auto by_ring = [&](auto epi_tag) {
    using W12 = std::integral_constant<int, 12>; using W16 = std::integral_constant<int, 16>;
    if (nw == 16) {
      if (ring == 8) go(epi_tag, std::integral_constant<int, 8>{}, W16{});
      else go(epi_tag, std::integral_constant<int, 4>{}, W16{});
    } else {
      if (ring == 8) go(epi_tag, std::integral_constant<int, 8>{}, W12{});
      else if (ring == 6) go(epi_tag, std::integral_constant<int, 6>{}, W12{});
      else go(epi_tag, std::integral_constant<int, 4>{}, W12{});
    }
  };
  if (L.epi == 0) by_ring(std::integral_constant<int, 0>{});
  else if (L.epi == 1) by_ring(std::integral_constant<int, 1>{});
  else by_ring(std::integral_constant<int, 2>{});
  if (stf) {
    std::vector<long long> h(128 * T2_MAXNW * 8);
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(h.data(), stamps->p, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* fp = std::fopen(stf, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
  }
  return true;
}

}  // namespace ntp
