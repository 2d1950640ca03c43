// SpGEMM numeric phase on the FP64 matrix cores (gfx950, v_mfma_f64_16x16x4_f64) for run-like real operands -- the
// operands and the results of the register-slab kernel (kernels.hip, MultiplyBlock.f90:9-36 + PruneList.f90:8-38),
// walked the other way round.
//
// A workgroup owns a block of 16 consecutive output columns and the row window [lo, lo + w) they can touch, as the
// slab kernel does.  The window is cut into TILES of 16 rows; a wave takes one tile at a time and walks the k range of
// the block that can reach it in groups of four consecutive k:
//
//     P(16 rows x 16 columns) += A(16 rows x 4 k) * B(4 k x 16 columns)        one v_mfma_f64_16x16x4_f64
//
// A comes from the expanded columns (dense runs, holes = 0; lane l reads row r0 + l % 16 of column k0 + l / 16, zero
// outside the run), B from the block's multiplier tile, copied to LDS once per block (lane l reads row k0 + l / 16,
// column l % 16: one conflict-free ds_read_b64).  The partial sums of a tile are 8 VGPRs and leave the registers when
// the tile's k range is done, so nothing about the window has to fit the register file: the epilogue (prune, the
// fused purification update, energy, trace, the result in slab form) runs per tile, hidden behind the other waves'
// matrix instructions.
//
// Arithmetic: the matrix instruction accumulates its four products in ascending k with ONE rounding each -- it is a
// chain of fma(), bit for bit (tools/micro/mfma_f64_probe.hip checks this on the device) -- and the groups follow in
// ascending k, so every C(i, j) is the FMA chain over ascending k that option spgemm_fma = 1 of the slab kernel
// computes and that the reference computes when it is built with FP contraction: the two
// kernels agree bit for bit, and both with the contracted reference build.  Zero padding is exact (fma(0, b, x) = x).
#include "spgemm_tile.hpp"
#include <string>

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
// R consecutive rows of a column as one value: a lane of the A operand loads them with ONE address computation and
// feeds R matrix instructions (rows r0 + R i + m, m = 0 .. R - 1, of a tile of 16 R rows)
template <int R> struct RowVec;
template <> struct RowVec<1> { typedef double type; };
template <> struct RowVec<2> { typedef v2d type; };
template <> struct RowVec<4> { typedef v4d type; };
template <int R> __device__ inline double rv_get(const typename RowVec<R>::type& x, int m) { return x[m]; }
template <> __device__ inline double rv_get<1>(const double& x, int) { return x; }
template <int R> __device__ inline void rv_set(typename RowVec<R>::type& x, int m, double v) { x[m] = v; }
template <> __device__ inline void rv_set<1>(double& x, int, double v) { x = v; }
// (loads through address space 1 are global_load, not flat_load)
template <int R> __device__ inline typename RowVec<R>::type rv_load(unsigned long long addr) {
  return *reinterpret_cast<const typename RowVec<R>::type __attribute__((address_space(1)))*>(addr);
}
template <int R> __device__ inline typename RowVec<R>::type rv_load(const double* p) { return rv_load<R>(reinterpret_cast<unsigned long long>(p)); }
template <int R> __device__ inline void rv_store(double* p, const typename RowVec<R>::type& v) {
  *reinterpret_cast<typename RowVec<R>::type __attribute__((address_space(1)))*>(reinterpret_cast<unsigned long long>(p)) = v;
}

// buffer loads (raw, offen): out-of-range offsets read as zero -- the "outside the run" case of the A operand costs no select
// of a 64-bit address and no load from a page of zeros
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr unsigned TILE_OOB = 0xffffff00u;   // (beyond every buffer: abytes < 4 GB - 4 KB)
template <int R> __device__ inline typename RowVec<R>::type rv_buffer_load(__amdgpu_buffer_rsrc_t rsrc, unsigned off);
template <> __device__ inline double rv_buffer_load<1>(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rsrc, off, 0, 0));
}
template <> __device__ inline v2d rv_buffer_load<2>(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
}
template <> __device__ inline v4d rv_buffer_load<4>(__amdgpu_buffer_rsrc_t rsrc, unsigned off) {
  const v2d lo = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
  const v2d hi = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + 16u, 0, 0));
  return v4d{lo[0], lo[1], hi[0], hi[1]};
}

// run of column k as the tile loop wants it: rz = address of (hypothetical) row 0, valid rows first .. last
struct alignas(16) TileRec {
  unsigned long long rz;
  int32_t first;
  uint32_t span;     // last - first; an empty run: first = INT_MAX, span = 0 (no row passes (unsigned)(r - first) <= span)
};
#ifndef NTP_TILE_PF
#define NTP_TILE_PF 6
#endif
constexpr int TILE_PF = NTP_TILE_PF;           // run loads (k groups) in flight per wave; a multiple of 3
constexpr int TILE_RPAD = 4;   // one group of empty records behind the last (the pipeline's look-ahead is clamped to it; OFF32: see the loop)
// The multiplier tile in LDS: row pitch 17 with one spare row in front and one behind: the tile is read as pairs of rows, a wave
// then writes consecutive rows of ONE column (at pitch 16 they would all fall on two banks), and the pairs start at an even row.
// (Label-aware instantiations kept pitch 16 and the element-wise path while three workgroups of four waves were preferred to two
// of eight -- with the labels of the window's rows in LDS as well the wider tile cost them the third workgroup.)
__host__ __device__ constexpr int tile_bp(bool) { return 17; }
__host__ __device__ constexpr int tile_bpad(bool) { return 2; }
// an element of X whose fate depends on the last kept row of the product column (decided when the block is done)
struct alignas(16) TileDefer {
  int32_t r, jj, prow, pad;
  double o, d;
};
constexpr int TILE_DEFER = 48;       // deferred elements per block (more: the step is refused)


struct TileArgs {
  const SlabRun* runs;
  const double* bblk;
  const int64_t* blk_boff;
  const int32_t *blk_kmin, *blk_kn, *blk_lo, *blk_w;
  const int64_t* blk_toff;
  double* out_val;
  int32_t* count;
  int32_t *ofirst, *olast;
  int64_t* ooff;
  int64_t* otoff;
  double alpha, threshold;
  int dense_rule, ncols, nblocks;
  int nrows;            // rows of the operands (labels are per row: panels have more rows than columns)
  int k4max, tmax;
  SlabFuseArgs fzv;     // EPI != 0: the fused epilogue's arguments, by value (kernel arguments: scalar loads, no upload before the launch)
  // EPI 0 only, optional: the right operand as the runs of its columns (slab algebra: no multiplier tiles were built) --
  // first / last row, offset of the first row and values; bblk / blk_boff are then unused
  const int32_t *brun_first, *brun_last;
  const int64_t* brun_off;
  const double* brun_val;
  int bpair;            // (with bbytes) the runs sit in zero-padded slots aligned to an even number of rows: the tile is read as pairs of rows, a wave per group of columns
  uint32_t bbytes;      // > 0: the runs brun_* lie in [brun_val, brun_val + bbytes), below 4 GB - 8 KB: the multiplier tile is read through a buffer resource
  const double* zero;   // 16 bytes of zeros: where the lanes outside a run load from
  // OFF32 instantiations: every run of A lies in [abase, abase + abytes), abytes < 4 GB -- the runs are read through a buffer
  // resource with 32-bit offsets, lanes outside a run get an offset beyond the buffer (the bounds check returns 0.0)
  const void* abase;
  uint32_t abytes;
  // OFF32 instantiations with a fused epilogue: X lies in the same allocation as the runs of A (it IS the left operand of a
  // TRS2 step), the expanded D operand in [dbase, dbase + dbytes) -- the epilogue's reads go through buffer resources too
  const void* dbase;
  uint32_t dbytes;
  int ablate;           // experiment build (-DNTP_ABLATIONS) only
#ifdef NTP_TILE_STAMPS
  long long* stamps;    // diagnostic build: [block][wave][64] s_memtime stamps
  long long* blkdur;    // diagnostic build: [block][4] = start, end (s_memtime), deferred elements, k range
#endif
};

// (lab_rows: label-aware instantiations keep the caller's labels of the window's rows in LDS)
__host__ __device__ inline size_t tile_lds_bytes(int k4max, int tmax, int lab_rows = 0) {
  return (size_t)lab_rows * 4 + (size_t)(k4max + tile_bpad(lab_rows > 0)) * tile_bp(lab_rows > 0) * 8 + (size_t)(k4max + TILE_RPAD) * sizeof(TileRec) + (size_t)tmax * 4 + (size_t)(k4max / 4 + 1) * 8 + 16 + TILE_DEFER * sizeof(TileDefer) + 16 * 5 * 4 +
         2 * 8 * 8 + (size_t)tmax * 16 + 8 + (size_t)(k4max + 8) * 4 + 64;
}

#ifdef NTP_TILE_STAMPS
#define STAMP(i) do { if (lane == 0 && b % 97 == 0 && b / 97 < 64 && (i) < 64) a.stamps[((b / 97) * 8 + wave) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

template <int EPI, int TILE_NW, int R, bool LAB, bool OFF32>
#ifdef NTP_TILE_WPE3
__global__ __launch_bounds__(TILE_NW* WAVE) __attribute__((amdgpu_waves_per_eu((((EPI == 0 || !LAB) && R <= 2) && TILE_NW != 4) ? 4 : 3, 8))) void k_spgemm_tile(const TileArgs a) {
#else
__global__ __launch_bounds__(TILE_NW* WAVE) __attribute__((amdgpu_waves_per_eu(R <= 2 ? 4 : 3, 8))) void k_spgemm_tile(const TileArgs a) {
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool LABK = LAB && EPI != 0;   // (what the launcher sizes the LDS by: labels of the window's rows)
  constexpr int BP = tile_bp(LABK), BPAD = tile_bpad(LABK);
  const int b = xcd_block(a.nblocks);
  if (b < 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  const int lo = a.blk_lo[b], w = a.blk_w[b], kmin = a.blk_kmin[b], kn = a.blk_kn[b];
  const int64_t tbase = a.blk_toff[b];
  // (the multiplier tile comes from the runs of the block's columns: their extents depend on the block's number only and are
  // requested HERE, together with the plan's scalars -- a memory round trip less in front of the tile's values)
  const bool brun = a.brun_val != nullptr;
  int bf0 = INT_MAX, bl0 = -1, bf1 = INT_MAX, bl1 = -1;
  const double *bp0 = nullptr, *bp1 = nullptr;
  unsigned bo0 = 0u, bo1 = 0u;   // (bbytes: byte offset of the hypothetical row 0 of the column's run in the buffer, modulo 2^32)
  auto thread_extents = [&]() {   // (a thread's column pair: the paths that read the tile element by element)
    const int c0 = b * SLAB_J + 2 * (tid & 7);
    if (c0 < a.ncols) {
      bf0 = a.brun_first[c0]; bl0 = a.brun_last[c0];
      const int64_t o = a.brun_off[c0];
      bp0 = a.brun_val + (o - bf0);
      bo0 = ((unsigned)o - (unsigned)bf0) * 8u;
    }
    if (c0 + 1 < a.ncols) {
      bf1 = a.brun_first[c0 + 1]; bl1 = a.brun_last[c0 + 1];
      const int64_t o = a.brun_off[c0 + 1];
      bp1 = a.brun_val + (o - bf1);
      bo1 = ((unsigned)o - (unsigned)bf1) * 8u;
    }
  };
  // (pairs of rows: a wave reads CPW whole columns -- their extents are wave-uniform: scalar loads)
  constexpr bool PAIR_OK = (TILE_NW == 4 || TILE_NW == 8) && BPAD == 2;
  constexpr int CPW = PAIR_OK ? SLAB_J / TILE_NW : 1;
  [[maybe_unused]] int cbf[CPW], cbl[CPW];
  [[maybe_unused]] unsigned cbo[CPW];
  const bool bpair = PAIR_OK && brun && a.bbytes != 0u && a.bpair != 0;
  if (bpair) {
#pragma unroll
    for (int m = 0; m < CPW; ++m) {
      const int c = b * SLAB_J + CPW * wave + m;
      cbf[m] = INT_MAX; cbl[m] = -1; cbo[m] = 0u;
      if (c < a.ncols) {
        cbf[m] = a.brun_first[c]; cbl[m] = a.brun_last[c];
        cbo[m] = ((unsigned)a.brun_off[c] - (unsigned)cbf[m]) * 8u;
      }
    }
  } else if (brun) thread_extents();   // (EPI 0: the slab algebra's right operand; EPI 1 / 2: the iterate itself -- its runs are the kernel's left operand already)
  // (what the fused epilogue needs of this lane's column -- extents and offsets of D and X -- depends on the block's number only:
  // requested here with the plan's scalars, used behind the barrier.  Loaded there, the test "does X fit the window" made every
  // wave wait for a memory round trip between the barrier and its first tile)
  // In TWO requests per wave (six before: a request costs a block in its prologue ~400 cycles whatever it carries): lane 16 s + jj
  // asks for value s of column jj -- dmin, dmax, xmin, xmax in one request, doff / xoff in the other -- and the lanes of a column
  // exchange them behind the barrier (ds_bpermute).
  [[maybe_unused]] int e_raw = 0, e_xpl = -1;
  [[maybe_unused]] int64_t e_raw8 = 0;
  if constexpr (EPI != 0) {
    const int jc0 = min(b * SLAB_J + (tid & 15), a.ncols - 1);
    const int sel = (tid >> 4) & 3;
    if constexpr (EPI == 2) {
      const int32_t* p4 = sel == 0 ? a.fzv.dmin : sel == 1 ? a.fzv.dmax : sel == 2 ? a.fzv.xmin : a.fzv.xmax;
      const int64_t* p8 = (sel & 1) ? a.fzv.xoff : a.fzv.doff;
      e_raw = p4[jc0];
      e_raw8 = p8[jc0];
      if constexpr (LAB) e_xpl = a.fzv.xplast[jc0];
    } else {
      const int32_t* p4 = (sel & 1) ? a.fzv.dmax : a.fzv.dmin;
      e_raw = p4[jc0];
      e_raw8 = a.fzv.doff[jc0];
    }
  }
  if (b == 0 && tid == 0) {   // (the end markers of the result's offset arrays)
    a.ooff[a.ncols] = a.blk_toff[a.nblocks];
    if (EPI != 0 && a.otoff) a.otoff[a.nblocks] = a.blk_toff[a.nblocks];
  }
  if (kn == 0) {   // no product entries in these columns
    const int j = b * SLAB_J + tid;
    if (tid < SLAB_J && j < a.ncols) {
      a.ofirst[j] = INT_MAX;
      a.olast[j] = -1;
      a.count[j] = 0;
      a.ooff[j] = tbase + (int64_t)tid * w;
      if constexpr (EPI != 0) {
        if (a.fzv.oplast) a.fzv.oplast[j] = -1;
        // (EPI 2: ... and none in the result only if the columns of X are empty as well; otherwise the step is not ours)
        if (EPI == 2 && a.fzv.xmax[j] >= a.fzv.xmin[j]) atomicOr(a.fzv.flag, 1);
      }
    }
    if (EPI != 0 && tid == 0) a.otoff[b] = tbase;
    return;
  }
  STAMP(0);
#ifdef NTP_TILE_STAMPS
  if (tid == 0 && a.blkdur) a.blkdur[4 * b] = __builtin_amdgcn_s_memtime();
#endif
  // ---- LDS
  double* Bs = reinterpret_cast<double*>(smem) + (BPAD / 2) * BP;                  // [-1 .. k4max][BP] (BPAD 2): row k - kmin, column j at Bs[(k - kmin) BP + j]
  TileRec* recs = reinterpret_cast<TileRec*>(reinterpret_cast<double*>(smem) + (size_t)(a.k4max + BPAD) * BP);   // [k4max + TILE_RPAD]
  int* grmin = reinterpret_cast<int*>(recs + a.k4max + TILE_RPAD);                 // [k4max / 4 + 1]: first / last row any
  int* grmax = grmin + (a.k4max / 4 + 1);                                          // column of a k group reaches
  unsigned* colmask = reinterpret_cast<unsigned*>(grmax + (a.k4max / 4 + 1));      // [tmax]
  TileDefer* dlist = reinterpret_cast<TileDefer*>(colmask + ((a.tmax + 2 * (a.k4max / 4 + 1) + 3) & ~3) - 2 * (a.k4max / 4 + 1));
  int* col_cnt = reinterpret_cast<int*>(dlist + TILE_DEFER);                       // [16] each
  int* col_first = col_cnt + 16;
  int* col_last = col_first + 16;
  int* col_pmax = col_last + 16;
  int* col_plast = col_pmax + 16;
  double* red = reinterpret_cast<double*>(col_plast + 16);                         // [2 * 8]
  double* tred = red + 2 * 8;                                                // [2 * tmax]: every tile's share of the two sums (EPI != 0)
  int* misc = reinterpret_cast<int*>(tred + 2 * a.tmax);                     // [0] deferred, [1] product entries, [2..3] products, [4] tiles taken
  [[maybe_unused]] int* cntl = misc + 6 + 1;                                  // [-1 .. k4max + 2]: entries of column kmin + i of the left operand (pair path: the product count)
  [[maybe_unused]] int* labs = misc + 6 + (a.k4max + 8);                                     // LAB: [tmax * 16 R] the caller's label of every row of the window

  const int KG = (kn + 3) >> 2, K4 = KG * 4;
  constexpr int TROWS = 16 * R;   // rows of a tile: R matrix instructions per k group (the window is a multiple, the host sees to it)
  typedef typename RowVec<R>::type VR;
  const int T = (w + TROWS - 1) / TROWS;
  // ---- block prologue: multiplier tile -> LDS (rows kn .. K4 zero; requested first, stored last: the loads are in
  // flight while the records are built), run records + row range of every k group (one thread per group), then the k
  // groups that can reach a tile are found by the wave that takes the tile (a ballot over the groups' row ranges)
  if (kn > 0 && w > 0 && tbase >= 0) STAMP(56);
  constexpr int NT = TILE_NW * WAVE, BCH = 3072 / NT;   // (in flight together: the tile of a k range of 384, 48 KB)
  const double2* __restrict__ bsrc = reinterpret_cast<const double2*>(a.bblk + (a.blk_boff ? a.blk_boff[b] : 0));
  double2 btmp[BCH];
  // (the multiplier tile from the runs of the block's columns: a thread always serves the same column pair -- NT is a
  // multiple of 8 -- and consecutive rows of a column go to threads 8 apart)
  auto brun_load = [&](int i) {
    const int r = kmin + (i >> 3);
    double2 v;
#ifdef NTP_ABL_NOBLOAD   // (ablation, wrong results: what the multiplier tile's loads cost)
    v.x = (i < kn * 8 && r >= bf0 && r <= bl0) ? 1.0 : 0.0;
    v.y = (i < kn * 8 && r >= bf1 && r <= bl1) ? 1.0 : 0.0;
    return v;
#endif
    v.x = (i < kn * 8 && r >= bf0 && r <= bl0) ? bp0[r] : 0.0;
    v.y = (i < kn * 8 && r >= bf1 && r <= bl1) ? bp1[r] : 0.0;
    return v;
  };
  // (the block's product count = sum over the tile's non-zeros B(k, j) of the entries of A(:, k): the entry counts of the rows a
  // thread holds are requested WITH the tile's values and multiplied while the values sit in registers -- the count used to
  // be a pass of its own over the LDS tile behind the barrier, 7 % of the kernel for a statistic, profiles/README.md round 6)
  [[maybe_unused]] int bcnt[BCH];
  [[maybe_unused]] const int32_t* __restrict__ prod_count = nullptr;
  [[maybe_unused]] bool want_prod = false;
  if constexpr (EPI != 0) {
    want_prod = a.fzv.prod != nullptr;
    prod_count = a.fzv.in_count;
  }
#ifndef NTP_TILE_NO_RECFIRST
  // (the run records requested BEFORE the tile's values: loads return in order, so records requested behind the tile's 36
  // loads per thread could not be worked on before the last of those had arrived)
  constexpr int RCH = (384 + 4 + NT - 1) / NT;
  uint4 rh0[RCH], rh1[RCH];
  {
    const uint4* __restrict__ rp_h = reinterpret_cast<const uint4*>(a.runs + kmin);
#pragma unroll
    for (int u = 0; u < RCH; ++u) {
      const int ic = min(tid + u * NT, kn - 1);
      rh0[u] = rp_h[2 * ic];
      rh1[u] = rp_h[2 * ic + 1];
    }
  }
  STAMP(50);
#ifdef NTP_TILE_STAMPS
  if (brun && bf0 + bl0 + bf1 + bl1 == 0x7ffffff1) STAMP(59);   // (forces the wait for the column extents between stamps 50 and 52)
  STAMP(52);
#endif
#endif
  const int bpath = (brun && a.bbytes != 0u) ? ((bpair && K4 + 2 <= 384) ? 2 : 1) : 0;
  if (bpath != 2 && bpair) thread_extents();   // (a k range beyond three requests of 128 rows: element by element after all)
  // (pair path: the entry counts behind the block's product count, ONE request per 64 rows and block -- a thread per row, to LDS
  // with the records; every wave reads the rows of its pairs behind the barrier.  Requested per wave they were as many requests
  // as the tile itself)
  constexpr int CCH = (384 + NT - 1) / NT;
  [[maybe_unused]] int ctmp[CCH];
  if constexpr (EPI != 0) {
    if (bpath == 2 && want_prod) {
      const int32_t* __restrict__ pc = prod_count;
      const int rmax = kmin + kn - 1;
#pragma unroll
      for (int u = 0; u < CCH; ++u) ctmp[u] = pc ? pc[min(kmin + tid + u * NT, rmax)] : 1;
    }
  }
  const int ke = kmin & ~1, pair_r = ke + 2 * lane;   // pairs: request u of a column holds rows pair_r + 128 u, + 1
  if (bpath == 2) {
    // Pairs of rows, a wave per CPW columns: 64 lanes x 16 bytes = 128 consecutive rows of ONE column per request -- twelve
    // requests per wave for the tile and six for the counts, where the element-wise paths issue 24 + 12 per THREAD (a CU's
    // request pipeline is what a block's prologue waits for, profiles/README.md round 6).  The slots are zero-padded to even
    // rows on both sides (bpair), so a pair that straddles an end of its run reads a stored zero.
    if constexpr (PAIR_OK) {
      const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.brun_val), 0, a.bbytes, 0x00020000);
      constexpr unsigned BOOB = 0xffffe000u;
#pragma unroll
      for (int m = 0; m < CPW; ++m) {
        const bool cv = cbl[m] >= cbf[m];
        const unsigned tb = cv ? (unsigned)(pair_r + 1 - cbf[m]) : 0x40000000u, sp = cv ? (unsigned)(cbl[m] - cbf[m] + 1) : 0u;
        const unsigned ob = cbo[m] + (unsigned)pair_r * 8u;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          const v2d v = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (tb + 128u * u <= sp ? ob : BOOB) + 1024u * u, 0, 0));
          btmp[m * 3 + u].x = v[0];
          btmp[m * 3 + u].y = v[1];
        }
      }
    }
  } else if (bpath == 1) {
    // The tile through ONE buffer resource over the operand's values: a thread's two columns are 32-bit offsets, a row outside a
    // column's run gets an offset beyond the buffer and reads as 0.0 -- three vector instructions and the load per element,
    // no branch, no 64-bit address (the prologue is bound by the instructions it issues, profiles/README.md round 6)
    const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.brun_val), 0, a.bbytes, 0x00020000);
    constexpr unsigned BOOB = 0xffffe000u;            // (+ 256 (BCH - 1) stays below 2^32 and beyond bbytes)
    const int rr = kmin + (tid >> 3);                 // chunk u: row rr + (NT / 8) u
    const bool v0 = bl0 >= bf0, v1 = bl1 >= bf1;
    const unsigned t0 = v0 ? (unsigned)(rr - bf0) : 0x40000000u, s0 = v0 ? (unsigned)(bl0 - bf0) : 0u;
    const unsigned t1 = v1 ? (unsigned)(rr - bf1) : 0x40000000u, s1 = v1 ? (unsigned)(bl1 - bf1) : 0u;
    const unsigned o0 = bo0 + (unsigned)rr * 8u, o1 = bo1 + (unsigned)rr * 8u;
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const unsigned du = (unsigned)(u * (NT / 8));
      btmp[u].x = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(brsrc, (t0 + du <= s0 ? o0 : BOOB) + 8u * du, 0, 0));
      btmp[u].y = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(brsrc, (t1 + du <= s1 ? o1 : BOOB) + 8u * du, 0, 0));
    }
    if constexpr (EPI != 0) {
      if (want_prod) {   // (rows beyond the k range hold no value of the tile: any count will do there)
        const int32_t* __restrict__ pc = prod_count;
        const int rmax = kmin + kn - 1;
#pragma unroll
        for (int u = 0; u < BCH; ++u) bcnt[u] = pc ? pc[min(rr + u * (NT / 8), rmax)] : 1;
      } else {
#pragma unroll
        for (int u = 0; u < BCH; ++u) bcnt[u] = 0;
      }
    }
  } else {
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int i = tid + u * NT;
      if (brun) btmp[u] = brun_load(i);
      else btmp[u] = i < kn * 8 ? bsrc[i] : make_double2(0.0, 0.0);
#ifdef NTP_ABL_NOBLOAD
      if constexpr (EPI != 0) bcnt[u] = (want_prod && i < kn * 8) ? 1 : 0;
#else
      if constexpr (EPI != 0) bcnt[u] = (want_prod && i < kn * 8) ? (prod_count ? prod_count[kmin + (i >> 3)] : 1) : 0;
#endif
    }
  }
  STAMP(51);
  // (label-aware kernels: the caller's labels of the window's rows are requested with the tile and the records -- not in a
  // round trip of their own behind the barrier -- and stored with them)
  constexpr int LABCH = LAB ? 4 : 0;
  [[maybe_unused]] int labtmp[LABCH > 0 ? LABCH : 1];
  if constexpr (LAB) {
    const int32_t* __restrict__ lab_g = a.fzv.lab;
#pragma unroll
    for (int u = 0; u < LABCH; ++u) labtmp[u] = lab_g[min(lo + tid + u * NT, a.nrows - 1)];
  }
  {
    // one thread per record (all loads independent and in flight together with the tile's), the row range of a k
    // group = min / max over its four records: two quad-permute steps
    const uint4* __restrict__ rp = reinterpret_cast<const uint4*>(a.runs + kmin);
    auto build_rec = [&](int i, const uint4 r0, const uint4 r1) {   // r0 = (addr_lo, addr_hi, nbytes, flags), r1 = (first8, first, span62, pad)
      STAMP(57);
      if (r0.z + r1.y == 0x7fffffffu) STAMP(59);
      STAMP(58);
      const int rows = i < kn ? (int)(r0.z >> 3) : 0;
      const int first = (int)r1.y;
      TileRec rec;
      rec.rz = 0;
      rec.first = INT_MAX;
      rec.span = 0u;
      int rmin = INT_MAX, rmax = -1;
      if (rows > 0) {
        const unsigned long long addr = (unsigned long long)r0.x | ((unsigned long long)r0.y << 32);
        rec.rz = addr - (unsigned long long)((long long)first * 8);
        if constexpr (OFF32) rec.rz -= reinterpret_cast<unsigned long long>(a.abase);   // (the low word is the offset, modulo 2^32)
        rec.first = first - (R - 1);               // a lane's rows ra .. ra + R - 1 touch the run iff (unsigned)(ra - rec.first) <= rec.span
        rec.span = (uint32_t)(rows - 1 + (R - 1));
        rmin = first;
        rmax = first + rows - 1;
      }
      rmin = min(rmin, __builtin_amdgcn_mov_dpp(rmin, 0xb1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
      rmax = max(rmax, __builtin_amdgcn_mov_dpp(rmax, 0xb1, 0xf, 0xf, false));
      rmin = min(rmin, __builtin_amdgcn_mov_dpp(rmin, 0x4e, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
      rmax = max(rmax, __builtin_amdgcn_mov_dpp(rmax, 0x4e, 0xf, 0xf, false));
      if (i < K4 + 4) {
        recs[i] = rec;
        if ((i & 3) == 0) {
          grmin[i >> 2] = rmin;
          grmax[i >> 2] = rmax;
        }
      }
    };
#ifndef NTP_TILE_NO_RECFIRST
#pragma unroll
    for (int u = 0; u < RCH; ++u)
      if (u * NT < K4 + 4) build_rec(u * NT + tid, rh0[u], rh1[u]);
    for (int i0 = RCH * NT; i0 < K4 + 4; i0 += NT) {
#else
    for (int i0 = 0; i0 < K4 + 4; i0 += NT) {
#endif
      const int ic = min(i0 + tid, kn - 1);
      build_rec(i0 + tid, rp[2 * ic], rp[2 * ic + 1]);
    }
  }
  STAMP(60);
  [[maybe_unused]] long long prod_p = 0;
  [[maybe_unused]] unsigned nzm = 0u;   // pair path: bit 2 (3 m + u) + e: element e of the pair of request u of column m is not zero
  if (bpath == 2) {
    if constexpr (PAIR_OK) {
      if constexpr (EPI != 0) {
        if (want_prod) {
#pragma unroll
          for (int u = 0; u < CCH; ++u)
            if (tid + u * NT <= K4) cntl[tid + u * NT] = tid + u * NT < K4 ? ctmp[u] : 0;
          if (tid == 0) cntl[-1] = 0;
        }
      }
#pragma unroll
      for (int m = 0; m < CPW; ++m) {
        double* const bcol = Bs + (pair_r - kmin) * BP + CPW * wave + m;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
          if (pair_r - kmin + 128 * u < K4) {   // (rows -1 and K4: the spare rows)
            bcol[128 * u * BP] = btmp[m * 3 + u].x;
            bcol[(128 * u + 1) * BP] = btmp[m * 3 + u].y;
          }
          if constexpr (EPI != 0)
            nzm |= ((btmp[m * 3 + u].x != 0.0) ? 1u : 0u) << (2 * (m * 3 + u)) | ((btmp[m * 3 + u].y != 0.0) ? 2u : 0u) << (2 * (m * 3 + u));
        }
      }
    }
  }
  else {
#pragma unroll
    for (int u = 0; u < BCH; ++u) {
      const int i = tid + u * NT;
      if (i < K4 * 8) { Bs[(i >> 3) * BP + 2 * (i & 7)] = btmp[u].x; Bs[(i >> 3) * BP + 2 * (i & 7) + 1] = btmp[u].y; }
      if constexpr (EPI != 0) prod_p += (long long)(((btmp[u].x != 0.0) ? 1 : 0) + ((btmp[u].y != 0.0) ? 1 : 0)) * bcnt[u];
    }
  }
  for (int i = tid + BCH * NT; i < K4 * 8; i += NT) {
    const double2 v = brun ? brun_load(i) : (i < kn * 8 ? bsrc[i] : make_double2(0.0, 0.0));
    Bs[(i >> 3) * BP + 2 * (i & 7)] = v.x;
    Bs[(i >> 3) * BP + 2 * (i & 7) + 1] = v.y;
    if constexpr (EPI != 0) {
      if (want_prod && i < kn * 8)
        prod_p += (long long)(((v.x != 0.0) ? 1 : 0) + ((v.y != 0.0) ? 1 : 0)) * (prod_count ? prod_count[kmin + (i >> 3)] : 1);
    }
  }
  STAMP(61);
  if constexpr (LAB) {
#pragma unroll
    for (int u = 0; u < LABCH; ++u)
      if (tid + u * NT < T * TROWS) labs[tid + u * NT] = labtmp[u];
  }
  for (int t = tid; t < T; t += NT) {
    colmask[t] = 0u;
    if constexpr (EPI != 0) { tred[2 * t] = 0.0; tred[2 * t + 1] = 0.0; }
  }
  if (tid < 16) {
    col_cnt[tid] = 0;
    col_first[tid] = INT_MAX;
    col_last[tid] = -1;
    col_pmax[tid] = -1;
    col_plast[tid] = -1;
  }
  if (tid < 6) misc[tid] = 0;
  __syncthreads();
  STAMP(62);
  STAMP(63);
  if constexpr (EPI != 0) {
    if (want_prod) {
      if constexpr (PAIR_OK) {
        if (bpath == 2) {   // (a pair's rows beyond the k range hold zeros: their count is not read -- index K4 is a zero)
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int rho = pair_r - kmin + 128 * u;
            const int c0 = cntl[min(rho, K4)], c1 = cntl[min(rho + 1, K4)];
#pragma unroll
            for (int m = 0; m < CPW; ++m)
              prod_p += (long long)(((nzm >> (2 * (m * 3 + u))) & 1u) ? c0 : 0) + (long long)(((nzm >> (2 * (m * 3 + u) + 1)) & 1u) ? c1 : 0);
          }
        }
      }
      prod_p = wave_sum_i64(prod_p);
      if (lane == 0 && prod_p) atomicAdd(reinterpret_cast<unsigned long long*>(misc + 2), (unsigned long long)prod_p);
    }
  }
  if constexpr (LAB) {   // (the labels beyond the first LABCH per thread, requested above)
    const int32_t* __restrict__ lab_g = a.fzv.lab;
    for (int i = tid + LABCH * NT; i < T * TROWS; i += NT) labs[i] = lab_g[min(lo + i, a.nrows - 1)];
    __syncthreads();
  }
  STAMP(1);

  // ---- per-lane constants: this lane's column is jj = lane % 16 in every tile
  const int jj = lane & 15, q = lane >> 4;
  const int j = b * SLAB_J + jj;
  const bool colv = j < a.ncols;
  const int jc = min(j, a.ncols - 1);
  const double* const zp = a.zero;
  const unsigned long long zaddr = reinterpret_cast<unsigned long long>(zp);
  double* const orun = a.out_val + (tbase + (int64_t)jj * w - lo);          // orun[r] = slot of row r of column j
  [[maybe_unused]] double* otile = nullptr;
  [[maybe_unused]] int xf = INT_MAX, xlrow = -1, xpl = -1, df = INT_MAX, dl = -1;
  [[maybe_unused]] const double *xrz = zp, *drz = zp;
  [[maybe_unused]] double am = 0, bm = 0, thr_m = 0;
  [[maybe_unused]] int diag = -1;
  if constexpr (EPI != 0) {
    // (fzv.tiles == nullptr: the result carries runs only -- the next step builds its multiplier tile from them)
    if (a.fzv.tiles) otile = a.fzv.tiles + (tbase - (int64_t)lo * SLAB_J + jj);   // otile[r * 16] = row r, column jj of the tile
    // (the values requested at the kernel's top, lane 16 s + jj: see there)
    const int pj = 4 * jj;
    auto pull64 = [&](int from, int64_t v) {
      const int lo32 = __builtin_amdgcn_ds_bpermute(from, (int)(unsigned)(unsigned long long)v);
      const int hi32 = __builtin_amdgcn_ds_bpermute(from, (int)(unsigned)((unsigned long long)v >> 32));
      return (int64_t)(((unsigned long long)(unsigned)hi32 << 32) | (unsigned long long)(unsigned)lo32);
    };
    const int e_d0 = __builtin_amdgcn_ds_bpermute(pj, e_raw), e_d1 = __builtin_amdgcn_ds_bpermute(pj + 64, e_raw);
    const int64_t e_doff = EPI == 2 ? pull64(pj, e_raw8) : e_raw8;
    const int d0 = e_d0, d1 = e_d1;
    if (colv && d1 >= d0) {
      df = d0;
      dl = d1;
      drz = a.fzv.dexp + (e_doff - d0);
    }
    diag = j + a.fzv.col_offset;
    if constexpr (EPI == 2) {
      am = a.fzv.am; bm = a.fzv.bm; thr_m = a.fzv.thr_m;
      const int e_x0 = __builtin_amdgcn_ds_bpermute(pj + 128, e_raw), e_x1 = __builtin_amdgcn_ds_bpermute(pj + 192, e_raw);
      const int64_t e_xoff = pull64(pj + 64, e_raw8);
      const int x0 = e_x0, x1 = e_x1;
      if (colv && x1 >= x0) {
        xf = x0;
        xlrow = x1;
        xrz = a.fzv.xexp + (e_xoff - x0);
        xpl = LAB ? e_xpl : x1;
        if (x0 < lo || x1 >= lo + w) atomicOr(a.fzv.flag, 1);   // every stored row of X(:, j) must be a row of this block's window
      }
    }
  }
  // (OFF32: the same columns as offsets into the two buffers; a lane's rows rb .. rb + R - 1 touch the run iff
  // (unsigned)(rb - first') <= span', as for the runs of A)
  [[maybe_unused]] unsigned xo0 = 0, xsp = 0, do0 = 0, dsp = 0;
  [[maybe_unused]] int xf1 = INT_MAX, df1 = INT_MAX;
  [[maybe_unused]] __amdgpu_buffer_rsrc_t xrsrc, drsrc;
  if constexpr (OFF32 && EPI != 0) {
    drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dbase), 0, a.dbytes, 0x00020000);
    if (dl >= df) {
      df1 = df - (R - 1);
      dsp = (unsigned)(dl - df + (R - 1));
      do0 = (unsigned)(reinterpret_cast<unsigned long long>(drz) - reinterpret_cast<unsigned long long>(a.dbase));
    }
    if constexpr (EPI == 2) {
      xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.abase), 0, a.abytes, 0x00020000);
      if (xlrow >= xf) {
        xf1 = xf - (R - 1);
        xsp = (unsigned)(xlrow - xf + (R - 1));
        xo0 = (unsigned)(reinterpret_cast<unsigned long long>(xrz) - reinterpret_cast<unsigned long long>(a.abase));
      }
    }
  }
  const double alpha = a.alpha, thr = a.threshold;
  const bool dense_rule = (a.dense_rule & 1) != 0;
  double dsum = 0.0, tsum = 0.0;
  int pn = 0;
  const int rend = lo + w;
  const int mid = (T - 1) >> 1;

  [[maybe_unused]] int sidx = 4;
#ifdef NTP_TILE_STATIC
  for (int ti = 0;; ++ti) {   // (snake order over the waves: every wave gets tiles from both ends of each round)
    const int ts = ti * TILE_NW + ((ti & 1) ? TILE_NW - 1 - wave : wave);
    if (ti * TILE_NW >= T) break;
    if (ts >= T) continue;
#else
  // A wave takes the next tile when it is through with its last (a counter in LDS): the tiles of a window cost between a few
  // and ~40 slots of the loop, and with a fixed deal the waves reached the block's closing barrier 6 k cycles apart on average.
  // The two sums stay reproducible: every tile leaves its share in LDS, the block adds them in tile order.
  for (;;) {
    int ts = 0;
    if (lane == 0) ts = atomicAdd(&misc[4], 1);
    ts = __builtin_amdgcn_readfirstlane(ts);
    if (ts >= T) break;
    if constexpr (EPI != 0) { dsum = 0.0; tsum = 0.0; }
#endif
    STAMP(sidx); ++sidx;   // centre first: the tiles in the middle of the window have the longest k ranges
    const int t = (ts & 1) ? mid + ((ts + 1) >> 1) : mid - (ts >> 1);
    const int r0 = lo + TROWS * t;
    int g0 = INT_MAX, g1 = -1;   // the k groups that can reach the tile: a ballot over the groups' row ranges
    for (int c = 0; c < KG; c += WAVE) {
      const int gq = min(c + lane, KG);
      const unsigned long long m = __ballot(grmin[gq] <= r0 + TROWS - 1 && grmax[gq] >= r0);   // (group KG: empty, never true)
      if (m) {
        if (g0 == INT_MAX) g0 = c + (int)__builtin_ctzll(m);
        g1 = c + 63 - (int)__builtin_clzll(m);
      }
    }
    // what the epilogue reads -- this lane's elements are rows r0 + R (4 v + q) + m of column jj -- is requested when
    // the main loop is through (before its last PF - 1 groups and the group-range bookkeeping of the epilogue): held
    // across the whole loop these values cost a wave of occupancy
    [[maybe_unused]] VR xv[4], dv[4];
    auto epilogue_loads = [&]() {
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int rb = r0 + R * (4 * v + q);
        if constexpr (OFF32) {
          if constexpr (EPI == 2) xv[v] = rv_buffer_load<R>(xrsrc, (unsigned)(rb - xf1) <= xsp ? xo0 + (unsigned)rb * 8u : TILE_OOB);
          if constexpr (EPI != 0) dv[v] = rv_buffer_load<R>(drsrc, (unsigned)(rb - df1) <= dsp ? do0 + (unsigned)rb * 8u : TILE_OOB);
        } else {
          if constexpr (EPI == 2) xv[v] = rv_load<R>(((rb + R - 1 >= xf) & (rb <= xlrow)) ? xrz + rb : zp);
          if constexpr (EPI != 0) {
            dv[v] = rv_load<R>(((rb + R - 1 >= df) & (rb <= dl)) ? drz + rb : zp);
          }
        }
      }
    };
    v4d acc[R];
#pragma unroll
    for (int m = 0; m < R; ++m) acc[m] = v4d{0.0, 0.0, 0.0, 0.0};
    if (g1 >= g0) {
      // Software pipeline over the k groups g0 .. g1 (records and multiplier rows are padded behind the last group, and
      // a group beyond g1 has no row in this tile, so nothing below needs a bound): the record of group g + PF + 1
      // is read from LDS while the run load of group g + PF is issued from the record read one step earlier, the
      // multiplier row of g + 1 is read, and group g -- operands landed PF steps / one step ago -- is multiplied.
      const int rl = r0 + R * jj;                  // A operand: rows rl .. rl + R - 1, column 4 g + q
      [[maybe_unused]] const unsigned long long r8 = (unsigned long long)((long long)rl * 8);
      const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(recs) + q;     // record of group g: rq[4 g]
      const double* __restrict__ bq = Bs + q * BP + jj;                            // multiplier of group g: bq[4 BP g]
#ifdef NTP_TILE_ABL_NOLOAD
      const unsigned long long abl_base = reinterpret_cast<unsigned long long>(a.out_val) & ~0x3fffull;   // (any mapped memory)
#endif
#ifdef NTP_ABLATIONS
      const unsigned long long abl_base = reinterpret_cast<unsigned long long>(a.out_val) & ~0x3fffull;   // (any mapped memory)
#endif
      [[maybe_unused]] const unsigned r8lo = (unsigned)rl * 8u;
      [[maybe_unused]] __amdgpu_buffer_rsrc_t arsrc;
      if constexpr (OFF32) arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.abase), 0, a.abytes, 0x00020000);
      auto run_load = [&](const uint4 raw) -> VR {
        const bool ok = (unsigned)(rl - (int)raw.z) <= raw.w;
        if constexpr (OFF32) {
          // (four vector instructions per load: row - first, compare, offset, select of the out-of-range offset)
          return rv_buffer_load<R>(arsrc, ok ? raw.x + r8lo : TILE_OOB);
        } else {
          const unsigned long long rz = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
#ifdef NTP_ABLATIONS
          // (experiment build, WRONG results: option spgemm_variant 601 makes every run load hit a 16 KB window -- what the
          // loop costs when the operand comes from L1)
          if (a.ablate == 1) return rv_load<R>(ok ? abl_base + ((rz + r8 - abl_base) & 0x3fe0ull) : zaddr);
#endif
          return rv_load<R>(ok ? rz + r8 : zaddr);
        }
      };
      // slot u of the ring holds the A operand of group g + u; it is refilled (group g + u + PF) right after the matrix
      // instruction that read it has been issued, so no value is ever copied from one register to another.  PF run
      // loads per wave stay in flight: they return in order, so one HBM miss holds back everything behind it and the
      // depth has to cover a miss, not a hit.  The multiplier rows come from LDS two slots ahead (three registers).
      // The look-ahead reads records up to PF + 1 groups and multiplier rows up to two groups beyond g1; what it finds there
      // is never multiplied (a ring slot is consumed only for groups <= g1).  With 64-bit addresses every such index is
      // clamped to the empty group KG behind the last record (a record must never be garbage: it becomes an address).
      // OFF32: NOT clamped -- behind the records lie other arrays of this block's LDS, and a "record" read from them can only
      // produce a buffer offset, whose load is bounds-checked; inside the unrolled body every LDS address is then the body's
      // base plus a constant (one address computation per six groups instead of two per group).
      auto gi = [&](int x) { return OFF32 ? x : min(x, KG); };
      auto bi = [&](int x) { return OFF32 ? x : min(x, KG - 1); };
      VR ring[TILE_PF];
      double bb[3];
#pragma unroll
      for (int u = 0; u < TILE_PF; ++u) ring[u] = run_load(rq[4 * gi(g0 + u)]);
      bb[0] = bq[4 * BP * g0];
      bb[1] = bq[4 * BP * bi(g0 + 1)];
      bb[2] = 0.0;
      uint4 raw = rq[4 * gi(g0 + TILE_PF)];
      int g = g0;
      STAMP(sidx);
      for (; g + TILE_PF - 1 <= g1; g += TILE_PF) {
#pragma unroll
        for (int u = 0; u < TILE_PF; ++u) {
          // (the order is pinned: record read one slot ahead | matrix instruction | refill of the slot it has read)
          const uint4 raw_n = rq[4 * gi(g + u + TILE_PF + 1)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < R; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(rv_get<R>(ring[u], m), bb[u % 3], acc[m], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          ring[u] = run_load(raw);
          bb[(u + 2) % 3] = bq[4 * BP * bi(g + u + 2)];
          raw = raw_n;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      epilogue_loads();
#pragma unroll
      for (int u = 0; u < TILE_PF - 1; ++u) {
        if (g + u <= g1) {
          const double bt = bq[4 * BP * (g + u)];
#pragma unroll
          for (int m = 0; m < R; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(rv_get<R>(ring[u], m), bt, acc[m], 0, 0, 0);
        }
      }
    }
    else epilogue_loads();   // (no k group reaches the tile: entries of X alone)
    ++sidx; STAMP(sidx); ++sidx;
#ifdef NTP_ABLATIONS
    if (a.ablate == 2) {   // (experiment build, WRONG results: no epilogue at all)
      if (acc[0][0] == 1.2345e300) colmask[t] = 1u;
      continue;
    }
#endif
    // ---- epilogue of the tile: lane holds rows r0 + R (4 v + q) + m (v = 0..3, m = 0..R-1) of column jj
    {  // tiles in which nothing can be kept (about half of a window: the products beyond the band that survives the
       // threshold, no entry of X) are done here: nothing to merge, count or store
      bool live = false;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const double vv = acc[m][v];
          live |= dense_rule ? (fabs(vv) > thr) : (fabs(__dmul_rn(alpha, vv)) > thr);
          if constexpr (EPI == 2) live |= rv_get<R>(xv[v], m) != 0.0;
        }
      }
      if (__ballot(live) == 0ull) {
        if (lane == 0) colmask[t] = 0u;
        STAMP(sidx); ++sidx;
        continue;
      }
    }
#ifndef NTP_TILE_NO_FULL_TILES
    if constexpr (EPI == 2 && !LAB) {
      // FULL tiles -- every element has a product entry above the threshold AND an entry of X AND a sum above the update's
      // threshold (the interior of the band: most of the live tiles of a purification step) -- keep everything: no
      // AddSparseVectors case analysis, no selects, no deferred elements; the values, the order of the sums and the column
      // statistics are those of the general path below (adding +0.0 to the trace sum, which it does for every off-diagonal
      // element, changes no bit)
      bool full = true;
      VR both[4];
#pragma unroll
      for (int v = 0; v < 4; ++v) {
#pragma unroll
        for (int m = 0; m < R; ++m) {
          const double vv = acc[m][v];
          const double sv = __dmul_rn(alpha, vv);
          const double bv = rv_get<R>(xv[v], m);
          const double bo = __dadd_rn(__dmul_rn(am, sv), __dmul_rn(bm, bv));
          full &= (dense_rule ? (fabs(vv) > thr) : (fabs(sv) > thr)) & (bv != 0.0) & (fabs(bo) > thr_m);
          rv_set<R>(both[v], m, bo);
        }
      }
      if (__ballot(!full) == 0ull) {
        pn += 64 * 4 * R;
        const bool has_diag = r0 <= b * SLAB_J + SLAB_J - 1 + a.fzv.col_offset && r0 + TROWS - 1 >= b * SLAB_J + a.fzv.col_offset;   // (wave-uniform)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int m = 0; m < R; ++m) {
            const double o = rv_get<R>(both[v], m);
            dsum = __dadd_rn(dsum, __dmul_rn(o, rv_get<R>(dv[v], m)));
            if (has_diag) tsum = __dadd_rn(tsum, (r0 + R * (4 * v + q) + m == diag) ? o : 0.0);
          }
        }
        if (q == 0) {   // (one lane per column leaves the column's statistics: 16 R rows kept, first r0, last r0 + 16 R - 1)
          atomicAdd(&col_cnt[jj], TROWS);
          atomicMin(&col_first[jj], r0);
          atomicMax(&col_last[jj], r0 + TROWS - 1);
          atomicMax(&col_pmax[jj], r0 + TROWS - 1);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) rv_store<R>(orun + (r0 + R * (4 * v + q)), both[v]);
        if (otile) {
#pragma unroll
          for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int m = 0; m < R; ++m) otile[(int64_t)(r0 + R * (4 * v + q) + m) * SLAB_J] = rv_get<R>(both[v], m);
          }
        }
        if (lane == 0) colmask[t] = 0xffffu;
#ifndef NTP_TILE_STATIC
        {
          const double td = wave_sum_f64(dsum), tt = wave_sum_f64(tsum);
          if (lane == 0) {
            tred[2 * t] = td;
            tred[2 * t + 1] = tt;
          }
        }
#endif
        STAMP(sidx); ++sidx;
        continue;
      }
    }
#endif
    VR res[4];
    unsigned long long anykeep = 0;
    int c_l = 0, f_l = INT_MAX, l_l = -1, pm_l = -1, pl_l = -1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
#pragma unroll
      for (int m = 0; m < R; ++m) {
        const int r = r0 + R * (4 * v + q) + m;
        const double vv = acc[m][v];
        const double sv = __dmul_rn(alpha, vv);
        const bool ha = dense_rule ? (fabs(vv) > thr) : (fabs(sv) > thr);
        bool keep;
        double o;
        [[maybe_unused]] double dval = 0.0;
        [[maybe_unused]] int pr = r;
        if constexpr (EPI != 0) {
          dval = rv_get<R>(dv[v], m);
          if constexpr (LAB) pr = labs[r - lo];   // ("beyond the other column's last entry" compares the caller's labels)
        }
        if constexpr (EPI != 2) {
          keep = ha;
          o = sv;
        } else {
          const double bv = rv_get<R>(xv[v], m);
          const bool hb = bv != 0.0;
          const double bs = __dmul_rn(bm, bv);
          const double wa = __dmul_rn(am, sv);
          const double both = __dadd_rn(wa, bs);
          o = ha ? (hb ? both : wa) : bs;                        // (neither: bs = 0)
          const bool big = fabs(o) > thr_m;
          // AddSparseVectors (inc_decide): both present -> threshold on the sum; one present -> threshold unless it lies
          // beyond the other column's last entry.  "Beyond the product column's last kept entry" is not known yet for an
          // element of X alone that fails the threshold: decided when the block is done (dlist)
          if (ha) {
            keep = (!hb && pr > xpl) || big;
          } else {
            keep = hb && big;
            if (hb && !big) {
              const int slot = atomicAdd(&misc[0], 1);
              if (slot < TILE_DEFER) {
                *reinterpret_cast<int4*>(&dlist[slot]) = make_int4(r, jj, pr, 0);
                dlist[slot].o = o;
                dlist[slot].d = dval;
              }
            }
          }
        }
        pn += (int)__popcll(__ballot(ha));
        anykeep |= __ballot(keep);
        if constexpr (EPI != 0) {
          dsum = __dadd_rn(dsum, __dmul_rn(keep ? o : 0.0, keep ? dval : 0.0));
          tsum = __dadd_rn(tsum, (keep && r == diag) ? o : 0.0);
          pm_l = max(pm_l, ha ? pr : -1);
          pl_l = max(pl_l, keep ? pr : -1);
        }
        c_l += keep ? 1 : 0;
        f_l = min(f_l, keep ? r : INT_MAX);
        l_l = max(l_l, keep ? r : -1);
        rv_set<R>(res[v], m, keep ? o : 0.0);
      }
    }
    const unsigned cm = (unsigned)((anykeep | (anykeep >> 16) | (anykeep >> 32) | (anykeep >> 48)) & 0xffffull);
    if (c_l) {
      atomicAdd(&col_cnt[jj], c_l);
      atomicMin(&col_first[jj], f_l);
      atomicMax(&col_last[jj], l_l);
      if constexpr (EPI != 0) {
        if constexpr (LAB) atomicMax(&col_plast[jj], pl_l);
      }
    }
    if constexpr (EPI == 2) {
      if (pm_l >= 0) atomicMax(&col_pmax[jj], pm_l);
    }
    if ((cm >> jj) & 1u) {   // the column has an entry in this tile: its 16 R rows are written (zeros = holes)
#pragma unroll
      for (int v = 0; v < 4; ++v) rv_store<R>(orun + (r0 + R * (4 * v + q)), res[v]);
    }
    if constexpr (EPI != 0) {
      if (cm && otile) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
          for (int m = 0; m < R; ++m) otile[(int64_t)(r0 + R * (4 * v + q) + m) * SLAB_J] = rv_get<R>(res[v], m);
        }
      }
    }
    if (lane == 0) colmask[t] = cm;
#ifndef NTP_TILE_STATIC
    if constexpr (EPI != 0) {
      const double td = wave_sum_f64(dsum), tt = wave_sum_f64(tsum);
      if (lane == 0) {
        tred[2 * t] = td;
        tred[2 * t + 1] = tt;
      }
    }
#endif
    STAMP(sidx); ++sidx;
  }
  STAMP(2);
  // ---- the block
#ifdef NTP_TILE_STATIC
  if constexpr (EPI != 0) {
    dsum = wave_sum_f64(dsum);
    tsum = wave_sum_f64(tsum);
    if (lane == 0) {
      red[2 * wave] = dsum;
      red[2 * wave + 1] = tsum;
    }
  }
#endif
  if (lane == 0 && pn) atomicAdd(&misc[1], pn);
  __syncthreads();
  STAMP(3);
  if constexpr (EPI == 2) {
    // the deferred elements: kept (unfiltered) where they lie beyond the last kept entry of the product column.  Only
    // the decision and the column statistics here; the values are stored after the holes have been zeroed (below)
    const int nd = misc[0];
    if (nd != 0) {   // (block-uniform; no deferred element -- the rule -- and the two barriers below are not entered)
    if (nd > TILE_DEFER) {
      if (tid == 0) atomicOr(a.fzv.flag, 1);
    } else {
      for (int i = tid; i < nd; i += TILE_NW * WAVE) {
        const int4 e = *reinterpret_cast<const int4*>(&dlist[i]);   // (r, jj, prow, pad)
        const bool kept = e.z > col_pmax[e.y];
        if (kept) {
          atomicAdd(&col_cnt[e.y], 1);
          atomicMin(&col_first[e.y], e.x);
          atomicMax(&col_last[e.y], e.x);
          if constexpr (LAB) atomicMax(&col_plast[e.y], e.z);
        }
        dlist[i].pad = kept ? 1 : 0;
      }
    }
    __syncthreads();
    // the kept ones contribute to the sums in (row, column) order -- reproducible, whatever order the list was filled
    // in: every element finds its rank among the kept ones and leaves its terms there (the multiplier tile is no longer
    // needed: k4max >= 8 rows of 16 hold 2 x TILE_DEFER values)
    if (nd <= TILE_DEFER && tid < nd) {
      const TileDefer e = dlist[tid];
      if (e.pad) {
        int rank = 0;
        for (int m2 = 0; m2 < nd; ++m2) {
          const TileDefer f = dlist[m2];
          rank += (f.pad && (f.r < e.r || (f.r == e.r && f.jj < e.jj))) ? 1 : 0;
        }
        Bs[rank] = __dmul_rn(e.o, e.d);
        Bs[TILE_DEFER + rank] = (e.r == b * SLAB_J + e.jj + a.fzv.col_offset) ? e.o : 0.0;
      }
    }
    __syncthreads();
    }
  }
  // entries, first and last row of every column; where its run and the block's tile rows start
  if (tid < SLAB_J) {
    const int jt = b * SLAB_J + tid;
    const int cf = col_first[tid], cl = col_last[tid];
    if (jt < a.ncols) {
      a.count[jt] = col_cnt[tid];
      a.ofirst[jt] = cf;
      a.olast[jt] = cl;
      a.ooff[jt] = tbase + (int64_t)tid * w + (cl >= cf ? cf - lo : 0);
      if constexpr (EPI != 0) {
        if constexpr (LAB) a.fzv.oplast[jt] = col_plast[tid];
      }
    }
  }
  int tk0 = INT_MAX, tk1 = -1;
#pragma unroll
  for (int c = 0; c < SLAB_J; ++c) {
    tk0 = min(tk0, col_first[c]);
    tk1 = max(tk1, col_last[c]);
  }
  if constexpr (EPI != 0) {
    if (tid == 0) {
      a.otoff[b] = tbase + (tk1 >= tk0 ? (int64_t)(tk0 - lo) * SLAB_J : 0);
      a.fzv.pnnz[b] = misc[1];
      if (a.fzv.prod) a.fzv.prod[b] = *reinterpret_cast<long long*>(misc + 2);
    }
    if (tid == 64) {
      double x = 0.0, y = 0.0;
#ifdef NTP_TILE_STATIC
      for (int qq = 0; qq < TILE_NW; ++qq) {
        x = __dadd_rn(x, red[2 * qq]);
        y = __dadd_rn(y, red[2 * qq + 1]);
      }
#else
      for (int t2 = 0; t2 < T; ++t2) {
        x = __dadd_rn(x, tred[2 * t2]);
        y = __dadd_rn(y, tred[2 * t2 + 1]);
      }
#endif
      if constexpr (EPI == 2) {   // kept deferred elements, in (row, column) order (ranked above)
        const int nd = min(misc[0], TILE_DEFER);
        int nk = 0;
        for (int m2 = 0; m2 < nd; ++m2) nk += dlist[m2].pad;
        for (int m2 = 0; m2 < nk; ++m2) {
          x = __dadd_rn(x, Bs[m2]);
          y = __dadd_rn(y, Bs[TILE_DEFER + m2]);
        }
      }
      a.fzv.part[2 * b] = x;
      a.fzv.part[2 * b + 1] = y;
    }
  }
  // holes: a tile strictly inside a column's run (inside the block's tile rows) that was skipped above holds zeros
  for (int p = tid; p < T * SLAB_J; p += TILE_NW * WAVE) {
    const int t = p >> 4, c = p & 15;
    const unsigned cmk = colmask[t];
    const int cf = col_first[c], cl = col_last[c];
    const int r0 = lo + TROWS * t;
    if (cl >= cf && r0 + TROWS - 1 >= cf && r0 <= cl && !((cmk >> c) & 1u)) {
      double* dst = a.out_val + (tbase + (int64_t)c * w - lo);
      for (int r = r0; r < min(r0 + TROWS, rend); ++r) dst[r] = 0.0;
    }
    if constexpr (EPI != 0) {
      if (a.fzv.tiles && tk1 >= tk0 && r0 + TROWS - 1 >= tk0 && r0 <= tk1 && cmk == 0u) {
        double* dst = a.fzv.tiles + (tbase - (int64_t)lo * SLAB_J + c);
        for (int r = r0; r < min(r0 + TROWS, rend); ++r) dst[(int64_t)r * SLAB_J] = 0.0;
      }
    }
  }
  if constexpr (EPI == 2) {
    const int nd = min(misc[0], TILE_DEFER);
    if (nd > 0) {
      __syncthreads();   // (the zeros above first)
      for (int i = tid; i < nd; i += TILE_NW * WAVE) {
        const TileDefer e = dlist[i];
        if (!e.pad) continue;
        a.out_val[tbase + (int64_t)e.jj * w + (e.r - lo)] = e.o;
        if (a.fzv.tiles) a.fzv.tiles[tbase + (int64_t)(e.r - lo) * SLAB_J + e.jj] = e.o;
      }
    }
  }
  STAMP(55);
#ifdef NTP_TILE_STAMPS
  if (tid == 0 && a.blkdur) {
    a.blkdur[4 * b + 1] = __builtin_amdgcn_s_memtime();
    a.blkdur[4 * b + 2] = EPI == 2 ? misc[0] : 0;
    a.blkdur[4 * b + 3] = kn;
  }
#endif
}

}  // namespace

#ifndef NTP_WITH_TILE2
// (the two-block geometry is an experiment outside the product build, csrc/experiments/spgemm_tile2.hip: declined here)
bool launch_spgemm_tile2(const TileLaunch&, int*) { return false; }
#endif

int tile_rows() {
  const int r = options().tile_rows;
  return r == 4 ? 4 : r == 1 ? 1 : 2;
}

bool spgemm_tile_fits(int max_kn, int max_w) {
  const int k4 = std::max(8, (max_kn + 3) & ~3), tm = (max_w + 15) >> 4;
  return max_kn > 0 && max_w > 0 && std::max(tile_lds_bytes(k4, tm, 16 * tm + 64), tile_lds_bytes(k4, tm, 0)) <= 150 * 1024;   // (with the labels of a label-aware step)
}

void launch_spgemm_tile(const TileLaunch& L) {
  TileArgs a;
  a.runs = static_cast<const SlabRun*>(L.runs);
  a.bblk = L.bblk; a.blk_boff = L.blk_boff; a.blk_kmin = L.blk_kmin; a.blk_kn = L.blk_kn; a.blk_lo = L.blk_lo; a.blk_w = L.blk_w;
  a.blk_toff = L.blk_toff; a.out_val = L.out_val; a.count = L.count; a.ofirst = L.ofirst; a.olast = L.olast; a.ooff = L.ooff;
  a.otoff = L.otoff; a.alpha = L.alpha; a.threshold = L.threshold; a.dense_rule = L.dense_rule; a.ncols = L.ncols;
  a.nblocks = L.nblocks;
  a.nrows = L.nrows > 0 ? L.nrows : L.ncols;
  a.k4max = std::max(8, (L.max_kn + 3) & ~3);   // (>= 8: the multiplier tile doubles as scratch for 2 x TILE_DEFER sums)
  const int trows = 16 * (L.rows == 4 ? 4 : L.rows == 2 ? 2 : 1);
  a.tmax = (L.max_w + trows - 1) / trows;
  if (L.fz) a.fzv = *static_cast<const SlabFuseArgs*>(L.fz);   // (a HOST copy: it travels with the kernel arguments)
  a.brun_first = L.brun_first; a.brun_last = L.brun_last; a.brun_off = L.brun_off; a.brun_val = L.brun_val;
  a.bbytes = (options().tile_bbuf != 0 && L.brun_val && L.bbytes > 0 && L.bbytes < 0xffffe000ull) ? (uint32_t)L.bbytes : 0u;
  a.bpair = (a.bbytes != 0u && options().tile_bbuf >= 2 && L.brun_pad >= 2 && L.brun_pad % 2 == 0) ? 1 : 0;
  static DevBuf<double>* zeros = nullptr;   // (never freed: lives as long as the library)
  if (!zeros) {
    zeros = new DevBuf<double>(8);
    zeros->zero();
  }
  a.zero = zeros->p;
  // 32-bit offsets: every run of A in one allocation below 4 GB; with a fused epilogue also X inside that allocation and the
  // expanded D operand in one of its own
  bool off32 = options().tile_off32 != 0 && L.abase != nullptr && L.abytes > 0 && L.abytes < 0xfffff000ull;
  if (off32 && L.epi != 0) {
    const SlabFuseArgs& fzh = *static_cast<const SlabFuseArgs*>(L.fz);
    const char *a0 = static_cast<const char*>(L.abase), *x0 = reinterpret_cast<const char*>(fzh.xexp);
    off32 = L.dbase != nullptr && L.dbytes > 0 && L.dbytes < 0xfffff000ull && (L.epi != 2 || (x0 >= a0 && x0 < a0 + L.abytes));
  }
  a.abase = off32 ? L.abase : nullptr;
  a.abytes = off32 ? (uint32_t)L.abytes : 0u;
  a.dbase = off32 ? L.dbase : nullptr;
  a.dbytes = off32 ? (uint32_t)L.dbytes : 0u;
  a.ablate = options().spgemm_variant == 601 ? 1 : options().spgemm_variant == 602 ? 2 : 0;
#ifdef NTP_TILE_STAMPS
  static DevBuf<long long>* stamps = nullptr;
  if (!stamps) stamps = new DevBuf<long long>(64 * 8 * 64);
  stamps->zero();
  a.stamps = stamps->p;
  static DevBuf<long long>* blkdur = nullptr;
  if (!blkdur || blkdur->n < (size_t)4 * L.nblocks) { delete blkdur; blkdur = new DevBuf<long long>((size_t)4 * L.nblocks); }
  blkdur->zero();
  a.blkdur = blkdur->p;
#endif
  const size_t lds = tile_lds_bytes(a.k4max, a.tmax, (L.labelled && L.epi != 0) ? a.tmax * trows : 0);
  // waves per workgroup (option tile_waves overrides): four when FOUR workgroups fit a CU's LDS (sixteen waves, four per SIMD --
  // what 128 registers per lane admit -- in four independent blocks: the prologue of one runs under the main loop of the
  // others), eight when fewer do: two workgroups of eight are again sixteen waves where three of four would be twelve.  (Up to
  // round 5 three workgroups of four were preferred to two of eight; since a wave takes its tiles from a counter, eight waves
  // share a window without waiting for one another at its end, and the fourth wave per SIMD pays: headline 683 -> 696
  // iterations/s, profiles/README.md round 6.  LDS is granted in units of 512 bytes.)
#ifdef NTP_TILE_WIDE3   // (A/B builds: the rule up to round 5 -- eight waves only where fewer than THREE workgroups of four fit)
  const bool wide = 3 * ((lds + 511) & ~(size_t)511) > 160 * 1024;
#else
  const bool wide = 4 * ((lds + 511) & ~(size_t)511) > 160 * 1024;
#endif
  const int tw = options().tile_waves;
  const int nw = (tw == 4 || tw == 5 || tw == 6 || tw == 8) ? tw : (wide ? 8 : 4);
  auto go = [&](auto epi_tag, auto nw_tag, auto r_tag, auto lab_tag, auto off_tag) {
    constexpr int E = decltype(epi_tag)::value, NW = decltype(nw_tag)::value, RR = decltype(r_tag)::value;
    constexpr bool LB = decltype(lab_tag)::value, OF = decltype(off_tag)::value;
    static size_t raised = 0;   // (one per instantiation)
    if (lds > 64 * 1024 && lds > raised) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spgemm_tile<E, NW, RR, LB, OF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    150 * 1024));
      raised = 150 * 1024;
    }
    hipLaunchKernelGGL((k_spgemm_tile<E, NW, RR, LB, OF>), dim3(xcd_grid(L.nblocks)), dim3(NW * WAVE), lds, stream(), a);
  };
  auto by_off = [&](auto epi_tag, auto nw_tag, auto r_tag, auto lab_tag) {
    // (32-bit offsets: the default geometries only -- four or eight waves, one or two rows per lane)
    constexpr int NW = decltype(nw_tag)::value, RR = decltype(r_tag)::value;
    if constexpr ((NW == 4 || NW == 8) && RR <= 2) {
      if (off32) return go(epi_tag, nw_tag, r_tag, lab_tag, std::true_type{});
    }
    go(epi_tag, nw_tag, r_tag, lab_tag, std::false_type{});
  };
  auto by_nw = [&](auto epi_tag, auto r_tag, auto lab_tag) {
    if (nw == 4) by_off(epi_tag, std::integral_constant<int, 4>{}, r_tag, lab_tag);
    else if (nw == 5) by_off(epi_tag, std::integral_constant<int, 5>{}, r_tag, lab_tag);
    else if (nw == 6) by_off(epi_tag, std::integral_constant<int, 6>{}, r_tag, lab_tag);
    else by_off(epi_tag, std::integral_constant<int, 8>{}, r_tag, lab_tag);
  };
  auto by_r = [&](auto epi_tag) {
    constexpr int E = decltype(epi_tag)::value;
    using F = std::false_type; using T = std::true_type;
    using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>; using R4 = std::integral_constant<int, 4>;
    if constexpr (E != 0) {
      if (L.labelled) {   // (two or one rows per lane; the layout of four is one of two as well)
        if (L.rows >= 2) by_nw(epi_tag, R2{}, T{});
        else by_nw(epi_tag, R1{}, T{});
        return;
      }
    }
    if (L.rows == 4) by_nw(epi_tag, R4{}, F{});
    else if (L.rows == 2) by_nw(epi_tag, R2{}, F{});
    else by_nw(epi_tag, R1{}, F{});
  };
  if (L.epi == 0) by_r(std::integral_constant<int, 0>{});
  else if (L.epi == 1) by_r(std::integral_constant<int, 1>{});
  else by_r(std::integral_constant<int, 2>{});
#ifdef NTP_TILE_STAMPS
  if (const char* f = std::getenv("NTP_TILE_STAMPS_FILE")) {
    std::vector<long long> h(64 * 8 * 64);
    HIP_CHECK(hipStreamSynchronize(stream()));
    HIP_CHECK(hipMemcpy(h.data(), stamps->p, h.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* fp = std::fopen(f, "wb")) { std::fwrite(h.data(), 8, h.size(), fp); std::fclose(fp); }
    std::vector<long long> hb((size_t)4 * L.nblocks);
    HIP_CHECK(hipMemcpy(hb.data(), blkdur->p, hb.size() * 8, hipMemcpyDeviceToHost));
    if (FILE* fp = std::fopen((std::string(f) + ".blocks").c_str(), "wb")) { std::fwrite(hb.data(), 8, hb.size(), fp); std::fclose(fp); }
  }
#endif
}

}  // namespace ntp
