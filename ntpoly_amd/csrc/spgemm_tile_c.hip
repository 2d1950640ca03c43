// SpGEMM numeric phase on the FP64 matrix cores for run-like COMPLEX operands (gfx950, v_mfma_f64_16x16x4_f64) -- the
// operands of the complex register-slab kernel (kernels.hip k_spgemm_slab_c; MultiplyBlock.f90:9-36 + PruneList.f90:8-38),
// walked tile by tile as spgemm_tile.hip walks the real ones.
//
// A workgroup owns a block of 8 consecutive complex output columns and the row window [lo, lo + w) they can touch.  The
// interleaved (re, im) multiplier tile of the block, B'(k, 2 c + p) = part p of B(k, c), IS a real k x 16 matrix, and
//
//     [ Re C(:, c) | Im C(:, c) ]  =  Re A * [ Re B(:, c) | Im B(:, c) ]  +  Im A * [ -Im B(:, c) | Re B(:, c) ]
//
// so a tile of 16 rows x 8 complex columns takes TWO matrix instructions per group of four k: the real plane of A
// against B', the imaginary plane against B'' = B' with the parts of every column swapped and the new real part negated
// (read from the same LDS tile at lane ^ 1, one v_xor for the sign).  A lane of the A operand loads (re, im) of one row
// with ONE 16-byte load and feeds both instructions; the two partial sums are kept apart (two independent accumulation
// chains) and added when the tile's k range is done.
//
// Arithmetic: every part of C(i, j) is the sum of two FMA chains over ascending k.  The reference's complex multiply-add
// rounds its four products, the subtraction / addition and the two accumulates one by one (the bit-for-bit kernel,
// k_spgemm_slab_c, option complex_tile = 0 or unfused arithmetic), so this kernel is a TOLERANCE mode: entries agree to
// 1e-13 relative to the largest entry, the pattern agrees except where |C(i, j)| lies within that distance of the
// threshold (tests/test_gpu_complex_tile.py), solver loops run the same iteration counts.
#include "spgemm_tile.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
constexpr int CJ = 8;            // complex columns of a block (16 real columns of the matrix instruction)
constexpr int CPF = 6;           // run loads (k groups) in flight per wave; a multiple of 3
constexpr int CRPAD = 4;         // one group of empty records behind the last

__device__ inline v2d ld_c(unsigned long long addr) { return *reinterpret_cast<const v2d __attribute__((address_space(1)))*>(addr); }
// the other part of the same complex number: lanes 2 c and 2 c + 1 hold (re, im) of column c
__device__ inline double partner(double x) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0xb1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0xb1, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

struct alignas(16) CRec {
  unsigned long long rz;   // address of (hypothetical) row 0 of the run
  int32_t first;
  uint32_t span;           // last - first; empty run: first = INT_MAX, span = 0
};

struct TileCArgs {
  const SlabRun* runs;
  const double2* bblk;
  const int64_t* blk_boff;
  const int32_t *blk_kmin, *blk_kn, *blk_lo, *blk_w;
  const int64_t* blk_toff;
  double* out_val;         // complex slots, (re, im) interleaved
  int32_t* count;
  int32_t *ofirst, *olast;
  int64_t* ooff;
  double alpha, threshold;
  int dense_rule, ncols, nblocks;
  int k4max, tmax;
  // optional: the right operand as the runs of its columns (complex slab sessions: no multiplier tiles were built) -- first /
  // last row, offset of the first row (in complex elements) and values; bblk / blk_boff are then unused
  const int32_t *brun_first, *brun_last;
  const int64_t* brun_off;
  const double2* brun_val;
  const double* zero;      // 16 bytes of zeros
};

__host__ __device__ inline size_t tile_c_lds_bytes(int k4max, int tmax) {
  return (size_t)k4max * 16 * 8 + (size_t)(k4max + CRPAD) * sizeof(CRec) + (size_t)(k4max / 4 + 1) * 8 + (size_t)((tmax + 3) & ~3) * 4 + 3 * CJ * 4 + 16 + 64;
}

// R: tiles of 16 rows a wave multiplies together (they share the multiplier rows read from LDS and the run records: 2 R
// matrix instructions per k group and R run loads per lane)
template <int NW, int R>
__global__ __launch_bounds__(NW* WAVE) __attribute__((amdgpu_waves_per_eu(R == 1 ? 4 : 3, 8))) void k_spgemm_tile_c(const TileCArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = xcd_block(a.nblocks);
  if (b < 0) return;
  const int tid = threadIdx.x, wave = uni_i32(tid / WAVE), lane = lane_id();
  const int lo = a.blk_lo[b], w = a.blk_w[b], kmin = a.blk_kmin[b], kn = a.blk_kn[b];
  const int64_t tbase = a.blk_toff[b];
  // (the multiplier tile from the runs of the block's columns: unit i of the tile is row kmin + i / 8 of column i % 8, and a
  // thread always serves the same column -- NT is a multiple of 8.  The columns' extents depend on the block's number only and
  // are requested HERE, with the plan's scalars: a memory round trip less in front of the tile's values)
  const bool brun = a.brun_val != nullptr;
  int bf0 = INT_MAX, bl0 = -1;
  const double2* bp0 = nullptr;
  if (brun) {
    const int c0 = b * CJ + (tid & 7);
    if (c0 < a.ncols) { bf0 = a.brun_first[c0]; bl0 = a.brun_last[c0]; bp0 = a.brun_val + (a.brun_off[c0] - bf0); }
  }
  if (b == 0 && tid == 0) a.ooff[a.ncols] = a.blk_toff[a.nblocks];
  if (kn == 0) {   // no product entries in these columns
    const int j = b * CJ + tid;
    if (tid < CJ && j < a.ncols) {
      a.ofirst[j] = INT_MAX;
      a.olast[j] = -1;
      a.count[j] = 0;
      a.ooff[j] = tbase + (int64_t)tid * w;
    }
    return;
  }
  // ---- LDS
  double* Bs = reinterpret_cast<double*>(smem);                                    // [k4max][16]: row k = (re, im) of the 8 columns
  CRec* recs = reinterpret_cast<CRec*>(Bs + (size_t)a.k4max * 16);                 // [k4max + CRPAD]
  int* grmin = reinterpret_cast<int*>(recs + a.k4max + CRPAD);                     // [k4max / 4 + 1]: first / last row any column of
  int* grmax = grmin + (a.k4max / 4 + 1);                                          // a k group reaches
  unsigned* colmask = reinterpret_cast<unsigned*>(grmax + (a.k4max / 4 + 1));      // [tmax]
  int* col_cnt = reinterpret_cast<int*>(colmask + ((a.tmax + 3) & ~3));            // [CJ] each
  int* col_first = col_cnt + CJ;
  int* col_last = col_first + CJ;
  int* tnext = col_last + CJ;                                                      // [1]: groups of tiles taken

  const int KG = (kn + 3) >> 2, K4 = KG * 4;
  const int T = (w + 15) >> 4;
  // ---- block prologue: multiplier tile -> LDS (rows kn .. K4 zero), run records + row range of every k group
  constexpr int NT = NW * WAVE, BCH = 3072 / NT;
  const double2* __restrict__ bsrc = brun ? a.brun_val : a.bblk + a.blk_boff[b];
  double2* bdst = reinterpret_cast<double2*>(Bs);
  double2 btmp[BCH];
  auto brun_load = [&](int i) {
    const int r = kmin + (i >> 3);
    return (i < kn * 8 && r >= bf0 && r <= bl0) ? bp0[r] : make_double2(0.0, 0.0);
  };
  // (the run records requested BEFORE the tile's values: loads return in order, records requested behind the tile's could not
  // be worked on before the last of those had arrived)
  const uint4* __restrict__ rp = reinterpret_cast<const uint4*>(a.runs + kmin);
  uint4 rh0, rh1;
  {
    const int ic = min(tid, kn - 1);
    rh0 = rp[2 * ic];
    rh1 = rp[2 * ic + 1];
  }
#pragma unroll
  for (int u = 0; u < BCH; ++u) {
    const int i = tid + u * NT;
    if (brun) btmp[u] = brun_load(i);
    else btmp[u] = i < kn * 8 ? bsrc[i] : make_double2(0.0, 0.0);
  }
  {
    for (int i0 = 0; i0 < K4 + 4; i0 += NT) {
      const int i = i0 + tid;
      const int ic = min(i, kn - 1);
      uint4 r0 = rh0, r1 = rh1;                              // (addr_lo, addr_hi, nbytes, flags), (first16, first, span62, pad)
      if (i0 != 0) { r0 = rp[2 * ic]; r1 = rp[2 * ic + 1]; }
      const int rows = i < kn ? (int)(r0.z >> 4) : 0;
      const int first = (int)r1.y;
      CRec rec;
      rec.rz = 0;
      rec.first = INT_MAX;
      rec.span = 0u;
      int rmin = INT_MAX, rmax = -1;
      if (rows > 0) {
        const unsigned long long addr = (unsigned long long)r0.x | ((unsigned long long)r0.y << 32);
        rec.rz = addr - (unsigned long long)((long long)first * 16);
        rec.first = first;
        rec.span = (uint32_t)(rows - 1);
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
    }
  }
#pragma unroll
  for (int u = 0; u < BCH; ++u) {
    const int i = tid + u * NT;
    if (i < K4 * 8) bdst[i] = btmp[u];
  }
  for (int i = tid + BCH * NT; i < K4 * 8; i += NT) {
    if (brun) bdst[i] = brun_load(i);
    else bdst[i] = i < kn * 8 ? bsrc[i] : make_double2(0.0, 0.0);
  }
  for (int t = tid; t < T; t += NT) colmask[t] = 0u;
  if (tid < CJ) {
    col_cnt[tid] = 0;
    col_first[tid] = INT_MAX;
    col_last[tid] = -1;
  }
  if (tid == 0) *tnext = 0;
  __syncthreads();

  // ---- per-lane constants: real column jj of the matrix instruction = part (jj & 1) of complex column jj >> 1
  const int jj = lane & 15, q = lane >> 4;
  const int part = lane & 1, cc = jj >> 1;
  const double* const zp = a.zero;
  const unsigned long long zaddr = reinterpret_cast<unsigned long long>(zp);
  double* const orun = a.out_val + 2 * (tbase + (int64_t)cc * w - lo) + part;   // orun[2 r] = this lane's part of row r of its column
  const double alpha = a.alpha, thr = a.threshold;
  const bool dense_rule = (a.dense_rule & 1) != 0;
  const int smask = part ? 0 : (int)0x80000000u;   // the multiplier of the imaginary plane: -Im B in the real columns, Re B in the imaginary ones
  const int rend = lo + w;
  const int TS = (T + R - 1) / R;   // groups of R tiles
  const int mid = (TS - 1) >> 1;

  // ---- epilogue of one tile of 16 rows (index t16, first row r0m): this lane holds part `part` of rows r0m + 4 v + q
  // (v = 0..3) of complex column cc
  auto epilogue = [&](const v4d& accr, const v4d& acci, int t16, int r0m) {
    double o[4];
    bool keep[4];
    bool amb_any = false, any = false;
    double tre[4], tim[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const double mine = __dadd_rn(accr[v], acci[v]);
      const double other = partner(mine);
      const double sv = __dmul_rn(alpha, mine);
      o[v] = sv;
      const double tm = dense_rule ? mine : sv;
      const double to = dense_rule ? other : __dmul_rn(alpha, other);
      tre[v] = part ? to : tm;
      tim[v] = part ? tm : to;
      // |z| > threshold (PruneList.f90:8-38 on a complex value): decided without the hypot wherever max(|re|, |im|) or
      // |re| + |im| already tells
      const double ax = fabs(tm), ay = fabs(to);
      const bool sure = fmax(ax, ay) > thr;
      const bool amb = !sure && __dadd_rn(ax, ay) > thr;
      keep[v] = sure;
      amb_any |= amb;
      any |= sure | amb;
    }
    if (__ballot(any) == 0ull) return;   // nothing of the tile survives: colmask[t16] stays 0
    if (__ballot(amb_any) != 0ull) {
#pragma unroll
      for (int v = 0; v < 4; ++v) keep[v] = hypot(tre[v], tim[v]) > thr;
    }
    unsigned long long anykeep = 0;
    int c_l = 0, f_l = INT_MAX, l_l = -1;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      const int r = r0m + 4 * v + q;
      anykeep |= __ballot(keep[v]);
      c_l += keep[v] ? 1 : 0;
      f_l = min(f_l, keep[v] ? r : INT_MAX);
      l_l = max(l_l, keep[v] ? r : -1);
      o[v] = keep[v] ? o[v] : 0.0;
    }
    const unsigned cm = (unsigned)((anykeep | (anykeep >> 16) | (anykeep >> 32) | (anykeep >> 48)) & 0xffffull);
    if (c_l && part == 0) {
      atomicAdd(&col_cnt[cc], c_l);
      atomicMin(&col_first[cc], f_l);
      atomicMax(&col_last[cc], l_l);
    }
    if ((cm >> jj) & 1u) {   // the column has an entry in this tile: its 16 rows are written (zeros = holes)
#pragma unroll
      for (int v = 0; v < 4; ++v) orun[2 * (int64_t)(r0m + 4 * v + q)] = o[v];
    }
    if (lane == 0) colmask[t16] = cm;
  };

  // A wave takes its next group of tiles from a counter in LDS when it is through with its last (as k_spgemm_tile does: the
  // tiles of a window cost between a few and all of the k groups, a fixed deal left the waves apart at the block's end); centre
  // first: the tiles in the middle of the window have the longest k ranges.  (The loop is counted: written as for (;;) the
  // kernel did not come back -- profiles/README.md round 6 -- and a wave takes at most TS groups anyway.)
  for (int taken = 0; taken <= TS; ++taken) {
    int ts = 0;
    if (lane == 0) ts = atomicAdd(tnext, 1);
    ts = __builtin_amdgcn_readfirstlane(ts);
    if (ts >= TS) break;
    const int t = (ts & 1) ? mid + ((ts + 1) >> 1) : mid - (ts >> 1);
    const int r0 = lo + 16 * R * t;
    int g0 = INT_MAX, g1 = -1;   // the k groups that can reach the tiles: a ballot over the groups' row ranges
    for (int c = 0; c < KG; c += WAVE) {
      const int gq = min(c + lane, KG);
      const unsigned long long m = __ballot(grmin[gq] <= r0 + 16 * R - 1 && grmax[gq] >= r0);   // (group KG: empty, never true)
      if (m) {
        if (g0 == INT_MAX) g0 = c + (int)__builtin_ctzll(m);
        g1 = c + 63 - (int)__builtin_clzll(m);
      }
    }
    if (g1 < g0) continue;   // (no k group reaches the tiles: their colmask entries stay 0)
    v4d accr[R], acci[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
      accr[m] = v4d{0.0, 0.0, 0.0, 0.0};
      acci[m] = v4d{0.0, 0.0, 0.0, 0.0};
    }
    {
      const int rl = r0 + jj;                      // A operand: rows rl + 16 m, column 4 g + q
      const unsigned long long r16 = (unsigned long long)((long long)rl * 16);
      const uint4* __restrict__ rq = reinterpret_cast<const uint4*>(recs) + q;     // record of group g: rq[4 g]
      const double* __restrict__ bq = Bs + lane;                                   // B'(4 g + q, jj) = bq[64 g]
      const double* __restrict__ bx = Bs + (lane ^ 1);                             // the other part of the same column
      auto run_load = [&](const uint4 raw, int m) -> v2d {
        const unsigned long long rz = (unsigned long long)raw.x | ((unsigned long long)raw.y << 32);
        const bool ok = (unsigned)(rl + 16 * m - (int)raw.z) <= raw.w;
        return ld_c(ok ? rz + r16 + 256ull * (unsigned)m : zaddr);
      };
      auto swz = [&](double x) { return __hiloint2double(__double2hiint(x) ^ smask, __double2loint(x)); };
      v2d ring[CPF][R];
      double bb[3], bs[3];
      const int KGm1 = KG - 1;
#pragma unroll
      for (int u = 0; u < CPF; ++u) {
        const uint4 rw = rq[4 * min(g0 + u, KG)];
#pragma unroll
        for (int m = 0; m < R; ++m) ring[u][m] = run_load(rw, m);
      }
      bb[0] = bq[64 * g0];
      bs[0] = swz(bx[64 * g0]);
      bb[1] = bq[64 * min(g0 + 1, KGm1)];
      bs[1] = swz(bx[64 * min(g0 + 1, KGm1)]);
      bb[2] = 0.0;
      bs[2] = 0.0;
      uint4 raw = rq[4 * min(g0 + CPF, KG)];
      int g = g0;
      for (; g + CPF - 1 <= g1; g += CPF) {
#pragma unroll
        for (int u = 0; u < CPF; ++u) {
          // (the order is pinned: record read one slot ahead | matrix instructions | refill of the slot they have read)
          const uint4 raw_n = rq[4 * min(g + u + CPF + 1, KG)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < R; ++m) {
            accr[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][m].x, bb[u % 3], accr[m], 0, 0, 0);
            acci[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][m].y, bs[u % 3], acci[m], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < R; ++m) ring[u][m] = run_load(raw, m);
          bb[(u + 2) % 3] = bq[64 * min(g + u + 2, KGm1)];
          bs[(u + 2) % 3] = swz(bx[64 * min(g + u + 2, KGm1)]);
          raw = raw_n;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int u = 0; u < CPF - 1; ++u) {
        if (g + u <= g1) {
          const double bt = bq[64 * (g + u)], bu = swz(bx[64 * (g + u)]);
#pragma unroll
          for (int m = 0; m < R; ++m) {
            accr[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][m].x, bt, accr[m], 0, 0, 0);
            acci[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(ring[u][m].y, bu, acci[m], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int m = 0; m < R; ++m) {
      const int t16 = R * t + m;
      if (t16 < T) epilogue(accr[m], acci[m], t16, r0 + 16 * m);   // (the window is a multiple of 16 rows, not of 16 R)
    }
  }
  __syncthreads();
  // entries, first and last row of every column; where its run starts
  if (tid < CJ) {
    const int jt = b * CJ + tid;
    const int cf = col_first[tid], cl = col_last[tid];
    if (jt < a.ncols) {
      a.count[jt] = col_cnt[tid];
      a.ofirst[jt] = cf;
      a.olast[jt] = cl;
      a.ooff[jt] = tbase + (int64_t)tid * w + (cl >= cf ? cf - lo : 0);
    }
  }
  // holes: a tile strictly inside a column's run that was skipped above holds zeros
  for (int p = tid; p < T * CJ; p += NT) {
    const int t = p >> 3, c = p & 7;
    const unsigned cmk = colmask[t];
    const int cf = col_first[c], cl = col_last[c];
    const int r0 = lo + 16 * t;
    if (cl >= cf && r0 + 15 >= cf && r0 <= cl && !((cmk >> (2 * c)) & 1u)) {
      double2* dst = reinterpret_cast<double2*>(a.out_val) + (tbase + (int64_t)c * w - lo);
      for (int r = r0; r < min(r0 + 16, rend); ++r) dst[r] = make_double2(0.0, 0.0);
    }
  }
}

}  // namespace

bool spgemm_tile_c_fits(int max_kn, int max_w) {
  const int k4 = std::max(8, (max_kn + 3) & ~3), tm = (max_w + 15) >> 4;
  return max_kn > 0 && max_w > 0 && tile_c_lds_bytes(k4, tm) <= 150 * 1024;
}

void launch_spgemm_tile_c(const TileLaunch& L) {
  TileCArgs a;
  a.runs = static_cast<const SlabRun*>(L.runs);
  a.bblk = reinterpret_cast<const double2*>(L.bblk);
  a.brun_first = L.brun_first; a.brun_last = L.brun_last; a.brun_off = L.brun_off;
  a.brun_val = reinterpret_cast<const double2*>(L.brun_val);
  a.blk_boff = L.blk_boff; a.blk_kmin = L.blk_kmin; a.blk_kn = L.blk_kn; a.blk_lo = L.blk_lo; a.blk_w = L.blk_w;
  a.blk_toff = L.blk_toff; a.out_val = L.out_val; a.count = L.count; a.ofirst = L.ofirst; a.olast = L.olast; a.ooff = L.ooff;
  a.alpha = L.alpha; a.threshold = L.threshold; a.dense_rule = L.dense_rule; a.ncols = L.ncols; a.nblocks = L.nblocks;
  a.k4max = std::max(8, (L.max_kn + 3) & ~3);
  a.tmax = (L.max_w + 15) / 16;
  static DevBuf<double>* zeros = nullptr;   // (never freed: lives as long as the library)
  if (!zeros) {
    zeros = new DevBuf<double>(8);
    zeros->zero();
  }
  a.zero = zeros->p;
  const size_t lds = tile_c_lds_bytes(a.k4max, a.tmax);
  const bool wide = 4 * ((lds + 511) & ~(size_t)511) > 160 * 1024;   // (as launch_spgemm_tile: eight waves where fewer than FOUR workgroups of four fit; LDS is granted in units of 512 bytes)
  const int tw = options().tile_waves;
  const int nw = (tw == 4 || tw == 8) ? tw : (wide ? 8 : 4);
  // R = 2 (two tiles per wave step) where the windows hold enough tiles for every wave to get a pair (option tile_rows = 1: one)
  const int rr = (options().tile_rows == 1 || a.tmax < 2 * nw) ? 1 : 2;
  auto go = [&](auto nw_tag, auto r_tag) {
    constexpr int NW = decltype(nw_tag)::value, RR = decltype(r_tag)::value;
    static size_t raised = 0;   // (one per instantiation)
    if (lds > 64 * 1024 && lds > raised) {
      HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spgemm_tile_c<NW, RR>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      raised = 150 * 1024;
    }
    hipLaunchKernelGGL((k_spgemm_tile_c<NW, RR>), dim3(xcd_grid(L.nblocks)), dim3(NW * WAVE), lds, stream(), a);
  };
  using R1 = std::integral_constant<int, 1>; using R2 = std::integral_constant<int, 2>;
  if (nw == 4 && rr == 1) go(std::integral_constant<int, 4>{}, R1{});
  else if (nw == 4) go(std::integral_constant<int, 4>{}, R2{});
  else if (rr == 1) go(std::integral_constant<int, 8>{}, R1{});
  else go(std::integral_constant<int, 8>{}, R2{});
}

}  // namespace ntp
