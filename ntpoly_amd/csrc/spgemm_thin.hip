// SpGEMM C = alpha A B (MultiplyBlock.f90:9-36 + PruneList.f90:8-38) for a THIN left operand: A holds a handful of
// entries per row -- an identity, a near-diagonal correction factor.  The solver loops multiply such factors into wide
// iterates near convergence (InverseSquareRoot / SquareRoot: T_k = (3 I - Z_k Y_k) / 2 -> I, SquareRootSolversModule.F90:
// 342-531; the polynomial and the Newton-Schulz loops alike).  The column-driven kernels walk column j of B and, for every
// entry B(k, j), column k of A with a whole wave -- one or two active lanes of 64 when A(:, k) holds one or two entries.
//
// Here the OUTPUT drives: one wave per column j, the column of B scattered into a dense LDS window over its row extent
// (zeros = no entry), one lane per candidate row i, which walks ROW i of A (column i of A^T, a few entries in ascending k)
// and gathers B(k, j) from the window:
//
//     C(i, j) = sum over the entries A(i, k) in ascending k of A(i, k) * B(k, j)
//
// -- the reference's accumulation order and its arithmetic (separate multiply and add, or one fma per product under option
// spgemm_fma for real operands; complex operands always unfused), so the result is the reference's bit for bit in both
// modes: a hole of B contributes a * 0, which changes no partial sum, and an all-zero sum never passes the prune.
// Two passes of the same kernel (count, then fill: the arithmetic is cheap, the traffic is the operands') write C
// straight into compressed columns.
#include <hip/hip_runtime.h>

#include <memory>

#include <algorithm>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {

constexpr int THIN_NW = 4;
constexpr int THIN_U = 4;    // row chunks of a wave in flight together

// extents of the operands: st[0] = max over the columns k of A of (k - first row), st[1] = max of (last row - k),
// st[2] = widest row extent of a column of B; bfirst / bext per column of B (bext = 0: empty)
__global__ void k_thin_extents(Csc A, Csc B, int32_t* __restrict__ bfirst, int32_t* __restrict__ bext, unsigned long long* __restrict__ st) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  int up = 0, dn = 0, ext = 0;
  if (t < A.cols) {
    const int64_t s = A.outer[t], e = A.outer[t + 1];
    if (e > s) {
      up = max(0, t - A.inner[s]);
      dn = max(0, A.inner[e - 1] - t);
    }
  }
  if (t < B.cols) {
    const int64_t s = B.outer[t], e = B.outer[t + 1];
    int f = 0;
    if (e > s) {
      f = B.inner[s];
      ext = B.inner[e - 1] - f + 1;
    }
    bfirst[t] = f;
    bext[t] = ext;
  }
  up = wave_max_i32(up);
  dn = wave_max_i32(dn);
  ext = wave_max_i32(ext);
  if (lane_id() == 0) {
    if (up) atomicMax(&st[0], (unsigned long long)up);
    if (dn) atomicMax(&st[1], (unsigned long long)dn);
    if (ext) atomicMax(&st[2], (unsigned long long)ext);
  }
}

template <typename T, bool FILL>
__global__ __launch_bounds__(THIN_NW* WAVE) void k_spgemm_thin(Csc AT, Csc B, const int32_t* __restrict__ bfirst, const int32_t* __restrict__ bext,
                                                               int up, int dn, int wmax, int32_t* __restrict__ count,
                                                               const int64_t* __restrict__ outer, int32_t* __restrict__ out_inner, T* __restrict__ out_val,
                                                               double alpha, double threshold, int dense_rule, int nblocks,
                                                               unsigned long long* __restrict__ products) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = xcd_block(nblocks);
  if (b < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = b * THIN_NW + wave;
  if (j >= B.cols) return;
  T* win = reinterpret_cast<T*>(smem) + (size_t)wave * wmax;
  const int ext = bext[j], bf = bfirst[j];
  if (ext == 0) {
    if (!FILL && lane == 0) count[j] = 0;
    return;
  }
  const bool fma = (dense_rule & 2) != 0;
  const T* __restrict__ Bv = static_cast<const T*>(B.val);
  const T* __restrict__ Av = static_cast<const T*>(AT.val);
  for (int s = lane; s < ext; s += WAVE) win[s] = Sc<T>::zero();
  __builtin_amdgcn_wave_barrier();
  for (int64_t p = B.outer[j] + lane, e = B.outer[j + 1]; p < e; p += WAVE) win[B.inner[p] - bf] = Bv[p];
  __builtin_amdgcn_wave_barrier();
  const int lo = max(0, bf - up), hi = min(AT.cols, bf + ext + dn);   // candidate rows [lo, hi)
  int64_t pos = FILL ? outer[j] : 0;
  int cnt = 0;
  long long np = 0;
  // (THIN_U chunks of 64 rows per step: their row walks are independent chains of dependent loads -- row pointers, column
  // index, values -- and proceed together)
  for (int r0 = lo; r0 < hi; r0 += THIN_U * WAVE) {
    T acc[THIN_U];
    int64_t p[THIN_U], e[THIN_U];
#pragma unroll
    for (int u = 0; u < THIN_U; ++u) {
      const int i = r0 + u * WAVE + lane;
      acc[u] = Sc<T>::zero();
      p[u] = 0;
      e[u] = 0;
      if (i < hi) {
        p[u] = AT.outer[i];
        e[u] = AT.outer[i + 1];
      }
    }
    for (;;) {
      // (one entry of each of the THIN_U rows per round; the loads of a round are issued together: index, then value and window)
      bool v[THIN_U], in[THIN_U], any = false;
      int k[THIN_U];
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) {
        v[u] = p[u] < e[u];
        any |= v[u];
      }
      if (__ballot(any) == 0ull) break;
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) k[u] = v[u] ? AT.inner[p[u]] : INT_MIN;
      T av[THIN_U], bv[THIN_U];
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) {
        in[u] = v[u] && (unsigned)(k[u] - bf) < (unsigned)ext;
        av[u] = in[u] ? Av[p[u]] : Sc<T>::zero();
        bv[u] = in[u] ? win[k[u] - bf] : Sc<T>::zero();
      }
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) {
        if (in[u]) {
          acc[u] = Sc<T>::fmadd(av[u], bv[u], acc[u], fma);
          if (!FILL && products) np += Sc<T>::is_zero(bv[u]) ? 0 : 1;
        }
        p[u] += v[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < THIN_U; ++u) {
      const int i = r0 + u * WAVE + lane;
      const T sv = Sc<T>::scale(alpha, acc[u]);
      const bool keep = i < hi && ((dense_rule & 1) ? (Sc<T>::mag(acc[u]) > threshold) : (Sc<T>::mag(sv) > threshold));
      const unsigned long long m = __ballot(keep);
      if (FILL && keep) {
        const int64_t q = pos + __popcll(m & lanemask_lt());
        out_inner[q] = i;
        out_val[q] = sv;
      }
      pos += __popcll(m);
      cnt += __popcll(m);
    }
  }
  if (!FILL) {
    if (lane == 0) count[j] = cnt;
    if (products) {
      np = wave_sum_i64(np);
      if (lane == 0 && np) atomicAdd(products, (unsigned long long)np);
    }
  }
}

// ---- thin operands inside a slab session (kernels.hpp slab algebra; real, FMA arithmetic).  Operands and result in slab
// form: column j of the result is written as a dense run over the row window of its block of 16 columns (the plan of
// slab_multiply: a superset of the rows any product of the column can reach), zeros = no entry.  Every entry is the fma
// chain over ascending k the MFMA tile kernel computes (zero padding adds exact zeros there), so the two agree bit for bit.
constexpr int THIN_MAXK = 64;   // non-zeros of a column of a thin right operand the kernel lists in LDS (more: the step is handed back)

__device__ inline void thin_finish(const ThinSlabArgs& a, int j, int64_t slot, int lo, int cnt, int f, int l) {
  a.count[j] = cnt;
  a.ofirst[j] = f;
  a.olast[j] = l;
  a.ooff[j] = slot + (l >= f ? f - lo : 0);
}

// A chunk of 64 rows [rb, re) with a kept entry is written whole (zeros = no entry), after the zeros of the chunks skipped
// since the last written one (gap = first unwritten row behind it; INT_MAX: nothing written yet).  Chunks before the first
// and behind the last kept entry are never written: a column's slot is only read over [first, last] widened to the row pad,
// which the chunks (multiples of it) cover.
__device__ inline void thin_store(double* __restrict__ col, int gap, int rb, int re, int lane, double v) {
  if (gap < rb)
    for (int r = gap + lane; r < rb; r += WAVE) col[r] = 0.0;
  if (rb + lane < re) col[rb + lane] = v;
}

// LEFT operand thin: lane = candidate row i, walks row i of A (column i of A^T) ONCE for THIN_CW adjacent columns of the
// result (they share the block's row window) and gathers B(k, j) from the runs of those columns
constexpr int THIN_CW = 4, THIN_UL = 2;
__global__ __launch_bounds__(THIN_NW* WAVE) void k_thin_slab_left(const ThinSlabArgs a) {
  const int ngroups = (a.ncols + THIN_CW - 1) / THIN_CW;
  const int nb4 = (ngroups + THIN_NW - 1) / THIN_NW;
  const int blk = xcd_block(nb4);
  if (blk < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int grp = blk * THIN_NW + wave;
  if (grp >= ngroups) return;
  const int j0 = grp * THIN_CW;               // (16 is a multiple of THIN_CW: the columns of a group belong to one block)
  const int b = j0 >> 4;
  const int lo = a.blk_lo[b], w = a.blk_w[b];
  int bf[THIN_CW];
  unsigned ext[THIN_CW];
  const double* bp[THIN_CW];
  int64_t slot[THIN_CW];
  bool live[THIN_CW];
  bool any_live = false;
#pragma unroll
  for (int c = 0; c < THIN_CW; ++c) {
    const int j = j0 + c;
    const int jc = min(j, a.ncols - 1);
    slot[c] = a.blk_toff[b] + (int64_t)(jc & 15) * w;
    const int f0 = a.bfirst[jc], l0 = a.blast[jc];
    live[c] = j < a.ncols && l0 >= f0 && w > 0;
    bf[c] = live[c] ? f0 : 0;
    ext[c] = live[c] ? (unsigned)(l0 - f0) : 0u;
    bp[c] = a.bval + (live[c] ? a.boff[jc] - f0 : 0);   // bp[c][k] = B(k, j0 + c), bf <= k <= bf + ext
    any_live |= live[c];
    if (!live[c] && j < a.ncols && lane == 0) thin_finish(a, j, slot[c], lo, 0, INT_MAX, -1);
  }
  if (!any_live) return;
  const bool dense_rule = (a.dense_rule & 1) != 0;
  int cnt[THIN_CW], f[THIN_CW], l[THIN_CW], gap[THIN_CW];
#pragma unroll
  for (int c = 0; c < THIN_CW; ++c) {
    cnt[c] = 0;
    f[c] = INT_MAX;
    l[c] = -1;
    gap[c] = INT_MAX;
  }
  for (int r0 = lo; r0 < lo + w; r0 += THIN_UL * WAVE) {
    double acc[THIN_UL][THIN_CW];
    int64_t p[THIN_UL], e[THIN_UL];
#pragma unroll
    for (int u = 0; u < THIN_UL; ++u) {
      const int i = r0 + u * WAVE + lane;
#pragma unroll
      for (int c = 0; c < THIN_CW; ++c) acc[u][c] = 0.0;
      p[u] = 0;
      e[u] = 0;
      if (i < lo + w && i < a.nrows) {
        p[u] = a.at_outer[i];
        e[u] = a.at_outer[i + 1];
      }
    }
    for (;;) {
      // (one entry of each of the THIN_UL rows per round; the loads of a round are issued together)
      bool v[THIN_UL], any = false;
      int k[THIN_UL];
      double av[THIN_UL];
#pragma unroll
      for (int u = 0; u < THIN_UL; ++u) {
        v[u] = p[u] < e[u];
        any |= v[u];
      }
      if (__ballot(any) == 0ull) break;
#pragma unroll
      for (int u = 0; u < THIN_UL; ++u) {
        k[u] = v[u] ? a.at_inner[p[u]] : INT_MIN;
        av[u] = v[u] ? a.at_val[p[u]] : 0.0;
      }
      double bv[THIN_UL][THIN_CW];
      bool in[THIN_UL][THIN_CW];
#pragma unroll
      for (int u = 0; u < THIN_UL; ++u) {
#pragma unroll
        for (int c = 0; c < THIN_CW; ++c) {
          in[u][c] = v[u] && live[c] && (unsigned)(k[u] - bf[c]) <= ext[c];
          bv[u][c] = in[u][c] ? bp[c][k[u]] : 0.0;
        }
      }
#pragma unroll
      for (int u = 0; u < THIN_UL; ++u) {
#pragma unroll
        for (int c = 0; c < THIN_CW; ++c)
          if (in[u][c]) acc[u][c] = __fma_rn(av[u], bv[u][c], acc[u][c]);
        p[u] += v[u] ? 1 : 0;
      }
    }
#pragma unroll
    for (int u = 0; u < THIN_UL; ++u) {
      const int rb = r0 + u * WAVE, i = rb + lane;
#pragma unroll
      for (int c = 0; c < THIN_CW; ++c) {
        const double sv = __dmul_rn(a.alpha, acc[u][c]);
        const bool keep = live[c] && i < lo + w && (dense_rule ? (fabs(acc[u][c]) > a.threshold) : (fabs(sv) > a.threshold));
        const unsigned long long m = __ballot(keep);
        if (m) {
          thin_store(a.out_val + slot[c] - lo, gap[c], rb, min(rb + WAVE, lo + w), lane, keep ? sv : 0.0);
          gap[c] = rb + WAVE;
          cnt[c] += __popcll(m);
          f[c] = min(f[c], rb + (int)__builtin_ctzll(m));
          l[c] = rb + 63 - (int)__builtin_clzll(m);
        }
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < THIN_CW; ++c)
      if (live[c]) thin_finish(a, j0 + c, slot[c], lo, cnt[c], f[c], l[c]);
  }
}

// RIGHT operand thin: the wave lists the non-zeros (k, b) of the run of column j of B, then lane = row i adds
// A(i, k) * b over the list in ascending k from the runs of A
__global__ __launch_bounds__(THIN_NW* WAVE) void k_thin_slab_right(const ThinSlabArgs a) {
  __shared__ int s_k[THIN_NW][THIN_MAXK], s_af[THIN_NW][THIN_MAXK], s_al[THIN_NW][THIN_MAXK];
  __shared__ double s_b[THIN_NW][THIN_MAXK];
  __shared__ long long s_ao[THIN_NW][THIN_MAXK];
  const int nb4 = (a.ncols + THIN_NW - 1) / THIN_NW;
  const int blk = xcd_block(nb4);
  if (blk < 0) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int j = blk * THIN_NW + wave;
  if (j >= a.ncols) return;
  const int b = j >> 4, jj = j & 15;
  const int lo = a.blk_lo[b], w = a.blk_w[b];
  const int64_t slot = a.blk_toff[b] + (int64_t)jj * w;
  const int bf = a.bfirst[j], bl = a.blast[j];
  if (bl < bf || w <= 0) {
    if (lane == 0) thin_finish(a, j, slot, lo, 0, INT_MAX, -1);
    return;
  }
  const double* __restrict__ bp = a.bval + (a.boff[j] - bf);
  int nk = 0;
  for (int k0 = bf; k0 <= bl; k0 += WAVE) {
    const int k = k0 + lane;
    const double v = k <= bl ? bp[k] : 0.0;
    const unsigned long long m = __ballot(v != 0.0);
    const int q = nk + (int)__popcll(m & lanemask_lt());
    if (v != 0.0 && q < THIN_MAXK) {
      s_k[wave][q] = k;
      s_b[wave][q] = v;
      s_af[wave][q] = a.afirst[k];
      s_al[wave][q] = a.alast[k];
      s_ao[wave][q] = a.aoff[k];
    }
    nk += (int)__popcll(m);
  }
  if (nk > THIN_MAXK) {   // not this kernel's column: the host repeats the product on the tile kernel
    if (lane == 0) atomicOr(a.flag, 1);
    return;
  }
  __builtin_amdgcn_wave_barrier();
  const bool dense_rule = (a.dense_rule & 1) != 0;
  int cnt = 0, f = INT_MAX, l = -1, gap = INT_MAX;
  for (int r0 = lo; r0 < lo + w; r0 += THIN_U * WAVE) {
    double acc[THIN_U];
#pragma unroll
    for (int u = 0; u < THIN_U; ++u) acc[u] = 0.0;
    for (int t = 0; t < nk; ++t) {
      const int af = s_af[wave][t], al = s_al[wave][t];
      const double bt = s_b[wave][t];
      const double* __restrict__ ap = a.aval + (s_ao[wave][t] - af);
      double av[THIN_U];
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) {
        const int i = r0 + u * WAVE + lane;
        av[u] = (i >= af && i <= al) ? ap[i] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < THIN_U; ++u) {
        const int i = r0 + u * WAVE + lane;
        if (i >= af && i <= al) acc[u] = __fma_rn(av[u], bt, acc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < THIN_U; ++u) {
      const int rb = r0 + u * WAVE, i = rb + lane;
      const double sv = __dmul_rn(a.alpha, acc[u]);
      const bool keep = i < lo + w && (dense_rule ? (fabs(acc[u]) > a.threshold) : (fabs(sv) > a.threshold));
      const unsigned long long m = __ballot(keep);
      if (m) {
        thin_store(a.out_val + slot - lo, gap, rb, min(rb + WAVE, lo + w), lane, keep ? sv : 0.0);
        gap = rb + WAVE;
        cnt += __popcll(m);
        f = min(f, rb + (int)__builtin_ctzll(m));
        l = rb + 63 - (int)__builtin_clzll(m);
      }
    }
  }
  if (lane == 0) thin_finish(a, j, slot, lo, cnt, f, l);
}

// rows of a thin left operand as columns, kept per matrix (spgemm_thin_left); released with the other operand caches
struct KeptTranspose {
  const void* val = nullptr;
  unsigned long long serial = 0, epoch = 0, used = 0;
  int64_t nnz = -1;
  std::shared_ptr<DevMat> at;
};
KeptTranspose* kept_transposes() {
  static KeptTranspose kept[2];
  return kept;
}
}  // namespace

void drop_thin_transposes() {
  KeptTranspose* const kept = kept_transposes();
  for (int ki = 0; ki < 2; ++ki) kept[ki] = KeptTranspose();
}

void launch_thin_slab(const ThinSlabArgs& a, bool left) {
  const int nb4 = cdiv(a.ncols, THIN_NW);
  if (left) hipLaunchKernelGGL(k_thin_slab_left, dim3(xcd_grid(cdiv(cdiv(a.ncols, THIN_CW), THIN_NW))), dim3(THIN_NW * WAVE), 0, stream(), a);
  else hipLaunchKernelGGL(k_thin_slab_right, dim3(xcd_grid(nb4)), dim3(THIN_NW * WAVE), 0, stream(), a);
}

// false: not taken (C untouched)
bool spgemm_thin_left(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, int dense_rule_bits, int64_t* products,
                      hipEvent_t ev_begin, hipEvent_t ev_end) {
  if (A.loose() || B.loose() || A.expanded() || B.expanded() || A.blocked() || B.blocked() || A.cplx != B.cplx) return false;
  if (A.cols != B.rows || A.nnz == 0 || B.nnz == 0) return false;
  const int32_t m = A.rows, n = B.cols;
  const Csc Av = view(A), Bvw = view(B);
  DevBuf<int32_t> bfirst((size_t)n), bext((size_t)n), count((size_t)n);
  DevBuf<unsigned long long> st(4);
  st.zero();
  const int nt = std::max(A.cols, n);
  hipLaunchKernelGGL(k_thin_extents, dim3(cdiv(nt, 256)), dim3(256), 0, stream(), Av, Bvw, bfirst.p, bext.p, st.p);
  unsigned long long hst[4] = {0, 0, 0, 0};
  {
    ScalarFetch f;
    f.add(st.p, 4, hst);
    f.run();
  }
  const int up = (int)hst[0], dn = (int)hst[1], wmax = (int)hst[2];
  const size_t esz = A.cplx ? 16 : 8;
  // the candidate rows of a column are its extent in B widened by the reach of A around its diagonal: a far entry of A
  // (a permuted operand) makes every column visit rows it has nothing in -- not this kernel's operand
  if ((int64_t)up + dn > 2048 || wmax <= 0 || (size_t)wmax * esz * THIN_NW > 64 * 1024) return false;
  if ((double)(up + dn) > 2.0 * (double)wmax + 64.0) return false;
  // rows of A as columns, ascending k inside each.  A solver loop multiplies by the same thin factor again and again (an
  // identity, a preconditioner): its transpose is kept per matrix (value buffer, allocation serial, value epoch), two slots
  KeptTranspose* const kept = kept_transposes();
  static unsigned long long clock = 0;
  const unsigned long long ser = dev_alloc_serial(A.val.p), ep = matrix_value_epoch();
  std::shared_ptr<DevMat> pAT;
  for (int ki = 0; ki < 2; ++ki) {
    KeptTranspose& k = kept[ki];
    if (k.at && k.val == A.val.p && k.serial == ser && ser != 0 && k.epoch == ep && k.nnz == A.nnz && k.at->rows == A.cols && k.at->cols == A.rows) {
      pAT = k.at;
      k.used = ++clock;
    }
  }
  if (!pAT) {
    pAT.reset(new DevMat(transpose(A)));
    KeptTranspose* slot = kept[0].used <= kept[1].used ? &kept[0] : &kept[1];
    slot->val = A.val.p; slot->serial = ser; slot->epoch = ep; slot->nnz = A.nnz; slot->at = pAT; slot->used = ++clock;
  }
  const DevMat& AT = *pAT;
  const Csc ATv = view(AT);
  const int nblocks = cdiv(n, THIN_NW);
  const size_t lds = (size_t)wmax * esz * THIN_NW;
  DevBuf<unsigned long long> prod(1);
  if (products) prod.zero();
  DevMat R;   // (C may be one of the operands)
  R.rows = m;
  R.cols = n;
  R.cplx = A.cplx;
  R.outer.alloc((size_t)n + 1);
  if (ev_begin) HIP_CHECK(hipEventRecord(ev_begin, stream()));
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_spgemm_thin<T, false>), dim3(xcd_grid(nblocks)), dim3(THIN_NW * WAVE), lds, stream(), ATv, Bvw, bfirst.p, bext.p, up, dn,
                       wmax, count.p, (const int64_t*)nullptr, (int32_t*)nullptr, (T*)nullptr, alpha, threshold, dense_rule_bits, nblocks,
                       products ? prod.p : (unsigned long long*)nullptr);
  });
  scan_i32_async(count.p, R.outer.p, (int64_t)n);
  int64_t nnz = 0;
  unsigned long long hp = 0;
  {
    ScalarFetch f;
    f.add(R.outer.p + n, 1, &nnz);
    if (products) f.add(prod.p, 1, &hp);
    f.run();
  }
  if (products) *products = (int64_t)hp;
  R.nnz = nnz;
  R.inner.alloc((size_t)nnz + kIndexSlack);
  R.val.alloc(((size_t)nnz + kIndexSlack) * R.wval());
  dispatch_type(A.cplx, [&](auto tag) {
    using T = decltype(tag);
    hipLaunchKernelGGL((k_spgemm_thin<T, true>), dim3(xcd_grid(nblocks)), dim3(THIN_NW * WAVE), lds, stream(), ATv, Bvw, bfirst.p, bext.p, up, dn,
                       wmax, count.p, R.outer.p, R.inner.p, reinterpret_cast<T*>(R.val.p), alpha, threshold, dense_rule_bits, nblocks,
                       (unsigned long long*)nullptr);
  });
  if (ev_end) HIP_CHECK(hipEventRecord(ev_end, stream()));
  C = std::move(R);
  return true;
}

}  // namespace ntp
