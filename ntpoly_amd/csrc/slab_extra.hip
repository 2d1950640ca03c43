// Small companions of the slab algebra (kernels.hip, last section).
// (1) Statistics that are only computed when the kernel timers are on: the intermediate products of C = A B for
//     operands in slab form -- sum over the entries B(k, j) of the entries of A(:, k) -- which the compressed-column
//     paths get from their plans (products per second).
// (2) IncrementMatrix(Identity, B, alpha) with threshold 0 on a slab-form B whose diagonal lies inside its runs: one
//     value per column changes, in place, instead of a merge pass over the whole matrix (AddSparseVectors rules for the
//     one row both columns can share: both present -> alpha + b kept unless exactly zero; B has a hole there -> alpha).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstring>
#include <memory>

#include "device_util.hpp"
#include "kernels.hpp"

namespace ntp {
namespace {
__global__ __launch_bounds__(256) void k_sa_products(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                     const int64_t* __restrict__ off, const double* __restrict__ val,
                                                     const int32_t* __restrict__ acount, int acols, unsigned long long* __restrict__ out) {
  __shared__ long long red[4];
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  long long p = 0;
  if (j < n) {
    const int f = first[j], l = last[j];
    if (l >= f) {
      const double* __restrict__ v = val + (off[j] - f);
      for (int k = f + lane_id(); k <= l; k += WAVE)
        if (v[k] != 0.0 && k < acols) p += acount[k];
    }
  }
  p = wave_sum_i64(p);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = p;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(out, (unsigned long long)t);
  }
}
}  // namespace

namespace {
// pass 1 (apply = 0): what would happen, without touching anything -- st[0] |= 1: a diagonal outside its column's run
// (the run would have to grow), |= 2: a diagonal that cancels at the end of its run (the run would have to shrink);
// st[1] += change of the entry count.  pass 2 (apply = 1): the values and the per-column counts.
__global__ __launch_bounds__(256) void k_sa_add_diagonal(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                         const int64_t* __restrict__ off, double* __restrict__ val,
                                                         int32_t* __restrict__ count, int col_offset, double alpha, int apply,
                                                         unsigned long long* __restrict__ st) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  int flag = 0, delta = 0;
  if (j < n) {
    const int d = j + col_offset, f = first[j], l = last[j];
    if (l < f || d < f || d > l) {
      flag = 1;
    } else {
      double* p = val + (off[j] + (d - f));
      const double old = *p;
      const bool hb = old != 0.0;
      const double s = hb ? __dadd_rn(alpha, old) : alpha;
      const bool keep = fabs(s) > 0.0;
      if (!keep && (d == f || d == l)) flag = 2;
      delta = (keep ? 1 : 0) - (hb ? 1 : 0);
      if (apply) {
        *p = keep ? s : 0.0;
        count[j] += delta;
      }
    }
  }
  if (!apply) {
    const unsigned long long any = __ballot(flag != 0);
    if (any) {
      int fl = flag;
      for (int o = 32; o > 0; o >>= 1) fl |= __shfl_xor(fl, o, WAVE);
      if (lane_id() == 0) atomicOr(st, (unsigned long long)fl);
    }
    const long long dsum = wave_sum_i64(delta);
    if (lane_id() == 0 && dsum) atomicAdd(st + 1, (unsigned long long)dsum);
  }
}
}  // namespace

// B <- B + alpha I (IncrementMatrix(Identity, B, alpha, 0)); false: not done (B untouched) -- the caller merges
bool slab_add_diagonal(DevMat& B, double alpha, int32_t col_offset) {
  if (!B.expanded() || B.cplx || (B.rows != B.cols && !slab_panels_ok()) || B.slab->labelled() || B.slab->origin || B.zero_free != 1 || alpha == 0.0) return false;
  SlabForm& f = *B.slab;
  const int n = B.cols;
  DevBuf<unsigned long long> st(2);
  st.zero();
  hipLaunchKernelGGL(k_sa_add_diagonal, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, f.val.p, f.count.p,
                     col_offset, alpha, 0, st.p);
  unsigned long long h[2] = {0, 0};
  {
    ScalarFetch ft;
    ft.add(st.p, 2, h);
    ft.run();
  }
  if (h[0] != 0) return false;
  hipLaunchKernelGGL(k_sa_add_diagonal, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, f.val.p, f.count.p,
                     col_offset, alpha, 1, st.p);
  B.nnz += (long long)h[1];
  f.tiles.release();      // (the multiplier tiles held the old diagonal; the next step's plan depends on the extents only)
  f.tile_off.release();
  return true;
}

// ------------------------------------------------------------------ TRS4's polynomial chain in two passes
// DensityMatrixSolversModule.F90:590-627 builds, with IncrementMatrix at threshold 0,
//   Fx = 4 X - 3 X2,   Gx = (I - 2 X) + X2,   trace_fx = dot(X2, Fx),   trace_gx = dot(X2, Gx),   P = Fx + sigma Gx
// element by element: fx = 4 x + (-3 x2), gx = x2 + ((-2 x) + d) (d = 1 on the diagonal), p = fx + sigma gx, every
// operation rounded on its own and an absent entry entering as zero (adding an exact zero changes nothing, so the values
// are those of the sequence of merges; an entry of the result is where the value is not zero).  Pass 1 reads X and X2
// and leaves the two dots; pass 2 reads them again and writes P -- instead of four merges and two dots over
// materialised Fx and Gx.
namespace {
struct Trs4Elem {
  double fx, gx;
};
__device__ inline Trs4Elem trs4_elem(double x, double x2, double d) {
  Trs4Elem e;
  e.fx = __dadd_rn(__dmul_rn(4.0, x), __dmul_rn(-3.0, x2));
  e.gx = __dadd_rn(x2, __dadd_rn(__dmul_rn(-2.0, x), d));
  return e;
}
// MODE 0: part[2 j] = sum x2 fx, part[2 j + 1] = sum x2 gx of column j (summed by k_sa_sum_pairs).  MODE 1: out = fx + sigma gx into the slot at
// base[j] (aligned union of the runs and the diagonal), kept count / first / last per column.
template <int MODE>
__global__ __launch_bounds__(256) void k_sa_trs4(int n, const int32_t* __restrict__ fa, const int32_t* __restrict__ la,
                                                 const int64_t* __restrict__ offa, const double* __restrict__ va,
                                                 const int32_t* __restrict__ fb, const int32_t* __restrict__ lb,
                                                 const int64_t* __restrict__ offb, const double* __restrict__ vb, int col_offset,
                                                 double sigma, int al, const int64_t* __restrict__ base, double* __restrict__ part,
                                                 double* __restrict__ out, int32_t* __restrict__ ofirst, int32_t* __restrict__ olast,
                                                 int32_t* __restrict__ ocount, int64_t* __restrict__ ooff, int64_t bound,
                                                 unsigned long long* __restrict__ stat) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int fA = fa[j], lA = la[j], fB = fb[j], lB = lb[j], dg = j + col_offset;
  const bool anyA = lA >= fA, anyB = lB >= fB;
  int f = dg, l = dg;   // (the identity's entry is always there)
  if (anyA) { f = min(f, fA); l = max(l, lA); }
  if (anyB) { f = min(f, fB); l = max(l, lB); }
  const int a0 = f / al * al, a1 = (l / al + 1) * al;
  const double* __restrict__ pa = anyA ? va + (offa[j] - fA) : va;
  const double* __restrict__ pb = anyB ? vb + (offb[j] - fB) : vb;
  if (MODE == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int r = a0 + lane; r < a1; r += WAVE) {
      const double x = (anyA && r >= fA && r <= lA) ? pa[r] : 0.0;
      const double x2 = (anyB && r >= fB && r <= lB) ? pb[r] : 0.0;
      const Trs4Elem e = trs4_elem(x, x2, r == dg ? 1.0 : 0.0);
      s0 = __dadd_rn(s0, __dmul_rn(x2, e.fx));
      s1 = __dadd_rn(s1, __dmul_rn(x2, e.gx));
    }
    s0 = wave_sum_f64(s0);
    s1 = wave_sum_f64(s1);
    if (lane == 0) { part[2 * (size_t)j] = s0; part[2 * (size_t)j + 1] = s1; }
  } else {
    const int64_t slot = base[j];
    if (slot + (int64_t)(a1 - a0) > bound) {   // (runs far apart: the union extent does not fit the output -- refused by the host)
      if (lane == 0) { ofirst[j] = INT_MAX; olast[j] = -1; ocount[j] = 0; ooff[j] = slot; atomicOr(stat, 2ull); }
      return;
    }
    double* __restrict__ dst = out + (slot - a0);
    int cnt = 0, kf = INT_MAX, kl = -1;
    for (int r = a0 + lane; r < a1; r += WAVE) {
      const double x = (anyA && r >= fA && r <= lA) ? pa[r] : 0.0;
      const double x2 = (anyB && r >= fB && r <= lB) ? pb[r] : 0.0;
      const Trs4Elem e = trs4_elem(x, x2, r == dg ? 1.0 : 0.0);
      const double p = __dadd_rn(e.fx, __dmul_rn(sigma, e.gx));
      const bool keep = p != 0.0;
      dst[r] = keep ? p : 0.0;
      cnt += keep ? 1 : 0;
      kf = min(kf, keep ? r : INT_MAX);
      kl = max(kl, keep ? r : -1);
    }
    cnt = (int)wave_sum_i64(cnt);
    kf = wave_min_i32(kf);
    kl = wave_max_i32(kl);
    if (lane == 0) {
      ofirst[j] = kf; olast[j] = kl; ocount[j] = cnt;
      ooff[j] = slot + (cnt ? kf - a0 : 0);
    }
  }
}
__global__ void k_sa_trs4_span(const int32_t* __restrict__ fa, const int32_t* __restrict__ la, const int32_t* __restrict__ fb,
                               const int32_t* __restrict__ lb, int n, int col_offset, int al, int32_t* __restrict__ span) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  int f = j + col_offset, l = j + col_offset;
  if (la[j] >= fa[j]) { f = min(f, fa[j]); l = max(l, la[j]); }
  if (lb[j] >= fb[j]) { f = min(f, fb[j]); l = max(l, lb[j]); }
  span[j] = (l / al + 1) * al - f / al * al;
}
__global__ __launch_bounds__(256) void k_sa_count_sum(const int32_t* __restrict__ v, int n, unsigned long long* __restrict__ out) {
  __shared__ long long red[4];
  long long s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += v[i];
  s = wave_sum_i64(s);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(out, (unsigned long long)t);
  }
}
// deterministic sum of n (a, b) pairs: fixed assignment of elements to threads, fixed tree -- the same bits every run
__global__ __launch_bounds__(256) void k_sa_sum_pairs(const double* __restrict__ in, int n, int chunk, double* __restrict__ out) {
  __shared__ double ra[256], rb[256];
  const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
  double a = 0.0, b = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += 256) {
    a = __dadd_rn(a, in[2 * (size_t)i]);
    b = __dadd_rn(b, in[2 * (size_t)i + 1]);
  }
  ra[threadIdx.x] = a;
  rb[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ra[threadIdx.x] = __dadd_rn(ra[threadIdx.x], ra[threadIdx.x + o]);
      rb[threadIdx.x] = __dadd_rn(rb[threadIdx.x], rb[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = ra[0]; out[2 * blockIdx.x + 1] = rb[0]; }
}
void sum_pairs_async(const double* part, int n, double* out2) {
  const int chunk = 2048, g = cdiv(n, chunk);
  if (g <= 1) {
    hipLaunchKernelGGL(k_sa_sum_pairs, dim3(1), dim3(256), 0, stream(), part, n, std::max(n, 1), out2);
    return;
  }
  DevBuf<double> lvl((size_t)2 * g);
  hipLaunchKernelGGL(k_sa_sum_pairs, dim3(g), dim3(256), 0, stream(), part, n, chunk, lvl.p);
  hipLaunchKernelGGL(k_sa_sum_pairs, dim3(1), dim3(256), 0, stream(), lvl.p, g, g, out2);
}
bool trs4_operands(const DevMat& X, const DevMat& X2) {
  auto ok = [](const DevMat& M) {
    return M.expanded() && !M.cplx && (M.rows == M.cols || slab_panels_ok()) && !M.slab->labelled() && !M.slab->origin && M.zero_free == 1;
  };
  return ok(X) && ok(X2) && X.cols == X2.cols && X.slab->row_pad == X2.slab->row_pad;
}
}  // namespace

bool slab_trs4_traces(const DevMat& X, const DevMat& X2, int32_t col_offset, double* trace_fx, double* trace_gx) {
  if (!trs4_operands(X, X2)) return false;
  const SlabForm &fa = *X.slab, &fb = *X2.slab;
  const int n = X.cols;
  DevBuf<double> part((size_t)2 * n), res(2);
  hipLaunchKernelGGL((k_sa_trs4<0>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p, fa.val.p,
                     fb.first.p, fb.last.p, fb.off.p, fb.val.p, col_offset, 0.0, std::max(1, fa.row_pad), (const int64_t*)nullptr, part.p,
                     (double*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                     (unsigned long long*)nullptr);
  sum_pairs_async(part.p, n, res.p);
  unsigned long long h[2] = {0, 0};
  ScalarFetch ft;
  ft.add(res.p, 2, h);
  ft.run();
  double d[2];
  std::memcpy(d, h, sizeof(h));
  *trace_fx = d[0];
  *trace_gx = d[1];
  return true;
}

bool slab_trs4_operand(const DevMat& X, const DevMat& X2, double sigma, int32_t col_offset, DevMat& Out) {
  if (!trs4_operands(X, X2)) return false;
  const SlabForm &fa = *X.slab, &fb = *X2.slab;
  const int n = X.cols, al = std::max(1, fa.row_pad);
  std::unique_ptr<SlabForm> fo(new SlabForm());
  fo->first.alloc((size_t)n); fo->last.alloc((size_t)n); fo->count.alloc((size_t)n); fo->off.alloc((size_t)n + 1);
  DevBuf<int32_t> span((size_t)n);
  DevBuf<int64_t> base((size_t)n + 1);
  hipLaunchKernelGGL(k_sa_trs4_span, dim3(cdiv(n, 256)), dim3(256), 0, stream(), fa.first.p, fa.last.p, fb.first.p, fb.last.p, n, col_offset,
                     al, span.p);
  scan_i32_async(span.p, base.p, (int64_t)n);
  const int64_t bound = fa.slots + fb.slots + 4LL * al * n;
  fo->val.alloc((size_t)bound + kIndexSlack);
  DevBuf<unsigned long long> stat(1);
  stat.zero();
  hipLaunchKernelGGL((k_sa_trs4<1>), dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p, fa.val.p,
                     fb.first.p, fb.last.p, fb.off.p, fb.val.p, col_offset, sigma, al, base.p, (double*)nullptr, fo->val.p, fo->first.p,
                     fo->last.p, fo->count.p, fo->off.p, bound, stat.p);
  DevBuf<unsigned long long> tot(1);
  tot.zero();
  hipLaunchKernelGGL(k_sa_count_sum, dim3(std::max(1, std::min(256, cdiv(n, 1024)))), dim3(256), 0, stream(), fo->count.p, n, tot.p);
  int64_t nnz = 0, slots = 0;
  unsigned long long hs = 0;
  {
    ScalarFetch ft;
    ft.add(tot.p, 1, &nnz);
    ft.add(base.p + n, 1, &slots);
    ft.add(stat.p, 1, &hs);
    ft.run();
  }
  if (hs != 0) return false;   // (a union extent beyond the output buffer: runs far apart -- the caller takes compressed columns)
  fo->row_pad = al;
  fo->slots = slots;
  DevMat R;
  R.rows = X.rows; R.cols = n; R.cplx = false; R.nnz = nnz; R.zero_free = 1;
  R.slab = std::move(fo);
  Out = std::move(R);
  return true;
}

// ------------------------------------------------------------------ MatrixNorm(alpha A + beta B) without the sum
// The loops of SignFunction, Invert and the square roots build a difference (Out - Temp2, I - Temp1, I - X) only to
// take its norm (max column abs-sum) for the convergence test.  One pass over the two runs of every column: the
// element is (alpha a) + (beta b) as the merge would compute it, an entry the merge drops (an exact zero) adds nothing.
namespace {
__global__ __launch_bounds__(256) void k_sa_norm_axpby(int n, const int32_t* __restrict__ fa, const int32_t* __restrict__ la,
                                                       const int64_t* __restrict__ offa, const double* __restrict__ va,
                                                       const int32_t* __restrict__ fb, const int32_t* __restrict__ lb,
                                                       const int64_t* __restrict__ offb, const double* __restrict__ vb, double alpha,
                                                       double beta, double* __restrict__ colsum) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int fA = fa[j], lA = la[j], fB = fb[j], lB = lb[j];
  const bool anyA = lA >= fA, anyB = lB >= fB;
  double s = 0.0;
  if (anyA || anyB) {
    const int f = anyA ? (anyB ? min(fA, fB) : fA) : fB, l = anyA ? (anyB ? max(lA, lB) : lA) : lB;
    const double* __restrict__ pa = anyA ? va + (offa[j] - fA) : va;
    const double* __restrict__ pb = anyB ? vb + (offb[j] - fB) : vb;
    for (int r = f + lane; r <= l; r += WAVE) {
      const double a = (anyA && r >= fA && r <= lA) ? pa[r] : 0.0;
      const double b = (anyB && r >= fB && r <= lB) ? pb[r] : 0.0;
      s = __dadd_rn(s, fabs(__dadd_rn(__dmul_rn(alpha, a), __dmul_rn(beta, b))));
    }
  }
  s = wave_sum_f64(s);
  if (lane == 0) colsum[j] = s;
}
}  // namespace

bool slab_norm_axpby(const DevMat& A, const DevMat& B, double alpha, double beta, double* out) {
  auto ok = [](const DevMat& M) { return M.expanded() && !M.cplx && (M.rows == M.cols || slab_panels_ok()) && !M.slab->labelled(); };
  if (!ok(A) || !ok(B) || A.cols != B.cols) return false;
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  const int n = A.cols;
  DevBuf<double> cs((size_t)n);
  hipLaunchKernelGGL(k_sa_norm_axpby, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p, fa.val.p,
                     fb.first.p, fb.last.p, fb.off.p, fb.val.p, alpha, beta, cs.p);
  *out = max_of(cs, (size_t)n);
  return true;
}

// ------------------------------------------------------------------ complex operands in slab form (kernels.hpp slab_enter_c)
namespace {
// column j: the diagonal entry (global row col_offset + j) must be stored; newval[j] = alpha + it (AddSparseVectors: alpha * 1
// rounded, then added); flag bit 0: no diagonal entry, bit 1: the sum is zero (the merge would drop it)
__global__ void k_sa_diag_c(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last, const int64_t* __restrict__ off,
                            double2* __restrict__ val, int col_offset, double alpha, int apply, double2* __restrict__ newval, int* __restrict__ flag) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int d = j + col_offset, f = first[j], l = last[j];
  if (apply) {
    val[off[j] + (d - f)] = newval[j];
    return;
  }
  int fl = 0;
  if (l < f || d < f || d > l) {
    fl = 1;
  } else {
    const double2 old = val[off[j] + (d - f)];
    if (old.x == 0.0 && old.y == 0.0) {
      fl = 1;
    } else {
      const double2 one = make_double2(1.0, 0.0);
      const double2 nv = Sc<double2>::add(Sc<double2>::scale(alpha, one), old);
      newval[j] = nv;
      if (nv.x == 0.0 && nv.y == 0.0) fl = 2;
    }
  }
  if (fl) atomicOr(flag, fl);
}
__global__ __launch_bounds__(256) void k_sa_norm_axpby_c(int n, const int32_t* __restrict__ fa, const int32_t* __restrict__ la,
                                                         const int64_t* __restrict__ offa, const double2* __restrict__ va,
                                                         const int32_t* __restrict__ fb, const int32_t* __restrict__ lb,
                                                         const int64_t* __restrict__ offb, const double2* __restrict__ vb, double alpha,
                                                         double beta, double* __restrict__ colsum) {
  const int j = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  if (j >= n) return;
  const int lane = lane_id();
  const int fA = fa[j], lA = la[j], fB = fb[j], lB = lb[j];
  const bool anyA = lA >= fA, anyB = lB >= fB;
  double s = 0.0;
  if (anyA || anyB) {
    const int f = anyA ? (anyB ? min(fA, fB) : fA) : fB, l = anyA ? (anyB ? max(lA, lB) : lA) : lB;
    const double2* __restrict__ pa = anyA ? va + (offa[j] - fA) : va;
    const double2* __restrict__ pb = anyB ? vb + (offb[j] - fB) : vb;
    for (int r = f + lane; r <= l; r += WAVE) {
      const double2 a = (anyA && r >= fA && r <= lA) ? pa[r] : make_double2(0.0, 0.0);
      const double2 b = (anyB && r >= fB && r <= lB) ? pb[r] : make_double2(0.0, 0.0);
      const double2 v = Sc<double2>::add(Sc<double2>::scale(alpha, a), Sc<double2>::scale(beta, b));
      s = __dadd_rn(s, Sc<double2>::mag(v));
    }
  }
  s = wave_sum_f64(s);
  if (lane == 0) colsum[j] = s;
}
}  // namespace

// B <- B + alpha I on a complex slab-form matrix, in place; false: a column without a stored diagonal entry or a zero sum
// (B untouched: the caller packs and merges)
bool slab_add_diagonal_c(DevMat& B, double alpha, int32_t col_offset) {
  if (!sa_operand_c(B) || B.zero_free != 1 || alpha == 0.0) return false;
  SlabForm& f = *B.slab;
  const int n = B.cols;
  DevBuf<double> newval((size_t)2 * n);
  DevBuf<int> flag(2);
  flag.zero();
  hipLaunchKernelGGL(k_sa_diag_c, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, reinterpret_cast<double2*>(f.val.p),
                     col_offset, alpha, 0, reinterpret_cast<double2*>(newval.p), flag.p);
  long long h = 0;
  {
    ScalarFetch ft;
    ft.add(flag.p, 1, &h);
    ft.run();
  }
  if ((int)(h & 0xffffffffll) != 0) return false;
  hipLaunchKernelGGL(k_sa_diag_c, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, reinterpret_cast<double2*>(f.val.p),
                     col_offset, alpha, 1, reinterpret_cast<double2*>(newval.p), flag.p);
  return true;
}

// MatrixNorm(alpha A + beta B) of two complex slab-form matrices, nothing built
bool slab_norm_axpby_c(const DevMat& A, const DevMat& B, double alpha, double beta, double* out) {
  if (!sa_operand_c(A) || !sa_operand_c(B) || A.cols != B.cols) return false;
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  const int n = A.cols;
  DevBuf<double> cs((size_t)n);
  hipLaunchKernelGGL(k_sa_norm_axpby_c, dim3(cdiv((int64_t)n * WAVE, 256)), dim3(256), 0, stream(), n, fa.first.p, fa.last.p, fa.off.p,
                     reinterpret_cast<const double2*>(fa.val.p), fb.first.p, fb.last.p, fb.off.p, reinterpret_cast<const double2*>(fb.val.p), alpha,
                     beta, cs.p);
  *out = max_of(cs, (size_t)n);
  return true;
}

// ------------------------------------------------------------------ MatrixTrace of a slab-form matrix
namespace {
__global__ __launch_bounds__(256) void k_sa_diag(int n, const int32_t* __restrict__ first, const int32_t* __restrict__ last,
                                                 const int64_t* __restrict__ off, const double* __restrict__ val, int col_offset,
                                                 double* __restrict__ part) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int d = j + col_offset, f = first[j], l = last[j];
  part[2 * (size_t)j] = (l >= f && d >= f && d <= l) ? val[off[j] + (d - f)] : 0.0;
  part[2 * (size_t)j + 1] = 0.0;
}
}  // namespace

bool slab_trace(const DevMat& A, int32_t col_offset, double* out) {
  if (!A.expanded() || A.cplx || A.slab->labelled()) return false;
  const SlabForm& f = *A.slab;
  const int n = A.cols;
  DevBuf<double> part((size_t)2 * n), res(2);
  hipLaunchKernelGGL(k_sa_diag, dim3(cdiv(n, 256)), dim3(256), 0, stream(), n, f.first.p, f.last.p, f.off.p, f.val.p, col_offset, part.p);
  sum_pairs_async(part.p, n, res.p);
  unsigned long long h[2] = {0, 0};
  ScalarFetch ft;
  ft.add(res.p, 2, h);
  ft.run();
  std::memcpy(out, &h[0], sizeof(double));
  return true;
}

// ------------------------------------------------------------------ how dense the runs are
namespace {
__global__ __launch_bounds__(256) void k_sa_span_sum(const int32_t* __restrict__ first, const int32_t* __restrict__ last, int n,
                                                     unsigned long long* __restrict__ out) {
  __shared__ long long red[4];
  long long s = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) s += last[i] >= first[i] ? last[i] - first[i] + 1 : 0;
  s = wave_sum_i64(s);
  if (lane_id() == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const long long t = red[0] + red[1] + red[2] + red[3];
    if (t) atomicAdd(out, (unsigned long long)t);
  }
}
}  // namespace

// rows covered by the runs of a slab-form matrix (kept in the form: its extents never change)
int64_t slab_span_sum(const DevMat& M) {
  if (!M.expanded()) return 0;
  const SlabForm& f = *M.slab;
  if (f.span_sum >= 0) return f.span_sum;
  DevBuf<unsigned long long> acc(1);
  acc.zero();
  hipLaunchKernelGGL(k_sa_span_sum, dim3(std::max(1, std::min(256, cdiv(M.cols, 1024)))), dim3(256), 0, stream(), f.first.p, f.last.p, M.cols,
                     acc.p);
  unsigned long long h = 0;
  ScalarFetch ft;
  ft.add(acc.p, 1, &h);
  ft.run();
  f.span_sum = (int64_t)h;
  return f.span_sum;
}

long long slab_product_count(const DevMat& A, const DevMat& B) {
  if (!A.expanded() || !B.expanded() || A.cplx || B.cplx) return 0;
  const SlabForm &fa = *A.slab, &fb = *B.slab;
  DevBuf<unsigned long long> acc(1);
  acc.zero();
  hipLaunchKernelGGL(k_sa_products, dim3(cdiv((int64_t)B.cols * WAVE, 256)), dim3(256), 0, stream(), B.cols, fb.first.p, fb.last.p,
                     fb.off.p, fb.val.p, fa.count.p, A.cols, acc.p);
  unsigned long long h = 0;
  ScalarFetch f;
  f.add(acc.p, 1, &h);
  f.run();
  return (long long)h;
}

}  // namespace ntp
