// Device-side helpers shared by the kernel translation units (kernels.hip, spgemm_grouped.hip): scalar traits with the
// reference's unfused arithmetic, wave64 cross-lane helpers, the XCD-aware block index and the kernel-side matrix view.
// Everything sits in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "common.hpp"

namespace ntp {
namespace {

constexpr int WAVE = 64;
constexpr int NXCD = 8;

// ------------------------------------------------------------------ scalar traits
template <typename T>
struct Sc;
template <>
struct Sc<double> {
  static constexpr bool cplx = false;
  __device__ static inline double zero() { return 0.0; }
  __device__ static inline double mul(double a, double b) { return __dmul_rn(a, b); }
  __device__ static inline double add(double a, double b) { return __dadd_rn(a, b); }
  // acc + a * b: two roundings (the reference's default build) or one (option spgemm_fma: its FP-contracted build)
  __device__ static inline double fmadd(double a, double b, double acc, bool fma) {
    return fma ? __fma_rn(a, b, acc) : __dadd_rn(acc, __dmul_rn(a, b));
  }
  __device__ static inline double scale(double s, double v) { return __dmul_rn(s, v); }
  __device__ static inline double mag(double v) { return fabs(v); }
  __device__ static inline double conj(double v) { return v; }
  __device__ static inline double re(double v) { return v; }
  __device__ static inline bool is_zero(double v) { return v == 0.0; }
};
template <>
struct Sc<double2> {
  static constexpr bool cplx = true;
  __device__ static inline double2 zero() { return make_double2(0.0, 0.0); }
  // (a.x + i a.y)(b.x + i b.y), no contraction: matches gcc/flang on baseline x86-64
  __device__ static inline double2 mul(double2 a, double2 b) {
    return make_double2(__dsub_rn(__dmul_rn(a.x, b.x), __dmul_rn(a.y, b.y)),
                        __dadd_rn(__dmul_rn(a.x, b.y), __dmul_rn(a.y, b.x)));
  }
  __device__ static inline double2 add(double2 a, double2 b) {
    return make_double2(__dadd_rn(a.x, b.x), __dadd_rn(a.y, b.y));
  }
  __device__ static inline double2 fmadd(double2 a, double2 b, double2 acc, bool) { return add(acc, mul(a, b)); }   // (complex: always unfused)
  __device__ static inline double2 scale(double s, double2 v) {
    return make_double2(__dmul_rn(s, v.x), __dmul_rn(s, v.y));
  }
  __device__ static inline double mag(double2 v) { return hypot(v.x, v.y); }
  __device__ static inline double2 conj(double2 v) { return make_double2(v.x, -v.y); }
  __device__ static inline double re(double2 v) { return v.x; }
  __device__ static inline bool is_zero(double2 v) { return v.x == 0.0 && v.y == 0.0; }
};

// double-double accumulation (error-free two-sums): (hi, lo) += (xh, xl).  A sum accumulated this way and rounded once at the end
// is the correctly rounded sum for ANY order of its terms, up to an error of ~1e-32 relative before that rounding -- used where a
// result must not depend on the order entries are stored in (the Gershgorin radius that starts a solve: a solve on a relabelled
// copy of a matrix, or on its slab form, has to start from the same spectral bounds as the solve on the caller's columns)
__device__ inline void dd_add(double& hi, double& lo, double xh, double xl) {
  const double s = __dadd_rn(hi, xh);
  const double bb = __dsub_rn(s, hi);
  double e = __dadd_rn(__dsub_rn(hi, __dsub_rn(s, bb)), __dsub_rn(xh, bb));
  e = __dadd_rn(e, __dadd_rn(lo, xl));
  hi = __dadd_rn(s, e);
  lo = __dsub_rn(e, __dsub_rn(hi, s));
}
// the sum of (hi, lo) over the lanes of a wave, every lane gets it
__device__ inline void dd_wave_sum(double& hi, double& lo) {
  for (int o = 32; o > 0; o >>= 1) {
    const double oh = __shfl_xor(hi, o, 64), ol = __shfl_xor(lo, o, 64);
    dd_add(hi, lo, oh, ol);
  }
}

// ------------------------------------------------------------------ wave helpers
__device__ inline int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ inline int readlane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline int64_t readlane_i64(int64_t v, int l) {
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ inline double readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
__device__ inline double readlane_T(double v, int l) { return readlane_f64(v, l); }
__device__ inline double2 readlane_T(double2 v, int l) {
  return make_double2(readlane_f64(v.x, l), readlane_f64(v.y, l));
}
__device__ inline int uni_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline int64_t uni_i64(int64_t v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (uint32_t)lo;
}
__device__ inline int wave_min_i32(int v) {
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ inline int wave_max_i32(int v) {
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ inline int64_t wave_sum_i64(int64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
// fixed-shape butterfly: deterministic for a given input
__device__ inline double wave_sum_f64(double v) {
  for (int o = 32; o > 0; o >>= 1) v = __dadd_rn(v, __shfl_xor(v, o, WAVE));
  return v;
}
__device__ inline unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// XCD-aware block index: hardware deals consecutive block ids round-robin over the 8 XCDs
// (MI355X_MICROARCH.md, "Workgroup dispatch"), each XCD has its own 4 MiB L2.  Give every XCD a
// contiguous range of columns so blocks that run together on one XCD read neighbouring operand
// columns (banded operands re-use the same A columns across ~2h consecutive output columns).
// grid must be launched with NXCD*ceil(nblocks/NXCD) blocks; returns -1 for padding blocks.
__device__ inline int xcd_block(int nblocks) {
  const int per = (nblocks + NXCD - 1) / NXCD;
  const int b = (int)(blockIdx.x % NXCD) * per + (int)(blockIdx.x / NXCD);
  return (b < nblocks && (int)(blockIdx.x / NXCD) < per) ? b : -1;
}
inline int xcd_grid(int nblocks) { return NXCD * ((nblocks + NXCD - 1) / NXCD); }

struct Csc {
  int32_t rows, cols;
  const int64_t* outer;
  const int32_t* inner;
  const void* val;
  const int32_t* cnt = nullptr;  // "loose" columns (a product left in its upper-bound slots): column j holds the
                                 // entries outer[j] .. outer[j] + cnt[j]; nullptr = packed (ends at outer[j + 1])
};
inline Csc view(const DevMat& m) {
  if (m.loose() || m.expanded() || m.blocked()) NTP_FATAL("internal: a loose matrix reached a kernel that reads packed columns (pack() it first)");
  return Csc{m.rows, m.cols, m.outer.p, m.inner.p, m.val.p};
}
inline Csc lview(const DevMat& m) {  // for the kernels that end a column at col_end()
  if (m.expanded() || m.blocked()) NTP_FATAL("internal: a matrix in slab / block form reached a kernel that reads compressed columns (pack() it first)");
  Csc v{m.rows, m.cols, m.outer.p, m.inner.p, m.val.p};
  v.cnt = m.cnt.p;
  return v;
}
__device__ inline int64_t col_end(const Csc& M, int j) { return M.cnt ? M.outer[j] + M.cnt[j] : M.outer[j + 1]; }

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

template <typename F>
void dispatch_type(bool cplx, F&& f) {
  if (cplx) f(double2{});
  else f(double{});
}

}  // namespace
}  // namespace ntp
