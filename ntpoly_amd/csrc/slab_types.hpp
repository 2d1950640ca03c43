// Types shared by the slab-form kernels (kernels.hip: register-slab SpGEMM and its fused purification epilogues;
// spgemm_tile.hip: the MFMA tile kernel on the same operands).  Anonymous namespace: one copy per translation unit.
#pragma once
#include <cstdint>

namespace ntp {
namespace {

constexpr int SLAB_J = 16, SLAB_SL = 3, SLAB_NW = 4;

// Run record of an expanded column of A, 32 bytes, fetched by ONE s_load_dwordx8: words 0-3 are the buffer
// descriptor of the run (base address, bytes, flags) used as-is by buffer_load; first8 = 8 * first row turns a
// row offset into a run offset; (first, span62 = rows + 62) give the one-compare test "does the run touch
// the slab that ends at row e": (unsigned)(e - first) <= span62.
struct alignas(32) SlabRun {
  uint32_t addr_lo, addr_hi, nbytes, flags;
  int32_t first8, first, span62, pad;
};
constexpr uint32_t kBufferFlags = 0x00020000u;  // raw buffer, 32-bit data format (gfx9 family word 3)

// Arguments of the fused purification epilogues (documented at k_spgemm_slab in kernels.hip).
struct SlabFuseArgs {
  double am = 0, bm = 0, thr_m = 0;
  const double* xexp = nullptr;      // expanded columns of X
  const int64_t* xoff = nullptr;
  const int32_t *xmin = nullptr, *xmax = nullptr;
  const double* dexp = nullptr;      // expanded columns of D
  const int64_t* doff = nullptr;
  const int32_t *dmin = nullptr, *dmax = nullptr;
  int32_t *ofirst = nullptr, *olast = nullptr;   // first / last row of every column of the result
  double* tiles = nullptr;           // the result as tiles (SlabForm::tiles), block b at blk_toff[b]
  double* part = nullptr;            // [2 * nblocks]: (dot, trace) of the block
  long long* pnnz = nullptr;         // [nblocks]: kept entries of the product
  const int32_t* in_count = nullptr; // statistics (operand in slab form): entries per column of X; with prod set, the
  long long* prod = nullptr;         // block counts its intermediate products from its multiplier tile before the loop
  // label-ordered steps (the data sits in a bandwidth-reducing order, the arithmetic follows the ORIGINAL labels
  // lab[index]): the k steps of a block come in ascending label -- per-block run records blkruns[rec_off(b) + t] and
  // the multiplier tile handed to the kernel are in that order, steps[...] names the column of step t -- and "beyond
  // the other column's last row" compares labels: xplast[j] = largest label in X(:, j), oplast[j] the result's
  const int32_t* lab = nullptr;
  const SlabRun* blkruns = nullptr;
  const int32_t* steps = nullptr;
  const int32_t* xplast = nullptr;
  int32_t* oplast = nullptr;
  int* flag = nullptr;
  int col_offset = 0;
};

}  // namespace
}  // namespace ntp
