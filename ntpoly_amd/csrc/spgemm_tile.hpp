// MFMA tile kernel for run-like real operands (spgemm_tile.hip): the same operands and results as the register-slab
// kernel of kernels.hip, computed tile by tile on the FP64 matrix cores.  Included by kernels.hip after slab_types.hpp.
#pragma once
#include "slab_types.hpp"

namespace ntp {

// One launch = one multiply of the local panel.  Operands exactly as the register-slab kernel takes them: run records
// of the expanded columns of A, the per-block multiplier tiles of B, the plan (row window, k range, output slots).
// Results (every EPI): the columns of the result as dense runs in their upper-bound slots -- column j of block b at
// out_val[blk_toff[b] + (j % 16) * blk_w[b] + (r - blk_lo[b])], zeros = no entry -- plus per column the number of kept
// entries, the first / last kept row and ooff[j] = where the run [first, last] starts.  EPI 1 / 2 (fused purification
// steps, SlabFuseArgs) also write the block as a row-major tile (fz.tiles, origin row blk_lo[b]; otoff[b] = where the
// rows [min first, max last] start) and the block's (dot, trace), product entries and product count.
struct TileLaunch {
  const void* runs = nullptr;        // SlabRun of column k of A at runs[k] (a pointer biased by the first column is fine)
  const double* bblk = nullptr;
  const int64_t* blk_boff = nullptr;
  const int32_t *blk_kmin = nullptr, *blk_kn = nullptr, *blk_lo = nullptr, *blk_w = nullptr;
  const int64_t* blk_toff = nullptr;
  double* out_val = nullptr;
  int32_t* count = nullptr;
  int32_t *ofirst = nullptr, *olast = nullptr;
  int64_t* ooff = nullptr;           // [ncols]
  int64_t* otoff = nullptr;          // [nblocks] (EPI != 0)
  double alpha = 1.0, threshold = 0.0;
  int dense_rule = 0, ncols = 0, nblocks = 0;
  int max_kn = 0, max_w = 0;         // largest k range / row window of any block (sizes the workgroup's LDS)
  int epi = 0;
  int rows = 1;                      // R = 1, 2 or 4 consecutive rows per lane of the A operand (tiles of 16 R rows).  R > 1 needs
                                     // (1) blk_lo and blk_w multiples of 16 R and (2) every expanded column of A (and of X, D in the
                                     // fused epilogues) readable and ZERO from the multiple of R below its first row to the one
                                     // above its last (tile_expand_align(): the expansions and this kernel's own results are)
  const void* fz = nullptr;          // HOST copy of SlabFuseArgs (EPI != 0): passed on by value with the kernel arguments
  // epi 0, optional: the right operand as the runs of its columns instead of multiplier tiles (bblk / blk_boff unused)
  const int32_t *brun_first = nullptr, *brun_last = nullptr;
  const int64_t* brun_off = nullptr;
  const double* brun_val = nullptr;
  size_t bbytes = 0;                 // optional: the allocation behind brun_val -- below 4 GB the multiplier tile is read through a buffer resource (option tile_bbuf)
  int brun_pad = 1;                  // SlabForm::row_pad of the operand behind brun_*: even -- the tile is read as pairs of rows (16-byte requests, a wave per group of columns)
  // optional: ONE allocation [abase, abase + abytes) that holds every run `runs` points into (no halo): below 4 GB the kernel
  // reads the runs through a buffer resource with 32-bit offsets (option tile_off32)
  const void* abase = nullptr;
  size_t abytes = 0;
  const void* dbase = nullptr;       // epi != 0: the allocation that holds the expanded D operand (fz.dexp), for the same purpose
  size_t dbytes = 0;
  int nrows = 0;                     // rows of the operands (0: ncols -- square); label-aware epilogues index fz.lab by row
  bool labelled = false;             // fz carries labels (SlabFuseArgs::lab, xplast, oplast): the epilogue's "beyond the last entry"
                                     // tests compare the caller's labels.  The k steps are walked in POSITION order (the rounding
                                     // of a product entry then differs from the label-ordered chain in its last bits: tolerance mode)
};
// false: the geometry does not fit (k range beyond the LDS tile); nothing was launched
bool spgemm_tile_fits(int max_kn, int max_w);
// rows per lane the engine uses (option tile_rows: 1, 2 or 4) and the alignment it implies for windows and expanded columns
int tile_rows();
inline int tile_expand_align() { return 16 * tile_rows(); }
void launch_spgemm_tile(const TileLaunch& a);
// The same multiply in the two-block geometry (spgemm_tile2.hip: pairs of column blocks share every fragment of A, the
// multiplier rows stream through LDS in chunks): rows == 2, the right operand given by its runs (brun_*), not labelled.
// false: not launched (the geometry cannot fit), call launch_spgemm_tile.  true: launched -- *fail (device, zeroed by the
// caller) is set when some pair of blocks did not fit after all (window beyond 1024 rows, or a wave's two slabs in
// progress together): NOTHING of the launch may be used then; the caller repeats it with launch_spgemm_tile.
bool launch_spgemm_tile2(const TileLaunch& a, int* fail);
// Complex run-like operands (spgemm_tile_c.hip): blocks of 8 complex columns, the operands of k_spgemm_slab_c (run records
// of 16-byte elements, interleaved multiplier tiles in bblk), windows that start at and are a multiple of 16 rows; epi 0
// only.  Results as above in complex slots: out_val[2 * slot + part], counts / first / last / ooff per column.
// A tolerance mode (two FMA chains per part instead of the reference's eight roundings per product).
bool spgemm_tile_c_fits(int max_kn, int max_w);
void launch_spgemm_tile_c(const TileLaunch& a);

}  // namespace ntp
