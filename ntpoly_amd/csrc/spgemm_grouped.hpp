// Grouped LDS-hash SpGEMM for operands without run structure (load-balanced / permuted / 3-D Hamiltonians):
// spgemm_grouped.hip.  Internal interface between spgemm() in kernels.hip and that translation unit.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace ntp {

// bin codes shared with kernels.hip (per output column): the grouped path marks the columns it has computed DONE and
// hands the others to the per-column LDS hash (BIN_HASH)
constexpr int GH_BIN_HASH = 5, GH_BIN_DONE = 8;

struct GroupedInfo {
  int64_t groups = 0;          // column groups formed
  int64_t failed_groups = 0;   // groups handed back to the per-column kernels (row union beyond the largest table)
  int64_t failed_cols = 0;
  int level = 0;               // largest table class used (0: 512 rows, 1: 1024, 2: 1536)
  int minhash = 0;             // 1: columns were clustered by their min-hash signature, 0: natural order
  double union_ratio = 0;      // sum over groups of |union of the B rows| / (nnz(B) / columns per group): 1 = identical columns
  int64_t tile_rows = 0;       // sum over groups of |union of the B rows| = steps of the numeric kernel
  int64_t products = 0;        // intermediate products of the multiply (counted by the tile builder)
};

// C = alpha * A * B for ALL columns, group by group.  tmpoff / tmp_inner / tmp_val / count are spgemm()'s upper-bound
// output slots.  Returns false (nothing written) when the columns of B show too little similarity for sharing to pay;
// otherwise bin_arr[j] is GH_BIN_DONE or GH_BIN_HASH on return.
bool spgemm_grouped(const DevMat& A, const DevMat& B, const int64_t* tmpoff, int32_t* tmp_inner, double* tmp_val,
                    int32_t* count, uint8_t* bin_arr, double alpha, double threshold, int dense_rule, int mode,
                    GroupedInfo* info, hipEvent_t numeric_begin = nullptr);   // recorded before the first numeric launch
// mode 0: as described; 1: forced (also with dissimilar columns); 2: only with the kept min-hash column order of the
// previous multiply of this dimension (returns false at once otherwise)

}  // namespace ntp
