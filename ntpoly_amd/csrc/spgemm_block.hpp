// Block-sparse SpGEMM on the FP64 matrix cores for operands WITHOUT run structure (spgemm_block.hip): 3-D Hamiltonians,
// bands hidden under a relabelling.  Internal interface between spgemm() in kernels.hip, psmatrix.cpp and that
// translation unit.  FMA arithmetic only (option spgemm_fma = 1), real square operands on one rank.
#pragma once
#include <memory>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace ntp {

// A clustering of the index set 0 .. n - 1 into blocks of <= 16 indices with similar rows / columns, found from the
// pattern and the magnitudes of one matrix (heavy-edge matching, five levels; then two more levels that pair blocks
// into SUPER-BLOCKS of <= 4 blocks, then as many as it takes to order the super-blocks along a nested, locality-
// preserving line).  Position p = 64 S + 16 b + o: super-block S, block b of it, index o of the block; positions
// without an index are padding (lab = -1).  The matrices of a dimension share one order.
struct BlockOrder {
  int32_t n = 0;
  int32_t ns = 0;              // super-blocks; 4 ns blocks, 64 ns positions
  DevBuf<int32_t> pos;         // [n]      pos[index] = position
  DevBuf<int32_t> lab;         // [64 ns]  lab[position] = index or -1
  int64_t built_from_nnz = 0;  // entries of the matrix whose pattern it was made from
  unsigned long long seed_fp = 0;   // fingerprint of that pattern (order-independent sum of per-entry hashes): the cache key beside n
  double seed_fill = 0;        // fill of that matrix in its own order (entries / (256 tiles)): what another pattern's fill is compared with
  unsigned long long serial = 0;
};

// A real square matrix as dense 16 x 16 tiles in a BlockOrder.  Super-tile (I, J) = the 64 x 64 positions of
// super-row I and super-column J, stored when it holds an entry, with a 16-bit mask of its 4 x 4 tiles
// (bit 4 cb + rb: row block rb, column block cb) and its tiles contiguous in bit order from slot `sbase`.
// A tile is 256 doubles; element (row o_r, column o_c) of it at phys(o_c) * 16 + phys(o_r), phys(o) = 4 (o & 3) + (o >> 2)
// (the order in which four lane groups of the matrix instruction hold consecutive words: spgemm_block.hip).
// Zero = no entry.
struct BlockForm {
  std::shared_ptr<BlockOrder> order;
  int32_t ns = 0;
  int64_t nst = 0, ntiles = 0, nnz = 0;
  DevBuf<int64_t> soff;        // [ns + 1] super-column J holds the super-tiles soff[J] .. soff[J + 1]
  DevBuf<int32_t> srow;        // [nst]    super-row, ascending inside a super-column
  DevBuf<int32_t> smask;       // [nst]    (low 16 bits)
  DevBuf<int64_t> sbase;       // [nst]    first tile slot
  DevBuf<double> tiles;        // [256 ntiles (capacity may be larger)]
  // the same super-tiles by super-ROW (what the left operand of a product is walked by): row I holds
  // roff[I] .. roff[I + 1], rcol ascending, ridx = index into srow / smask / sbase.  Built on first use.
  DevBuf<int64_t> roff;
  DevBuf<int32_t> rcol, ridx;
  bool have_rows = false;
  // per column POSITION (64 ns): entries of the column and the largest LABEL among its rows (-1: empty) -- the "last row"
  // of the AddSparseVectors rules in the caller's labels.  Written by the merge that produced the matrix, or on first use.
  DevBuf<int32_t> ccount, plast;
  bool have_stat = false;
  // per super-tile two words of 16 x 4 bits (tile bit t at bits 4 t .. 4 t + 3): [2 s] which of the four 4-column slices
  // (in-block positions 4 q .. 4 q + 3) of tile t hold an entry, [2 s + 1] which 4-row slices.  A matrix instruction q of a
  // tile pair multiplies column slice q of the A tile with row slice q of the B tile: skipped when either is empty.
  // Computed on first use as an operand (one pass over the tiles).
  DevBuf<unsigned long long> quads;
  bool have_quads = false;
};

struct BlockInfo {
  int used = 0;                // 1: the block path computed the product
  double fill_a = 0, fill_b = 0;      // entries / (256 * tiles) of the operands
  int64_t tiles_a = 0, tiles_b = 0, tiles_c = 0;
  int64_t cand = 0;            // candidate output super-tiles
  int64_t tile_products = 0;   // 16 x 16 x 16 tile products issued (4 matrix instructions each)
  int64_t nnz_c = 0;
  int64_t products = 0;        // intermediate products of the multiply (counted only with option time_kernels)
  float ms_numeric = 0.f;
};

// C = alpha A B pruned (PruneList.f90:8-38) through the block path.  false: not taken (operands complex / not square /
// the clustering finds no blocks worth the matrix cores); C untouched.  ev_begin / ev_end (optional): recorded around the
// numeric kernel.
// Operands: compressed columns (their block form is cached per matrix: value buffer, its allocation serial, the value
// epoch) or block form (DevMat::blk).  keep_blocked: C is left in block form (C.blk; pack() converts).
bool spgemm_block(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule, BlockInfo* info,
                  hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr, bool keep_blocked = false);
DevMat block_unpack(const DevMat& M);   // block form -> compressed columns under the caller's labels
// One TRS2 step on an iterate of a dimension the block path multiplies (DensityMatrixSolversModule.F90:380-404): mode 1:
// X <- X X; mode 2: X <- 2 X - X X merged by the AddSparseVectors rules -- with out[0] = dot(X_new, D), out[2] = trace.
// X: compressed columns or block form; it is left in BLOCK form.  false: not taken (X unchanged).
bool block_trs2_step(DevMat& X, int mode, double threshold, bool dense_rule, const DevMat& D, double out[4], BlockInfo* info = nullptr,
                     hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
// the block order the engine holds for matrices of M's dimension, made from M if there is none (tests / tools);
// pos_host[index] = position
bool block_order_for(const DevMat& M, std::vector<int32_t>& pos_host);
// the block order made FOR the pattern of M (kept under its fingerprint; made now if there is none): positions and super-blocks
bool block_order_of_pattern(const DevMat& M, std::vector<int32_t>& pos_host, int32_t* ns_out);
// an order given by explicit positions (pos[index] in [0, 64 ns), at most 16 indices per block of 16 positions) becomes the current
// order of dimension n: a solve that has redistributed its operands in a block order (band_scope.cpp) multiplies in it
void install_block_positions(int32_t n, int32_t ns, const std::vector<int32_t>& pos);
void drop_block_caches();
// Block algebra (one rank, real, FMA arithmetic): the vocabulary of the solver loops on matrices in block form -- the
// counterpart of the slab algebra (kernels.hpp) for operands without runs.  Operands are in block form or in compressed
// columns (converted through the per-matrix cache: an identity, the Hamiltonian).  The same element rules as on
// compressed columns (AddSparseVectors with the tail rule in the caller's labels); dots and traces as fixed-shape sums.
// Every function returns false and leaves its operands alone when it cannot take them.
bool block_axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold);   // B <- alpha A + beta B
bool block_scale(DevMat& A, double c);
bool block_clone(const DevMat& A, DevMat& Out);
bool block_dot_trace(const DevMat& A, const DevMat& B, double* dot, double* trace_a);   // sum a b; trace(A)
bool block_norm(const DevMat& A, double* out);   // max column abs-sum

}  // namespace ntp
