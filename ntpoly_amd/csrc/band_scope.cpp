// Solver-level band recovery across ranks.
//
// The reference's own load balancer (LoadBalancerModule.F90:14-92: PermuteMatrix / UndoPermuteMatrix around a solver when
// SolverParameters::do_load_balancing is set) runs a solver on P^T A P for a random permutation P and undoes the
// permutation on the result: the arithmetic of the whole solve -- the order of the k steps of every product, the "beyond
// the other column's last row" tests of every merge -- happens in the PERMUTED index space.  An operand that reaches this
// engine under such a permutation (a banded Hamiltonian relabelled at random) has no runs: on one rank the fused steps
// recover the band under the labels (kernels.hip relabel_enter); on several ranks a recovered order also has to MOVE
// columns between ranks, so it is done here, once per solve: the band order of the first operand's pattern is found on the
// gathered matrix (the same deterministic search on every rank, kept per pattern), the operands are redistributed in that
// order, the solver runs on banded panels -- halo exchanges of a bandwidth instead of whole-matrix gathers, fused steps in
// slab form -- and the results are carried back.  What comes out is the reference's result for the solve run under its load
// balancer with this permutation: bit for bit the reference's on the relabelled operands, and within the threshold of
// the solve on the caller's labels (entries below the threshold survive a merge where they lie beyond the other
// column's last row: which ones do depends on the labels).
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include <algorithm>
#include <numeric>

#include "engine.hpp"
#include "kernels.hpp"
#include "spgemm_block.hpp"

namespace ntp {
namespace {
struct BandOrderCache {
  bool searched = false, found = false;
  bool block_tried = false, block_found = false;   // no band: the pattern's block order (spgemm_block.hip) instead
  int32_t block_ns = 0;
  std::vector<int32_t> block_pos;                  // block_pos[label] = position in [0, 64 block_ns)
  unsigned long long fingerprint = 0;
  int32_t n = 0;
  int64_t nnz = 0;
  std::vector<int32_t> pos;   // pos[label] = position
};
BandOrderCache& band_cache() {
  static BandOrderCache* c = new BandOrderCache();
  return *c;
}
int g_scope_depth = 0;
bool g_block_scope = false;             // the solve in progress runs on operands redistributed in a BLOCK order
long long g_scope_counts[3] = {0, 0, 0};   // solves run in a recovered band order, operands looked at, solves run in a block order

// out(map[r], map[c]) = in(r, c), column panels as the grid says (ps_permute with an explicit map)
void relabel_ps(const PSMatrix& in, PSMatrix& out, const int32_t* d_map) {
  DevMat R;
  if (world().active()) {
    DevMat full = ps_gather_full(in);
    R = remap_general(full, d_map, d_map, in.dim, in.c0, in.c1, false);
  } else {
    DevMat packed = packed_copy(in.loc);
    R = remap_general(packed, d_map, d_map, in.dim, 0, in.dim, false);
  }
  sync_stream();
  out.grid = in.grid; out.dim = in.dim; out.cplx = in.cplx; out.c0 = in.c0; out.c1 = in.c1;
  out.loc = std::move(R);
}
}  // namespace

const long long* band_scope_counts() { return g_scope_counts; }
bool block_scope_active() { return g_block_scope; }

bool band_scope_try(const std::vector<const PSMatrix*>& ins, const std::vector<PSMatrix*>& outs,
                    const std::function<void(const std::vector<const PSMatrix*>&, const std::vector<PSMatrix*>&)>& run) {
  if (g_scope_depth > 0 || ins.empty() || !world().active() || options().label_order == 0 || options().band_scope == 0) return false;
  // Unfused arithmetic is the mode whose point is the BITS of the reference's default build on the caller's labels, on any
  // number of ranks: a solve in a permuted index space keeps other entries below the threshold alive (section 4 of
  // DESIGN.md), so the scope is taken there only on request (band_scope = 2); the label-ordered kernels run instead.
  if (options().spgemm_fma == 0 && options().band_scope != 2) return false;
  const PSMatrix& H = *ins[0];
  const int32_t n = H.dim;
  if (n < 1024 || (H.grid && H.grid->num_slices > 1)) return false;
  for (const PSMatrix* m : ins)
    if (m->dim != n) return false;
  // run-like as it stands (collective): the spans of the local columns against their entries, summed over the ranks
  {
    int64_t span = 0;
    DevMat packed;
    const DevMat* P = &H.loc;
    if (H.loc.expanded() || H.loc.loose() || H.loc.blocked()) {
      packed = packed_copy(H.loc);
      P = &packed;
    }
    span = column_span_sum(*P);
    double v[2] = {(double)span, (double)P->nnz};
    comm_allreduce_sum(v, 2);
    if (v[1] <= 0.0 || v[1] < 8.0 * (double)n || v[0] <= 2.0 * v[1]) return false;
  }
  g_scope_counts[1] += 1;
  // the band order of this pattern: searched once (the gathered matrix is the same on every rank, the search deterministic)
  BandOrderCache& c = band_cache();
  DevMat full = ps_gather_full(H);
  const unsigned long long fp = pattern_fingerprint_of(full);
  if (!(c.searched && c.fingerprint == fp && c.n == n && c.nnz == full.nnz)) {
    c.searched = true;
    c.found = false;
    c.block_tried = false;
    c.block_found = false;
    c.block_pos.clear();
    c.fingerprint = fp;
    c.n = n;
    c.nnz = full.nnz;
    c.pos.clear();
    DevBuf<int32_t> pos;
    int64_t bw = 0;
    if (find_band_order(full, pos, &bw) && bw <= 700 && (double)(2 * bw + 1) <= 3.0 * (double)full.nnz / (double)n) {
      c.pos.resize((size_t)n);
      pos.download(c.pos.data(), (size_t)n);
      c.found = true;
    }
    if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
      std::fprintf(stderr, "[band scope] pattern %016llx: %s (bandwidth %lld)\n", fp, c.found ? "band recovered" : "no band", (long long)bw);
  }
  // No band (a 3-D operand): the solve in the pattern's BLOCK order instead -- the index set clustered into blocks of 16 and
  // super-blocks of 64 (spgemm_block.hip), the operands redistributed so that a rank owns a contiguous range of positions;
  // its panel products then run on the block path (psmatrix.cpp multiply_panel) instead of the LDS hash.  Same contract as the
  // band order: the reference's solve under its load balancer with this permutation.
  bool block_mode = false;
  std::vector<int32_t> perm;   // perm[label] = new label
  if (!c.found && options().block_scope != 0 && options().block_path != 0 && options().spgemm_fma == 1 && !H.cplx) {
    if (!c.block_tried) {
      c.block_tried = true;
      c.block_found = block_order_of_pattern(full, c.block_pos, &c.block_ns);
      if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
        std::fprintf(stderr, "[band scope] pattern %016llx: %s\n", fp, c.block_found ? "block order" : "no block order");
    }
    if (c.block_found) {
      // new labels = ranks of the positions (the padding slots of the order drop out); the order in the new labels keeps its
      // positions -- monotone, so a panel of consecutive labels is a range of positions
      std::vector<int32_t> idx((size_t)n);
      std::iota(idx.begin(), idx.end(), 0);
      std::sort(idx.begin(), idx.end(), [&](int32_t x, int32_t y) { return c.block_pos[(size_t)x] < c.block_pos[(size_t)y]; });
      perm.resize((size_t)n);
      std::vector<int32_t> pos_new((size_t)n);
      for (int32_t i = 0; i < n; ++i) {
        perm[(size_t)idx[(size_t)i]] = i;
        pos_new[(size_t)i] = c.block_pos[(size_t)idx[(size_t)i]];
      }
      install_block_positions(n, c.block_ns, pos_new);
      block_mode = true;
    }
  }
  full = DevMat();
  if (!c.found && !block_mode) return false;
  const std::vector<int32_t>& fwd = block_mode ? perm : c.pos;
  std::vector<int32_t> inv((size_t)n);
  for (int32_t i = 0; i < n; ++i) inv[(size_t)fwd[(size_t)i]] = i;
  DevBuf<int32_t> d_pos((size_t)n), d_inv((size_t)n);
  d_pos.upload(fwd.data(), (size_t)n);
  d_inv.upload(inv.data(), (size_t)n);
  std::vector<PSMatrix> in_b(ins.size()), out_b(outs.size());
  std::vector<const PSMatrix*> in_p;
  std::vector<PSMatrix*> out_p;
  for (size_t i = 0; i < ins.size(); ++i) {
    relabel_ps(*ins[i], in_b[i], d_pos.p);
    in_p.push_back(&in_b[i]);
  }
  for (size_t i = 0; i < outs.size(); ++i) {
    ps_construct_empty(out_b[i], n, H.grid, outs[i]->cplx);
    out_p.push_back(&out_b[i]);
  }
  {
    struct ScopeGuard {   // (restored on every way out of run(), an exception included)
      ScopeGuard(bool block, const int32_t* labels) { g_scope_depth += 1; g_block_scope = block; set_scope_labels(labels); }
      ~ScopeGuard() { set_scope_labels(nullptr); g_block_scope = false; g_scope_depth -= 1; }
    // band order: the fused TRS2 panel steps merge on the CALLER'S labels (d_inv[position] = label), as the one-rank steps on a
    // relabelled operand do -- the solve keeps the entries the one-rank solve keeps (kernels.hpp scope_labels)
    } guard(block_mode, block_mode ? nullptr : d_inv.p);
    run(in_p, out_p);
  }
  for (size_t i = 0; i < outs.size(); ++i) relabel_ps(out_b[i], *outs[i], d_inv.p);
  g_scope_counts[block_mode ? 2 : 0] += 1;
  return true;
}

}  // namespace ntp
