// C ABI of the engine: the reference's own wrapper surface (Source/Wrapper/*_wrp.F90, declared in
// Source/C/*_c.h) re-exported with identical names, argument order and calling convention
// (every scalar by reference, handles = caller-owned int[SIZE_wrp] buffers, no status codes),
// plus a few ntpoly_amd_* extension entry points (RCCL bootstrap, statistics, options).
// See include/*.h for the declarations with the reference line each symbol replaces.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "engine.hpp"
#include "spgemm_block.hpp"
#include "io.hpp"

using namespace ntp;

namespace {
constexpr int SIZE_wrp = 12;  // Source/C/Wrapper.h:4

template <typename T>
T* get(const int* ih) {
  T* p;
  std::memcpy(&p, ih, sizeof(p));
  if (!p) NTP_FATAL("null handle passed to the C ABI");
  return p;
}
// every entry point sees packed matrices; only the TRS2 step below passes its iterate on as it is (kernels.hpp, pack())
template <>
PSMatrix* get<PSMatrix>(const int* ih) {
  PSMatrix* p;
  std::memcpy(&p, ih, sizeof(p));
  if (!p) NTP_FATAL("null handle passed to the C ABI");
  if (p->loc.loose() || p->loc.expanded() || p->loc.blocked()) pack(p->loc);
  if (p->grid) use_grid_comm(p->grid);   // (a matrix on a sub-grid: the call's collectives run on that grid's communicator)
  return p;
}
PSMatrix* get_unpacked(const int* ih) {
  PSMatrix* p;
  std::memcpy(&p, ih, sizeof(p));
  if (!p) NTP_FATAL("null handle passed to the C ABI");
  if (p->grid) use_grid_comm(p->grid);
  return p;
}
// The vocabulary entry points (MatrixMultiply, IncrementMatrix, ScaleMatrix, CopyMatrix, DotMatrix, MatrixNorm) run
// inside a slab session of their own (engine.hpp SlabSession; FMA arithmetic, one rank, real, option slab_algebra):
// a caller's own loop over the C ABI then keeps its matrices in the tile kernel's operand form between its calls --
// operands are converted where they are on first use, products stay where the kernel wrote them -- and every OTHER
// entry point still sees compressed columns (get<PSMatrix> packs on access).  Session not open: packed as ever.
struct ApiSession {
  SlabSession s;
  ApiSession() : s(true, true) {}
  PSMatrix* mat(const int* ih) const { return s.opened ? get_unpacked(ih) : get<PSMatrix>(ih); }
};
template <typename T>
void put(int* ih, T* p) {
  std::memset(ih, 0, sizeof(int) * SIZE_wrp);
  std::memcpy(ih, &p, sizeof(p));
}

struct Pool {  // MatrixMemoryPool_p / _lr / _lc: the engine keeps its own HBM workspace
  int dummy = 0;
};
struct LocalMat {  // Matrix_lsr / Matrix_lsc
  DevMat m;
};

const ProcessGrid* default_grid() {
  if (!global_grid_constructed()) NTP_FATAL("the global process grid has not been constructed");
  return &global_grid();
}
std::string fstring(const char* s, const int* n) { return std::string(s, (size_t)*n); }

// MatrixDiagonalScale of a local matrix (sparse_includes/MatrixDiagonalScale.f90): values of column `index_column`
// are multiplied by the triplet's value, triplet after triplet
void local_diagonal_scale(DevMat& m, const HostTriplets& t) {
  if (m.cols == 0) return;
  const size_t w = m.wval();
  std::vector<double> f((size_t)m.cols * w, 0.0);
  for (int32_t j = 0; j < m.cols; ++j) f[(size_t)j * w] = 1.0;
  for (size_t i = 0; i < t.size(); ++i) {
    const int32_t col = t.col[i] - 1;
    if (col < 0 || col >= m.cols) continue;
    const double re = t.cplx ? t.val[2 * i] : t.val[i], im = t.cplx ? t.val[2 * i + 1] : 0.0;
    double& fr = f[(size_t)col * w];
    if (m.cplx) {
      double& fi = f[(size_t)col * w + 1];
      const double nr = fr * re - fi * im, ni = fr * im + fi * re;
      fr = nr;
      fi = ni;
    } else {
      fr *= re;
    }
  }
  DevBuf<double> d(f.size());
  d.upload(f.data(), f.size());
  scale_columns(m, d.p);
  sync_stream();
}
// PrintMatrix (sparse_includes/PrintMatrix.f90): MatrixMarket coordinate text, entries in column order
void local_print(const DevMat& m, const char* path) {
  HostTriplets t;
  to_triplets(m, 0, t);
  FILE* f = path ? std::fopen(path, "w") : stdout;
  if (!f) NTP_FATAL(std::string("cannot open ") + path + " for writing");
  std::fprintf(f, "%%%%MatrixMarket matrix coordinate %s general\n%%\n", m.cplx ? "complex" : "real");
  std::fprintf(f, "%d %d %lld\n", m.rows, m.cols, (long long)m.nnz);
  for (size_t i = 0; i < t.size(); ++i) {
    if (m.cplx) std::fprintf(f, "%d %d %.17g %.17g\n", t.row[i], t.col[i], t.val[2 * i], t.val[2 * i + 1]);
    else std::fprintf(f, "%d %d %.17g\n", t.row[i], t.col[i], t.val[i]);
  }
  if (path) std::fclose(f);
  else std::fflush(f);
}
}  // namespace

extern "C" {

// ===================================================================== extensions
// RCCL bootstrap: rank 0 calls ntpoly_amd_get_unique_id, the launcher broadcasts the 128 bytes,
// every rank calls ntpoly_amd_init_comm before constructing a process grid.
void ntpoly_amd_get_unique_id(char* out128) { comm_get_unique_id(out128); }
void ntpoly_amd_init_comm(const char* id128, const int* rank, const int* nranks) { comm_init(id128, *rank, *nranks); }
void ntpoly_amd_finalize_comm() { comm_finalize(); }
int ntpoly_amd_comm_rank() { return world().rank; }
int ntpoly_amd_comm_size() { return world().nranks; }
void ntpoly_amd_barrier() { use_comm(nullptr); comm_barrier(); }   // (all processes, whatever grid was worked on last)
// max over all ranks of n host doubles, in place (timing of a distributed region: the slowest rank counts)
void ntpoly_amd_allreduce_max(double* values, const int* n) { use_comm(nullptr); comm_allreduce_max(values, *n); }
void ntpoly_amd_synchronize() {
  ensure_init();
  sync_stream();
  HIP_CHECK(hipDeviceSynchronize());
}
// number of visible GPUs (0 without failing: used by test collection)
int ntpoly_amd_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
void ntpoly_amd_panel_range(const int* dim, const int* nranks, const int* rank, int* c0, int* c1) {
  panel_range(*dim, *nranks, *rank, c0, c1);
}
// columns [a, b) of rank s's panel that a requester whose B panel has rows [kmin, kmax] receives
void ntpoly_amd_halo_segment(const int* dim, const int* nranks, const int* s, const int* kmin, const int* kmax, int* a,
                             int* b) {
  halo_segment(*dim, *nranks, *s, *kmin, *kmax, a, b);
}
// the layout of one panel exchange on rank `me` (psmatrix.cpp panel_exchange_layout: the host arithmetic of the halo exchange of
// the fused panel steps): req = 4 words per rank (first, last row of its panel of B, two unused here), cnt[s P + q] = doubles
// rank s sends to rank q
void ntpoly_amd_panel_exchange_layout(const int* dim, const int* nranks, const int* me, const long long* req, const long long* cnt,
                                      int* sa, int* sb, long long* soff, int* ra, int* rb, long long* zoff) {
  panel_exchange_layout(*dim, *nranks, *me, reinterpret_cast<const int64_t*>(req), reinterpret_cast<const int64_t*>(cnt), sa, sb,
                        reinterpret_cast<int64_t*>(soff), ra, rb, reinterpret_cast<int64_t*>(zoff));
}
void ntpoly_amd_set_option(const char* name, const int* value) {
  const std::string n(name);
  if (n == "spgemm_force_bin") options().spgemm_force_bin = *value;
  else if (n == "increment_force_seq") options().increment_force_seq = *value;
  else if (n == "spgemm_fma") options().spgemm_fma = *value;
  else if (n == "plan_ahead") options().plan_ahead = *value;
  else if (n == "slab_algebra") options().slab_algebra = *value;
  else if (n == "operand_cache") options().operand_cache = *value;
  else if (n == "tile_rows") options().tile_rows = *value;
  else if (n == "tile_waves") options().tile_waves = *value;
  else if (n == "time_kernels") options().time_kernels = *value;
  else if (n == "spgemm_variant") options().spgemm_variant = *value;
  else if (n == "halo_overlap") options().halo_overlap = *value;
  else if (n == "load_balance") options().load_balance = *value;
  else if (n == "virtual_grid") options().virtual_grid = *value;
  else if (n == "loose_iterates") options().loose_iterates = *value;
  else if (n == "fused_update") options().fused_update = *value;
  else if (n == "label_order") options().label_order = *value;
  else if (n == "band_scope") options().band_scope = *value;
  else if (n == "exchange_ahead") options().exchange_ahead = *value;
  else if (n == "plan_fused") options().plan_fused = *value;
  else if (n == "tile_off32") options().tile_off32 = *value;
  else if (n == "tile_bbuf") options().tile_bbuf = *value;
  else if (n == "ghash_mfma") options().ghash_mfma = *value;
  else if (n == "block_unfused") options().block_unfused = *value;
  else if (n == "block_match") options().block_match = *value;
  else if (n == "block_scope") options().block_scope = *value;
  else if (n == "panel_sessions") options().panel_sessions = *value;
  else if (n == "label_rowoff") options().label_rowoff = *value;
  else if (n == "block_path") options().block_path = *value;
  else if (n == "tile_runs_only") options().tile_runs_only = *value;
  else if (n == "tile2") options().tile2 = *value;
  else if (n == "complex_tile") options().complex_tile = *value;
  else if (n == "thin_left") options().thin_left = *value;
  else if (n == "column_fused") options().column_fused = *value;
  else if (n == "complex_sessions") options().complex_sessions = *value;
  else NTP_FATAL("unknown option " + n);
}
// the current value of the options a caller may want to report (bench.py prints the arithmetic a drop-in caller gets)
int ntpoly_amd_get_option(const char* name) {
  const std::string n(name);
  if (n == "spgemm_fma") return options().spgemm_fma;
  if (n == "slab_algebra") return options().slab_algebra;
  if (n == "plan_ahead") return options().plan_ahead;
  if (n == "tile_rows") return options().tile_rows;
  if (n == "tile2") return options().tile2;
  if (n == "block_path") return options().block_path;
  if (n == "label_order") return options().label_order;
  if (n == "band_scope") return options().band_scope;
  if (n == "exchange_ahead") return options().exchange_ahead;
  if (n == "plan_fused") return options().plan_fused;
  if (n == "tile_off32") return options().tile_off32;
  if (n == "tile_bbuf") return options().tile_bbuf;
  if (n == "ghash_mfma") return options().ghash_mfma;
  if (n == "block_unfused") return options().block_unfused;
  if (n == "block_match") return options().block_match;
  if (n == "block_scope") return options().block_scope;
  if (n == "panel_sessions") return options().panel_sessions;
  if (n == "fused_update") return options().fused_update;
  if (n == "loose_iterates") return options().loose_iterates;
  if (n == "complex_tile") return options().complex_tile;
  if (n == "thin_left") return options().thin_left;
  if (n == "column_fused") return options().column_fused;
  if (n == "complex_sessions") return options().complex_sessions;
  NTP_FATAL("unknown option " + n);
}
// statistics of the last SpGEMM: out[0..12]: nnzA, nnzB, nnzC, products, tmp_entries, bins[6], overflow, slab kernel used
void ntpoly_amd_last_spgemm_stats(long long* out, float* ms_numeric, float* ms_total) {
  flush_spgemm_timers();
  const SpgemmStats& s = last_spgemm_stats();
  out[0] = s.nnz_a; out[1] = s.nnz_b; out[2] = s.nnz_c; out[3] = s.products; out[4] = s.tmp_entries;
  for (int i = 0; i < 6; ++i) out[5 + i] = s.bin_cols[i];
  out[11] = s.overflow_cols;
  out[12] = s.slab;
  *ms_numeric = s.ms_numeric;
  *ms_total = s.ms_total;
}
// grouped LDS-hash path of the last SpGEMM: out[0..5] = used, columns handed back to the per-column kernels, groups,
// table class, min-hash clustering used, steps (sum of the groups' row unions of B); ratio = steps / (nnz(B) / columns per group)
void ntpoly_amd_last_grouped_stats(long long* out, double* ratio) {
  const SpgemmStats& s = last_spgemm_stats();
  out[0] = s.grouped; out[1] = s.gh_failed_cols; out[2] = s.gh_groups; out[3] = s.gh_level; out[4] = s.gh_minhash;
  out[5] = s.gh_tile_rows;
  *ratio = s.gh_union_ratio;
}
// 1: the last SpGEMM ran on the thin-left kernel (spgemm_thin.hip)
int ntpoly_amd_last_spgemm_thin() { return last_spgemm_stats().thin; }
// block path of the last SpGEMM (spgemm_block.hip): out[0..2] = used, 16 x 16 x 16 tile products issued, candidate output
// super-tiles; fill = entries / (256 tiles) of the left operand
void ntpoly_amd_last_block_stats(long long* out, double* fill) {
  const SpgemmStats& s = last_spgemm_stats();
  out[0] = s.block; out[1] = s.block_tile_products; out[2] = s.block_cand;
  *fill = s.block_fill;
}
// tests: the block order the engine multiplies matrices of this dimension in (made from this matrix if there is none):
// position[index] (0-based; positions ascend along the order of the k steps); returns 1 on success
int ntpoly_amd_block_order(const int* ih, int* position) {
  PSMatrix& m = *get<PSMatrix>(ih);
  std::vector<int32_t> pos;
  if (!block_order_for(m.loc, pos)) return 0;
  std::memcpy(position, pos.data(), sizeof(int32_t) * pos.size());
  return 1;
}
void ntpoly_amd_drop_block_caches() { drop_block_caches(); }
// out[0] = halo exchanges of distributed multiplies so far, out[1] = host synchronisations inside them (counted where the
// host waits: sync_stream), out[2] = ALL host synchronisations of the process so far: a caller brackets a call with two
// reads to learn what the whole call cost (exchange, plan, totals)
void ntpoly_amd_exchange_stats(long long* out) {
  out[0] = exchange_stats().exchanges;
  out[1] = exchange_stats().host_syncs;
  out[2] = host_sync_count();
}
// tests: the bandwidth-reducing order of a (one-rank, real or complex) matrix' pattern; newpos[old] = new (0-based),
// returns 1 on success
int ntpoly_amd_band_order(const int* ih, int* newpos, long long* bandwidth) {
  PSMatrix& m = *get<PSMatrix>(ih);
  DevBuf<int32_t> pos;
  int64_t bw = 0;
  if (!find_band_order(m.loc, pos, &bw)) return 0;
  HIP_CHECK(hipMemcpyAsync(newpos, pos.p, sizeof(int32_t) * (size_t)m.loc.cols, hipMemcpyDeviceToHost, stream()));
  sync_stream();
  *bandwidth = bw;
  return 1;
}
// out[0..2]: purification steps computed inside the SpGEMM kernel's epilogue (X*X; 2X - X*X) and fused steps that had
// to be repeated on the unfused path, since start
void ntpoly_amd_fusion_counts(long long* out) {
  for (int q = 0; q < 3; ++q) out[q] = fusion_counts()[q];
}
// out[0] = multiplies computed in the two-block geometry of the MFMA kernel (spgemm_tile2.hip) since start, out[1] = launches
// of it whose geometry did not fit after all and were repeated on k_spgemm_tile
void ntpoly_amd_tile2_counts(long long* out) {
  for (int q = 0; q < 2; ++q) out[q] = tile2_counts()[q];
}
// GatherMatrixToProcess (PSMatrixModule.F90:1704-1808, distributed_includes/GatherMatrixToProcess.f90, GatherMatrixToAll.f90):
// the whole distributed matrix as a LOCAL matrix (Matrix_lsr / Matrix_lsc of its scalar type) -- on every process
// (*within_slice_id < 0: the _all variants) or on the process with that rank inside its slice only (the _id variants: data
// stays replicated across slices; the other processes' ih_local is left as it was).  The callee constructs the handle.
void ntpoly_amd_gather_matrix_to_process(const int* ih_this, int* ih_local, const int* within_slice_id) {
  const PSMatrix& m = *get<PSMatrix>(ih_this);
  DevMat full = ps_gather_full(m);      // (collective: the panels travel to every rank; a process that is not the target drops them)
  const ProcessGrid& g = m.grid ? *m.grid : global_grid();
  const int slice_size = g.num_rows * g.num_cols;
  const int in_slice = g.global_rank - slice_size * g.my_slice;
  if (*within_slice_id >= 0 && *within_slice_id != in_slice) return;
  auto* L = new LocalMat();
  L->m = std::move(full);
  put(ih_local, L);
}
// CommSplitMatrix (PSMatrixModule.F90:1489-1541, distributed_includes/CommSplitMatrix.f90): a copy of the WHOLE matrix on each
// half of its process grid (SplitProcessGrid, ProcessGridModule.F90:430-515: along the slices where there are several, else
// along the longer of rows / columns).  The half lives on a sub-communicator (ncclCommSplit); every later call on the copy --
// products, reductions, solvers -- runs inside that half (engine.hpp use_grid_comm).  One process: the copy itself, colour 0,
// "split along the slices" (the reference's base case, :11-14).
void ntpoly_amd_comm_split_matrix(const int* ih_this, int* ih_split, int* my_color, bool* split_slice) {
  const PSMatrix& m = *get<PSMatrix>(ih_this);
  if (!m.grid) NTP_FATAL("CommSplitMatrix of a matrix that was never constructed");
  auto* out = new PSMatrix();
  ps_comm_split(m, *out, my_color, split_slice);
  put(ih_split, out);
}
// SplitProcessGrid (ProcessGridModule.F90:430-515): the grid of this process's half (colour 0 / 1); collective over the old grid.
// The handle is owned by the engine (not to be handed to DestructProcessGrid_wrp); matrices constructed on it live on the
// half's communicator.  The reference's fifth result, the between-grid communicator, has no counterpart: what CommSplitMatrix
// uses it for is done by ntpoly_amd_comm_split_matrix.
void ntpoly_amd_split_process_grid(const int* ih_old_grid, int* ih_new_grid, int* my_color, bool* split_slice) {
  put(ih_new_grid, split_process_grid(*get<ProcessGrid>(ih_old_grid), my_color, split_slice));
}
// out[0] = rank of this process on the grid's communicator, out[1] = its size, out[2] = 1 when that is a sub-communicator
void ntpoly_amd_grid_comm_info(const int* ih_grid, int* out) {
  const ProcessGrid* g = get<ProcessGrid>(ih_grid);
  out[0] = g->global_rank;
  out[1] = g->total;
  out[2] = g->comm != nullptr ? 1 : 0;
}
// out[0] = solves that ran in a recovered band order across ranks (band_scope.cpp), out[1] = operands searched for one
void ntpoly_amd_band_scope_counts(long long* out) {
  for (int q = 0; q < 2; ++q) out[q] = band_scope_counts()[q];
}
// solves that ran in a BLOCK order across ranks (operands without runs or band: 3-D Hamiltonians), panel products of such
// solves that took the block path
void ntpoly_amd_block_scope_counts(long long* out) {
  out[0] = band_scope_counts()[2];
  out[1] = block_scope_products();
}
// out[0..3]: operations the solver loops did on matrices in slab form since start (products, merges / copies, scalings
// / dots / norms) and operations that had to go back to compressed columns
// out[0] = vocabulary operations (copy, scale, merge, dot, trace, norm) done on matrices in block form, out[1] = fallbacks
// [2] since start: identity increments done in place, norms of differences taken from the operands (column_fused.hip)
void ntpoly_amd_column_fused_counts(long long* out) {
  out[0] = column_fused_counts()[0];
  out[1] = column_fused_counts()[1];
}
// tests: the two vocabulary operations of the solver loops that have no entry point of their own in the reference's C ABI --
// IncrementMatrix(Identity, B, alpha) as the loops call it, and the norm of alpha A + beta B (returns 0 when the engine would
// form the difference instead)
void ntpoly_amd_increment_identity(const int* ih_identity, int* ih_matB, const double* alpha) {
  ps_increment_identity(*get<PSMatrix>(ih_identity), *get<PSMatrix>(ih_matB), *alpha);
}
int ntpoly_amd_norm_axpby(const int* ih_matA, const int* ih_matB, const double* alpha, const double* beta, double* norm) {
  return ps_norm_axpby(*get<PSMatrix>(ih_matA), *get<PSMatrix>(ih_matB), *alpha, *beta, norm) ? 1 : 0;
}
void ntpoly_amd_block_algebra_counts(long long* out) {
  out[0] = block_algebra_counts()[0];
  out[1] = block_algebra_counts()[1];
}
void ntpoly_amd_slab_algebra_counts(long long* out) {
  for (int q = 0; q < 4; ++q) out[q] = slab_algebra_counts()[q];
}
// products of slab sessions on more than one rank since start: done in slab form on every rank, declined (compressed columns),
// host synchronisations inside the former (measured)
void ntpoly_amd_panel_product_counts(long long* out) {
  out[0] = panel_product_counts()[0];
  out[1] = panel_product_counts()[1];
  out[2] = panel_product_counts()[2];
}
// searches for a bandwidth-reducing order since start: one per sparsity pattern, not per operand (the next cycle of an
// SCF loop -- same pattern, other values -- reuses the order)
void ntpoly_amd_band_searches(long long* out) { *out = band_searches(); }
void ntpoly_amd_reset_spgemm_accum() {
  flush_spgemm_timers();
  spgemm_accum() = SpgemmAccum();
}
// out: calls, products, nnz_c ; dout: alg_bytes, ms_numeric, ms_total
void ntpoly_amd_get_spgemm_accum(long long* out, double* dout) {
  flush_spgemm_timers();
  const SpgemmAccum& a = spgemm_accum();
  out[0] = a.calls; out[1] = a.products; out[2] = a.nnz_c;
  dout[0] = a.alg_bytes; dout[1] = a.ms_numeric; dout[2] = a.ms_total;
}
// per-iteration trace of the last solver call
int ntpoly_amd_trace_iterations() { return last_trace().iterations; }
void ntpoly_amd_trace_get(double* value, double* energy, double* sigma, long long* nnz) {
  const SolverTrace& t = last_trace();
  for (int i = 0; i < t.iterations; ++i) {
    value[i] = t.value[(size_t)i];
    energy[i] = t.energy[(size_t)i];
    sigma[i] = t.sigma[(size_t)i];
    nnz[i] = t.nnz[(size_t)i];
  }
}
void ntpoly_amd_trace_times(double* setup_ms, double* loop_ms) {
  *setup_ms = last_trace().setup_ms;
  *loop_ms = last_trace().loop_ms;
}
void ntpoly_amd_memory(long long* in_use, long long* cached) {
  *in_use = (long long)dev_bytes_in_use();
  *cached = (long long)dev_bytes_cached();
}
void ntpoly_amd_release_cache() {
  drop_operand_caches();
  dev_release_cache();
}
// hipMalloc calls the caching allocator had to make so far, and the host milliseconds they took
void ntpoly_amd_malloc_stats(long long* calls, double* ms) { dev_malloc_stats(calls, ms); }
// bulk triplet transfer (the reference ABI moves triplets one at a time)
void ntpoly_amd_triplets_set_r(int* ih_list, const long long* n, const int* col, const int* row, const double* val) {
  HostTriplets* t = get<HostTriplets>(ih_list);
  t->cplx = false;
  t->col.assign(col, col + *n);
  t->row.assign(row, row + *n);
  t->val.assign(val, val + *n);
}
void ntpoly_amd_triplets_set_c(int* ih_list, const long long* n, const int* col, const int* row, const double* val_ri) {
  HostTriplets* t = get<HostTriplets>(ih_list);
  t->cplx = true;
  t->col.assign(col, col + *n);
  t->row.assign(row, row + *n);
  t->val.assign(val_ri, val_ri + 2 * *n);
}
void ntpoly_amd_triplets_get(const int* ih_list, int* col, int* row, double* val) {
  const HostTriplets* t = get<HostTriplets>(ih_list);
  std::copy(t->col.begin(), t->col.end(), col);
  std::copy(t->row.begin(), t->row.end(), row);
  std::copy(t->val.begin(), t->val.end(), val);
}

// ===================================================================== ProcessGrid_c.h
void ConstructGlobalProcessGrid_wrp(const int* world_comm, const int* process_rows, const int* process_columns,
                                    const int* process_slices) {
  comm_bind_mpi(*world_comm);  // the reference's MPI communicator (ProcessGrid.cc:14): honoured when MPI is initialised
  construct_grid(global_grid(), *process_rows, *process_columns, *process_slices);
}
void ConstructGlobalProcessGrid_onlyslice_wrp(const int* world_comm, const int* process_slices) {
  comm_bind_mpi(*world_comm);
  construct_grid_default(global_grid(), *process_slices);
}
void ConstructGlobalProcessGrid_default_wrp(const int* world_comm) {
  comm_bind_mpi(*world_comm);
  construct_grid_default(global_grid(), 1);
}
void CopyProcessGrid_wrp(const int* ih_old_grid, int* ih_new_grid) {
  put(ih_new_grid, new ProcessGrid(*get<ProcessGrid>(ih_old_grid)));
}
int GetGlobalMySlice_wrp() { return global_grid().my_slice; }
int GetGlobalMyColumn_wrp() { return global_grid().my_col; }
int GetGlobalMyRow_wrp() { return global_grid().my_row; }
bool GetGlobalIsRoot_wrp() { return global_grid().is_root(); }
int GetGlobalNumSlices_wrp() { return global_grid().num_slices; }
int GetGlobalNumColumns_wrp() { return global_grid().num_cols; }
int GetGlobalNumRows_wrp() { return global_grid().num_rows; }
void WriteGlobalProcessGridInfo_wrp() { write_grid_info(global_grid()); }
void DestructGlobalProcessGrid_wrp() {}
void ConstructProcessGrid_wrp(int* ih_grid, const int* world_comm, const int* process_rows, const int* process_columns,
                              const int* process_slices) {
  comm_bind_mpi(*world_comm);
  ProcessGrid* g = new ProcessGrid();
  construct_grid(*g, *process_rows, *process_columns, *process_slices);
  put(ih_grid, g);
}
void ConstructProcessGrid_onlyslice_wrp(int* ih_grid, const int* world_comm, const int* process_slices) {
  comm_bind_mpi(*world_comm);
  ProcessGrid* g = new ProcessGrid();
  construct_grid_default(*g, *process_slices);
  put(ih_grid, g);
}
void ConstructProcessGrid_default_wrp(int* ih_grid, const int* world_comm) {
  comm_bind_mpi(*world_comm);
  ProcessGrid* g = new ProcessGrid();
  construct_grid_default(*g, 1);
  put(ih_grid, g);
}
int GetMySlice_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->my_slice; }
int GetMyColumn_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->my_col; }
int GetMyRow_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->my_row; }
int GetNumSlices_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->num_slices; }
int GetNumColumns_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->num_cols; }
int GetNumRows_wrp(const int* ih_grid) { return get<ProcessGrid>(ih_grid)->num_rows; }
void WriteProcessGridInfo_wrp(const int* ih_grid) { write_grid_info(*get<ProcessGrid>(ih_grid)); }
void DestructProcessGrid_wrp(int* ih_grid) { delete get<ProcessGrid>(ih_grid); }

// ===================================================================== TripletList_c.h
void ConstructTripletList_r_wrp(int* ih_this, const int* size) {
  HostTriplets* t = new HostTriplets();
  t->cplx = false;
  t->col.assign((size_t)*size, 0);
  t->row.assign((size_t)*size, 0);
  t->val.assign((size_t)*size, 0.0);
  put(ih_this, t);
}
void ResizeTripletList_r_wrp(int* ih_this, const int* size) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  t->col.resize((size_t)*size);
  t->row.resize((size_t)*size);
  t->val.resize((size_t)*size);
}
void AppendToTripletList_r_wrp(int* ih_this, const int* index_column, const int* index_row, const double* point_value) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  t->col.push_back(*index_column);
  t->row.push_back(*index_row);
  t->val.push_back(*point_value);
}
void SetTripletAt_r_wrp(int* ih_this, const int* index, const int* index_column, const int* index_row,
                        const double* point_value) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  const size_t i = (size_t)*index - 1;  // 1-based (TripletList.cc:44-48 adds 1)
  t->col[i] = *index_column;
  t->row[i] = *index_row;
  t->val[i] = *point_value;
}
void GetTripletAt_r_wrp(const int* ih_this, const int* index, int* index_column, int* index_row, double* point_value) {
  const HostTriplets* t = get<HostTriplets>(ih_this);
  const size_t i = (size_t)*index - 1;
  *index_column = t->col[i];
  *index_row = t->row[i];
  *point_value = t->val[i];
}
void DestructTripletList_r_wrp(int* ih_this) { delete get<HostTriplets>(ih_this); }
static void sort_triplets(const HostTriplets& in, HostTriplets& out) {
  const size_t n = in.size(), w = in.cplx ? 2 : 1;
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    if (in.col[a] != in.col[b]) return in.col[a] < in.col[b];
    return in.row[a] < in.row[b];
  });
  out.cplx = in.cplx;
  out.col.resize(n);
  out.row.resize(n);
  out.val.resize(n * w);
  for (size_t i = 0; i < n; ++i) {
    out.col[i] = in.col[order[i]];
    out.row[i] = in.row[order[i]];
    for (size_t k = 0; k < w; ++k) out.val[i * w + k] = in.val[order[i] * w + k];
  }
}
void SortTripletList_r_wrp(const int* ih_this, const int* matrix_size, int* h_sorted) {
  (void)matrix_size;
  HostTriplets* s = new HostTriplets();
  sort_triplets(*get<HostTriplets>(ih_this), *s);
  put(h_sorted, s);
}
int GetTripletListSize_r_wrp(const int* ih_this) { return (int)get<HostTriplets>(ih_this)->size(); }

void ConstructTripletList_c_wrp(int* ih_this, const int* size) {
  HostTriplets* t = new HostTriplets();
  t->cplx = true;
  t->col.assign((size_t)*size, 0);
  t->row.assign((size_t)*size, 0);
  t->val.assign((size_t)*size * 2, 0.0);
  put(ih_this, t);
}
void ResizeTripletList_c_wrp(int* ih_this, const int* size) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  t->col.resize((size_t)*size);
  t->row.resize((size_t)*size);
  t->val.resize((size_t)*size * 2);
}
void AppendToTripletList_c_wrp(int* ih_this, const int* index_column, const int* index_row, const double* re,
                               const double* im) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  t->col.push_back(*index_column);
  t->row.push_back(*index_row);
  t->val.push_back(*re);
  t->val.push_back(*im);
}
void SetTripletAt_c_wrp(int* ih_this, const int* index, const int* index_column, const int* index_row, const double* re,
                        const double* im) {
  HostTriplets* t = get<HostTriplets>(ih_this);
  const size_t i = (size_t)*index - 1;
  t->col[i] = *index_column;
  t->row[i] = *index_row;
  t->val[2 * i] = *re;
  t->val[2 * i + 1] = *im;
}
// the reference declares the outputs `const double*` (TripletList_c.h:26-28) but writes them
void GetTripletAt_c_wrp(const int* ih_this, const int* index, int* index_column, int* index_row, const double* re,
                        const double* im) {
  const HostTriplets* t = get<HostTriplets>(ih_this);
  const size_t i = (size_t)*index - 1;
  *index_column = t->col[i];
  *index_row = t->row[i];
  *const_cast<double*>(re) = t->val[2 * i];
  *const_cast<double*>(im) = t->val[2 * i + 1];
}
void DestructTripletList_c_wrp(int* ih_this) { delete get<HostTriplets>(ih_this); }
void SortTripletList_c_wrp(const int* ih_this, const int* matrix_size, int* h_sorted) {
  (void)matrix_size;
  HostTriplets* s = new HostTriplets();
  sort_triplets(*get<HostTriplets>(ih_this), *s);
  put(h_sorted, s);
}
int GetTripletListSize_c_wrp(const int* ih_this) { return (int)get<HostTriplets>(ih_this)->size(); }

// ===================================================================== Permutation_c.h
void ConstructDefaultPermutation_wrp(int* ih_this, const int* matrix_dimension) {
  Permutation* p = new Permutation();
  permutation_default(*p, *matrix_dimension);
  put(ih_this, p);
}
void ConstructReversePermutation_wrp(int* ih_this, const int* matrix_dimension) {
  Permutation* p = new Permutation();
  permutation_reverse(*p, *matrix_dimension);
  put(ih_this, p);
}
void ConstructRandomPermutation_wrp(int* ih_this, const int* matrix_dimension) {
  Permutation* p = new Permutation();
  permutation_random(*p, *matrix_dimension);
  put(ih_this, p);
}
void DestructPermutation_wrp(int* ih_this) { delete get<Permutation>(ih_this); }
// extension: explicit permutation (index_lookup, 1-based)
void ntpoly_amd_permutation_set(int* ih_this, const int* n, const int* index_lookup) {
  Permutation* p = get<Permutation>(ih_this);
  permutation_default(*p, *n);
  for (int i = 0; i < *n; ++i) p->index_lookup[(size_t)i] = index_lookup[i];
  for (int i = 0; i < *n; ++i) p->reverse_index_lookup[(size_t)p->index_lookup[(size_t)i] - 1] = i + 1;
}

// ===================================================================== memory pools
void ConstructMatrixMemoryPool_p_wrp(int* ih_this, const int* ih_matrix) {
  (void)ih_matrix;
  put(ih_this, new Pool());
}
void DestructMatrixMemoryPool_p_wrp(int* ih_this) { delete get<Pool>(ih_this); }
void ConstructMatrixMemoryPool_lr_wrp(int* ih_this, const int* columns, const int* rows) {
  (void)columns; (void)rows;
  put(ih_this, new Pool());
}
void DestructMatrixMemoryPool_lr_wrp(int* ih_this) { delete get<Pool>(ih_this); }
void ConstructMatrixMemoryPool_lc_wrp(int* ih_this, const int* columns, const int* rows) {
  (void)columns; (void)rows;
  put(ih_this, new Pool());
}
void DestructMatrixMemoryPool_lc_wrp(int* ih_this) { delete get<Pool>(ih_this); }

// ===================================================================== SolverParameters_c.h
void ConstructSolverParameters_wrp(int* ih_this) { put(ih_this, new SolverParameters()); }
void SetParametersConvergeDiff_wrp(int* ih_this, const double* v) { get<SolverParameters>(ih_this)->converge_diff = *v; }
void SetParametersMaxIterations_wrp(int* ih_this, const int* v) { get<SolverParameters>(ih_this)->max_iterations = *v; }
void SetParametersBeVerbose_wrp(int* ih_this, const bool* v) { get<SolverParameters>(ih_this)->be_verbose = *v; }
void SetParametersThreshold_wrp(int* ih_this, const double* v) { get<SolverParameters>(ih_this)->threshold = *v; }
void SetParametersLoadBalance_wrp(int* ih_this, const int* ih_permutation) {
  SolverParameters* p = get<SolverParameters>(ih_this);
  if (options().load_balance == 0) return;  // engine option: keep the caller's ordering (kernels.hpp)
  p->do_load_balancing = true;  // SolverParametersModule.F90:160-167
  p->balance_permutation = *get<Permutation>(ih_permutation);
}
void SetParametersStepThreshold_wrp(int* ih_this, const double* v) { get<SolverParameters>(ih_this)->step_thresh = *v; }
void SetParametersMonitorConvergence_wrp(int* ih_this, const bool* v) {
  get<SolverParameters>(ih_this)->monitor_convergence = *v;
}
void DestructSolverParameters_wrp(int* ih_this) { delete get<SolverParameters>(ih_this); }

// ===================================================================== Logging_c.h
void ActivateLogger_wrp(const bool* start_document) { log_activate(*start_document, nullptr); }
void ActivateLoggerFile_wrp(const bool* start_document, const char* file_name, const int* name_size) {
  log_activate(*start_document, fstring(file_name, name_size).c_str());
}
void DeactivateLogger_wrp() { log_deactivate(); }
// extension: the writing half of LoggingModule.F90 (WriteHeader / WriteElement / WriteListElement / Enter / ExitSubLog),
// which the reference exposes to Fortran callers only; used by the Fortran module layer (fortran/ntpoly_amd_modules.f90)
void ntpoly_amd_log_header(const char* text, const int* n) { log_header(fstring(text, n).c_str()); }
void ntpoly_amd_log_enter() { log_enter(); }
void ntpoly_amd_log_exit() { log_exit(); }
void ntpoly_amd_log_element_string(const char* key, const int* nk, const char* v, const int* nv) {
  log_element(fstring(key, nk).c_str(), fstring(v, nv).c_str());
}
void ntpoly_amd_log_element_int(const char* key, const int* nk, const int* v) { log_element(fstring(key, nk).c_str(), *v); }
void ntpoly_amd_log_element_real(const char* key, const int* nk, const double* v) { log_element(fstring(key, nk).c_str(), *v); }
void ntpoly_amd_log_element_bool(const char* key, const int* nk, const bool* v) { log_element(fstring(key, nk).c_str(), *v); }
void ntpoly_amd_log_list_element(const char* key, const int* nk) { log_list_element(fstring(key, nk).c_str()); }

// ===================================================================== PSMatrix_c.h
void ConstructEmptyMatrix_ps_wrp(int* ih_this, const int* matrix_dim) {
  PSMatrix* m = new PSMatrix();
  ps_construct_empty(*m, *matrix_dim, default_grid(), false);
  put(ih_this, m);
}
void ConstructEmptyMatrixPG_ps_wrp(int* ih_this, const int* matrix_dim, const int* ih_grid) {
  PSMatrix* m = new PSMatrix();
  ps_construct_empty(*m, *matrix_dim, get<ProcessGrid>(ih_grid), false);
  put(ih_this, m);
}
void CopyMatrix_ps_wrp(const int* ih_matA, int* ih_matB) {
  ApiSession ses;
  ps_copy(*ses.mat(ih_matA), *ses.mat(ih_matB));
}
void DestructMatrix_ps_wrp(int* ih_this) { delete get_unpacked(ih_this); }
void ConstructMatrixFromMatrixMarket_ps_wrp(int* ih_this, const char* file_name, const int* name_size) {
  PSMatrix* m = new PSMatrix();
  ps_read_matrix_market(*m, fstring(file_name, name_size), default_grid());
  put(ih_this, m);
}
void ConstructMatrixFromBinary_ps_wrp(int* ih_this, const char* file_name, const int* name_size) {
  PSMatrix* m = new PSMatrix();
  ps_read_binary(*m, fstring(file_name, name_size), default_grid());
  put(ih_this, m);
}
void ConstructMatrixFromMatrixMarketPG_ps_wrp(int* ih_this, const char* file_name, const int* name_size,
                                              const int* ih_grid) {
  PSMatrix* m = new PSMatrix();
  ps_read_matrix_market(*m, fstring(file_name, name_size), get<ProcessGrid>(ih_grid));
  put(ih_this, m);
}
void ConstructMatrixFromBinaryPG_ps_wrp(int* ih_this, const char* file_name, const int* name_size, const int* ih_grid) {
  PSMatrix* m = new PSMatrix();
  ps_read_binary(*m, fstring(file_name, name_size), get<ProcessGrid>(ih_grid));
  put(ih_this, m);
}
void WriteMatrixToBinary_ps_wrp(const int* ih_this, const char* file_name, const int* name_size) {
  ps_write_binary(*get<PSMatrix>(ih_this), fstring(file_name, name_size));
}
void WriteMatrixToMatrixMarket_ps_wrp(const int* ih_this, const char* file_name, const int* name_size) {
  ps_write_matrix_market(*get<PSMatrix>(ih_this), fstring(file_name, name_size));
}
void FillMatrixFromTripletList_psr_wrp(const int* ih_this, const int* ih_triplet_list) {
  PSMatrix* m = get<PSMatrix>(ih_this);
  if (m->cplx) {  // a real fill of a (so far) complex-typed handle makes it real, as the Fortran generic does
    m->cplx = false;
    m->loc.reset_empty(m->dim, m->c1 - m->c0, false);
  }
  ps_fill_from_triplets(*m, *get<HostTriplets>(ih_triplet_list));
}
void FillMatrixFromTripletList_psc_wrp(const int* ih_this, const int* ih_triplet_list) {
  PSMatrix* m = get<PSMatrix>(ih_this);
  if (!m->cplx) {  // FillMatrixFromTripletList_psc converts the matrix to complex (PSMatrixModule.F90:846-850)
    m->cplx = true;
    m->loc.reset_empty(m->dim, m->c1 - m->c0, true);
  }
  ps_fill_from_triplets(*m, *get<HostTriplets>(ih_triplet_list));
}
// extension: the Fortran API's prepartitioned_in = .TRUE. (PSMatrixModule.F90:796-825): every rank
// passes only entries of its own column panel, no exchange
void ntpoly_amd_fill_prepartitioned(const int* ih_this, const int* ih_triplet_list) {
  PSMatrix* m = get<PSMatrix>(ih_this);
  const HostTriplets* t = get<HostTriplets>(ih_triplet_list);
  if (m->cplx != t->cplx) {
    m->cplx = t->cplx;
  }
  m->loc = from_triplets(*t, m->dim, m->c1 - m->c0, m->c0);
}
void FillMatrixPermutation_ps_wrp(int* ih_this, const int* ih_permutation, const bool* permuterows) {
  ps_fill_permutation(*get<PSMatrix>(ih_this), get<Permutation>(ih_permutation)->index_lookup, *permuterows);
}
void FillMatrixIdentity_ps_wrp(int* ih_this) { ps_fill_identity(*get<PSMatrix>(ih_this)); }
void GetMatrixActualDimension_ps_wrp(const int* ih_this, int* size) { *size = get<PSMatrix>(ih_this)->dim; }
void GetMatrixLogicalDimension_ps_wrp(const int* ih_this, int* size) { *size = get<PSMatrix>(ih_this)->dim; }
void GetMatrixSize_ps_wrp(const int* ih_this, long int* size) { *size = (long int)ps_size(*get_unpacked(ih_this)); }   // (the entry count is kept in every storage form)
// PSMatrix_c.h:31 (wrapper PSMatrixModule_wrp.F90:259-266)
void FillMatrixDense_ps_wrp(int* ih_this) { ps_fill_dense(*get<PSMatrix>(ih_this)); }
// PSMatrix_c.h:37-44 (wrapper :344-391): blocks are [start, end) with 1-based indices, slices are inclusive
void GetMatrixBlock_psr_wrp(const int* ih_this, int* ih_triplet_list, int* start_row, int* end_row, int* start_column,
                            int* end_column) {
  const PSMatrix* m = get<PSMatrix>(ih_this);
  HostTriplets* t = get<HostTriplets>(ih_triplet_list);
  if (m->cplx) {
    PSMatrix r;
    ps_to_real(*m, r);
    ps_get_block(r, *start_row, *end_row, *start_column, *end_column, *t);
  } else {
    ps_get_block(*m, *start_row, *end_row, *start_column, *end_column, *t);
  }
}
void GetMatrixBlock_psc_wrp(const int* ih_this, int* ih_triplet_list, int* start_row, int* end_row, int* start_column,
                            int* end_column) {
  const PSMatrix* m = get<PSMatrix>(ih_this);
  HostTriplets* t = get<HostTriplets>(ih_triplet_list);
  if (!m->cplx) {
    PSMatrix c;
    ps_to_complex(*m, c);
    ps_get_block(c, *start_row, *end_row, *start_column, *end_column, *t);
  } else {
    ps_get_block(*m, *start_row, *end_row, *start_column, *end_column, *t);
  }
}
void GetMatrixSlice_wrp(const int* ih_this, int* ih_submatrix, int* start_row, int* end_row, int* start_column,
                        int* end_column) {
  ps_get_slice(*get<PSMatrix>(ih_this), *get<PSMatrix>(ih_submatrix), *start_row, *end_row, *start_column, *end_column);
}
// PSMatrix_c.h:48 (wrapper :417-425)
void ResizeMatrix_ps_wrp(int* ih_this, const int* new_size) { ps_resize(*get<PSMatrix>(ih_this), *new_size); }
// PSMatrix_c.h:67-68 (wrapper PSMatrixAlgebraModule_wrp.F90)
void MatrixDiagonalScale_psr_wrp(int* ih_mat, const int* ih_tlist) {
  ps_diagonal_scale(*get<PSMatrix>(ih_mat), *get<HostTriplets>(ih_tlist));
}
void MatrixDiagonalScale_psc_wrp(int* ih_mat, const int* ih_tlist) {
  PSMatrix* m = get<PSMatrix>(ih_mat);
  if (!m->cplx) {  // a complex diagonal turns the matrix complex, as ScaleMatrix_psc does (PSMatrixAlgebraModule.F90:486-504)
    PSMatrix c;
    ps_to_complex(*m, c);
    m->cplx = true;
    m->loc = std::move(c.loc);
  }
  ps_diagonal_scale(*m, *get<HostTriplets>(ih_tlist));
}
void GetMatrixTripletList_psr_wrp(const int* ih_this, int* ih_triplet_list) {
  const PSMatrix* m = get<PSMatrix>(ih_this);
  HostTriplets* t = get<HostTriplets>(ih_triplet_list);
  if (m->cplx) {  // ConvertMatrixToReal (PSMatrixModule.F90:1003-1004)
    PSMatrix r;
    ps_to_real(*m, r);
    ps_get_triplets(r, *t);
  } else {
    ps_get_triplets(*m, *t);
  }
}
void GetMatrixTripletList_psc_wrp(const int* ih_this, int* ih_triplet_list) {
  const PSMatrix* m = get<PSMatrix>(ih_this);
  HostTriplets* t = get<HostTriplets>(ih_triplet_list);
  if (!m->cplx) {
    PSMatrix c;
    ps_to_complex(*m, c);
    ps_get_triplets(c, *t);
  } else {
    ps_get_triplets(*m, *t);
  }
}
void TransposeMatrix_ps_wrp(const int* ih_matA, int* ih_transmat) {
  ps_transpose(*get<PSMatrix>(ih_matA), *get<PSMatrix>(ih_transmat));
}
void ConjugateMatrix_ps_wrp(int* ih_matA) { ps_conjugate(*get<PSMatrix>(ih_matA)); }
void GetMatrixProcessGrid_ps_wrp(const int* ih_this, int* ih_grid) {
  put(ih_grid, const_cast<ProcessGrid*>(get<PSMatrix>(ih_this)->grid));
}
int ntpoly_amd_matrix_is_complex(const int* ih_this) { return get<PSMatrix>(ih_this)->cplx ? 1 : 0; }
void ntpoly_amd_matrix_local_columns(const int* ih_this, int* c0, int* c1) {
  *c0 = get<PSMatrix>(ih_this)->c0;
  *c1 = get<PSMatrix>(ih_this)->c1;
}

void DotMatrix_psr_wrp(const int* ih_matA, const int* ih_matB, double* product) {
  double out[2];
  ApiSession ses;
  ps_dot(*ses.mat(ih_matA), *ses.mat(ih_matB), out);
  *product = out[0];
}
void DotMatrix_psc_wrp(const int* ih_matA, const int* ih_matB, double* product_real, double* product_imag) {
  double out[2];
  ps_dot(*get<PSMatrix>(ih_matA), *get<PSMatrix>(ih_matB), out);
  *product_real = out[0];
  *product_imag = out[1];
}
void IncrementMatrix_ps_wrp(const int* ih_matA, int* ih_matB, const double* alpha_in, const double* threshold_in) {
  ApiSession ses;
  ps_increment(*ses.mat(ih_matA), *ses.mat(ih_matB), *alpha_in, *threshold_in);
}
void MatrixPairwiseMultiply_ps_wrp(const int* ih_matA, const int* ih_matB, int* ih_matC) {
  ps_pairwise(*get<PSMatrix>(ih_matA), *get<PSMatrix>(ih_matB), *get<PSMatrix>(ih_matC));
}
void MatrixMultiply_ps_wrp(const int* ih_matA, const int* ih_matB, int* ih_matC, const double* alpha_in,
                           const double* beta_in, const double* threshold_in, int* ih_memory_pool_in) {
  (void)ih_memory_pool_in;
  ApiSession ses;
  ps_multiply(*ses.mat(ih_matA), *ses.mat(ih_matB), *ses.mat(ih_matC), *alpha_in, *beta_in, *threshold_in);
}
void ScaleMatrix_ps_wrp(int* ih_this, const double* constant) {
  ApiSession ses;
  ps_scale(*ses.mat(ih_this), *constant);
}
double MatrixNorm_ps_wrp(const int* ih_this) {
  ApiSession ses;
  return ps_norm(*ses.mat(ih_this));
}
double MeasureAsymmetry_ps_wrp(const int* ih_this) { return ps_measure_asymmetry(*get<PSMatrix>(ih_this)); }
void MatrixTrace_ps_wrp(const int* ih_this, double* trace_val) {
  ApiSession ses;
  *trace_val = ps_trace(*ses.mat(ih_this));
}
int IsIdentity_ps_wrp(const int* ih_this) { return ps_is_identity(*get<PSMatrix>(ih_this)) ? 1 : 0; }
void SymmetrizeMatrix_ps_wrp(int* ih_this) { ps_symmetrize(*get<PSMatrix>(ih_this)); }

// ===================================================================== LoadBalancer_c.h, EigenBounds_c.h
void PermuteMatrix_wrp(const int* ih_mat_in, int* ih_mat_out, const int* ih_permutation, int* ih_memorypool) {
  (void)ih_memorypool;
  ps_permute(*get<PSMatrix>(ih_mat_in), *get<PSMatrix>(ih_mat_out), *get<Permutation>(ih_permutation), false);
}
void UndoPermuteMatrix_wrp(const int* ih_mat_in, int* ih_mat_out, const int* ih_permutation, int* ih_memorypool) {
  (void)ih_memorypool;
  ps_permute(*get<PSMatrix>(ih_mat_in), *get<PSMatrix>(ih_mat_out), *get<Permutation>(ih_permutation), true);
}
// EigenBounds_c.h:4-5 names the outputs (max_value, min_value) but the wrapper forwards them
// positionally to GershgorinBounds(this, min_value, max_value) (EigenBoundsModule_wrp.F90): first = min
void GershgorinBounds_wrp(const int* ih_Hamiltonian, double* max_value, double* min_value) {
  ps_gershgorin(*get<PSMatrix>(ih_Hamiltonian), max_value, min_value);
}

// ===================================================================== solvers
// outputs are declared `const double*` in DensityMatrixSolvers_c.h:4-23 but are written
// (DensityMatrixSolversModule_wrp.F90:57-58)
#define DENSITY_SOLVER(NAME, FN)                                                                              \
  void NAME(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, const double* trace, int* ih_Density, \
            const double* energy_value_out, const double* chemical_potential_out,                             \
            const int* ih_solver_parameters) {                                                                \
    FN(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *trace, *get<PSMatrix>(ih_Density), \
       const_cast<double*>(energy_value_out), const_cast<double*>(chemical_potential_out),                    \
       *get<SolverParameters>(ih_solver_parameters));                                                         \
  }
DENSITY_SOLVER(PM_wrp, solver_pm)
DENSITY_SOLVER(TRS2_wrp, solver_trs2)
DENSITY_SOLVER(TRS4_wrp, solver_trs4)
DENSITY_SOLVER(HPCP_wrp, solver_hpcp)
// DensityMatrixSolvers_c.h:24-28 (wrapper DensityMatrixSolversModule_wrp.F90:129-153)
void ScaleAndFold_wrp(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, const double* trace, int* ih_Density,
                      const double* homo, const double* lumo, const double* energy_value_out,
                      const int* ih_solver_parameters) {
  solver_scale_and_fold(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *trace,
                        *get<PSMatrix>(ih_Density), *homo, *lumo, const_cast<double*>(energy_value_out),
                        *get<SolverParameters>(ih_solver_parameters));
}
// DensityMatrixSolvers_c.h:34-38 (wrapper :183-231)
void EnergyDensityMatrix_wrp(const int* ih_Hamiltonian, const int* ih_Density, int* ih_EnergyDensity, const double* threshold) {
  energy_density_matrix(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_Density), *get<PSMatrix>(ih_EnergyDensity), *threshold);
}
void McWeenyStep_wrp(const int* ih_D, int* ih_DOut, const double* threshold) {
  mcweeny_step(*get<PSMatrix>(ih_D), *get<PSMatrix>(ih_DOut), nullptr, *threshold);
}
void McWeenyStepS_wrp(const int* ih_D, int* ih_DOut, const int* ih_S, const double* threshold) {
  mcweeny_step(*get<PSMatrix>(ih_D), *get<PSMatrix>(ih_DOut), get<PSMatrix>(ih_S), *threshold);
}
// The dense family of the reference (gather + LAPACK eigensolver, EigenSolversModule.F90:74-131): here the
// eigendecomposition runs on the GPU (two-sided Jacobi, solvers_extra.cpp, dense.hip); f(A) = V f(L) V^H.
void DenseDensity_wrp(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, const double* trace, int* ih_Density,
                      const double* energy_value_out, const double* chemical_potential_out, const int* ih_solver_parameters) {
  // DensityMatrixSolversModule.F90:1120-1160: the step-function occupation (no smearing)
  compute_dense_foe(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *trace, *get<PSMatrix>(ih_Density),
                    nullptr, const_cast<double*>(energy_value_out), const_cast<double*>(chemical_potential_out),
                    *get<SolverParameters>(ih_solver_parameters));
}
static void dense_function(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters, const char* header,
                           double (*f)(double)) {
  const SolverParameters& p = *get<SolverParameters>(ih_solver_parameters);
  if (p.be_verbose && header) {
    log_header(header);
    log_enter();
  }
  dense_matrix_function(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), f, p);
  if (p.be_verbose && header) log_exit();
}
void DenseSquareRoot_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Square Root Solver", [](double v) { return std::sqrt(v); });
}
void DenseInverseSquareRoot_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Inverse Square Root Solver",
                 [](double v) { return 1.0 / std::sqrt(v); });
}
// FermiOperator_c.h:4-16 (FermiOperatorModule_wrp.F90): the wrapper always passes the inverse temperature
void ComputeDenseFOE_wrp(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, const double* trace, int* ih_Density,
                         const double* inv_temp_in, const double* energy_value_out, const double* chemical_potential_out,
                         const int* ih_solver_parameters) {
  compute_dense_foe(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *trace, *get<PSMatrix>(ih_Density),
                    inv_temp_in, const_cast<double*>(energy_value_out), const_cast<double*>(chemical_potential_out),
                    *get<SolverParameters>(ih_solver_parameters));
}
void WOM_GC_wrp(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, int* ih_Density, const double* chemical_potential,
                const double* inv_temp, const double* energy_value_out, const int* ih_solver_parameters) {
  solver_wom(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *get<PSMatrix>(ih_Density), *inv_temp,
             nullptr, chemical_potential, const_cast<double*>(energy_value_out), *get<SolverParameters>(ih_solver_parameters));
}
void WOM_C_wrp(const int* ih_Hamiltonian, const int* ih_InverseSquareRoot, int* ih_Density, const double* trace,
               const double* inv_temp, const double* energy_value_out, const int* ih_solver_parameters) {
  solver_wom(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_InverseSquareRoot), *get<PSMatrix>(ih_Density), *inv_temp,
             trace, nullptr, const_cast<double*>(energy_value_out), *get<SolverParameters>(ih_solver_parameters));
}
// EigenSolvers_c.h:4-15 -- positions as the Fortran wrapper binds them (EigenSolversModule_wrp.F90:19-62: matrix,
// eigenvalues, nvals, eigenvectors), which is how the C++ layer calls them (EigenSolvers.cc:12-21); the parameter
// NAMES in the C header are swapped
void EigenDecomposition_wrp(const int* ih_this, int* ih_eigenvalues, const int* nvals, int* ih_eigenvectors,
                            const int* ih_solver_parameters) {
  ps_eigendecomposition(*get<PSMatrix>(ih_this), *get<PSMatrix>(ih_eigenvalues), get<PSMatrix>(ih_eigenvectors), *nvals,
                        *get<SolverParameters>(ih_solver_parameters));
}
void EigenDecomposition_novec_wrp(const int* ih_this, int* ih_eigenvalues, const int* nvals, const int* ih_solver_parameters) {
  ps_eigendecomposition(*get<PSMatrix>(ih_this), *get<PSMatrix>(ih_eigenvalues), nullptr, *nvals,
                        *get<SolverParameters>(ih_solver_parameters));
}
void SingularValueDecompostion_wrp(const int* ih_this, int* ih_leftvectors, int* ih_rightvectors, int* ih_singularvalues,
                                   const int* ih_solver_parameters) {
  ps_svd(*get<PSMatrix>(ih_this), *get<PSMatrix>(ih_leftvectors), *get<PSMatrix>(ih_rightvectors),
         *get<PSMatrix>(ih_singularvalues), *get<SolverParameters>(ih_solver_parameters));
}
void EstimateGap_wrp(const int* ih_H, const int* ih_K, const double* chemical_potential, double* gap,
                     const int* ih_solver_parameters) {
  estimate_gap(*get<PSMatrix>(ih_H), *get<PSMatrix>(ih_K), *chemical_potential, gap, *get<SolverParameters>(ih_solver_parameters));
}
// LinearSolvers_c.h:4-7, Analysis_c.h:4-8
void CGSolver_wrp(const int* ih_MatA, int* ih_MatX, const int* ih_matB, const int* ih_solver_parameters) {
  solver_cg(*get<PSMatrix>(ih_MatA), *get<PSMatrix>(ih_MatX), *get<PSMatrix>(ih_matB), *get<SolverParameters>(ih_solver_parameters));
}
void CholeskyDecomposition_wrp(const int* ih_MatA, int* ih_MatL, const int* ih_solver_parameters) {
  ps_cholesky(*get<PSMatrix>(ih_MatA), *get<PSMatrix>(ih_MatL), -1, *get<SolverParameters>(ih_solver_parameters));
}
void PivotedCholeskyDecomposition_wrp(const int* ih_MatA, int* ih_MatL, const int* rank_in, const int* ih_solver_parameters) {
  ps_cholesky(*get<PSMatrix>(ih_MatA), *get<PSMatrix>(ih_MatL), *rank_in, *get<SolverParameters>(ih_solver_parameters));
}
void ReduceDimension_wrp(const int* ih_this, const int* dim, int* ih_reduced, const int* ih_solver_parameters) {
  reduce_dimension(*get<PSMatrix>(ih_this), *dim, *get<PSMatrix>(ih_reduced), *get<SolverParameters>(ih_solver_parameters));
}
// GeometryOptimization_c.h:4-10, MatrixConversion_c.h:4
void PurificationExtrapolate_wrp(const int* ih_PreviousDensity, const int* Overlap, const double* trace, int* ih_NewDensity,
                                 const int* ih_solver_parameters) {
  purification_extrapolate(*get<PSMatrix>(ih_PreviousDensity), *get<PSMatrix>(Overlap), *trace, *get<PSMatrix>(ih_NewDensity),
                           *get<SolverParameters>(ih_solver_parameters));
}
void LowdinExtrapolate_wrp(const int* ih_PreviousDensity, const int* OldOverlap, const int* NewOverlap, int* ih_NewDensity,
                           const int* ih_solver_parameters) {
  lowdin_extrapolate(*get<PSMatrix>(ih_PreviousDensity), *get<PSMatrix>(OldOverlap), *get<PSMatrix>(NewOverlap),
                     *get<PSMatrix>(ih_NewDensity), *get<SolverParameters>(ih_solver_parameters));
}
void SnapMatrixToSparsityPattern_wrp(int* ih_matA, const int* ih_matB) {
  snap_to_sparsity_pattern(*get<PSMatrix>(ih_matA), *get<PSMatrix>(ih_matB));
}

// ---- matrix polynomials: Polynomial_c.h:4-13, ChebyshevSolvers_c.h:4-15, HermiteSolvers_c.h:4-11
// (wrappers PolynomialSolversModule_wrp.F90:27-90, ChebyshevSolversModule_wrp.F90, HermiteSolversModule_wrp.F90).
// A polynomial handle holds `degree` coefficients; SetCoefficient takes a 1-based position (the C++ layer adds 1).
struct PolyHandle {
  std::vector<double> c;
};
static void poly_construct(int* ih_polynomial, int degree) {
  auto* h = new PolyHandle();
  h->c.assign((size_t)std::max(0, degree), 0.0);
  put(ih_polynomial, h);
}
static void poly_destruct(int* ih_polynomial) {
  delete get<PolyHandle>(ih_polynomial);
  std::memset(ih_polynomial, 0, sizeof(int) * SIZE_wrp);
}
static void poly_set(int* ih_polynomial, int degree, double coefficient) {
  PolyHandle* h = get<PolyHandle>(ih_polynomial);
  if (degree < 1 || degree > (int)h->c.size()) NTP_FATAL("SetCoefficient: degree out of range");
  h->c[(size_t)degree - 1] = coefficient;
}
void ConstructPolynomial_wrp(int* ih_polynomial, const int* degree) { poly_construct(ih_polynomial, *degree); }
void DestructPolynomial_wrp(int* ih_polynomial) { poly_destruct(ih_polynomial); }
void SetCoefficient_wrp(int* ih_polynomial, const int* degree, const double* coefficient) {
  poly_set(ih_polynomial, *degree, *coefficient);
}
void ConstructChebyshevPolynomial_wrp(int* ih_polynomial, const int* degree) { poly_construct(ih_polynomial, *degree); }
void DestructChebyshevPolynomial_wrp(int* ih_polynomial) { poly_destruct(ih_polynomial); }
void SetChebyshevCoefficient_wrp(int* ih_polynomial, const int* degree, const double* coefficient) {
  poly_set(ih_polynomial, *degree, *coefficient);
}
void ConstructHermitePolynomial_wrp(int* ih_polynomial, const int* degree) { poly_construct(ih_polynomial, *degree); }
void DestructHermitePolynomial_wrp(int* ih_polynomial) { poly_destruct(ih_polynomial); }
void SetHermiteCoefficient_wrp(int* ih_polynomial, const int* degree, const double* coefficient) {
  poly_set(ih_polynomial, *degree, *coefficient);
}
void HornerCompute_wrp(const int* ih_InputMat, int* ih_OutputMat, const int* ih_polynomial, const int* ih_solver_parameters) {
  polynomial_horner(*get<PSMatrix>(ih_InputMat), *get<PSMatrix>(ih_OutputMat), get<PolyHandle>(ih_polynomial)->c,
     *get<SolverParameters>(ih_solver_parameters));
}
void PatersonStockmeyerCompute_wrp(const int* ih_InputMat, int* ih_OutputMat, const int* ih_polynomial, const int* ih_solver_parameters) {
  polynomial_paterson_stockmeyer(*get<PSMatrix>(ih_InputMat), *get<PSMatrix>(ih_OutputMat), get<PolyHandle>(ih_polynomial)->c,
     *get<SolverParameters>(ih_solver_parameters));
}
void ChebyshevCompute_wrp(const int* ih_InputMat, int* ih_OutputMat, const int* ih_polynomial, const int* ih_solver_parameters) {
  chebyshev_compute(*get<PSMatrix>(ih_InputMat), *get<PSMatrix>(ih_OutputMat), get<PolyHandle>(ih_polynomial)->c,
     *get<SolverParameters>(ih_solver_parameters));
}
void FactorizedChebyshevCompute_wrp(const int* ih_InputMat, int* ih_OutputMat, const int* ih_polynomial, const int* ih_solver_parameters) {
  chebyshev_factorized(*get<PSMatrix>(ih_InputMat), *get<PSMatrix>(ih_OutputMat), get<PolyHandle>(ih_polynomial)->c,
     *get<SolverParameters>(ih_solver_parameters));
}
void HermiteCompute_wrp(const int* ih_InputMat, int* ih_OutputMat, const int* ih_polynomial, const int* ih_solver_parameters) {
  hermite_compute(*get<PSMatrix>(ih_InputMat), *get<PSMatrix>(ih_OutputMat), get<PolyHandle>(ih_polynomial)->c,
     *get<SolverParameters>(ih_solver_parameters));
}

// ---- matrix functions: ExponentialSolvers_c.h, TrigonometrySolvers_c.h, RootSolvers_c.h, EigenBounds_c.h
void ComputeExponential_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  compute_exponential(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters));
}
void ComputeLogarithm_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  compute_logarithm(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters));
}
void Sine_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  compute_sine(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters));
}
void Cosine_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  compute_cosine(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters));
}
void ComputeRoot_wrp(const int* ih_inputmat, int* ih_outputmat, const int* root, const int* ih_solver_parameters) {
  compute_root(*get<PSMatrix>(ih_inputmat), *get<PSMatrix>(ih_outputmat), *root, *get<SolverParameters>(ih_solver_parameters));
}
void ComputeInverseRoot_wrp(const int* ih_inputmat, int* ih_outputmat, const int* root, const int* ih_solver_parameters) {
  compute_inverse_root(*get<PSMatrix>(ih_inputmat), *get<PSMatrix>(ih_outputmat), *root,
                       *get<SolverParameters>(ih_solver_parameters));
}
void PowerBounds_wrp(const int* ih_Hamiltonian, double* max_value, const int* ih_solver_parameters) {
  power_bounds(*get<PSMatrix>(ih_Hamiltonian), max_value, *get<SolverParameters>(ih_solver_parameters), false);
}
// dense (eigendecomposition) variants: ExponentialSolversModule.F90:372-404,637-669, TrigonometrySolversModule.F90:66-150,
// InverseSolversModule.F90:152-181, SignSolversModule.F90:67-96
void ComputeExponentialPade_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  compute_exponential_pade(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters));
}
void ComputeDenseExponential_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Exponential Solver", [](double v) { return std::exp(v); });
}
void ComputeDenseLogarithm_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Logarithm Solver", [](double v) { return std::log(v); });
}
void DenseSine_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Trigonometry Solver", [](double v) { return std::sin(v); });
}
void DenseCosine_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Trigonometry Solver", [](double v) { return std::cos(v); });
}
void DenseInvert_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Inverse Solver", [](double v) { return 1.0 / v; });
}
void DenseSignFunction_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  dense_function(ih_Input, ih_Output, ih_solver_parameters, "Sign Function Solver", [](double v) { return v < 0.0 ? -1.0 : 1.0; });
}

void SignFunction_wrp(const int* ih_mat1, int* ih_signmat, const int* ih_solver_parameters) {
  solver_sign(*get<PSMatrix>(ih_mat1), *get<PSMatrix>(ih_signmat), *get<SolverParameters>(ih_solver_parameters));
}
void PolarDecomposition_wrp(const int* ih_mat1, int* ih_umat, int* ih_hmat, const int* ih_solver_parameters) {
  solver_polar(*get<PSMatrix>(ih_mat1), *get<PSMatrix>(ih_umat), get<PSMatrix>(ih_hmat),
               *get<SolverParameters>(ih_solver_parameters));
}
void Invert_wrp(const int* ih_Hamiltonian, int* ih_Inverse, const int* ih_solver_parameters) {
  solver_invert(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_Inverse), *get<SolverParameters>(ih_solver_parameters));
}
void PseudoInverse_wrp(const int* ih_Hamiltonian, int* ih_Inverse, const int* ih_solver_parameters) {
  solver_pseudoinverse(*get<PSMatrix>(ih_Hamiltonian), *get<PSMatrix>(ih_Inverse),
                       *get<SolverParameters>(ih_solver_parameters));
}
void SquareRoot_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  solver_square_root(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters),
                     false, 5);
}
void InverseSquareRoot_wrp(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters) {
  solver_square_root(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters),
                     true, 5);
}
// extension: one TRS2 iteration on caller-held matrices (what TRS2_wrp runs inside its loop); lets a
// driver time exactly K iterations.  X and X2 must be constructed; returns energy and sigma.
// trace_io: in = trace(X) when the caller has it from the previous step (NaN: computed here), out = trace of the new X
// (accumulated in the pass that evaluates the energy), exactly what the solver loop hands from iteration to iteration.
void ntpoly_amd_trs2_step(int* ih_X, int* ih_X2, const int* ih_WH, const double* trace, const double* threshold,
                          double* energy_out, double* sigma_out, double* trace_io) {
  *energy_out = trs2_step(*get_unpacked(ih_X), *get<PSMatrix>(ih_X2), *get<PSMatrix>(ih_WH), *trace, *threshold, sigma_out,
                          trace_io);
}
// extension: the reference's optional order_in argument (SquareRootSolversModule.F90:30-61) is not
// reachable through its C ABI; expose it for tests
void ntpoly_amd_square_root_order(const int* ih_Input, int* ih_Output, const int* ih_solver_parameters,
                                  const int* inverse, const int* order) {
  solver_square_root(*get<PSMatrix>(ih_Input), *get<PSMatrix>(ih_Output), *get<SolverParameters>(ih_solver_parameters),
                     *inverse != 0, *order);
}

// ===================================================================== SMatrix_c.h (local matrices, config 2)
#define LOCAL_API(SUF, CPLX)                                                                                     \
  void ConstructMatrixFromTripletList_##SUF##_wrp(int* ih_this, const int* ih_triplet_list, const int* rows,     \
                                                  const int* columns) {                                          \
    ensure_init();                                                                                               \
    LocalMat* m = new LocalMat();                                                                                \
    m->m = from_triplets(*get<HostTriplets>(ih_triplet_list), *rows, *columns, 0);                               \
    put(ih_this, m);                                                                                             \
  }                                                                                                              \
  void ConstructZeroMatrix_##SUF##_wrp(int* ih_this, const int* rows, const int* columns) {                      \
    ensure_init();                                                                                               \
    LocalMat* m = new LocalMat();                                                                                \
    m->m.reset_empty(*rows, *columns, CPLX);                                                                     \
    put(ih_this, m);                                                                                             \
  }                                                                                                              \
  void ConstructMatrixFromFile_##SUF##_wrp(int* ih_this, const char* file_name, const int* name_size) {          \
    ensure_init();                                                                                               \
    LocalMat* m = new LocalMat();                                                                                \
    int rows = 0, cols = 0;                                                                                      \
    HostTriplets t;                                                                                              \
    read_matrix_market_file(fstring(file_name, name_size), t, &rows, &cols, CPLX);                               \
    m->m = from_triplets(t, rows, cols, 0);                                                                      \
    put(ih_this, m);                                                                                             \
  }                                                                                                              \
  void DestructMatrix_##SUF##_wrp(int* ih_this) { delete get<LocalMat>(ih_this); }                               \
  void CopyMatrix_##SUF##_wrp(const int* ih_matA, int* ih_matB) {                                                \
    get<LocalMat>(ih_matB)->m = get<LocalMat>(ih_matA)->m.clone();                                               \
  }                                                                                                              \
  void GetMatrixRows_##SUF##_wrp(const int* ih_this, int* rows) { *rows = get<LocalMat>(ih_this)->m.rows; }      \
  void GetMatrixColumns_##SUF##_wrp(const int* ih_this, int* columns) { *columns = get<LocalMat>(ih_this)->m.cols; } \
  void ScaleMatrix_##SUF##_wrp(int* ih_this, const double* constant) { scale(get<LocalMat>(ih_this)->m, *constant); } \
  void IncrementMatrix_##SUF##_wrp(const int* ih_matA, int* ih_matB, const double* alpha_in,                     \
                                   const double* threshold_in) {                                                 \
    increment(get<LocalMat>(ih_matA)->m, get<LocalMat>(ih_matB)->m, *alpha_in, *threshold_in);                   \
  }                                                                                                              \
  void PairwiseMultiplyMatrix_##SUF##_wrp(const int* ih_matA, const int* ih_matB, int* ih_matC) {                \
    pairwise(get<LocalMat>(ih_matA)->m, get<LocalMat>(ih_matB)->m, get<LocalMat>(ih_matC)->m, false);            \
  }                                                                                                              \
  void TransposeMatrix_##SUF##_wrp(const int* ih_matA, int* ih_matAT) {                                          \
    get<LocalMat>(ih_matAT)->m = transpose(get<LocalMat>(ih_matA)->m);                                           \
  }                                                                                                              \
  void MatrixToTripletList_##SUF##_wrp(const int* ih_this, int* ih_triplet_list) {                               \
    to_triplets(get<LocalMat>(ih_this)->m, 0, *get<HostTriplets>(ih_triplet_list));                              \
  }                                                                                                              \
  /* ExtractMatrixRow.f90 / ExtractMatrixColumn.f90 (1-based number), MatrixDiagonalScale (column col *= value), */ \
  /* PrintMatrix.f90 (MatrixMarket text to stdout or a file) */                                                  \
  void ExtractMatrixColumn_##SUF##_wrp(const int* ih_this, int* column_number, int* ih_column_out) {             \
    get<LocalMat>(ih_column_out)->m = column_slice(get<LocalMat>(ih_this)->m, *column_number - 1, *column_number); \
  }                                                                                                              \
  void ExtractMatrixRow_##SUF##_wrp(const int* ih_this, int* row_number, int* ih_row_out) {                      \
    DevMat t = transpose(get<LocalMat>(ih_this)->m);                                                             \
    DevMat c = column_slice(t, *row_number - 1, *row_number);                                                    \
    get<LocalMat>(ih_row_out)->m = transpose(c);                                                                 \
  }                                                                                                              \
  void MatrixDiagonalScale_##SUF##_wrp(int* ih_mat, const int* ih_tlist) {                                       \
    local_diagonal_scale(get<LocalMat>(ih_mat)->m, *get<HostTriplets>(ih_tlist));                                \
  }                                                                                                              \
  void PrintMatrix_##SUF##_wrp(const int* ih_this) { local_print(get<LocalMat>(ih_this)->m, nullptr); }          \
  void PrintMatrixF_##SUF##_wrp(const int* ih_this, const char* file_name, const int* name_size) {                \
    const std::string path = fstring(file_name, name_size);                                                      \
    local_print(get<LocalMat>(ih_this)->m, path.c_str());                                                        \
  }                                                                                                              \
  /* GemmMatrix (sparse_includes/GemmMatrix.f90:1-101) */                                                        \
  void MatrixMultiply_##SUF##_wrp(const int* ih_matA, const int* ih_matB, int* ih_matC, const bool* IsATransposed, \
                                  const bool* IsBTransposed, const double* alpha, const double* beta,            \
                                  const double* threshold, int* ih_matrix_memory_pool) {                         \
    (void)ih_matrix_memory_pool;                                                                                 \
    const DevMat& A0 = get<LocalMat>(ih_matA)->m;                                                                \
    const DevMat& B0 = get<LocalMat>(ih_matB)->m;                                                                \
    DevMat& C = get<LocalMat>(ih_matC)->m;                                                                       \
    const double sa = (double)A0.nnz / ((double)A0.rows * (double)A0.cols);                                      \
    const double sb = (double)B0.nnz / ((double)B0.rows * (double)B0.cols);                                      \
    const bool dense_rule = std::min(sa, sb) > 0.1;                                                              \
    DevMat At, Bt;                                                                                               \
    if (*IsATransposed) At = transpose(A0);                                                                      \
    if (*IsBTransposed) Bt = transpose(B0);                                                                      \
    DevMat AB;                                                                                                   \
    spgemm(*IsATransposed ? At : A0, *IsBTransposed ? Bt : B0, AB, *alpha, *threshold, dense_rule);              \
    if (std::fabs(*beta) > 0 && C.rows == AB.rows && C.cols == AB.cols) {                                        \
      scale(C, *beta);                                                                                           \
      increment(AB, C, 1.0, 0.0);                                                                                \
    } else {                                                                                                     \
      C = std::move(AB);                                                                                         \
    }                                                                                                            \
  }

LOCAL_API(lsr, false)
LOCAL_API(lsc, true)

void DotMatrix_lsr_wrp(const int* ih_matA, const int* ih_matB, double* product) {
  double out[2];
  dot(get<LocalMat>(ih_matA)->m, get<LocalMat>(ih_matB)->m, out);
  *product = out[0];
}
void DotMatrix_lsc_wrp(const int* ih_matA, const int* ih_matB, double* product_real, double* product_complex) {
  double out[2];
  dot(get<LocalMat>(ih_matA)->m, get<LocalMat>(ih_matB)->m, out);
  *product_real = out[0];
  *product_complex = out[1];
}
void ConjugateMatrix_lsc_wrp(int* ih_matA) { conjugate(get<LocalMat>(ih_matA)->m); }

}  // extern "C"
