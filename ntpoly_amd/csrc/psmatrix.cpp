// Distributed matrix (Matrix_ps) and distributed algebra on column panels.
// Reference: PSMatrixModule.F90, PSMatrixAlgebraModule.F90, ProcessGridModule.F90,
// LoadBalancerModule.F90, PermutationModule.F90.
#include <chrono>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <numeric>
#include <random>

#include "engine.hpp"
#include "spgemm_block.hpp"

namespace ntp {

namespace {
// non-owning window on the arrays of another DevMat: a column range of it, or the same entries under another
// column-offset array.  The kernels only ever index inner / val through the (absolute) offsets in `outer`.
struct MatView {
  DevMat m;
  void alias(const DevMat& src, int64_t* outer, int32_t cols, int64_t nnz) {
    m.rows = src.rows;
    m.cols = cols;
    m.cplx = src.cplx;
    m.nnz = nnz;
    m.outer.p = outer;
    m.inner.p = src.inner.p;
    m.val.p = src.val.p;
  }
  ~MatView() { m.outer.p = nullptr; m.inner.p = nullptr; m.val.p = nullptr; }
};
}  // namespace

// ------------------------------------------------------------------ process grid
namespace {
ProcessGrid g_grid;
bool g_grid_built = false;
}  // namespace
ProcessGrid& global_grid() { return g_grid; }
bool global_grid_constructed() { return g_grid_built; }

void construct_grid(ProcessGrid& g, int rows, int cols, int slices) {
  const Comm& c = world();
  // grid sanity check (ProcessGridModule.F90:162-176): fatal if the shape does not match
  if (rows * cols * slices != c.nranks && !options().virtual_grid)
    NTP_FATAL("process grid " + std::to_string(rows) + "x" + std::to_string(cols) + "x" + std::to_string(slices) +
              " does not match " + std::to_string(c.nranks) + " processes");
  g.num_rows = rows;
  g.num_cols = cols;
  g.num_slices = slices;
  g.total = c.nranks;
  g.global_rank = c.rank;
  // rank -> (slice, row, column) as ProcessGridModule.F90:180-183
  const int slice_size = rows * cols;
  g.my_slice = c.rank / slice_size;
  const int in_slice = c.rank - slice_size * g.my_slice;
  g.my_row = in_slice / cols;
  g.my_col = in_slice % cols;
  if (&g == &g_grid) g_grid_built = true;
}

void construct_grid_default(ProcessGrid& g, int slices) {
  // ComputeGridSize (ProcessGridModule.F90:576-601): most square rows x cols for the given slices
  const int total = world().nranks;
  if (slices <= 0) slices = 1;
  while (total % slices != 0) --slices;
  const int slice_size = total / slices;
  int rows = 1, cols = slice_size;
  for (int r = (int)std::floor(std::sqrt((double)slice_size)); r >= 1; --r) {
    if (slice_size % r == 0) {
      rows = r;
      cols = slice_size / r;
      break;
    }
  }
  construct_grid(g, rows, cols, slices);
}

void use_grid_comm(const ProcessGrid* g) { use_comm(g ? g->comm : nullptr); }

ProcessGrid* split_process_grid(const ProcessGrid& old_grid, int* my_color, bool* split_slice) {
  static std::vector<ProcessGrid*>* kept = new std::vector<ProcessGrid*>();   // (grids made here live as long as the library)
  use_grid_comm(&old_grid);
  int rows = 1, cols = 1, slices = 1, color = 0;
  *split_slice = false;
  if (old_grid.total == 1) {
    // base case (:447-453)
  } else if (old_grid.num_slices > 1) {          // preferably along the slices (:455-468)
    const int mid = old_grid.num_slices / 2;
    rows = old_grid.num_rows;
    cols = old_grid.num_cols;
    color = old_grid.my_slice < mid ? 0 : 1;
    slices = color == 0 ? mid : old_grid.num_slices - mid;
    *split_slice = true;
  } else if (old_grid.num_rows > old_grid.num_cols) {   // else the longer direction (:470-482)
    const int mid = old_grid.num_rows / 2;
    cols = old_grid.num_cols;
    color = old_grid.my_row < mid ? 0 : 1;
    rows = color == 0 ? mid : old_grid.num_rows - mid;
  } else {                                       // default: the columns (:484-496)
    const int mid = old_grid.num_cols / 2;
    rows = old_grid.num_rows;
    color = old_grid.my_col < mid ? 0 : 1;
    cols = color == 0 ? mid : old_grid.num_cols - mid;
  }
  *my_color = color;
  // MPI_COMM_SPLIT(global_comm, my_color, global_rank) (:499-501): the processes of a colour in the order of their old ranks
  Comm* nc = comm_split(color, old_grid.global_rank);
  auto* g = new ProcessGrid();
  kept->push_back(g);
  use_comm(nc);
  construct_grid(*g, rows, cols, slices);
  g->comm = nc;
  use_grid_comm(&old_grid);
  return g;
}

void ps_comm_split(const PSMatrix& m, PSMatrix& split, int* my_color, bool* split_slice) {
  const ProcessGrid& g = *m.grid;
  use_grid_comm(&g);
  if (g.total == 1) {   // (distributed_includes/CommSplitMatrix.f90:11-14)
    ps_copy(m, split);
    *my_color = 0;
    *split_slice = true;
    return;
  }
  // every process of either half ends up with a share of the WHOLE matrix (:20-60: the triplets of the other half travel over
  // the between-grid communicator); here: the panels gathered once over the old grid, each process cuts the column panel it
  // owns on its half
  DevMat full = ps_gather_full(m);
  ProcessGrid* ng = split_process_grid(g, my_color, split_slice);
  use_grid_comm(ng);
  ps_construct_empty(split, m.dim, ng, m.cplx);
  split.loc = column_slice(full, split.c0, split.c1);
  sync_stream();
  use_grid_comm(&g);
}

void write_grid_info(const ProcessGrid& g) {
  log_header("Process Grid");
  log_enter();
  log_element("Process Rows", g.num_rows);
  log_element("Process Columns", g.num_cols);
  log_element("Process Slices", g.num_slices);
  log_element("Column Panels (GPUs)", g.total);
  log_exit();
}

// ------------------------------------------------------------------ construction / fill
void panel_range(int32_t dim, int nranks, int rank, int32_t* c0, int32_t* c1) {
  *c0 = (int32_t)(((int64_t)dim * rank) / nranks);
  *c1 = (int32_t)(((int64_t)dim * (rank + 1)) / nranks);
}

// The layout of one panel exchange on rank `me`, from what every rank knows after the step's all-gather: req[4 q ..] = (first,
// last row of rank q's panel of B, ...), cnt[s P + q] = doubles rank s sends to rank q.  sa / sb [q]: my columns [sa, sb) go to
// rank q, packed from soff[q] in my send buffer; ra / rb [s]: the columns [ra, rb) of rank s arrive at zoff[s] of my receive
// buffer (nothing travels from a rank to itself).  Pure host arithmetic (tests/test_distributed_cpu.py drives it over gloo).
void panel_exchange_layout(int32_t dim, int P, int me, const int64_t* req, const int64_t* cnt, int32_t* sa, int32_t* sb, int64_t* soff,
                           int32_t* ra, int32_t* rb, int64_t* zoff) {
  auto kmin_of = [&](int q) { const int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? 0 : (int32_t)lo; };
  auto kmax_of = [&](int q) { const int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? -1 : (int32_t)hi; };
  soff[0] = 0;
  zoff[0] = 0;
  for (int q = 0; q < P; ++q) {
    halo_segment(dim, P, me, kmin_of(q), kmax_of(q), &sa[q], &sb[q]);
    soff[q + 1] = soff[q] + (q == me ? 0 : cnt[(size_t)me * P + q]);
  }
  for (int s = 0; s < P; ++s) {
    halo_segment(dim, P, s, kmin_of(me), kmax_of(me), &ra[s], &rb[s]);
    zoff[s + 1] = zoff[s] + (s == me ? 0 : cnt[(size_t)s * P + me]);
  }
}

void ps_construct_empty(PSMatrix& m, int32_t dim, const ProcessGrid* g, bool cplx) {
  if (!g) NTP_FATAL("matrix constructed without a process grid (construct the global grid first)");
  ensure_init();
  use_grid_comm(g);   // (the matrix lives on its grid's communicator: what follows on it runs there)
  m.grid = g;
  m.dim = dim;
  m.cplx = cplx;
  panel_range(dim, world().nranks, world().rank, &m.c0, &m.c1);
  m.loc.reset_empty(dim, m.c1 - m.c0, cplx);
}

void ps_construct_like(PSMatrix& m, const PSMatrix& ref) {
  use_grid_comm(ref.grid); ps_construct_empty(m, ref.dim, ref.grid, ref.cplx); }

namespace {
int g_slab_depth = 0;
bool g_slab_failed = false;
bool slab_on() { return g_slab_depth > 0 && !g_slab_failed; }
// (the representation of an operand, not its value, changes: const operands are converted in place)
DevMat& mut(const PSMatrix& m) { return const_cast<DevMat&>(m.loc); }
int g_slab_refusals = 0;
bool g_complex_session = false;   // the open session's loop takes complex operands in slab form (SlabSession complex_ok)
long long g_slab_counts[4] = {0, 0, 0, 0};   // products, merges / copies, other operations in slab form; refusals
// an operation that cannot be done in slab form: its operands go back to compressed columns and the general path does
// it (a Hamiltonian with stored zeros in the first merge of a loop); the session goes on, unless this keeps happening
void slab_refused(std::initializer_list<const PSMatrix*> ms) {
  g_slab_refusals += 1;
  g_slab_counts[3] += 1;
  if (g_slab_refusals > 4) g_slab_failed = true;
  if (std::getenv("NTPOLY_AMD_DEBUG_SPGEMM"))
    std::fprintf(stderr, "[slab session] an operation was refused (%d so far)%s\n", g_slab_refusals,
                 g_slab_failed ? ": compressed columns from here on" : "");
  for (const PSMatrix* m : ms) pack(mut(*m));
}
// Block form (DevMat::blk: what the block path's products leave behind for the C ABI's MatrixMultiply and the TRS2 loop)
// is understood by ps_multiply and the TRS2 steps only: every other operation takes compressed columns
void unblock(std::initializer_list<const PSMatrix*> ms) {
  for (const PSMatrix* m : ms)
    if (m->loc.blocked()) pack(mut(*m));
}
// the block algebra (spgemm_block.hpp) takes an operation when an operand is in block form (one rank)
bool blk_any(std::initializer_list<const PSMatrix*> ms) {
  if (world().active()) return false;
  for (const PSMatrix* m : ms)
    if (m->loc.blocked()) return true;
  return false;
}
long long g_column_fused[2] = {0, 0};   // in-place identity increments, norms of differences (column_fused.hip)
long long g_block_counts[2] = {0, 0};   // operations of the block algebra; fallbacks to compressed columns
// an operation outside the session, or after a refusal: no operand may stay in slab form
void slab_pack_if(std::initializer_list<const PSMatrix*> ms) {
  if (g_slab_depth == 0) return;   // (outside a session nothing is left in slab form by one)
  for (const PSMatrix* m : ms)
    if (m->loc.expanded() || m->loc.loose() || m->loc.blocked()) pack(mut(*m));
}
}  // namespace

const long long* slab_algebra_counts() { return g_slab_counts; }
const long long* block_algebra_counts() { return g_block_counts; }
const long long* column_fused_counts() { return g_column_fused; }

SlabSession::SlabSession(bool eligible, bool api, bool complex_ok) {
  // (more than one rank: the loops' matrices are column panels in slab form, a product exchanges the runs of its left
  // operand's halo -- panel_slab_multiply below; FMA arithmetic, real, the solvers' own sessions only: option panel_sessions)
  const bool multi = world().active();
  opened = eligible && options().slab_algebra != 0 && (options().spgemm_fma == 1 || options().spgemm_fma == 0) && options().spgemm_variant < 0 &&
           options().spgemm_force_bin <= 0 && (!multi || (options().panel_sessions != 0 && options().spgemm_fma == 1 && !api));
  if (opened && multi && g_slab_depth == 0) slab_allow_panels(true);
  if (opened && !multi && complex_ok && options().complex_sessions != 0 && options().spgemm_fma == 1 && options().complex_tile != 0 && !g_complex_session) {
    g_complex_session = true;
    set_complex = true;
  }
  // (a one-call session of the C ABI on a matrix that is not run-like pays the refused conversion once: slab_enter leaves a
  // mark on the matrix, DevMat::slab_hint, and says no at once when asked again)
  if (opened) {
    if (g_slab_depth == 0) { g_slab_failed = false; g_slab_refusals = 0; }
    g_slab_depth += 1;
  }
}
SlabSession::~SlabSession() { close(); }
void SlabSession::close() {
  if (set_complex) { g_complex_session = false; set_complex = false; }
  if (opened) {
    g_slab_depth -= 1;
    if (g_slab_depth == 0) slab_allow_panels(false);
  }
  opened = false;
}
void ps_slab_leave(PSMatrix& m) {
  if (m.loc.expanded() || m.loc.blocked()) pack(m.loc);
  drop_pending_exchange();   // (the iterate leaves slab form: no step follows the last one)
}

void ps_copy(const PSMatrix& a, PSMatrix& b) {
  use_grid_comm(a.grid);
  if (&a == &b) return;
  if (blk_any({&a})) {
    DevMat t;
    if (block_clone(a.loc, t)) {
      g_block_counts[0] += 1;
      b.grid = a.grid; b.dim = a.dim; b.cplx = a.cplx; b.c0 = a.c0; b.c1 = a.c1;
      b.loc = std::move(t);
      return;
    }
    g_block_counts[1] += 1;
  }
  unblock({&a});
  if (slab_on() && g_complex_session && a.cplx && a.loc.expanded()) {   // (a session that takes complex operands)
    DevMat t;
    if (slab_clone_c(a.loc, t)) {
      g_slab_counts[1] += 1;
      b.grid = a.grid; b.dim = a.dim; b.cplx = true; b.c0 = a.c0; b.c1 = a.c1;
      b.loc = std::move(t);
      return;
    }
  }
  if (slab_on() && a.loc.expanded()) {
    DevMat t;
    if (slab_clone(a.loc, t)) {
      g_slab_counts[1] += 1;
      b.grid = a.grid; b.dim = a.dim; b.cplx = a.cplx; b.c0 = a.c0; b.c1 = a.c1;
      b.loc = std::move(t);
      return;
    }
    slab_refused({&a});
  }
  DevMat t = a.loc.clone();
  b.grid = a.grid;
  b.dim = a.dim;
  b.cplx = a.cplx;
  b.c0 = a.c0;
  b.c1 = a.c1;
  b.loc = std::move(t);
}

void ps_fill_identity(PSMatrix& m) {
  use_grid_comm(m.grid);  // FillMatrixIdentity (O(N) here, O(N^2/P) in the reference)
  m.loc = identity(m.dim, m.c0, m.c1 - m.c0, m.cplx);
}

void ps_fill_permutation(PSMatrix& m, const std::vector<int32_t>& lookup, bool rows) {
  use_grid_comm(m.grid);
  // distributed_includes/FillMatrixPermutation.f90:1-35
  HostTriplets t;
  t.cplx = m.cplx;
  const size_t w = m.cplx ? 2 : 1;
  for (int32_t ii = 1; ii <= m.dim; ++ii) {
    const int32_t col = rows ? lookup[(size_t)ii - 1] : ii;
    const int32_t row = rows ? ii : lookup[(size_t)ii - 1];
    if (col - 1 >= m.c0 && col - 1 < m.c1) {
      t.col.push_back(col);
      t.row.push_back(row);
      t.val.push_back(1.0);
      if (w == 2) t.val.push_back(0.0);
    }
  }
  m.loc = from_triplets(t, m.dim, m.c1 - m.c0, m.c0);
}

void ps_fill_from_triplets(PSMatrix& m, const HostTriplets& t) {
  use_grid_comm(m.grid);
  // FillMatrixFromTripletList (distributed_includes/FillMatrixFromTripletList.f90:14-47): any rank
  // may hold any triplet.  Ranks exchange what they hold (setup path, host triplets travel through
  // device buffers because RCCL moves device memory) and keep their own columns.
  if (t.cplx != m.cplx) {
    HostTriplets conv;
    conv.cplx = m.cplx;
    conv.col = t.col;
    conv.row = t.row;
    if (m.cplx) {
      conv.val.resize(t.size() * 2);
      for (size_t i = 0; i < t.size(); ++i) {
        conv.val[2 * i] = t.val[i];
        conv.val[2 * i + 1] = 0.0;
      }
    } else {
      conv.val.resize(t.size());
      for (size_t i = 0; i < t.size(); ++i) conv.val[i] = t.val[2 * i];
    }
    ps_fill_from_triplets(m, conv);
    return;
  }
  for (size_t i = 0; i < t.size(); ++i)
    if (t.col[i] < 1 || t.col[i] > m.dim)
      NTP_FATAL("triplet " + std::to_string(i) + " names column " + std::to_string(t.col[i]) + " of a matrix of dimension " +
                std::to_string(m.dim));
  if (!world().active()) {
    m.loc = from_triplets(t, m.dim, m.c1 - m.c0, m.c0);
    return;
  }
  // multi-rank: every rank turns what it holds into a full-width matrix, all of them are gathered
  // (one grouped broadcast per rank) and each rank sums the pieces that fall into its own panel.
  // Contributions of different ranks are disjoint by contract.
  const int P = world().nranks;
  DevMat mine = from_triplets(t, m.dim, m.dim, 0);
  std::vector<int32_t> widths((size_t)P, m.dim);
  DevMat all = gather_panels(mine, widths);  // dim x (P*dim), rank r's matrix at columns [r*dim, (r+1)*dim)
  DevMat acc;
  acc.reset_empty(m.dim, m.c1 - m.c0, m.cplx);
  for (int r = 0; r < P; ++r) {
    DevMat part = column_slice(all, r * m.dim + m.c0, r * m.dim + m.c1);
    if (part.nnz) increment(part, acc, 1.0, 0.0);
  }
  m.loc = std::move(acc);
}

void ps_get_triplets(const PSMatrix& m, HostTriplets& t) {
  use_grid_comm(m.grid); to_triplets(m.loc, m.c0, t); }

// FillMatrixDense (PSMatrixModule.F90:958-990, distributed_includes/FillMatrixDense.f90): every element of the
// local panel is 1; dense by definition, so the O(dim * width) host triplets are what the caller asked for
void ps_fill_dense(PSMatrix& m) {
  use_grid_comm(m.grid);
  HostTriplets t;
  t.cplx = m.cplx;
  const size_t w = m.cplx ? 2 : 1;
  const size_t n = (size_t)m.dim * (size_t)(m.c1 - m.c0);
  t.col.reserve(n); t.row.reserve(n); t.val.reserve(n * w);
  for (int32_t c = m.c0; c < m.c1; ++c)
    for (int32_t r = 0; r < m.dim; ++r) {
      t.col.push_back(c + 1);
      t.row.push_back(r + 1);
      t.val.push_back(1.0);
      if (m.cplx) t.val.push_back(0.0);
    }
  m.loc = from_triplets(t, m.dim, m.c1 - m.c0, m.c0);
}

// MatrixDiagonalScale (PSMatrixAlgebraModule.F90:507-532, ScaleDiagonal.f90, sparse_includes/DiagonalScale.f90):
// for every triplet whose column is stored here, the values of that column are multiplied by its value
void ps_diagonal_scale(PSMatrix& m, const HostTriplets& t) {
  use_grid_comm(m.grid);
  const int32_t width = m.c1 - m.c0;
  if (width == 0) return;
  const size_t w = m.cplx ? 2 : 1;
  std::vector<double> f((size_t)width * w, 0.0);
  for (int32_t j = 0; j < width; ++j) f[(size_t)j * w] = 1.0;
  for (size_t i = 0; i < t.size(); ++i) {
    const int32_t col = t.col[i] - 1;
    if (col < m.c0 || col >= m.c1) continue;
    double re, im = 0.0;
    if (t.cplx) { re = t.val[2 * i]; im = t.val[2 * i + 1]; } else { re = t.val[i]; }
    // several triplets naming the same column multiply one after the other, as the reference loop does
    double& fr = f[(size_t)(col - m.c0) * w];
    if (m.cplx) {
      double& fi = f[(size_t)(col - m.c0) * w + 1];
      const double nr = fr * re - fi * im, ni = fr * im + fi * re;
      fr = nr; fi = ni;
    } else {
      fr *= re;
    }
  }
  DevBuf<double> d(f.size());
  d.upload(f.data(), f.size());
  scale_columns(m.loc, d.p);
  sync_stream();  // f / d go out of scope
}

// GetMatrixBlock (PSMatrixModule.F90:1036-1150, distributed_includes/GetMatrixBlock.f90): every rank names a block
// [start_row, end_row) x [start_column, end_column) (1-based) and receives its entries with absolute coordinates;
// an entry goes to the FIRST rank whose block contains it (the EXIT in the reference's routing loop)
void ps_get_block(const PSMatrix& m, int sr, int er, int sc, int ec, HostTriplets& out) {
  use_grid_comm(m.grid);
  const int P = world().active() ? world().nranks : 1, me = world().active() ? world().rank : 0;
  std::vector<int64_t> box((size_t)4 * P);
  const int64_t mine[4] = {sr, er, sc, ec};
  comm_allgather_i64(mine, 4, box.data());
  DevMat full = ps_gather_full(m);
  HostTriplets all;
  to_triplets(full, 0, all);
  out = HostTriplets();
  out.cplx = m.cplx;
  const size_t w = m.cplx ? 2 : 1;
  for (size_t i = 0; i < all.size(); ++i) {
    const int r = all.row[i], c = all.col[i];
    for (int p = 0; p < P; ++p) {
      if (r >= box[(size_t)4 * p] && r < box[(size_t)4 * p + 1] && c >= box[(size_t)4 * p + 2] && c < box[(size_t)4 * p + 3]) {
        if (p == me) {
          out.col.push_back(c);
          out.row.push_back(r);
          for (size_t k = 0; k < w; ++k) out.val.push_back(all.val[i * w + k]);
        }
        break;
      }
    }
  }
}

// GetMatrixSlice (PSMatrixModule.F90:1153-1225, distributed_includes/SliceMatrix.f90): inclusive bounds, result of
// dimension max(rows, columns) of the slice on the same grid
void ps_get_slice(const PSMatrix& m, PSMatrix& sub, int sr, int er, int sc, int ec) {
  use_grid_comm(m.grid);
  HostTriplets t, s;
  ps_get_triplets(m, t);
  s.cplx = m.cplx;
  const size_t w = m.cplx ? 2 : 1;
  for (size_t i = 0; i < t.size(); ++i) {
    if (t.row[i] >= sr && t.row[i] <= er && t.col[i] >= sc && t.col[i] <= ec) {
      s.row.push_back(t.row[i] - sr + 1);
      s.col.push_back(t.col[i] - sc + 1);
      for (size_t k = 0; k < w; ++k) s.val.push_back(t.val[i * w + k]);
    }
  }
  const int new_dim = std::max(er - sr + 1, ec - sc + 1);
  const ProcessGrid* g = m.grid;
  const bool cplx = m.cplx;
  ps_construct_empty(sub, new_dim, g, cplx);
  ps_fill_from_triplets(sub, s);
}

// ResizeMatrix (PSMatrixModule.F90:1704-1741, distributed_includes/ResizeMatrix.f90): entries beyond the new size
// are dropped
void ps_resize(PSMatrix& m, int new_size) {
  use_grid_comm(m.grid);
  HostTriplets t, s;
  ps_get_triplets(m, t);
  s.cplx = m.cplx;
  const size_t w = m.cplx ? 2 : 1;
  for (size_t i = 0; i < t.size(); ++i) {
    if (t.row[i] <= new_size && t.col[i] <= new_size) {
      s.row.push_back(t.row[i]);
      s.col.push_back(t.col[i]);
      for (size_t k = 0; k < w; ++k) s.val.push_back(t.val[i * w + k]);
    }
  }
  const ProcessGrid* g = m.grid;
  const bool cplx = m.cplx;
  ps_construct_empty(m, new_size, g, cplx);
  ps_fill_from_triplets(m, s);
}

int64_t ps_size(const PSMatrix& m) {
  use_grid_comm(m.grid);  // GetMatrixSize (PSMatrixModule.F90:1360-1389)
  int64_t n = m.loc.nnz;
  comm_allreduce_sum_i64(&n, 1);
  return n;
}

void ps_to_complex(const PSMatrix& a, PSMatrix& out) {
  use_grid_comm(a.grid);
  DevMat t = to_complex(a.loc);
  out.grid = a.grid; out.dim = a.dim; out.c0 = a.c0; out.c1 = a.c1;
  out.cplx = true;
  out.loc = std::move(t);
}
void ps_to_real(const PSMatrix& a, PSMatrix& out) {
  use_grid_comm(a.grid);
  DevMat t = to_real(a.loc);
  out.grid = a.grid; out.dim = a.dim; out.c0 = a.c0; out.c1 = a.c1;
  out.cplx = false;
  out.loc = std::move(t);
}

// ------------------------------------------------------------------ algebra
// MatrixMultiply_ps (PSMatrixAlgebraModule.F90:108-211).  Column j of C needs column j of B
// (local) and the columns of A named by B's row indices: rank r gathers the panels of A
// (M1-M3) and multiplies them with its own panel of B; C comes out in the same panel layout, so
// there is no reduction step (slices == 1 semantics: working_threshold = threshold,
// distributed_algebra_includes/MatrixMultiply.f90:25-29).
namespace {
bool exchange_fits_fetch(int P);
int panel_pitch(int32_t dim, int P, bool with_counts, int* wcols_out);
long long g_block_scope_products = 0;   // panel products of block-order solves (band_scope.cpp) that took the block path
long long g_panel_products[3] = {0, 0, 0};   // products of slab sessions across ranks: done in slab form on every rank; declined; host synchronisations inside the former

// C = alpha A B of a slab session on more than one rank: A, B column panels in slab form, the result a column panel in slab
// form (MatrixMultiply.f90:92-267 gathers blocks of both operands along the grid; here the rows of B's panel name the columns
// of A that have to travel, as dense runs).  Protocol, all on the engine stream: (1) ONE all-gather of a record per rank --
// request (first / last row of B's panel, entries of B, entries of A and the alignment of its runs; -1: this rank's panels are
// not in slab form), packed extents of A's columns, prefix sums of their spans; (2) who sends how many doubles to whom, one
// read-back; (3) the runs of the requested columns packed per requester, one send / recv group; (4) layout of the columns this
// rank multiplies with, the tile kernel on them (slab_multiply with a left halo); (5) one reduction: did every rank's kernel
// take its panel.  Collective; false (every rank alike): nothing done, the caller takes the compressed-column path.
bool panel_slab_multiply(const PSMatrix& A, const PSMatrix& B, DevMat& AB, double alpha, double threshold) {
  Comm& c = world();
  Transport& tr = *c.tr;
  const int P = c.nranks, me = c.rank;
  const int32_t dim = A.dim;
  const long long syncs_before = host_sync_count();
  const bool mine_ok = slab_on() && A.loc.nnz > 0 && B.loc.nnz > 0 && A.c0 == B.c0 && A.c1 == B.c1 && slab_enter(mut(A)) &&
                       (&A == &B || slab_enter(mut(B))) && A.loc.slab->row_pad % 16 == 0 && A.loc.slab->row_pad < 4096;
  int wcols = 0;
  const int pitch = panel_pitch(dim, P, false, &wcols);
  DevBuf<int64_t> d_all((size_t)P * pitch), d_req((size_t)4 * P), d_bound((size_t)2 * P), d_cnt((size_t)P * P);
  int64_t* mine = d_all.p + (size_t)me * pitch;
  if (mine_ok) {
    slab_request_async(B.loc, mine);
    slab_extents_async(A.loc, mine + 4, mine + 4 + wcols);
    const long long w3 = (long long)A.loc.nnz * 4096 + A.loc.slab->row_pad;
    HIP_CHECK(hipMemcpyAsync(mine + 3, &w3, sizeof(w3), hipMemcpyHostToDevice, stream()));
  } else {
    const long long rec[4] = {INT_MAX, -1, -1, 0};
    HIP_CHECK(hipMemsetAsync(mine, 0, (size_t)pitch * sizeof(int64_t), stream()));
    HIP_CHECK(hipMemcpyAsync(mine, rec, sizeof(rec), hipMemcpyHostToDevice, stream()));
  }
  tr.allgather(mine, d_all.p, (size_t)pitch * sizeof(int64_t));
  HIP_CHECK(hipMemcpy2DAsync(d_req.p, 4 * sizeof(int64_t), d_all.p, (size_t)pitch * sizeof(int64_t), 4 * sizeof(int64_t), (size_t)P,
                             hipMemcpyDeviceToDevice, stream()));
  const int64_t *d_ext_all = d_all.p + 4, *d_pre_all = d_all.p + 4 + wcols;
  halo_counts_async(d_req.p, d_pre_all, pitch, dim, P, me, d_cnt.p, d_bound.p);   // (counts in doubles)
  // the product's plan (block windows and k ranges follow from the extents alone) from the gathered extents of A and the
  // extents of this rank's panel of B: its sizes come back with the exchange layout -- the product itself then needs ONE
  // more host round trip (its entry count), as on one rank
  SlabPlan plan;
  DevBuf<int32_t> gfirst, glast;
  DevBuf<int64_t> plan_stats(24);
  unsigned long long plan_hs[2] = {0, 0};
  const int snb = (B.loc.cols + 15) / 16;
  if (mine_ok) {
    plan_stats.zero();
    slab_plan_panel_async(B.loc, d_ext_all, pitch, dim, P, plan, gfirst, glast, reinterpret_cast<unsigned long long*>(plan_stats.p));
  }
  std::vector<int64_t> req((size_t)4 * P, 0), cnt((size_t)P * P, 0);
  if (exchange_fits_fetch(P)) {
    ScalarFetch f;
    f.add(d_req.p, 4 * P, req.data());
    f.add(d_cnt.p, P * P, cnt.data());
    if (mine_ok) {
      f.add(plan.blk_toff.p + snb, 1, &plan.total);
      f.add(plan_stats.p + 16, 2, plan_hs);
    }
    f.run();
  } else {
    HIP_CHECK(hipMemcpyAsync(req.data(), d_req.p, (size_t)4 * P * 8, hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.p, (size_t)P * P * 8, hipMemcpyDeviceToHost, stream()));
    if (mine_ok) {
      HIP_CHECK(hipMemcpyAsync(&plan.total, plan.blk_toff.p + snb, 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(plan_hs, plan_stats.p + 16, 16, hipMemcpyDeviceToHost, stream()));
    }
    sync_stream();
  }
  plan.max_w = (int)plan_hs[0];
  plan.max_kn = (int)plan_hs[1];
  exchange_stats().exchanges += 1;
  int64_t nnz_a = 0, nnz_b = 0;
  bool all_ok = true;
  for (int q = 0; q < P; ++q) {
    if (req[(size_t)4 * q + 2] < 0) { all_ok = false; break; }
    nnz_b += req[(size_t)4 * q + 2];
    nnz_a += req[(size_t)4 * q + 3] / 4096;
    if (req[(size_t)4 * q + 3] % 4096 != req[3] % 4096) all_ok = false;   // (every panel's runs aligned alike)
  }
  if (!all_ok) {
    g_panel_products[1] += 1;
    return false;
  }
  const int row_pad = (int)(req[3] % 4096);
  auto kmin_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? 0 : (int32_t)lo; };
  auto kmax_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? -1 : (int32_t)hi; };
  const int32_t kmin = kmin_of(me), kmax = kmax_of(me);
  // what I send (my columns inside every requester's range, packed per requester) and what I receive (the segments of the
  // other owners tile [kmin, kmax] in rank order): panel_exchange_layout, host arithmetic on the gathered requests and counts
  std::vector<int32_t> sa((size_t)P), sb((size_t)P), ra((size_t)P), rb((size_t)P);
  std::vector<int64_t> soff((size_t)P + 1, 0), zoff((size_t)P + 1, 0);
  panel_exchange_layout(dim, P, me, req.data(), cnt.data(), sa.data(), sb.data(), soff.data(), ra.data(), rb.data(), zoff.data());
  DevBuf<double> sendbuf((size_t)soff[(size_t)P] + 1);
  for (int q = 0; q < P; ++q)
    if (q != me && cnt[(size_t)me * P + q] > 0)
      slab_pack_runs_async(A.loc, d_pre_all + (size_t)me * pitch, sa[(size_t)q] - A.c0, sb[(size_t)q] - A.c0, sendbuf.p + soff[(size_t)q]);
  DevBuf<double> recvbuf((size_t)zoff[(size_t)P] + kIndexSlack);
  tr.group_begin();
  for (int q = 0; q < P; ++q) {
    const int64_t m = cnt[(size_t)me * P + q];
    if (q != me && m > 0) tr.send(sendbuf.p + soff[(size_t)q], (size_t)m * sizeof(double), q);
  }
  for (int s = 0; s < P; ++s) {
    const int64_t m = cnt[(size_t)s * P + me];
    if (s != me && m > 0) tr.recv(recvbuf.p + zoff[(size_t)s], (size_t)m * sizeof(double), s);
  }
  tr.group_end();
  // "did every rank's kernel take its panel": known to a rank once its plan is back; the sum over the ranks is enqueued
  // here and read back with the product's entry count
  // (plan.max_w / max_kn are back: the SAME predicate slab_multiply applies -- a rank that agrees here cannot decline later)
  plan.max_w = (int)plan_hs[0];
  plan.max_kn = (int)plan_hs[1];
  bool ok = mine_ok && kmax >= kmin && slab_multiply_takes_panel(A.loc, B.loc, row_pad, kmin, kmax + 1, &plan);
  DevBuf<double> d_declined(4);
  double declined[4] = {ok ? 0.0 : 1.0, 0.0, 0.0, 0.0};
  d_declined.upload(declined, 4);
  tr.allreduce(d_declined.p, 4, true, 0);
  bool fetched = false;
  if (ok) {
    const int32_t ka = kmin, kb = kmax + 1;
    DevBuf<int32_t> d_ra((size_t)P), nfirst((size_t)(kb - ka)), nlast((size_t)(kb - ka));
    DevBuf<int64_t> d_zoff((size_t)P);
    DevBuf<unsigned long long> naddr((size_t)(kb - ka));
    if (P > 16) {
      d_ra.upload(ra.data(), (size_t)P);
      d_zoff.upload(zoff.data(), (size_t)P);
    }
    slab_halo_layout_async(d_ext_all, d_pre_all, pitch, dim, P, me, ka, kb, d_ra.p, d_zoff.p, recvbuf.p, A.loc, nfirst.p, nlast.p, naddr.p,
                           nullptr, nullptr, ra.data(), zoff.data());
    SlabHalo halo;
    halo.ka = ka;
    halo.kb = kb;
    halo.first = nfirst.p;
    halo.last = nlast.p;
    halo.addr = naddr.p;
    halo.row_pad = row_pad;
    halo.plan = &plan;
    halo.on_fetch = [&](ScalarFetch& f) {
      f.add(d_declined.p, 1, reinterpret_cast<unsigned long long*>(&declined[0]));
      fetched = true;
    };
    const double denom = (double)dim * (double)dim;
    const bool dense_rule = denom > 0 && std::min((double)nnz_a / denom, (double)nnz_b / denom) > 0.1;
    // (the buffers above are released on return: the allocator is stream ordered, and slab_multiply ends with a read-back)
    if (!slab_multiply(A.loc, B.loc, AB, alpha, threshold, dense_rule, &halo))
      NTP_FATAL("internal: a panel product was declined after its rank had agreed to it");
  }
  if (!fetched) {
    ScalarFetch f;
    f.add(d_declined.p, 1, reinterpret_cast<unsigned long long*>(&declined[0]));
    f.run();
  }
  if (declined[0] != 0.0) {
    g_panel_products[1] += 1;
    return false;
  }
  g_panel_products[0] += 1;
  g_panel_products[2] += host_sync_count() - syncs_before;   // (measured in sync_stream: the exchange's and the product's)
  return true;
}
}  // namespace
const long long* panel_product_counts() { return g_panel_products; }
long long block_scope_products() { return g_block_scope_products; }

namespace {
// alpha * A * B with the threshold, local panel of the result (slices == 1 semantics: every output entry is one sum
// over the whole inner dimension in ascending order)
DevMat multiply_panel(const PSMatrix& A, const PSMatrix& B, double alpha, double threshold, double a_fraction = 1.0) {
  // dense-branch rule of the local multiply (GemmMatrix.f90:49-61): only the order of threshold and
  // alpha differs here, the arithmetic is the same hash-free sparse kernel.  a_fraction: share of the inner dimension
  // A is populated over (a process slice's operand: its density is that of the populated part, as in the
  // reference's per-block test)
  int64_t nz[2] = {A.loc.nnz, B.loc.nnz};
  const double denom = (double)A.dim * (double)A.dim;
  DevMat AB;
  if (world().active()) {
    // only the columns of A named by the rows of the local B panel travel (halo for banded operands); the same
    // exchange returns the global nnz for the dense-branch rule
    HaloExchange hx;
    gather_needed_begin(hx, A, B.loc, nz, true);
    const bool dense_rule = denom > 0 && std::min((double)nz[0] / (denom * a_fraction), (double)nz[1] / denom) > 0.1;
    const ColRange need{hx.kmin, hx.kmax + 1};   // the rows of the B panel name these columns only
    // A solve in a block order (band_scope.cpp: 3-D operands on several ranks): the panel of B as the columns c0 .. c1 of a
    // square matrix whose other columns are empty, the product through the block path -- tiles of the order the scope
    // installed, candidates in this rank's super-columns only -- and the panel's columns cut out of the result
    const bool block_route = block_scope_active() && !A.cplx && !B.cplx && options().spgemm_fma == 1 && options().block_path != 0 &&
                             a_fraction == 1.0;
    if (block_route) {
      hx.finish();
      DevMat Csq;
      {
        MatView Bsq;
        DevBuf<int64_t> pad_outer((size_t)B.dim + 1);
        fill_i64(pad_outer.p, B.c0, 0);
        copy_shift_i64(B.loc.outer.p, pad_outer.p + B.c0, (int64_t)(B.c1 - B.c0) + 1, 0);
        fill_i64(pad_outer.p + B.c1 + 1, (int64_t)B.dim - B.c1, B.loc.nnz);
        Bsq.alias(B.loc, pad_outer.p, B.dim, B.loc.nnz);
        hx.full.block_hint = 1;   // (straight to the block path: no run statistics, no band search on the gathered operand)
        spgemm(hx.full, Bsq.m, Csq, alpha, threshold, dense_rule);
        sync_stream();   // (the padded offsets are released on leaving the scope)
      }
      if (last_spgemm_stats().block) g_block_scope_products += 1;
      if (Csq.loose() || Csq.expanded() || Csq.blocked()) pack(Csq);
      MatView Cv;
      Cv.alias(Csq, Csq.outer.p + B.c0, B.c1 - B.c0, Csq.nnz);
      AB = concat_columns({&Cv.m});
      sync_stream();
    } else if (!hx.overlapped) {
      hx.finish();
      spgemm(hx.full, B.loc, AB, alpha, threshold, dense_rule, nullptr, &need);
    } else {
      // the halo is travelling on the communication stream: multiply the interior columns of the B panel (they
      // name local columns of A only) meanwhile, then the boundary columns against the gathered operand.  A
      // column's arithmetic does not depend on which call computes it, so the result is bit-identical.
      const int32_t width = B.c1 - B.c0;
      DevMat Cint, Cleft, Cright;
      {
        MatView Apad, Bint;
        DevBuf<int64_t> pad_outer((size_t)A.dim + 1);
        fill_i64(pad_outer.p, A.c0, 0);
        copy_shift_i64(A.loc.outer.p, pad_outer.p + A.c0, (int64_t)(A.c1 - A.c0) + 1, 0);
        fill_i64(pad_outer.p + A.c1 + 1, (int64_t)A.dim - A.c1, A.loc.nnz);
        Apad.alias(A.loc, pad_outer.p, A.dim, A.loc.nnz);
        Bint.alias(B.loc, B.loc.outer.p + hx.jl, hx.jr - hx.jl, hx.off_r - hx.off_l);
        spgemm(Apad.m, Bint.m, Cint, alpha, threshold, dense_rule);
      }
      hx.finish();
      if (hx.jl > 0) {
        MatView Bl;
        Bl.alias(B.loc, B.loc.outer.p, hx.jl, hx.off_l);
        spgemm(hx.full, Bl.m, Cleft, alpha, threshold, dense_rule, nullptr, &need);
      } else {
        Cleft.reset_empty(A.dim, 0, A.cplx);
      }
      if (hx.jr < width) {
        MatView Br;
        Br.alias(B.loc, B.loc.outer.p + hx.jr, width - hx.jr, B.loc.nnz - hx.off_r);
        spgemm(hx.full, Br.m, Cright, alpha, threshold, dense_rule, nullptr, &need);
      } else {
        Cright.reset_empty(A.dim, 0, A.cplx);
      }
      std::vector<const DevMat*> parts;
      if (Cleft.cols) parts.push_back(&Cleft);
      parts.push_back(&Cint);
      if (Cright.cols) parts.push_back(&Cright);
      AB = parts.size() == 1 ? std::move(Cint) : concat_columns(parts);
    }
  } else {
    const bool dense_rule = denom > 0 && std::min((double)nz[0] / (denom * a_fraction), (double)nz[1] / denom) > 0.1;
    spgemm(A.loc, B.loc, AB, alpha, threshold, dense_rule);
  }
  return AB;
}
}  // namespace

void ps_multiply(const PSMatrix& A, const PSMatrix& B, PSMatrix& C, double alpha, double beta, double threshold) {
  use_grid_comm(A.grid);
  if (A.dim != B.dim) NTP_FATAL("MatrixMultiply: dimension mismatch");
  // up-casting of mixed real/complex operands (PSMatrixAlgebraModule.F90:171-188)
  if (A.cplx != B.cplx) {
    // (a real operand that an earlier call of a slab session left in slab form or loose: to_complex copies the three
    // arrays of compressed columns, so the operands are packed first)
    if (A.loc.expanded() || A.loc.loose() || A.loc.blocked()) pack(mut(A));
    if (B.loc.expanded() || B.loc.loose() || B.loc.blocked()) pack(mut(B));
    PSMatrix Ac, Bc;
    ps_to_complex(A, Ac);
    ps_to_complex(B, Bc);
    ps_multiply(Ac, Bc, C, alpha, beta, threshold);
    return;
  }
  DevMat AB;
  const int S = A.grid ? A.grid->num_slices : 1;
  // A one-call session of the C ABI on operands without run structure (or already in block form): the product goes
  // through the block path and STAYS in block form (DevMat::blk) -- the next product of the caller's loop multiplies it
  // as it is, every other entry point packs on access.  Solver loops (sessions of their own) take compressed columns.
  const bool block_first = slab_on() && !A.cplx && S <= 1 && std::fabs(beta) < 2.2250738585072014e-308 &&
                           (A.loc.blocked() || B.loc.blocked() || A.loc.block_hint || B.loc.block_hint) && !A.loc.expanded() && !B.loc.expanded() &&
                           !A.loc.loose() && !B.loc.loose();
  if (block_first) {
    BlockKeepScope keep;
    const double denom = (double)A.dim * (double)A.dim;
    const bool dense_rule = denom > 0 && std::min((double)A.loc.nnz / denom, (double)B.loc.nnz / denom) > 0.1;
    spgemm(A.loc, B.loc, AB, alpha, threshold, dense_rule);
   
    C.grid = A.grid; C.dim = A.dim; C.c0 = B.c0; C.c1 = B.c1;
    C.cplx = false;
    C.loc = std::move(AB);
    return;
  }
  unblock({&A, &B});
  if (slab_on() && g_complex_session && A.cplx && S <= 1 && std::fabs(beta) < 2.2250738585072014e-308 && A.loc.nnz > 0 && B.loc.nnz > 0) {
    // (a session that takes complex operands: the iterates stay in the complex tile kernel's operand form)
    const double denom = (double)A.dim * (double)A.dim;
    const bool dense_rule = denom > 0 && std::min((double)A.loc.nnz / denom, (double)B.loc.nnz / denom) > 0.1;
    if (slab_enter_c(mut(A)) && (&A == &B || slab_enter_c(mut(B))) && slab_multiply_c(A.loc, B.loc, AB, alpha, threshold, dense_rule)) {
      g_slab_counts[0] += 1;
      C.grid = A.grid; C.dim = A.dim; C.c0 = B.c0; C.c1 = B.c1;
      C.cplx = true;
      C.loc = std::move(AB);
      return;
    }
    slab_refused({&A, &B});
  }
  if (g_slab_depth > 0 && world().active() && !A.cplx && S <= 1 && std::fabs(beta) < 2.2250738585072014e-308) {
    // (a slab session across ranks: collective -- every rank reports whether its panels are in slab form with the halo request)
    if (panel_slab_multiply(A, B, AB, alpha, threshold)) {
      g_slab_counts[0] += 1;
      C.grid = A.grid; C.dim = A.dim; C.c0 = B.c0; C.c1 = B.c1;
      C.cplx = false;
      C.loc = std::move(AB);
      return;
    }
    g_slab_counts[3] += 1;
    for (const PSMatrix* m : {&A, &B})
      if (m->loc.expanded() || m->loc.loose()) pack(mut(*m));
  } else if (slab_on() && !world().active() && !A.cplx && S <= 1 && std::fabs(beta) < 2.2250738585072014e-308 && A.loc.nnz > 0 && B.loc.nnz > 0) {
    // (a slab session: operands are turned into slab form where they are, the product stays in it)
    const double denom = (double)A.dim * (double)A.dim;
    const bool dense_rule = denom > 0 && std::min((double)A.loc.nnz / denom, (double)B.loc.nnz / denom) > 0.1;
    // (unfused arithmetic: the register-slab kernel multiplies whole runs, zeros included -- operands that have become
    // sparse inside wide extents, 3 I - X^2 near the end of a sign iteration, are better served by the general kernels,
    // which the compressed-column path picks per product; the MFMA tile kernel of the FMA mode takes them as they are)
    auto runs_dense = [](const DevMat& M) {
      return options().spgemm_fma != 0 || !M.expanded() ||
             (double)slab_span_sum(M) <= 1.5 * (double)M.nnz + 64.0 * (double)M.cols;
    };
    if (slab_enter(mut(A)) && (&A == &B || slab_enter(mut(B))) && runs_dense(A.loc) && runs_dense(B.loc) &&
        slab_multiply(A.loc, B.loc, AB, alpha, threshold, dense_rule)) {
      g_slab_counts[0] += 1;
      if (options().time_kernels != 0) {   // (statistics mode: the products a plan over compressed columns would have counted)
        const long long pr = slab_product_count(A.loc, B.loc);
        last_spgemm_stats().products = pr;
        spgemm_accum().products += pr;
      }
      C.grid = A.grid; C.dim = A.dim; C.c0 = B.c0; C.c1 = B.c1;
      C.cplx = false;
      C.loc = std::move(AB);
      return;
    }
    slab_refused({&A, &B});
  } else {
    slab_pack_if({&A, &B});
  }
  if ((C.loc.expanded() || C.loc.loose() || C.loc.blocked()) && &C != &A && &C != &B) pack(C.loc);   // (beta != 0 reads it below; otherwise it is replaced)
  if (S <= 1) {
    // (a one-call session of the C ABI: should the product turn out to belong to the block path -- decided inside spgemm()
    // once the run-based kernels have declined -- it stays in block form, and the refusal above was not one)
    const bool keep_block = g_slab_depth > 0 && !g_slab_failed && !A.cplx && std::fabs(beta) < 2.2250738585072014e-308;
    if (keep_block) {
      BlockKeepScope keep;
      AB = multiply_panel(A, B, alpha, threshold);
    } else {
      AB = multiply_panel(A, B, alpha, threshold);
    }
    if (AB.blocked() && g_slab_refusals > 0) { g_slab_refusals -= 1; }
  } else {
    // Process slices (the reference's 2.5-D algorithm, MatrixMultiply.f90:25-29, 74-80, 230-267): slice s multiplies
    // its share of the inner dimension -- the blocks g with g % S == s, block = padded dimension / (max(rows, columns)
    // * S * block multiplier) -- with threshold / (1000 S); the partial products are then summed in slice order by
    // IncrementMatrix, only the last addition with the caller's threshold (comm_includes/ReduceAndSumMatrixCleanup.f90
    // :11-32), so entries below the threshold survive where the rule copies tails unfiltered.  This engine keeps its
    // column panels and reproduces those sums: S multiplies over masked copies of A, S increments.  (Block multiplier 1
    // = the reference run with fewer threads than blocks.)
    const ProcessGrid& g = *A.grid;
    const int64_t lcm = (int64_t)S * g.num_cols * g.num_rows;
    const int64_t padded = ((int64_t)A.dim + lcm - 1) / lcm * lcm;
    const int32_t block = (int32_t)(padded / ((int64_t)std::max(g.num_rows, g.num_cols) * S));
    const double working = threshold / ((double)S * 1000.0);
    AB.reset_empty(A.dim, B.c1 - B.c0, A.cplx);
    for (int s = 0; s < S; ++s) {
      PSMatrix As;
      As.grid = A.grid; As.dim = A.dim; As.cplx = A.cplx; As.c0 = A.c0; As.c1 = A.c1;
      As.loc = mask_columns(A.loc, A.c0, block, S, s);
      int64_t kcount = 0;   // columns of the whole matrix in this slice's share
      for (int64_t k0 = (int64_t)s * block; k0 < A.dim; k0 += (int64_t)S * block) kcount += std::min<int64_t>(block, A.dim - k0);
      DevMat part = multiply_panel(As, B, alpha, working, std::max(1e-300, (double)kcount / (double)A.dim));
      increment_blocked(part, AB, 1.0, s == S - 1 ? threshold : 0.0, block);   // (row blocks = inner-dimension blocks)
    }
  }
  // beta handling (MatrixMultiply.f90:324-329)
  if (std::fabs(beta) < 2.2250738585072014e-308 || !C.constructed() || C.dim != A.dim) {
    C.grid = A.grid; C.dim = A.dim; C.c0 = B.c0; C.c1 = B.c1;
    C.cplx = A.cplx;
    C.loc = std::move(AB);
  } else {
    if (C.cplx != A.cplx) {
      PSMatrix Cc;
      ps_to_complex(C, Cc);
      C.cplx = true;
      C.loc = std::move(Cc.loc);
      if (!A.cplx) AB = to_complex(AB);
    }
    scale(C.loc, beta);
    increment(AB, C.loc, 1.0, 0.0);
  }
}

// IncrementMatrix_ps (PSMatrixAlgebraModule.F90:414-460)
void ps_increment(const PSMatrix& A, PSMatrix& B, double alpha, double threshold) {
  use_grid_comm(A.grid);
  if (A.dim != B.dim) NTP_FATAL("IncrementMatrix: dimension mismatch");
  if (blk_any({&A, &B}) && &A != &B && !A.cplx && !B.cplx) {
    ps_axpby(A, B, alpha, 1.0, threshold);
    return;
  }
  unblock({&A, &B});
  if (slab_on() && (A.loc.expanded() || B.loc.expanded()) && &A != &B) {
    ps_axpby(A, B, alpha, 1.0, threshold);
    return;
  }
  slab_pack_if({&A, &B});
  if (A.cplx && !B.cplx) {
    DevMat bc = to_complex(B.loc);
    increment(A.loc, bc, alpha, threshold);
    B.cplx = true;
    B.loc = std::move(bc);
  } else if (!A.cplx && B.cplx) {
    DevMat ac = to_complex(A.loc);
    increment(ac, B.loc, alpha, threshold);
  } else if (&A == &B) {
    DevMat a2 = A.loc.clone();
    increment(a2, B.loc, alpha, threshold);
  } else {
    increment(A.loc, B.loc, alpha, threshold);
  }
}

void ps_scale(PSMatrix& A, double c) {
  use_grid_comm(A.grid);
  if (blk_any({&A})) {
    if (block_scale(A.loc, c)) { g_block_counts[0] += 1; return; }
    g_block_counts[1] += 1;
  }
  unblock({&A});
  if (slab_on() && g_complex_session && A.cplx && A.loc.expanded() && slab_scale_c(A.loc, c)) { g_slab_counts[2] += 1; return; }
  if (slab_on() && A.loc.expanded()) {
    if (slab_scale(A.loc, c)) { g_slab_counts[2] += 1; return; }
    slab_refused({&A});
  }
  scale(A.loc, c);
}

// B <- alpha*A + beta*B: ScaleMatrix(B, beta) followed by IncrementMatrix(A, B, alpha, threshold) in one pass (the
// merge kernels scale B's values as they read them: the same products, the same rules, bit for bit)
void ps_axpby(const PSMatrix& A, PSMatrix& B, double alpha, double beta, double threshold) {
  use_grid_comm(A.grid);
  if (A.dim != B.dim) NTP_FATAL("IncrementMatrix: dimension mismatch");
  if (blk_any({&A, &B}) && !A.cplx && !B.cplx && &A != &B && !A.loc.expanded() && !B.loc.expanded() && !A.loc.loose() && !B.loc.loose()) {
    if (block_axpby(A.loc, B.loc, alpha, beta, threshold)) { g_block_counts[0] += 1; return; }
    g_block_counts[1] += 1;
  }
  unblock({&A, &B});
  if (slab_on() && g_complex_session && (A.loc.expanded() || B.loc.expanded()) && A.cplx && B.cplx && &A != &B) {
    // (a session that takes complex operands: the merge on runs of (re, im) pairs)
    if (slab_enter_c(mut(A)) && slab_enter_c(B.loc) && slab_axpby_c(A.loc, B.loc, alpha, beta, threshold)) { g_slab_counts[1] += 1; return; }
    slab_refused({&A, &B});
  } else if (slab_on() && (A.loc.expanded() || B.loc.expanded()) && !A.cplx && !B.cplx && &A != &B) {
    // (a slab session: the operand still in compressed columns -- an identity, the Hamiltonian -- is turned into slab form)
    if (slab_enter(mut(A)) && slab_enter(B.loc) && slab_axpby(A.loc, B.loc, alpha, beta, threshold)) { g_slab_counts[1] += 1; return; }
    slab_refused({&A, &B});
  } else {
    slab_pack_if({&A, &B});
  }
  if (A.cplx != B.cplx || &A == &B) {
    ps_scale(B, beta);
    ps_increment(A, B, alpha, threshold);
    return;
  }
  axpby(A.loc, B.loc, alpha, beta, threshold, nullptr, nullptr);
}

void ps_increment_identity(const PSMatrix& Identity, PSMatrix& B, double alpha) {
  use_grid_comm(Identity.grid);
  if (slab_on() && B.loc.expanded() && !B.cplx && !Identity.cplx && Identity.dim == B.dim && slab_add_diagonal(B.loc, alpha, B.c0)) {
    g_slab_counts[1] += 1;
    return;
  }
  if (slab_on() && g_complex_session && B.loc.expanded() && B.cplx && Identity.dim == B.dim && slab_add_diagonal_c(B.loc, alpha, B.c0)) {
    g_slab_counts[1] += 1;
    return;
  }
  // compressed columns that store their diagonal: one value per column changes, in place
  if (options().column_fused != 0 && Identity.dim == B.dim && (!Identity.cplx || B.cplx) && !B.loc.expanded() && !B.loc.loose() && !B.loc.blocked() &&
      add_identity_inplace(B.loc, alpha, B.c0)) {
    g_column_fused[0] += 1;
    return;
  }
  ps_increment(Identity, B, alpha, 0.0);
}

bool ps_norm_axpby(const PSMatrix& A, const PSMatrix& B, double alpha, double beta, double* norm) {
  use_grid_comm(A.grid);
  if (blk_any({&A, &B})) return false;   // (the caller spells it with the vocabulary, which knows the block form)
  if (world().active() && g_slab_depth > 0 && !A.cplx && !B.cplx && A.dim == B.dim && &A != &B) {
    // (a session across ranks: the decision is collective -- one reduction carries the norm and "some rank declined")
    double v = 0.0;
    const bool ok = slab_on() && (A.loc.expanded() || B.loc.expanded()) && slab_enter(mut(A)) && slab_enter(mut(B)) &&
                    slab_norm_axpby(A.loc, B.loc, alpha, beta, &v);
    double pair[2] = {ok ? v : 0.0, ok ? 0.0 : 1.0};
    comm_allreduce_max(pair, 2);
    if (pair[1] != 0.0) return false;
    *norm = pair[0];
    g_slab_counts[2] += 1;
    return true;
  }
  if (slab_on() && !A.cplx && !B.cplx && A.dim == B.dim && &A != &B && (A.loc.expanded() || B.loc.expanded())) {
    if (!(slab_enter(mut(A)) && slab_enter(mut(B)) && slab_norm_axpby(A.loc, B.loc, alpha, beta, norm))) return false;
    g_slab_counts[2] += 1;
    return true;
  }
  if (slab_on() && g_complex_session && A.cplx && B.cplx && A.dim == B.dim && &A != &B && (A.loc.expanded() || B.loc.expanded())) {
    if (!(slab_enter_c(mut(A)) && slab_enter_c(mut(B)) && slab_norm_axpby_c(A.loc, B.loc, alpha, beta, norm))) return false;
    g_slab_counts[2] += 1;
    return true;
  }
  // compressed columns (complex loops, real ones outside a session): a dense window per column, the difference is never formed
  auto cols = [](const PSMatrix& M) { return !M.loc.expanded() && !M.loc.loose() && !M.loc.blocked(); };
  if (options().column_fused != 0 && beta == 1.0 && A.cplx == B.cplx && A.dim == B.dim && &A != &B && cols(A) && cols(B) && A.c0 == B.c0 && A.c1 == B.c1) {
    // (the decision is collective: a rank whose columns do not fit the kernel's window must not fall back alone -- one
    // reduction carries the norm and "some rank declined" together, and every rank takes the same branch)
    double v = 0.0;
    const bool ok = norm_axpy_columns(A.loc, B.loc, alpha, &v);
    double pair[2] = {ok ? v : 0.0, ok ? 0.0 : 1.0};
    comm_allreduce_max(pair, 2);
    if (pair[1] == 0.0) {
      *norm = pair[0];
      g_column_fused[1] += 1;
      return true;
    }
  }
  return false;
}

bool ps_trs4_traces(const PSMatrix& X, const PSMatrix& X2, double* trace_fx, double* trace_gx) {
  use_grid_comm(X.grid);
  if (blk_any({&X, &X2})) return false;   // (the caller spells it with the vocabulary, which knows the block form)
  if (world().active()) {
    // (a session across ranks: the decision is collective -- the sums and "some rank declined" in one reduction)
    if (g_slab_depth == 0 || X.cplx || X2.cplx) return false;
    double t[3] = {0.0, 0.0, 0.0};
    const bool ok = slab_on() && X.loc.expanded() && X2.loc.expanded() && slab_trs4_traces(X.loc, X2.loc, X.c0, &t[0], &t[1]);
    if (!ok) { t[0] = t[1] = 0.0; t[2] = 1.0; }
    comm_allreduce_sum(t, 3);
    if (t[2] != 0.0) return false;
    *trace_fx = t[0];
    *trace_gx = t[1];
    g_slab_counts[2] += 1;
    return true;
  }
  if (!slab_on() || !X.loc.expanded() || !X2.loc.expanded() || X.cplx || X2.cplx) return false;
  if (!slab_trs4_traces(X.loc, X2.loc, X.c0, trace_fx, trace_gx)) return false;
  g_slab_counts[2] += 1;
  return true;
}
bool ps_trs4_operand(const PSMatrix& X, const PSMatrix& X2, double sigma, PSMatrix& P) {
  use_grid_comm(X.grid);
  if (blk_any({&X, &X2})) return false;   // (the caller spells it with the vocabulary, which knows the block form)
  if (!slab_on() || !X.loc.expanded() || !X2.loc.expanded() || X.cplx || X2.cplx || sigma == 0.0) return false;
  DevMat R;
  if (!slab_trs4_operand(X.loc, X2.loc, sigma, X.c0, R)) return false;
  P.grid = X.grid; P.dim = X.dim; P.cplx = false; P.c0 = X.c0; P.c1 = X.c1;
  P.loc = std::move(R);
  g_slab_counts[1] += 1;
  return true;
}

void ps_copy_axpby(const PSMatrix& B, const PSMatrix& A, PSMatrix& Out, double alpha, double beta, double threshold) {
  use_grid_comm(B.grid);
  if (blk_any({&A, &B}) && !A.cplx && !B.cplx && &A != &B && &Out != &A && &Out != &B) {
    ps_copy(B, Out);
    ps_axpby(A, Out, alpha, beta, threshold);
    return;
  }
  unblock({&A, &B});
  if (slab_on() && g_complex_session && (A.loc.expanded() || B.loc.expanded()) && A.cplx && B.cplx && &A != &B && &Out != &A && &Out != &B &&
      A.dim == B.dim) {
    DevMat R;
    if (slab_enter_c(mut(A)) && slab_enter_c(mut(B)) && slab_axpby_to_c(A.loc, B.loc, R, alpha, beta, threshold)) {
      g_slab_counts[1] += 1;
      Out.grid = B.grid; Out.dim = B.dim; Out.cplx = true; Out.c0 = B.c0; Out.c1 = B.c1;
      Out.loc = std::move(R);
      return;
    }
    slab_refused({&A, &B});
  } else if (slab_on() && (A.loc.expanded() || B.loc.expanded()) && !A.cplx && !B.cplx && &A != &B && &Out != &A && &Out != &B && A.dim == B.dim) {
    DevMat R;
    if (slab_enter(mut(A)) && slab_enter(mut(B)) && slab_axpby_to(A.loc, B.loc, R, alpha, beta, threshold)) {
      g_slab_counts[1] += 1;
      Out.grid = B.grid; Out.dim = B.dim; Out.cplx = false; Out.c0 = B.c0; Out.c1 = B.c1;
      Out.loc = std::move(R);
      return;
    }
    slab_refused({&A, &B});
  }
  ps_copy(B, Out);
  if (beta == 1.0) ps_increment(A, Out, alpha, threshold);
  else ps_axpby(A, Out, alpha, beta, threshold);
}

void ps_axpby_dot(const PSMatrix& A, PSMatrix& B, double alpha, double beta, double threshold, const PSMatrix& D, double out[4],
                  bool want_trace) {
  use_grid_comm(A.grid);
  out[2] = out[3] = 0.0;
  if (A.cplx != B.cplx || A.cplx != D.cplx || &A == &B) {  // mixed types: unfused sequence
    ps_scale(B, beta);
    ps_increment(A, B, alpha, threshold);
    ps_dot(B, D, out);
    if (want_trace) out[2] = ps_trace(B);
    return;
  }
  axpby(A.loc, B.loc, alpha, beta, threshold, &D.loc, out, want_trace ? &out[2] : nullptr, B.c0);
  comm_allreduce_sum(out, want_trace ? 3 : 2);
}

namespace {
// One TRS2 step of a rank whose panel is in slab form (kernels.hpp SlabForm): the halo travels as dense column runs
// (8 bytes per row of a column's span, no row ids, no offsets: every rank derives the layout from the all-gathered
// column extents), the run records of the kernel address the received runs where they land, the multiplier tiles are
// local.  Protocol, all on the engine stream with ONE host synchronisation: (1) ONE all-gather of a record per rank:
// request (first / last row of the panel, nnz), packed extents, prefix sums of the spans; (2) a kernel
// derives who sends how many doubles to whom, one read-back; (3) the runs of the requested columns are packed per
// requester and exchanged in one send / recv group; (4) layout kernel, then slab_step.  Collective: every rank calls it.
// Returns this rank's success; the result is in fu.result (the caller installs it when all ranks succeeded).
void dev_allreduce4(double* d) { world().tr->allreduce(d, 4, true, 0); }

// What a panel step must know before its halo can travel -- every rank's request, the column extents of the whole iterate,
// who sends how much to whom, the step's plan -- is a function of the iterate's column extents alone.  It is PREPARED on the
// device (one all-gather, three small kernels) and read back in one round trip: at the start of a step, or -- for every step
// after the first -- by the step BEFORE it, from its result's extents, on that step's own read-back (SlabHalo::before_fetch):
// a panel step then costs ONE host round trip, as a step on one rank does.
struct PanelExchange {
  int P = 0, me = 0, pitch = 0, wcols = 0, snb = 0;
  int32_t dim = 0;
  bool with_counts = false;
  DevBuf<int64_t> d_all, d_req, d_bound, d_cnt, plan_stats;
  SlabPlan plan;
  DevBuf<int32_t> gfirst, glast;
  std::vector<int64_t> req, bound, cnt;
  unsigned long long plan_hs[2] = {0, 0};
  const void* owner = nullptr;   // the value buffer of the iterate (in slab form) it was prepared for ...
  unsigned long long owner_serial = 0;   // ... and that buffer's allocation serial: an address alone comes back from the caching allocator
  const int64_t* d_ext_all() const { return d_all.p + 4; }
  const int64_t* d_pre_all() const { return d_all.p + 4 + wcols; }
  const int64_t* d_cnt_all() const { return with_counts ? d_all.p + 4 + 2 * (size_t)wcols : nullptr; }
};
std::unique_ptr<PanelExchange> g_pending_exchange;
long long g_exchange_prefetched = 0;   // panel steps whose exchange layout came with the step before
}  // namespace
// the preparation a step left for a successor that never came (the last step of a solve or of a bench block): P * pitch * 8
// bytes of device memory and a stale layout -- dropped where a solve begins and ends and with the operand caches
void drop_pending_exchange() { g_pending_exchange.reset(); }
namespace {

int panel_pitch(int32_t dim, int P, bool with_counts, int* wcols_out) {
  int32_t maxw = 0;
  for (int q = 0; q < P; ++q) {
    int32_t a0, a1;
    panel_range(dim, P, q, &a0, &a1);
    maxw = std::max(maxw, a1 - a0);
  }
  // ONE all-gather per step: every rank contributes a record of `pitch` 8-byte words -- its request (4 words), the
  // packed extents of its columns, the prefix sums of their spans and, for the statistics (timers on), their entry
  // counts; the kernels below read the sections through the common stride
  *wcols_out = maxw + 1;
  return 4 + (with_counts ? 3 : 2) * (maxw + 1);
}
// enqueues the preparation for the panel Xl (slab form) and adds what the host needs of it to the fetch (collective)
void exchange_prepare(const DevMat& Xl, int32_t dim, const long long* d_nnz, PanelExchange& pe, ScalarFetch& f) {
  Comm& c = world();
  Transport& tr = *c.tr;
  pe.P = c.nranks;
  pe.me = c.rank;
  pe.dim = dim;
  pe.with_counts = options().time_kernels != 0;
  pe.pitch = panel_pitch(dim, pe.P, pe.with_counts, &pe.wcols);
  pe.snb = (Xl.cols + 15) / 16;
  const int P = pe.P;
  pe.req.assign((size_t)4 * P, 0);
  pe.bound.assign((size_t)2 * P, 0);
  pe.cnt.assign((size_t)P * P, 0);
  pe.d_all.alloc((size_t)P * pe.pitch);
  pe.d_req.alloc((size_t)4 * P);
  pe.d_bound.alloc((size_t)2 * P);
  pe.d_cnt.alloc((size_t)P * P);
  int64_t* mine = pe.d_all.p + (size_t)pe.me * pe.pitch;
  slab_export_async(Xl, mine, d_nnz, mine + 4, mine + 4 + pe.wcols, pe.with_counts ? mine + 4 + 2 * (size_t)pe.wcols : nullptr);
  tr.allgather(mine, pe.d_all.p, (size_t)pe.pitch * sizeof(int64_t));
  HIP_CHECK(hipMemcpy2DAsync(pe.d_req.p, 4 * sizeof(int64_t), pe.d_all.p, (size_t)pe.pitch * sizeof(int64_t), 4 * sizeof(int64_t), (size_t)P,
                             hipMemcpyDeviceToDevice, stream()));
  halo_counts_async(pe.d_req.p, pe.d_pre_all(), pe.pitch, dim, P, pe.me, pe.d_cnt.p, pe.d_bound.p);   // (counts in doubles here)
  // the step's plan (block windows and k ranges follow from the extents alone) is made here, from the gathered
  // extents, so that its sizes come back in the same read-back as the exchange layout
  pe.plan_stats.alloc(24);
  pe.plan_stats.zero();
  slab_plan_panel_async(Xl, pe.d_ext_all(), pe.pitch, dim, P, pe.plan, pe.gfirst, pe.glast, reinterpret_cast<unsigned long long*>(pe.plan_stats.p));
  f.add(pe.d_req.p, 4 * P, pe.req.data());
  f.add(pe.d_bound.p, 2 * P, pe.bound.data());
  f.add(pe.d_cnt.p, P * P, pe.cnt.data());
  f.add(pe.plan.blk_toff.p + pe.snb, 1, &pe.plan.total);
  f.add(pe.plan_stats.p + 16, 2, pe.plan_hs);
  pe.owner = Xl.slab->val.p;
  pe.owner_serial = dev_alloc_serial(Xl.slab->val.p);
}
bool exchange_fits_fetch(int P) { return (size_t)4 * P + (size_t)P * P + 2 * P <= 400; }

bool slab_exchange_and_step(PSMatrix& B, SlabFusion& fu, double threshold, SlabReduce& red, bool* owed_prefetch) {
  static const bool step_times = std::getenv("NTPOLY_AMD_DEBUG_STEPTIME") != nullptr;   // (host clock of a panel step's phases, rank 0)
  const auto st0 = std::chrono::steady_clock::now();
  auto st_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - st0).count(); };
  double st_prep = 0, st_xchg = 0;
  const long long syncs_before = host_sync_count();
  Comm& c = world();
  Transport& tr = *c.tr;
  const int P = c.nranks, me = c.rank;
  const int32_t dim = B.dim;
  const bool with_counts = options().time_kernels != 0;
  const bool ahead = options().exchange_ahead != 0 && exchange_fits_fetch(P);
  // the preparation: left behind by the step that produced this iterate, or made (and read back) here
  std::unique_ptr<PanelExchange> pe;
  // (matched by ALLOCATION, not by address: the allocator hands a freed address to the next matrix of the size, and a hit
  // decided from an address could differ between ranks -- one rank skipping an all-gather the others enter.  An allocation
  // serial names one buffer of one step's result; the steps are collective, so either every rank holds the preparation
  // of this iterate or none does)
  if (g_pending_exchange && g_pending_exchange->owner == B.loc.slab->val.p && g_pending_exchange->owner_serial != 0 &&
      g_pending_exchange->owner_serial == dev_alloc_serial(B.loc.slab->val.p) && g_pending_exchange->dim == dim && g_pending_exchange->P == P &&
      g_pending_exchange->with_counts == with_counts) {
    pe = std::move(g_pending_exchange);
    g_exchange_prefetched += 1;
  } else {
    g_pending_exchange.reset();
    pe.reset(new PanelExchange());
    if (exchange_fits_fetch(P)) {
      ScalarFetch f;
      exchange_prepare(B.loc, dim, nullptr, *pe, f);
      f.run();
    } else {
      ScalarFetch f;   // (too many ranks for one fetch: plain copies)
      exchange_prepare(B.loc, dim, nullptr, *pe, f);
      f.n = 0;
      HIP_CHECK(hipMemcpyAsync(pe->req.data(), pe->d_req.p, (size_t)4 * P * 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(pe->bound.data(), pe->d_bound.p, (size_t)2 * P * 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(pe->cnt.data(), pe->d_cnt.p, (size_t)P * P * 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(&pe->plan.total, pe->plan.blk_toff.p + pe->snb, 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(pe->plan_hs, pe->plan_stats.p + 16, 16, hipMemcpyDeviceToHost, stream()));
      sync_stream();
    }
  }
  st_prep = st_ms();
  const int pitch = pe->pitch;
  std::vector<int64_t>&req = pe->req, &cnt = pe->cnt;
  const int64_t *d_ext_all = pe->d_ext_all(), *d_pre_all = pe->d_pre_all(), *d_cnt_all = pe->d_cnt_all();
  SlabPlan& plan = pe->plan;
  plan.max_w = (int)pe->plan_hs[0];
  plan.max_kn = (int)pe->plan_hs[1];
  exchange_stats().host_syncs += host_sync_count() - syncs_before;   // (the exchange's own: measured in sync_stream)
  exchange_stats().exchanges += 1;
  int64_t nnz_global = 0;
  for (int q = 0; q < P; ++q) nnz_global += req[(size_t)4 * q + 2];
  auto kmin_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? 0 : (int32_t)lo; };
  auto kmax_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? -1 : (int32_t)hi; };
  const int32_t kmin = kmin_of(me), kmax = kmax_of(me);
  // what I send (my columns inside every requester's range, packed per requester) and what I receive (the segments of the
  // other owners tile [kmin, kmax] in rank order): panel_exchange_layout, host arithmetic on the gathered requests and counts
  std::vector<int32_t> sa((size_t)P), sb((size_t)P), ra((size_t)P), rb((size_t)P);
  std::vector<int64_t> soff((size_t)P + 1, 0), zoff((size_t)P + 1, 0);
  panel_exchange_layout(dim, P, me, req.data(), cnt.data(), sa.data(), sb.data(), soff.data(), ra.data(), rb.data(), zoff.data());
  DevBuf<double> sendbuf((size_t)soff[(size_t)P] + 1);
  for (int q = 0; q < P; ++q)
    if (q != me && cnt[(size_t)me * P + q] > 0)
      slab_pack_runs_async(B.loc, d_pre_all + (size_t)me * pitch, sa[(size_t)q] - B.c0, sb[(size_t)q] - B.c0,
                           sendbuf.p + soff[(size_t)q]);
  DevBuf<double> recvbuf((size_t)zoff[(size_t)P] + kIndexSlack);
  tr.group_begin();
  for (int q = 0; q < P; ++q) {
    const int64_t m = cnt[(size_t)me * P + q];
    if (q != me && m > 0) tr.send(sendbuf.p + soff[(size_t)q], (size_t)m * sizeof(double), q);
  }
  for (int s = 0; s < P; ++s) {
    const int64_t m = cnt[(size_t)s * P + me];
    if (s != me && m > 0) tr.recv(recvbuf.p + zoff[(size_t)s], (size_t)m * sizeof(double), s);
  }
  tr.group_end();
  st_xchg = st_ms();
  if (kmax < kmin) {   // (an empty panel: nothing to multiply; the caller's consensus takes the other path)
    if (ahead) *owed_prefetch = true;
    return false;
  }
  // layout of the columns I need
  const int32_t ka = kmin, kb = kmax + 1;
  DevBuf<int32_t> d_ra((size_t)P), nfirst((size_t)(kb - ka)), nlast((size_t)(kb - ka));
  DevBuf<int64_t> d_zoff((size_t)P);
  DevBuf<unsigned long long> naddr((size_t)(kb - ka));
  if (P > 16) {   // (up to 16 ranks the segments travel as kernel arguments)
    d_ra.upload(ra.data(), (size_t)P);
    d_zoff.upload(zoff.data(), (size_t)P);
  }
  DevBuf<int32_t> ncount;
  if (d_cnt_all) ncount.alloc((size_t)(kb - ka));
  slab_halo_layout_async(d_ext_all, d_pre_all, pitch, dim, P, me, ka, kb, d_ra.p, d_zoff.p, recvbuf.p, B.loc, nfirst.p,
                         nlast.p, naddr.p, d_cnt_all, ncount.p, ra.data(), zoff.data());
  SlabHalo halo;
  halo.ka = ka;
  halo.kb = kb;
  halo.first = nfirst.p;
  halo.last = nlast.p;
  halo.addr = naddr.p;
  halo.count = ncount.p;
  halo.row_pad = std::max(1, B.loc.slab->row_pad);   // (every rank packs with the alignment of its panel: the same option everywhere)
  halo.plan = &plan;
  red.allreduce = &dev_allreduce4;
  halo.reduce = &red;
  const double denom = (double)dim * (double)dim;
  const bool dense_rule = denom > 0 && (double)nnz_global / denom > 0.1;
  // the NEXT step's preparation, from this step's result, on this step's read-back (option exchange_ahead)
  std::unique_ptr<PanelExchange> next;
  if (ahead) {
    halo.before_fetch = [&](const DevMat& R, const long long* d_nnz, ScalarFetch& f) {
      next.reset(new PanelExchange());
      exchange_prepare(R, dim, d_nnz, *next, f);
    };
  }
  // (the buffers above are released on return: the allocator is stream ordered, and slab_step ends with a read-back)
  const bool ok = slab_step(B.loc, fu, threshold, dense_rule, &halo);
  if (step_times && me == 0)
    std::fprintf(stderr, "[panel step] preparation %.3f ms (%s), halo exchange %.3f ms (%lld doubles out, %lld in), step %.3f ms\n", st_prep,
                 g_exchange_prefetched ? "left by the step before or made" : "made", st_xchg - st_prep, (long long)soff[(size_t)P],
                 (long long)zoff[(size_t)P], st_ms() - st_xchg);
  if (ahead && !next) *owed_prefetch = true;   // (this rank gave up before its kernel: it owes the others the all-gather)
  if (ok && next) g_pending_exchange = std::move(next);
  return ok;
}

// TRS2 step across ranks with the fused kernel (mode 1: X <- X*X, mode 2: X <- 2X - X*X; energy and trace in out).
// true: done on every rank.  false: nothing changed (the panel is in compressed columns again), the caller runs
// the separate passes.  Collective.
bool dist_fused_step(PSMatrix& B, int mode, double threshold, const PSMatrix& D, double out[4]) {
  if (!B.loc.expanded()) return false;
  const int P = world().nranks;
  SlabFusion fu;
  fu.mode = mode;
  fu.am = -1.0;
  fu.bm = 2.0;
  fu.threshold = threshold;
  fu.D = &D.loc;
  fu.col_offset = B.c0;
  fu.panel_c0 = B.c0;
  SlabReduce red;
  bool owed_prefetch = false;
  const bool ok = slab_exchange_and_step(B, fu, threshold, red, &owed_prefetch);
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (red.done) {   // the sums came back with the step's totals
    for (int q = 0; q < 4; ++q) v[q] = red.reduced[q];
  } else {          // this rank gave up before its kernel: it still owes the other ranks the collective
    comm_allreduce_sum(v, 4);
  }
  if (owed_prefetch && !red.done) {   // ... and the all-gather of the next step's preparation, which the others will discard
    int wcols = 0;
    const int pitch = panel_pitch(B.dim, P, options().time_kernels != 0, &wcols);
    DevBuf<int64_t> dummy((size_t)P * pitch);
    dummy.zero();
    world().tr->allgather(dummy.p + (size_t)world().rank * pitch, dummy.p, (size_t)pitch * sizeof(int64_t));
    sync_stream();
  }
  (void)ok;
  if (v[3] != (double)P) g_pending_exchange.reset();
  if (v[3] == (double)P) {
    B.loc = std::move(fu.result);
    out[0] = v[0];
    out[1] = 0.0;
    out[2] = v[2];
    out[3] = 0.0;
    return true;
  }
  pack(B.loc);   // (some rank could not: every rank repeats the step on the unfused, equally collective path)
  return false;
}
}  // namespace

// TRS2, sigma > 0 (DensityMatrixSolversModule.F90:388-396): X2 = X*X; X = 2X - X2; energy = dot(X, D).  When the
// register-slab kernel computes X*X the product is never compacted: the merge kernel reads it from its slots.
namespace {
// a fused TRS2 step reads the iterate's multiplier tiles; an iterate that the slab algebra left in slab form (runs only,
// or the read-only view of a matrix with stored zeros) goes back to compressed columns first
void trs2_iterate_form(PSMatrix& B) {
  if (!B.loc.expanded()) return;
  // (the MFMA tile kernel on one rank builds its multiplier tiles from the runs; the unfused loop and the panel steps
  // read them from memory)
  const bool need_tiles = options().spgemm_fma != 1 || world().active();
  const bool no_tiles = B.loc.slab->tiles.p == nullptr || B.loc.slab->tile_off.p == nullptr;
  if (B.loc.slab->origin || (need_tiles && no_tiles)) pack(B.loc);
}
}  // namespace

namespace {
// TRS2 steps of an iterate without run structure: block form from step to step (spgemm_block.hip), one rank, real
bool trs2_block(PSMatrix& B, int mode, double threshold, const PSMatrix& D, double out[4]) {
  if (world().active() || B.cplx || D.cplx || (B.grid && B.grid->num_slices > 1) || options().fused_update == 0 || options().loose_iterates == 0) {
    if (B.loc.blocked()) pack(B.loc);
    return false;
  }
  if (!B.loc.blocked() && (B.loc.expanded() || B.loc.loose() || !B.loc.block_hint)) return false;
  const double denom = (double)B.dim * (double)B.dim;
  const bool dense_rule = denom > 0 && (double)B.loc.nnz / denom > 0.1;
  if (trs2_block_step(B.loc, mode, threshold, dense_rule, D.loc, out)) return true;
  if (B.loc.blocked()) pack(B.loc);
  return false;
}
}  // namespace

void ps_square_update_dot(PSMatrix& B, PSMatrix& scratch, double threshold, const PSMatrix& D, double out[4], bool want_trace) {
  use_grid_comm(D.grid);
  out[2] = out[3] = 0.0;
  if (trs2_block(B, 2, threshold, D, out)) return;
  trs2_iterate_form(B);
  // (process slices: the K-split sums of ps_multiply; a solve in a block order across ranks, band_scope.cpp: the panel product
  // on the block path of multiply_panel)
  if (B.cplx || D.cplx != B.cplx || (B.grid && B.grid->num_slices > 1) || (block_scope_active() && world().active())) {
    pack(B.loc);
    ps_multiply(B, B, scratch, 1.0, 0.0, threshold);
    ps_axpby_dot(scratch, B, -1.0, 2.0, threshold, D, out, want_trace);
    return;
  }
  const double denom = (double)B.dim * (double)B.dim;
  int64_t nz[2] = {B.loc.nnz, B.loc.nnz};
  LooseProduct L;
  DevMat AB;
  // one rank: the iterate may stay loose from step to step (kernels.hpp, axpby keep_loose)
  const bool keep_loose = !world().active() && options().loose_iterates != 0;
  const bool dist_fused = world().active() && options().fused_update != 0 && options().loose_iterates != 0;
  if (dist_fused && dist_fused_step(B, 2, threshold, D, out)) return;
  if (!keep_loose) pack(B.loc);
  if (world().active()) {
    HaloExchange hx;
    gather_needed_begin(hx, B, B.loc, nz, false);
    hx.finish();
    const bool dense_rule = denom > 0 && std::min((double)nz[0] / denom, (double)nz[1] / denom) > 0.1;
    const ColRange need{hx.kmin, hx.kmax + 1};
    SlabFusion fu;   // (as on one rank; B is the panel [c0, c1) of the iterate whose needed columns hx.full holds)
    fu.mode = dist_fused ? 2 : 0;
    fu.am = -1.0;
    fu.bm = 2.0;
    fu.threshold = threshold;
    fu.D = &D.loc;
    fu.col_offset = B.c0;
    fu.panel_c0 = B.c0;
    spgemm(hx.full, B.loc, AB, 1.0, threshold, dense_rule, &L, &need, fu.mode ? &fu : nullptr);
    if (dist_fused) {
      // the ranks must agree on the form of the iterate (the next step's exchange depends on it): slab form only if
      // every rank's kernel produced it
      if (fu.done) {
        out[0] = fu.dot;
        out[1] = 0.0;
        out[2] = fu.trace;
      } else if (L.valid) {
        axpby(L, B.loc, -1.0, 2.0, threshold, &D.loc, out, &out[2], B.c0, nullptr, false);
      } else {
        scratch.grid = B.grid; scratch.dim = B.dim; scratch.c0 = B.c0; scratch.c1 = B.c1; scratch.cplx = B.cplx;
        scratch.loc = std::move(AB);
        axpby(scratch.loc, B.loc, -1.0, 2.0, threshold, &D.loc, out, &out[2], B.c0);
      }
      out[3] = fu.done ? 1.0 : 0.0;
      comm_allreduce_sum(out, 4);
      if (fu.done) {
        B.loc = std::move(fu.result);
        if (out[3] != (double)world().nranks) pack(B.loc);
      }
      out[3] = 0.0;
      return;
    }
  } else {
    const bool dense_rule = denom > 0 && std::min((double)nz[0] / denom, (double)nz[1] / denom) > 0.1;
    SlabFusion fu;   // the whole update inside the multiply's epilogue when the register-slab kernel takes it
    fu.mode = keep_loose && options().fused_update ? 2 : 0;
    fu.am = -1.0;
    fu.bm = 2.0;
    fu.threshold = threshold;
    fu.D = &D.loc;
    fu.col_offset = B.c0;
    // an operand without run structure may hide a band under its labels (kernels.hpp relabel_enter)
    if (fu.mode && !B.loc.expanded() && !B.loc.loose()) relabel_enter(B.loc, D.loc);
    if (B.loc.expanded() && B.loc.slab->labelled()) {
      fu.D = relabelled_operand(D.loc);
      if (!fu.D) {   // (D changed under the loop: back to the caller's labels)
        pack(B.loc);
        fu.D = &D.loc;
      }
    }
    if (B.loc.expanded()) {   // the iterate is in the kernel's own form already: no preparation pass at all
      if (fu.mode && slab_step(B.loc, fu, threshold, dense_rule)) {
        out[0] = fu.dot;
        out[1] = 0.0;
        out[2] = fu.trace;
        return;
      }
      if (B.loc.slab->labelled()) relabel_giveup(D.loc);
      pack(B.loc);
      fu.D = &D.loc;
    }
    spgemm(B.loc, B.loc, AB, 1.0, threshold, dense_rule, &L, nullptr, fu.mode ? &fu : nullptr);
    if (fu.done) {
      B.loc = std::move(fu.result);
      out[0] = fu.dot;
      out[1] = 0.0;
      out[2] = fu.trace;
      return;
    }
  }
  if (L.valid) {
    axpby(L, B.loc, -1.0, 2.0, threshold, &D.loc, out, want_trace ? &out[2] : nullptr, B.c0, nullptr, keep_loose);
  } else {
    pack(B.loc);
    scratch.grid = B.grid; scratch.dim = B.dim; scratch.c0 = B.c0; scratch.c1 = B.c1; scratch.cplx = B.cplx;
    scratch.loc = std::move(AB);
    axpby(scratch.loc, B.loc, -1.0, 2.0, threshold, &D.loc, out, want_trace ? &out[2] : nullptr, B.c0);
  }
  if (last_spgemm_stats().block) B.loc.block_hint = 1;   // (the next step takes the iterate in block form: trs2_block)
  comm_allreduce_sum(out, want_trace ? 3 : 2);
}

// B <- B * B, out = dot(B_new, D) (+ trace(B_new)): the sigma < 0 step of TRS2.  On one rank with real operands the
// product stays loose (no compaction pass); otherwise multiply, swap and reduce as before.
void ps_square_dot(PSMatrix& B, PSMatrix& scratch, double threshold, const PSMatrix& D, double out[4], bool want_trace) {
  use_grid_comm(D.grid);
  out[2] = out[3] = 0.0;
  if (trs2_block(B, 1, threshold, D, out)) return;
  trs2_iterate_form(B);
  const bool keep_loose = !world().active() && options().loose_iterates != 0 && !B.cplx && !D.cplx &&
                          !(B.grid && B.grid->num_slices > 1);
  if (keep_loose) {
    const double denom = (double)B.dim * (double)B.dim;
    const bool dense_rule = denom > 0 && (double)B.loc.nnz / denom > 0.1;
    if (square_keep_loose(B.loc, threshold, dense_rule, D.loc, out, want_trace ? &out[2] : nullptr, B.c0)) return;
  }
  const bool dist_fused = world().active() && options().fused_update != 0 && options().loose_iterates != 0 && !B.cplx &&
                          !D.cplx && !(B.grid && B.grid->num_slices > 1) && !block_scope_active();
  if (dist_fused) {
    if (dist_fused_step(B, 1, threshold, D, out)) return;
    // from compressed panels: the product with the fused epilogue (energy, trace, slab form) where the kernel takes it
    pack(B.loc);
    const double denom = (double)B.dim * (double)B.dim;
    int64_t nz[2] = {B.loc.nnz, B.loc.nnz};
    HaloExchange hx;
    gather_needed_begin(hx, B, B.loc, nz, false);
    hx.finish();
    const bool dense_rule = denom > 0 && std::min((double)nz[0] / denom, (double)nz[1] / denom) > 0.1;
    const ColRange need{hx.kmin, hx.kmax + 1};
    SlabFusion fu;
    fu.mode = 1;
    fu.D = &D.loc;
    fu.col_offset = B.c0;
    fu.panel_c0 = B.c0;
    DevMat AB;
    spgemm(hx.full, B.loc, AB, 1.0, threshold, dense_rule, nullptr, &need, &fu);
    if (fu.done) {
      out[0] = fu.dot;
      out[1] = 0.0;
      out[2] = fu.trace;
    } else {
      B.loc = std::move(AB);
      dot_trace(B.loc, D.loc, out, &out[2], B.c0);
    }
    out[3] = fu.done ? 1.0 : 0.0;
    comm_allreduce_sum(out, 4);
    if (fu.done) {
      B.loc = std::move(fu.result);
      if (out[3] != (double)world().nranks) pack(B.loc);
    }
    out[3] = 0.0;
    return;
  }
  pack(B.loc);
  ps_multiply(B, B, scratch, 1.0, 0.0, threshold);
  std::swap(B.loc, scratch.loc);  // B <- B*B; scratch is recomputed by the next multiply, so no copy
  if (last_spgemm_stats().block) B.loc.block_hint = 1;
  ps_dot_trace(B, D, out, want_trace);
}

// dot(A, B) and trace(A) from one pass
void ps_dot_trace(const PSMatrix& A, const PSMatrix& B, double out[4], bool want_trace) {
  use_grid_comm(A.grid);
  unblock({&A, &B});
  out[2] = out[3] = 0.0;
  if (A.cplx != B.cplx) {
    ps_dot(A, B, out);
    if (want_trace) out[2] = ps_trace(A);
    return;
  }
  dot_trace(A.loc, B.loc, out, want_trace ? &out[2] : nullptr, A.c0);
  comm_allreduce_sum(out, want_trace ? 3 : 2);
}

void ps_pairwise(const PSMatrix& A, const PSMatrix& B, PSMatrix& C) {
  use_grid_comm(A.grid);
  unblock({&A, &B});
  if (A.cplx != B.cplx) {
    PSMatrix Ac, Bc;
    ps_to_complex(A, Ac);
    ps_to_complex(B, Bc);
    ps_pairwise(Ac, Bc, C);
    return;
  }
  DevMat R;
  pairwise(A.loc, B.loc, R, false);
  C.grid = A.grid; C.dim = A.dim; C.c0 = A.c0; C.c1 = A.c1;
  C.cplx = A.cplx;
  C.loc = std::move(R);
}

// DotMatrix_psr/psc (PSMatrixAlgebraModule.F90:387-410, distributed_algebra_includes/DotMatrix.f90):
// sum conj(A).B; fused, no Hadamard temporary.
void ps_dot(const PSMatrix& A, const PSMatrix& B, double out[2]) {
  use_grid_comm(A.grid);
  if (blk_any({&A, &B}) && !A.cplx && !B.cplx) {
    double d = 0.0;
    // (the sum runs over the super-tiles of its first operand: the one in block form)
    const bool ok = A.loc.blocked() ? block_dot_trace(A.loc, B.loc, &d, nullptr) : block_dot_trace(B.loc, A.loc, &d, nullptr);
    if (ok) { g_block_counts[0] += 1; out[0] = d; out[1] = 0.0; return; }
    g_block_counts[1] += 1;
  }
  unblock({&A, &B});
  if (slab_on() && (A.loc.expanded() || B.loc.expanded()) && !A.cplx && !B.cplx) {
    // (a rank whose operands left slab form computes its share from compressed columns: the sum over the ranks follows either way)
    if (slab_enter(mut(A)) && slab_enter(mut(B)) && slab_dot(A.loc, B.loc, out)) {
      g_slab_counts[2] += 1;
      comm_allreduce_sum(out, 2);
      return;
    }
    slab_refused({&A, &B});
  } else {
    slab_pack_if({&A, &B});
  }
  if (A.cplx != B.cplx) {
    PSMatrix Ac, Bc;
    ps_to_complex(A, Ac);
    ps_to_complex(B, Bc);
    dot(Ac.loc, Bc.loc, out);
  } else {
    dot(A.loc, B.loc, out);
  }
  comm_allreduce_sum(out, 2);
}

double ps_trace(const PSMatrix& A) {
  use_grid_comm(A.grid);  // MatrixTrace (distributed_algebra_includes/MatrixTrace.f90)
  if (blk_any({&A}) && !A.cplx) {
    double t = 0.0;
    if (block_dot_trace(A.loc, A.loc, nullptr, &t)) { g_block_counts[0] += 1; return t; }
    g_block_counts[1] += 1;
  }
  unblock({&A});
  if (slab_on() && A.loc.expanded() && !A.cplx) {
    double v = 0.0;
    if (slab_trace(A.loc, A.c0, &v)) {
      g_slab_counts[2] += 1;
      comm_allreduce_sum(&v, 1);
      return v;
    }
    slab_refused({&A});
  } else {
    slab_pack_if({&A});
  }
  double t = trace(A.loc, A.c0);
  comm_allreduce_sum(&t, 1);
  return t;
}

double ps_norm(const PSMatrix& A) {
  use_grid_comm(A.grid);  // MatrixNorm: max column abs-sum; columns are local
  if (blk_any({&A}) && !A.cplx) {
    double v = 0.0;
    if (block_norm(A.loc, &v)) { g_block_counts[0] += 1; return v; }
    g_block_counts[1] += 1;
  }
  unblock({&A});
  if (slab_on() && g_complex_session && A.cplx && A.loc.expanded()) {
    double v = 0.0;
    if (slab_norm_c(A.loc, &v)) { g_slab_counts[2] += 1; return v; }
  }
  if (slab_on() && A.loc.expanded()) {
    double v = 0.0;
    if (slab_norm(A.loc, &v)) {
      g_slab_counts[2] += 1;
      comm_allreduce_max(&v, 1);
      return v;
    }
    slab_refused({&A});
  } else {
    slab_pack_if({&A});
  }
  DevBuf<double> cs;
  column_abs_sums(A.loc, cs);
  double n = max_of(cs, (size_t)A.loc.cols);
  comm_allreduce_max(&n, 1);
  return n;
}

double ps_sigma(const PSMatrix& A) {
  use_grid_comm(A.grid);  // MatrixSigma (distributed_algebra_includes/MatrixSigma.f90)
  const double n = ps_norm(A);
  return 1.0 / (n * n);
}

void ps_gershgorin(const PSMatrix& A, double* e_min, double* e_max) {
  use_grid_comm(A.grid);  // GershgorinBounds.f90:1-41
  unblock({&A});
  double mn, mx;
  if (slab_on() && A.loc.expanded()) {
    if (slab_gershgorin(A.loc, A.c0, &mn, &mx)) {
      g_slab_counts[2] += 1;
      comm_allreduce_min(&mn, 1);
      comm_allreduce_max(&mx, 1);
      *e_min = mn;
      *e_max = mx;
      return;
    }
    slab_refused({&A});
  } else {
    slab_pack_if({&A});
  }
  gershgorin(A.loc, A.c0, &mn, &mx);
  comm_allreduce_min(&mn, 1);
  comm_allreduce_max(&mx, 1);
  *e_min = mn;
  *e_max = mx;
}

void ps_transpose(const PSMatrix& A, PSMatrix& AT) {
  use_grid_comm(A.grid);  // TransposeMatrix_ps
  unblock({&A});
  DevMat R;
  if (world().active()) {
    DevMat full = ps_gather_full(A);
    R = transpose_slice(full, A.c0, A.c1);
  } else {
    R = transpose(A.loc);
  }
  AT.grid = A.grid; AT.dim = A.dim; AT.c0 = A.c0; AT.c1 = A.c1;
  AT.cplx = A.cplx;
  AT.loc = std::move(R);
}

void ps_conjugate(PSMatrix& A) {
  use_grid_comm(A.grid); conjugate(A.loc); }

bool ps_is_identity(const PSMatrix& A) {
  use_grid_comm(A.grid);  // distributed_includes/IsIdentity.f90:7-38
  int64_t d = identity_check(A.loc, A.c0);
  int64_t v[2] = {d < 0 ? 1 : 0, d < 0 ? 0 : d};
  comm_allreduce_sum_i64(v, 2);
  return v[0] == 0 && v[1] == A.dim;
}

double ps_measure_asymmetry(const PSMatrix& A) {
  use_grid_comm(A.grid);  // MeasureAsymmetry: norm(A - A^H)
  PSMatrix T;
  ps_transpose(A, T);
  ps_conjugate(T);
  ps_increment(A, T, -1.0, 0.0);
  return ps_norm(T);
}

void ps_symmetrize(PSMatrix& A) {
  use_grid_comm(A.grid);  // SymmetrizeMatrix: A <- (A + A^H)/2
  PSMatrix T;
  ps_transpose(A, T);
  ps_conjugate(T);
  ps_increment(T, A, 1.0, 0.0);
  ps_scale(A, 0.5);
}

void ps_similarity(const PSMatrix& A, const PSMatrix& P, const PSMatrix& PInv, PSMatrix& Res, double threshold) {
  use_grid_comm(A.grid);
  // SimilarityTransform (PSMatrixAlgebraModule.F90:603-654)
  if (ps_is_identity(P)) {
    ps_copy(A, Res);
    return;
  }
  PSMatrix Temp, Out;
  ps_multiply(P, A, Temp, 1.0, 0.0, threshold);
  ps_multiply(Temp, PInv, Out, 1.0, 0.0, threshold);
  Res = std::move(Out);
}

// ------------------------------------------------------------------ permutations
void permutation_default(Permutation& p, int n) {
  p.index_lookup.resize((size_t)n);
  p.reverse_index_lookup.resize((size_t)n);
  for (int i = 0; i < n; ++i) p.index_lookup[(size_t)i] = p.reverse_index_lookup[(size_t)i] = i + 1;
}
void permutation_reverse(Permutation& p, int n) {
  p.index_lookup.resize((size_t)n);
  p.reverse_index_lookup.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    p.index_lookup[(size_t)i] = n - i;
    p.reverse_index_lookup[(size_t)i] = i + 1;  // as PermutationModule.F90:66 (only index_lookup is used)
  }
}
void permutation_random(Permutation& p, int n) {
  // Fisher-Yates on the root, broadcast to everybody (PermutationModule.F90:92-107 shares the
  // root's permutation the same way).  The reference draws from the Fortran RNG; any permutation
  // gives the same result matrix up to the threshold.
  permutation_default(p, n);
  static std::mt19937_64 rng(42);
  for (int i = n - 1; i > 0; --i) {
    const int j = (int)(rng() % (uint64_t)(i + 1));
    std::swap(p.index_lookup[(size_t)i], p.index_lookup[(size_t)j]);
  }
  comm_bcast_i32(p.index_lookup.data(), n, 0);
  for (int i = 0; i < n; ++i) p.reverse_index_lookup[(size_t)p.index_lookup[(size_t)i] - 1] = i + 1;
}

// PermuteMatrix / UndoPermuteMatrix (LoadBalancerModule.F90:14-92).  The reference multiplies by two
// permutation matrices; out(i,j) = in(perm(i), perm(j)) is computed here by re-indexing and a device
// sort, with the same result (including the pruning of stored zeros by the threshold-0 products).
void ps_permute(const PSMatrix& in, PSMatrix& out, const Permutation& perm, bool undo) {
  use_grid_comm(in.grid);
  const int n = in.dim;
  if ((int)perm.index_lookup.size() != n) NTP_FATAL("permutation size does not match the matrix");
  std::vector<int32_t> map((size_t)n);
  if (!undo) {
    // entry (r, c) moves to (inv[r], inv[c])
    for (int i = 0; i < n; ++i) map[(size_t)perm.index_lookup[(size_t)i] - 1] = i;
  } else {
    for (int i = 0; i < n; ++i) map[(size_t)i] = perm.index_lookup[(size_t)i] - 1;
  }
  DevBuf<int32_t> dmap((size_t)n);
  dmap.upload(map.data(), (size_t)n);
  DevMat R;
  if (world().active()) {
    DevMat full = ps_gather_full(in);
    R = remap_general(full, dmap.p, dmap.p, n, in.c0, in.c1, true);
  } else {
    R = remap_general(in.loc, dmap.p, dmap.p, n, 0, n, true);
  }
  sync_stream();
  const ProcessGrid* g = in.grid;
  const bool cplx = in.cplx;
  const int32_t c0 = in.c0, c1 = in.c1;
  out.grid = g; out.dim = n; out.cplx = cplx; out.c0 = c0; out.c1 = c1;
  out.loc = std::move(R);
}

}  // namespace ntp
