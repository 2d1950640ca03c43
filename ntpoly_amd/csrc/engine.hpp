// Host-side engine: process grid, distributed matrix (column panels, one per GPU),
// distributed algebra, solver parameters, convergence monitor, logger and the solvers.
// Mirrors the reference's module API (names cite the Fortran modules) in C++ because the
// reference host is compiled code; the C ABI in wrp.cpp is the reference's own *_wrp surface.
#pragma once
#include <cstdio>
#include <string>
#include <functional>
#include <vector>

#include "common.hpp"
#include "kernels.hpp"

namespace ntp {

// ------------------------------------------------------------------ communication
// One communicator over all ranks (one process per GPU).  nranks == 1 needs no transport.
// The data plane is RCCL over xGMI.  A second transport exists for TESTS only (NTPOLY_AMD_COMM=shm:<name>): the
// ranks are processes sharing ONE GPU and exchange through a POSIX shared-memory segment, which lets the multi-rank
// code paths (halo exchange, panel gathers, distributed solvers) run on a single-GPU box.
struct Transport {
  virtual ~Transport() {}
  // all pointers are device pointers; sizes in bytes; operations are ordered on the engine stream
  virtual void allgather(const void* send, void* recv, size_t bytes_per_rank) = 0;
  virtual void allreduce(void* buf, size_t count, bool is_f64, int op /*0 sum, 1 min, 2 max*/) = 0;
  virtual void bcast(const void* send, void* recv, size_t bytes, int root) = 0;
  virtual void group_begin() = 0;
  virtual void send(const void* p, size_t bytes, int peer) = 0;
  virtual void recv(void* p, size_t bytes, int peer) = 0;
  virtual void group_end() = 0;
  // stream the following point-to-point group is enqueued on (RCCL); the shared-memory test transport is synchronous
  virtual void set_stream(hipStream_t) {}
  // a transport over the sub-group `members` (ranks of THIS transport, ascending new rank; this rank is members[new_rank]);
  // collective over this transport's ranks: every rank calls it with its own colour (RCCL: ncclCommSplit)
  virtual Transport* split(int color, int key, int new_rank, const std::vector<int>& members) = 0;
};
struct Comm {
  int rank = 0, nranks = 1;
  Transport* tr = nullptr;
  bool force = false;    // tests: run the multi-rank code paths even with a single rank
  bool user_init = false;  // ntpoly_amd_init_comm was called (also with one rank): the caller's MPI communicator is not consulted
  bool active() const { return tr != nullptr && (nranks > 1 || force); }
};
// The communicator the engine's collectives run on.  Normally the one over all processes; while matrices hosted on a
// SUB-grid are worked on (SplitProcessGrid / CommSplitMatrix, ProcessGridModule.F90:430-515) that grid's communicator: the C
// ABI selects it from the grid of the matrices it is handed (use_grid_comm), so the multiply, the reductions and the solvers
// of a half of the processes stay inside that half.
Comm& world();
Comm& base_world();                 // the communicator over all processes, whatever is selected
void use_comm(Comm* c);             // nullptr: the communicator over all processes
// MPI_Comm_split on the current communicator (collective): the ranks of one colour, ordered by (key, rank); the result is
// owned by the engine (kept until comm_finalize)
Comm* comm_split(int color, int key);
// halo exchanges of the distributed multiply since the start, and the host synchronisations they needed
struct ExchangeStats { long long exchanges = 0, host_syncs = 0; };
ExchangeStats& exchange_stats();
void comm_get_unique_id(char out[128]);
void comm_init(const char id[128], int rank, int nranks);
void comm_finalize();
// ranks from the caller's MPI communicator (Fortran handle), when the process has initialised an MPI library and the
// engine has no communicator yet; true when a multi-rank communicator exists afterwards
bool comm_bind_mpi(int fortran_comm);
void comm_allreduce_sum(double* host_vals, int n);
void comm_allreduce_min(double* host_vals, int n);
void comm_allreduce_max(double* host_vals, int n);
void comm_allreduce_sum_i64(int64_t* host_vals, int n);
void comm_bcast_i32(int32_t* host_vals, int n, int root);
void comm_allgather_i64(const int64_t* mine, int n, int64_t* all /* n * nranks */);
void comm_barrier();

// ProcessGrid_t (ProcessGridModule.F90:15-56).  The reference's rows x columns x slices shape is
// kept for the API (getters, consistency check); the data decomposition of this engine is always
// 1-D column panels over the global rank (DESIGN.md "Multi-GPU").
struct ProcessGrid {
  int num_rows = 1, num_cols = 1, num_slices = 1;
  int my_row = 0, my_col = 0, my_slice = 0;
  int global_rank = 0, total = 1;
  Comm* comm = nullptr;   // the communicator the grid lives on (nullptr: all processes); set by split_process_grid
  bool is_root() const { return global_rank == 0; }
};
void use_grid_comm(const ProcessGrid* g);   // selects the communicator of g (collectives that follow run on it)
// SplitProcessGrid (ProcessGridModule.F90:430-515): two grids of about half the size, preferably along the slices, else along
// the longer of rows / columns; collective over the old grid.  The new grid (owned by the engine) lives on a sub-communicator
// made of the processes of this process's colour in the order of their old ranks.
ProcessGrid* split_process_grid(const ProcessGrid& old_grid, int* my_color, bool* split_slice);
// CommSplitMatrix (PSMatrixModule.F90:1489-1541, distributed_includes/CommSplitMatrix.f90): a copy of the WHOLE matrix on each
// of the two halves of its grid (column panels over the half's processes)
struct PSMatrix;
void ps_comm_split(const PSMatrix& m, PSMatrix& split, int* my_color, bool* split_slice);
ProcessGrid& global_grid();
bool global_grid_constructed();
void construct_grid(ProcessGrid& g, int rows, int cols, int slices);
void construct_grid_default(ProcessGrid& g, int slices /* <=0: choose */);
void write_grid_info(const ProcessGrid& g);

// ------------------------------------------------------------------ distributed matrix
// Matrix_ps (PSMatrixModule.F90:33-51): rank r owns the column panel [c0, c1) (all rows).
struct PSMatrix {
  const ProcessGrid* grid = nullptr;
  int32_t dim = 0;  // actual == logical dimension (no padding needed for 1-D panels)
  bool cplx = false;
  int32_t c0 = 0, c1 = 0;
  DevMat loc;       // dim x (c1-c0)
  bool constructed() const { return grid != nullptr; }
};
void panel_range(int32_t dim, int nranks, int rank, int32_t* c0, int32_t* c1);
void ps_construct_empty(PSMatrix& m, int32_t dim, const ProcessGrid* g, bool cplx);
void ps_construct_like(PSMatrix& m, const PSMatrix& ref);
void ps_copy(const PSMatrix& a, PSMatrix& b);
// Slab session (psmatrix.cpp): while one is open -- a solver loop, one rank, real operands, FMA arithmetic, option
// slab_algebra -- ps_multiply / ps_axpby / ps_increment / ps_copy / ps_scale / ps_dot / ps_norm / ps_gershgorin keep
// their operands and results in slab form (kernels.hpp, slab algebra) instead of compressed columns; the loop's owner
// packs what leaves it (ps_slab_leave).  Any operation that cannot be done in slab form packs its operands and takes
// the general path; after a handful of refusals the session stays off.
struct SlabSession {
  // api: a one-call session of the C ABI's vocabulary entry points.  complex_ok: the loop's products, identity increments and
  // norms of differences take COMPLEX operands in slab form too (FMA arithmetic, options complex_tile / complex_sessions);
  // every other operation packs them first
  explicit SlabSession(bool eligible, bool api = false, bool complex_ok = false);
  ~SlabSession();
  SlabSession(const SlabSession&) = delete;
  SlabSession& operator=(const SlabSession&) = delete;
  void close();   // (the loop is over: what follows works on compressed columns again)
  bool opened = false;
  bool set_complex = false;
};
// Out = alpha A + beta B: CopyMatrix(B, Out); ScaleMatrix(Out, beta); IncrementMatrix(A, Out, alpha, threshold) -- in a
// slab session one pass without the copy (and B, an identity say, is turned into slab form once instead of its copies)
void ps_copy_axpby(const PSMatrix& B, const PSMatrix& A, PSMatrix& Out, double alpha, double beta, double threshold);
// IncrementMatrix(Identity, B, alpha, 0) where the caller KNOWS its first operand is the identity (built by
// FillMatrixIdentity, possibly under the load balancer's permutation, which leaves it the identity): in a slab session one
// value per column changes in place; otherwise the ordinary merge
void ps_increment_identity(const PSMatrix& Identity, PSMatrix& B, double alpha);
// TRS4's polynomial chain without its intermediates, for an X and X2 in slab form inside a slab session (false: not
// done, the caller runs the sequence of merges and dots): the two traces, then P = Fx + sigma Gx
bool ps_trs4_traces(const PSMatrix& X, const PSMatrix& X2, double* trace_fx, double* trace_gx);
bool ps_trs4_operand(const PSMatrix& X, const PSMatrix& X2, double sigma, PSMatrix& P);
// MatrixNorm of alpha A + beta B (ScaleMatrix(B, beta); IncrementMatrix(A, B, alpha, 0); MatrixNorm(B)) for the loops
// that build the sum only for its norm; false: not done (outside a slab session, operands in compressed columns ...)
bool ps_norm_axpby(const PSMatrix& A, const PSMatrix& B, double alpha, double beta, double* norm);
void ps_slab_leave(PSMatrix& m);
const long long* column_fused_counts();   // [2] since start: IncrementMatrix(Identity, .) done in place, norms of differences taken without forming them (column_fused.hip)
const long long* block_algebra_counts();  // [2] since start: operations done in block form (spgemm_block.hpp block algebra); fallbacks
long long block_scope_products();   // panel products of block-order solves that took the block path (psmatrix.cpp)
bool block_scope_active();   // band_scope.cpp: the solve in progress runs on operands redistributed in a block order (several ranks)
const long long* panel_product_counts();  // [3] products of slab sessions across ranks: in slab form on every rank; declined; host synchronisations inside the former
const long long* slab_algebra_counts();   // [4] since start: products, merges / copies, other operations done in slab form; refusals   // back to compressed columns (no-op for a matrix that is not in slab form)
void ps_fill_identity(PSMatrix& m);
void ps_fill_permutation(PSMatrix& m, const std::vector<int32_t>& lookup /*1-based*/, bool rows);
void ps_fill_from_triplets(PSMatrix& m, const HostTriplets& t);
void ps_get_triplets(const PSMatrix& m, HostTriplets& t);
int64_t ps_size(const PSMatrix& m);
// PSMatrixModule utilities on the caller side of the path (triplet based, as in the reference)
void ps_fill_dense(PSMatrix& m);                                          // FillMatrixDense: every element 1
void ps_diagonal_scale(PSMatrix& m, const HostTriplets& t);               // MatrixDiagonalScale: column col *= value
void ps_get_block(const PSMatrix& m, int sr, int er, int sc, int ec, HostTriplets& out);  // [sr,er) x [sc,ec), 1-based
void ps_get_slice(const PSMatrix& m, PSMatrix& sub, int sr, int er, int sc, int ec);      // inclusive bounds, 1-based
void ps_resize(PSMatrix& m, int new_size);
void ps_to_complex(const PSMatrix& a, PSMatrix& out);
void ps_to_real(const PSMatrix& a, PSMatrix& out);
void panel_exchange_layout(int32_t dim, int P, int me, const int64_t* req, const int64_t* cnt, int32_t* sa, int32_t* sb, int64_t* soff,
                           int32_t* ra, int32_t* rb, int64_t* zoff);   // psmatrix.cpp: who sends which columns where in a panel exchange (host)
DevMat ps_gather_full(const PSMatrix& m);  // every rank gets the whole matrix (dim x dim)
// range-restricted exchange: a dim x dim matrix holding only the columns [kmin, kmax] of the distributed matrix
// halo exchange for C = A*B: the columns of A named by the rows of the local B panel; also returns the global
// nnz of A and B (collected in the same exchange)
DevMat gather_needed(const PSMatrix& m, const DevMat& Bloc, int64_t nnz_global[2]);
// The same exchange in two steps, so that the caller can multiply the INTERIOR columns of its B panel (those that
// name only local columns of A) while the halo travels on the communication stream: begin() does the two size
// round trips and posts the send / recv group -- on comm_stream when `overlapped` comes back true --, finish() makes
// the engine stream wait for it and builds the column offsets of `full`.
struct HaloExchange {
  DevMat full;                    // dim x dim, columns outside the needed range empty; valid after finish()
  bool overlapped = false;
  int32_t kmin = 0, kmax = -1;    // columns of the distributed matrix present in `full` (the requested range)
  int32_t jl = 0, jr = 0;         // interior columns [jl, jr) of the local B panel (overlapped mode)
  int64_t off_l = 0, off_r = 0;   // their entry offsets in the panel
  // state between begin and finish
  DevBuf<int64_t> stage;
  std::vector<int32_t> ra, rb;
  std::vector<int64_t> zoff, soff, cnt_from;
  int32_t dim = 0;
  int P = 1;
  void finish();
};
void gather_needed_begin(HaloExchange& hx, const PSMatrix& m, const DevMat& Bloc, int64_t nnz_global[2], bool may_overlap);
void halo_segment(int32_t dim, int P, int s, int32_t kmin, int32_t kmax, int32_t* a, int32_t* b);
// concatenate the column panels of all ranks (widths[r] = columns held by rank r, known to all)
DevMat gather_panels(const DevMat& loc, const std::vector<int32_t>& widths);

// PSMatrixAlgebraModule
void ps_multiply(const PSMatrix& A, const PSMatrix& B, PSMatrix& C, double alpha, double beta, double threshold);
void ps_increment(const PSMatrix& A, PSMatrix& B, double alpha, double threshold);
void ps_scale(PSMatrix& A, double c);
void ps_axpby(const PSMatrix& A, PSMatrix& B, double alpha, double beta, double threshold);   // ScaleMatrix + IncrementMatrix, one pass
// B <- alpha*A + beta*B with the increment rules and, fused, out = dot(B_new, D) (TRS2 update + energy)
void ps_axpby_dot(const PSMatrix& A, PSMatrix& B, double alpha, double beta, double threshold, const PSMatrix& D, double out[4],
                  bool want_trace = false);  // out[2] = trace(B_new) on request
void ps_dot_trace(const PSMatrix& A, const PSMatrix& B, double out[4], bool want_trace);
void ps_square_dot(PSMatrix& B, PSMatrix& scratch, double threshold, const PSMatrix& D, double out[4], bool want_trace);
// B <- 2B - B*B (threshold on the product and on the merge, TRS2's sigma > 0 update), out = dot(B_new, D), trace(B_new);
// the product goes from the numeric kernel's slots straight into the merge when the slab kernel computes it
void ps_square_update_dot(PSMatrix& B, PSMatrix& scratch, double threshold, const PSMatrix& D, double out[4], bool want_trace);
void ps_pairwise(const PSMatrix& A, const PSMatrix& B, PSMatrix& C);
void ps_dot(const PSMatrix& A, const PSMatrix& B, double out[2]);
double ps_trace(const PSMatrix& A);
double ps_norm(const PSMatrix& A);
double ps_sigma(const PSMatrix& A);
void ps_gershgorin(const PSMatrix& A, double* e_min, double* e_max);
void ps_transpose(const PSMatrix& A, PSMatrix& AT);
void ps_conjugate(PSMatrix& A);
bool ps_is_identity(const PSMatrix& A);
double ps_measure_asymmetry(const PSMatrix& A);
void ps_symmetrize(PSMatrix& A);
void ps_similarity(const PSMatrix& A, const PSMatrix& P, const PSMatrix& PInv, PSMatrix& Res, double threshold);
// LoadBalancerModule
struct Permutation {
  std::vector<int32_t> index_lookup, reverse_index_lookup;  // 1-based values
};
void permutation_default(Permutation& p, int n);
void permutation_reverse(Permutation& p, int n);
void permutation_random(Permutation& p, int n);
void ps_permute(const PSMatrix& in, PSMatrix& out, const Permutation& perm, bool undo);

// ------------------------------------------------------------------ logger (LoggingModule.F90)
struct Logger {
  bool active = false;
  int level = 0;
  FILE* out = stdout;
  bool owns_file = false;
};
Logger& logger();
void log_activate(bool start_document, const char* file_name);
void log_deactivate();
void log_enter();
void log_exit();
void log_header(const char* h);
void log_element(const char* key, double v);
void log_element(const char* key, int v);
void log_element(const char* key, const char* v);
void log_element(const char* key, bool v);
void log_list_element(const char* key, double v);
void log_list_element(const char* key);

// ------------------------------------------------------------------ solver parameters / monitor
struct Monitor {  // ConvergenceMonitorModule.F90:14-27
  double win_short[3] = {0, 0, 0}, win_long[6] = {0, 0, 0, 0, 0, 0};
  int nval = 0;
  double loose_cutoff = 1e-2, tight_cutoff = 1e-8;
  bool automatic = true;
};
void monitor_construct(Monitor& m, bool automatic, double tight_cutoff);
void monitor_append(Monitor& m, double v);
bool monitor_converged(const Monitor& m, bool be_verbose);

struct SolverParameters {  // SolverParametersModule.F90:14-33, defaults :48-50
  double converge_diff = 1e-6;
  int max_iterations = 1000;
  double threshold = 0.0;
  bool be_verbose = false;
  bool do_load_balancing = false;
  Permutation balance_permutation;
  double step_thresh = 1e-2;
  bool monitor_convergence = true;
};
void print_parameters(const SolverParameters& p);
void print_matrix_information(const PSMatrix& m);

// per-iteration record of the last solver call (tests/bench read it through the C ABI extension)
struct SolverTrace {
  std::vector<double> value, energy, sigma;
  std::vector<int64_t> nnz;
  int iterations = 0;
  double setup_ms = 0, loop_ms = 0;
};
SolverTrace& last_trace();
// one TRS2 iteration (DensityMatrixSolversModule.F90:380-404): returns the energy, sets sigma
// matrix polynomials (solvers_poly.cpp): coefficient i of the vector multiplies x^i / T_i(x) / H_i(x)
void polynomial_horner(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p);
void polynomial_paterson_stockmeyer(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p);
void chebyshev_compute(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p);
void chebyshev_factorized(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p);
void hermite_compute(const PSMatrix& In, PSMatrix& Out, const std::vector<double>& c, const SolverParameters& p);
// matrix functions (solvers_func.cpp)
void power_bounds(const PSMatrix& A, double* max_value, const SolverParameters& p, bool defaults);
void compute_exponential(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p);
void compute_logarithm(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p);
void compute_sine(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p);
void compute_cosine(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p);
void compute_root(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p);
void compute_inverse_root(const PSMatrix& In, PSMatrix& Out, int root, const SolverParameters& p);
// DensityMatrixSolversModule.F90:953-1117, :1165-1187, :1190-1231
void solver_scale_and_fold(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double homo, double lumo,
                           double* energy_out, const SolverParameters& p);
void energy_density_matrix(const PSMatrix& H, const PSMatrix& D, PSMatrix& ED, double threshold);
void mcweeny_step(const PSMatrix& D, PSMatrix& DOut, const PSMatrix* S, double threshold);
double trs2_step(PSMatrix& X, PSMatrix& X2, const PSMatrix& WH, double trace_target, double threshold, double* sigma,
                 double* trace_io = nullptr);

// band_scope.cpp: a solver on several ranks whose first operand hides a band under its labels runs on operands redistributed
// in the recovered order (collective; false: not applicable -- the caller solves as it stands).  run(ins, outs) is the
// solver itself on the relabelled operands; the outputs are carried back to the caller's labels
bool band_scope_try(const std::vector<const PSMatrix*>& ins, const std::vector<PSMatrix*>& outs,
                    const std::function<void(const std::vector<const PSMatrix*>&, const std::vector<PSMatrix*>&)>& run);
const long long* band_scope_counts();

// ------------------------------------------------------------------ solvers
void solver_trs2(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy, double* mu,
                 const SolverParameters& p);
void solver_trs4(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy, double* mu,
                 const SolverParameters& p);
void solver_pm(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy, double* mu,
               const SolverParameters& p);
void solver_hpcp(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, double* energy, double* mu,
                 const SolverParameters& p);
void solver_sign(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p);
void solver_polar(const PSMatrix& A, PSMatrix& U, PSMatrix* Hm, const SolverParameters& p);
void solver_invert(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p);
void solver_pseudoinverse(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p);
void solver_square_root(const PSMatrix& A, PSMatrix& Out, const SolverParameters& p, bool inverse, int order);

// solvers_extra.cpp: linear solvers, Pade exponential, geometry extrapolation, the dense (eigendecomposition) family,
// Fermi-operator solvers, Cholesky factorisations
void ps_filter(PSMatrix& m, double threshold);
void ps_gather_triplets(const PSMatrix& m, HostTriplets& t);
void solver_cg(const PSMatrix& A, PSMatrix& X, const PSMatrix& B, const SolverParameters& p);
void compute_exponential_pade(const PSMatrix& In, PSMatrix& Out, const SolverParameters& p);
void purification_extrapolate(const PSMatrix& PreviousDensity, const PSMatrix& Overlap, double trace, PSMatrix& NewDensity,
                              const SolverParameters& p);
void lowdin_extrapolate(const PSMatrix& PreviousDensity, const PSMatrix& OldOverlap, const PSMatrix& NewOverlap,
                        PSMatrix& NewDensity, const SolverParameters& p);
void snap_to_sparsity_pattern(PSMatrix& mat, const PSMatrix& pattern);
void ps_eigendecomposition(const PSMatrix& A, PSMatrix& eigenvalues, PSMatrix* eigenvectors, int nvals,
                           const SolverParameters& p);
void dense_matrix_function(const PSMatrix& A, PSMatrix& Result, const std::function<double(double)>& func,
                           const SolverParameters& p);
void ps_svd(const PSMatrix& A, PSMatrix& left, PSMatrix& right, PSMatrix& singular, const SolverParameters& p);
void estimate_gap(const PSMatrix& H, const PSMatrix& K, double chemical_potential, double* gap, const SolverParameters& p);
void compute_dense_foe(const PSMatrix& H, const PSMatrix& ISQ, double trace, PSMatrix& K, const double* inv_temp_in,
                       double* energy_out, double* mu_out, const SolverParameters& p);
void solver_wom(const PSMatrix& H, const PSMatrix& ISQ, PSMatrix& K, double inv_temp, const double* trace_in,
                const double* mu_in, double* energy_out, const SolverParameters& p);
void ps_cholesky(const PSMatrix& A, PSMatrix& L, int rank, const SolverParameters& p);
void reduce_dimension(const PSMatrix& A, int dim, PSMatrix& Reduced, const SolverParameters& p);

}  // namespace ntp
