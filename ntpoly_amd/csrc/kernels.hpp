// Launchers of the hand-written gfx950 kernels (kernels.hip).  Everything operates on
// device-resident column-compressed matrices (DevMat) on ctx().stream.
#pragma once
#include <functional>
#include "common.hpp"

namespace ntp {

// Per-call statistics of the SpGEMM (for bench.py's roofline accounting).
struct SpgemmStats {
  int64_t nnz_a = 0, nnz_b = 0, nnz_c = 0;
  int64_t products = 0;        // intermediate products IP = sum_j sum_{k in B(:,j)} nnz(A(:,k)); on the register-slab path
                               // only counted when the time_kernels option is on (it costs a gather per entry of B)
  int64_t tmp_entries = 0;     // upper-bound entries reserved for the numeric pass
  int slab = 0;                // 1 when the register-slab kernel computed the product
  int fused = 0;               // 1 / 2: with the fused epilogue of a purification step (SlabFusion::mode)
  int64_t bin_cols[6] = {0, 0, 0, 0, 0, 0};
  int64_t overflow_cols = 0;   // columns that left the LDS hash for the HBM accumulator
  // grouped LDS-hash path (spgemm_grouped.hip): 1 when it computed the product; columns handed back to the per-column
  // kernels, table class reached, min-hash clustering used, union ratio (1 = the columns of a group are identical)
  int grouped = 0;
  int strips = 0;              // > 0: the product was computed in that many row strips of A (columns with too many distinct rows for
                               // the grouped kernel's tables: kernels.hip spgemm_striped)
  int64_t gh_failed_cols = 0, gh_groups = 0, gh_tile_rows = 0;
  int gh_level = 0, gh_minhash = 0;
  double gh_union_ratio = 0;
  // block path (spgemm_block.hip): 1 when it computed the product; tile fill of A, 16 x 16 x 16 tile products issued
  // (4 matrix instructions each), candidate output super-tiles
  int block = 0;
  double block_fill = 0;
  int64_t block_tile_products = 0, block_cand = 0;
  int thin = 0;                // 1: the thin-left kernel (spgemm_thin.hip) computed the product
  float ms_total = 0.f;        // filled only when timing is enabled
  float ms_numeric = 0.f;
};

struct EngineOptions {
  int spgemm_force_bin = -1;   // tests: force every non-empty column through one path (1..5), 6 = HBM fallback
  int increment_force_seq = 0; // tests: 1 force the sequential-merge fallback, 2 force the rank-merge kernel (columns of <= 2048 entries)
  int time_kernels = 0;        // record HIP-event timings in SpgemmStats
  int spgemm_fma = 1;          // arithmetic of the real products (NTPOLY_AMD_ARITHMETIC=fma|unfused in the environment).
                               // 1 (default since round 4): every product entry is the chain fma(a, b, acc) over ascending k, one
                               // rounding per product -- the reference built with FP contraction, bit for bit; run-like operands on
                               // the FP64 matrix cores (spgemm_tile.hip), operands without runs as 16 x 16 blocks on the same
                               // instruction (spgemm_block.hip).  0: separate multiply and add, bit-identical to the reference's
                               // default x86-64 build (register-slab / LDS kernels).  3: the v_fma_f64 vector loop (tests)
  int block_path = 1;          // FMA arithmetic, one rank, real square operands WITHOUT run structure (3-D Hamiltonians, relabelled bands outside
                               // a TRS2 loop): products on 16 x 16 blocks of a clustered index order through v_mfma_f64_16x16x4_f64
                               // (spgemm_block.hip) instead of the grouped LDS-hash kernel; 0: never; 2: every real square one-rank
                               // product that no run-based kernel takes, whatever the fill (tests)
  int slab_algebra = 1;        // solver loops (TRS4, sign, inverse, square roots; one rank, real, FMA arithmetic): iterates stay in slab form
                               // between products, merges, dots and norms (psmatrix.cpp SlabSession); 0: compressed columns between the operations
  int plan_ahead = 1;          // TRS2 steps on the slab form (tile kernel, one rank): the step plans its successor behind its own kernel and
                               // reads the sizes back with its results -- one host round trip per step instead of two;
                               // 0: every step makes its plan and reads it back before the launch
  int operand_cache = 1;       // 1: the expanded WH (and WH in a recovered band order) of a purification solve stays on the
                               // device for the next solve on the same operand (SCF loops): at most 8 (nnz + 32 n) + 12 nnz + 8 n
                               // bytes, replaced when the operand changes; 0: freed at the end of every solve;
                               // ntpoly_amd_release_cache() frees it at any time
  int tile_rows = 2;           // MFMA tile kernel (spgemm_fma = 1): consecutive rows per lane of the A operand, 1 / 2 / 4 (spgemm_tile.hpp)
  int tile_waves = 0;          // ... waves per workgroup, 4 / 8 (0: chosen from the LDS footprint)
  int thin_left = 1;           // products whose left operand holds a handful of entries per row (identities, near-diagonal factors of the
                               // square-root loops): the output-driven gather kernel of spgemm_thin.hip, both arithmetic modes, bit for bit
 int complex_sessions = 1;    // the complex SignFunction loop keeps its iterates in the complex tile kernel's operand form between products (no
                               // expansion, no pack; FMA arithmetic with complex_tile); 0: compressed columns between the operations
  int column_fused = 1;        // IncrementMatrix(Identity, B) in place and the norm of a difference without forming it, on compressed columns
                               // (column_fused.hip: complex solver loops, real ones outside slab sessions); 0: the merge kernels
  int complex_tile = 1;        // FMA arithmetic: run-like COMPLEX operands on the matrix cores (spgemm_tile_c.hip: two FMA chains per part of an
                               // entry -- a tolerance mode, 1e-13 of the largest entry); 0: the register-slab kernel with the reference's
                               // complex multiply-add, bit for bit (what unfused arithmetic always runs)
  int tile2 = 0;               // run-like real operands in FMA arithmetic: the two-block chunk-streaming geometry of the MFMA kernel (spgemm_tile2.hip) (1) where it fits -- slower than k_spgemm_tile as measured in round 5 (profiles/README.md 83): 0 by default
  int tile_runs_only = 1;      // TRS2 steps on the tile kernel (one rank): the result is written as runs only and the next step builds its
                               // multiplier tiles from them (1.5 GB -> 1.0 GB written per launch at the headline size, no tile read);
                               // 0: runs + multiplier tiles as the unfused loop needs them
  int load_balance = 1;        // 1: solvers permute with the caller's load-balancing permutation as the reference does;
                               // 0: SetParametersLoadBalance is ignored -- the same results up to summation order (a
                               // symmetric permutation only relabels entries), but banded operands stay on the run-based
                               // kernels (a random permutation costs ~25x in SpGEMM time and turns the halo into a full gather)
  int virtual_grid = 0;        // tests: a process grid of any rows x columns x slices may be constructed on the ranks there
                               // are (the shape only selects the summation semantics of the multiply: slices > 1)
  int fused_update = 1;        // TRS2 on one rank, real operands: the update X <- 2X - X*X (or X*X), its energy and its trace
                               // come out of the epilogue of the register-slab kernel; 0: separate merge / reduction passes
  int block_scope = 1;         // several ranks, FMA arithmetic: a solve whose first operand has neither runs nor a hidden band (a 3-D Hamiltonian) runs in the pattern's block order -- operands redistributed so that a rank owns a range of positions, panel products on the block path instead of the LDS hash (band_scope.cpp); 0: the operands as they are
  int block_match = 0;         // block path, an experiment (profiles/README.md 94): 1 = the matches of every candidate super-tile (the K with super-tiles on both sides) are found once, by a kernel of its own (k_bs_match), and read by the numeric kernel; 0: the numeric kernel searches itself.  Measured: no difference (37.4 against 37.2 iterations/s on the 64^3 lattice).  Results do not depend on it
  int block_unfused = 0;       // the block path (spgemm_block.hip) in UNFUSED arithmetic too: products rounded, then added, on the vector units, in ascending POSITION of the block order -- the reference's default build on the matrix relabelled by that order (what its own load balancer does), 1e-13 of the sums over ascending labels.  0 (default): operands without runs keep the label-ordered kernels in unfused arithmetic, bit for bit the reference on the caller's labels
  int panel_sessions = 1;      // slab sessions (TRS4, sign, inverse, square roots, polynomials ...) on more than one rank: the loops' matrices stay in slab form as column panels, a product exchanges the runs of the left operand's halo (psmatrix.cpp panel_slab_multiply); 0: compressed columns across ranks
  int ghash_mfma = 1;          // grouped LDS-hash SpGEMM, real operands, FMA arithmetic: the products of a phase (four steps) as ONE v_mfma_f64_16x16x4_f64 per tile of 16 slots x 16 columns instead of 64 vector FMAs -- the same chain of fma() over ascending k, bit for bit; 0: vector units
  int tile_off32 = 1;          // MFMA tile kernel: the runs of the left operand read through a buffer resource with 32-bit offsets where they lie in ONE allocation below 4 GB (no halo): lanes outside a run get an out-of-range offset and the bounds check returns 0.0 -- four vector instructions per run load instead of seven; 0: 64-bit addresses everywhere
  int tile_bbuf = 2;           // MFMA tile kernel: the multiplier tile of a block read from the runs of its columns through a buffer resource (operand below 4 GB): a row outside a run reads as 0.0 by the bounds check -- no branch and no 64-bit address per element (1); 2 (default): as PAIRS of rows, a wave per group of columns, where the operand's slots are padded to even rows -- a third of the requests; 0: per-element address selection
  int plan_fused = 1;          // the maxima and prefix sums of a slab step's plan in ONE launch (k_slab_offsets: every workgroup sums what lies before its part itself) instead of four to seven; 0: separate launches
  int exchange_ahead = 1;      // panel steps across ranks: a step prepares the NEXT step's exchange (extents all-gathered, counts, plan) from its result and reads it back with its own totals -- one host round trip per panel step (psmatrix.cpp PanelExchange); 0: two
  int band_scope = 1;          // solvers on SEVERAL ranks, FMA arithmetic: an operand without run structure is searched for a hidden band once per solve, the operands are redistributed in the recovered order, the results carried back (band_scope.cpp).  2: in unfused arithmetic too (the results are then the reference's under its load balancer with that permutation, not its bits on the caller's labels); 0: never
  int label_order = 1;         // TRS2 on one rank: an operand without run structure is searched for a hidden band (relabel.hip)
                               // and, if there is one, the loop runs in that order with label-ordered arithmetic; 0: never
  int label_rowoff = 1;        // label-ordered steps: the loop takes a step's multiplier row from its run record (no copy of
                               // the tiles in step order); 0: the tiles are copied in step order before every launch
  int loose_iterates = 1;      // TRS2 on one rank, real operands: the iterate X stays in the slots the update / the slab
                               // kernel wrote it to between the steps (no compaction pass per iteration); 0: packed
  int halo_overlap = 1;        // distributed multiply: 0 exchange then multiply, 1 overlap the exchange with the interior
                               // columns when the halo is a sizeable part of the panel, 2 always split, 3 split even
                               // with an empty halo (tests; also NTPOLY_AMD_HALO_OVERLAP in the environment)
  int spgemm_variant = -1;     // numeric kernel: -1 automatic (register-slab kernels for run-like operands, real and complex,
                               // geometry by the widest row window; else the column-pair kernel for real operands, one
                               // column per wave for complex ones; LDS hash beyond 4096-row windows); 0 one column per wave
                               // for everything (first generation); 3<MAXCH><NW> column-pair kernel with that geometry;
                               // 400 register-slab kernel whenever it fits (also when its run-density test says no);
                               // timing experiments on the three-slab real kernel: 401..404 ablations (WRONG results:
                               // no slab loads / no multiplier loads / no arithmetic / cache-hot multipliers; compiled only
                               // with -DNTP_ABLATIONS, NTPOLY_AMD_EXTRA_FLAGS of ntpoly_amd/_build.py -- the product library
                               // runs the ordinary loop for these values), 405 plain
                               // loop + rotating prefetch, 406 lean periods, 407 both (= the default loop), 408 / 409
                               // two / three workgroups per CU, 410 plain loop and three slabs also for narrow windows;
                               // 500 grouped LDS-hash kernel for every multiply the slab kernels do not take (automatic: when
                               // at least half of the columns have row windows beyond the direct-mapped LDS kernels),
                               // 501 never the grouped kernel (one column per wave LDS hash, the previous path)
};
EngineOptions& options();
SpgemmStats& last_spgemm_stats();
// accumulated since reset: number of calls, products, algorithmic bytes, numeric-kernel ms
struct SpgemmAccum { int64_t calls = 0, products = 0, nnz_c = 0; double alg_bytes = 0, ms_numeric = 0, ms_total = 0; };
SpgemmAccum& spgemm_accum();
// resolve the HIP events recorded by the timed multiplies since the last call (time_kernels option)
void flush_spgemm_timers();

// A product left "loose": every column sits in the upper-bound slot the numeric kernel wrote it to (entries
// start[j] .. start[j] + count[j]), the compaction pass has not run.  Produced by spgemm(..., &loose) on the
// register-slab path and consumed by axpby(loose, ...): the TRS2 update reads X*X straight from the slots.
struct LooseProduct {
  bool valid = false;
  int32_t rows = 0, cols = 0;
  int64_t slots = 0;            // capacity of inner / val (= start[cols])
  DevBuf<int64_t> start;        // cols + 1
  DevBuf<int32_t> count;        // cols
  DevBuf<int32_t> inner;
  DevBuf<double> val;           // real only
  DevBuf<int64_t> prod_scan;    // statistics: [prod_index] = product count of the multiply (when counted)
  int64_t prod_index = -1;
};

// C = alpha * A * B with NTPoly's prune rule.  A: (m x k), B: (k x n), same scalar type.
// dense_rule = the reference's dense-branch order (threshold before alpha, DenseBranch.f90:14-15).
// loose != nullptr: if the register-slab kernel computes the product, leave it uncompacted in *loose (loose->valid,
// C untouched); otherwise C is produced as usual and loose->valid stays false.
// arange: the columns [a, b) of A that the rows of B can name, when the caller knows them (gathered operands are
// dim wide but populated over the halo range only): the per-column planning work then covers that range only
struct ColRange { int32_t a, b; };
// A purification step computed inside the register-slab kernel (A = B = X, real, one rank): the product never exists as
// a matrix.  mode 1: result = X * X; mode 2: result = am * (X * X) + bm * X, merged by the AddSparseVectors rules with
// `threshold`.  Either way dot = sum result .* D and trace = trace(result) come out of the same kernel and the result
// is left LOOSE (no compaction).  done = false on return: the multiply took another path, or the kernel met a case it
// does not decide (SlabFuseArgs in kernels.hip) -- the product was then computed the ordinary way (C / *loose).
struct SlabFusion {
  int mode = 0;
  double am = 0, bm = 0, threshold = 0;
  const DevMat* D = nullptr;
  int32_t col_offset = 0;
  int32_t panel_c0 = -1;      // >= 0: B is the column panel [panel_c0, panel_c0 + cols) of the iterate, A holds those columns
  bool done = false;
  DevMat result;
  double dot = 0, trace = 0;
  int64_t product_nnz = 0;
  int64_t refused = 0;
};
void spgemm(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold,
            bool dense_rule, LooseProduct* loose = nullptr, const ColRange* arange = nullptr, SlabFusion* fuse = nullptr);
// while one is alive, a product that the block path computes (spgemm_block.hip) is left in block form (DevMat::blk): for
// callers that hand it on to another product or pack() it themselves
// One TRS2 step with the iterate in block form (X: compressed columns of a dimension the block path multiplies, or block
// form; left in block form): mode 1: X <- X X, mode 2: X <- 2 X - X X by the AddSparseVectors rules; out[0] = dot(X, D),
// out[2] = trace(X).  false: not taken, X unchanged.
bool trs2_block_step(DevMat& X, int mode, double threshold, bool dense_rule, const DevMat& D, double out[4]);
struct BlockKeepScope {
  BlockKeepScope();
  ~BlockKeepScope();
  BlockKeepScope(const BlockKeepScope&) = delete;
  BlockKeepScope& operator=(const BlockKeepScope&) = delete;
};
// the same step on an iterate already in slab form (DevMat::slab, written by a previous fused step): X is replaced by
// the result (again in slab form).  false: not taken (X unchanged; pack() it and use spgemm)
struct SlabReduce {   // panel steps: the sum over the ranks rides on the step's own read-back
  void (*allreduce)(double* dev4) = nullptr;   // enqueues an all-reduce (sum) of 4 doubles on the engine stream
  double reduced[4] = {0, 0, 0, 0};            // (dot, 0, trace, number of ranks whose kernel succeeded)
  bool done = false;                           // the collective was entered (false: the step gave up before its kernel)
};
struct SlabHalo {   // A side of a panel step: the columns ka .. kb (global numbers) of the distributed iterate
  int32_t ka = 0, kb = 0;
  const int32_t* first = nullptr;            // [kb - ka] device: extents of column ka + i
  const int32_t* last = nullptr;
  const unsigned long long* addr = nullptr;  // [kb - ka] device: address of its run (own buffer or receive buffer)
  const int32_t* count = nullptr;            // [kb - ka] device, optional (statistics): entries of the column
  int row_pad = 1;                           // the received runs sit in slots aligned to this many rows, zero padded (SlabForm::row_pad)
  SlabReduce* reduce = nullptr;
  // optional: called once the step's kernel, totals and reduction are enqueued, BEFORE their read-back, with the result as
  // it stands on the device (slab form; its entry count still on the device: d_nnz) -- what the caller enqueues and adds to
  // the fetch rides on the step's own host round trip (psmatrix.cpp: the NEXT step's exchange layout and plan)
  std::function<void(const DevMat& result, const long long* d_nnz, ScalarFetch& fetch)> before_fetch;
  // optional: the plan of this step, made by the caller from the all-gathered extents and read back together with the
  // exchange layout (slab_plan_panel_async) -- the step then launches without a read-back of its own (and may keep
  // the plan's buffers for its result)
  SlabPlan* plan = nullptr;
  // optional (slab_multiply with a left halo): called with the fetch of the product's entry count before it runs -- what the
  // caller adds comes back on the same host round trip
  std::function<void(ScalarFetch& fetch)> on_fetch;
};
// (could the tile kernel take a plan with these maxima: psmatrix.cpp decides a panel product before its exchange is over)
bool slab_plan_fits_tile(int max_kn, int max_w);
struct SlabPlan;
// true: slab_multiply with a left halo of these columns, this alignment and this plan will NOT decline (the one predicate it uses itself)
bool slab_multiply_takes_panel(const DevMat& A, const DevMat& B, int left_row_pad, int32_t ka, int32_t kb, const SlabPlan* plan);
// the plan of a panel step from the all-gathered packed extents (record stride `pitch`, extents at d_ext_all): flat
// extent arrays of all `dim` columns are left in gfirst / glast, the plan's sizes on the device in plan.blk_toff[blocks]
// and stats24[16..17] (stats24: 24 zeroed words)
void slab_plan_panel_async(const DevMat& X, const int64_t* d_ext_all, int pitch, int32_t dim, int P, SlabPlan& plan,
                           DevBuf<int32_t>& gfirst, DevBuf<int32_t>& glast, unsigned long long* stats24);
// halo != nullptr: X is this rank's column panel; on success the result is left in fuse.result (not installed)
bool slab_step(DevMat& X, SlabFusion& fuse, double threshold, bool dense_rule, const SlabHalo* halo = nullptr);
// halo exchange of a panel in slab form (psmatrix.cpp ps_slab_step_dist): request record (first row, last row, nnz, nnz),
// packed extents (first | last << 32) + prefix of the spans, runs of the local columns [ja, jb) packed back to back, and
// the layout of the columns a rank needs (extents, run addresses in its own buffer or in the receive buffer)
void slab_request_async(const DevMat& X, int64_t* d_out4, const long long* d_nnz = nullptr);   // (d_nnz: the entry count is still on the device)
void slab_extents_async(const DevMat& X, int64_t* d_ext, int64_t* d_pre);
void slab_pack_runs_async(const DevMat& X, const int64_t* d_pre, int32_t ja, int32_t jb, double* dst);
void slab_halo_layout_async(const int64_t* d_ext_all, const int64_t* d_pre_all, int pitch, int32_t dim, int P, int me,
                            int32_t ka, int32_t kb, const int32_t* d_ra, const int64_t* d_zoff, const double* d_recv,
                            const DevMat& X, int32_t* d_first, int32_t* d_last, unsigned long long* d_addr,
                            const int64_t* d_cnt_all = nullptr, int32_t* d_count = nullptr,   // (statistics: entries per column)
                            const int32_t* h_ra = nullptr, const int64_t* h_zoff = nullptr);   // (host copies of d_ra / d_zoff: passed by value up to 16 ranks, the device arrays are then not read)
// request, extents + prefix sums of the spans and (d_cnt64 != nullptr) entry counts of a panel in one pass
void slab_export_async(const DevMat& X, int64_t* d_out4, const long long* d_nnz, int64_t* d_ext, int64_t* d_pre, int64_t* d_cnt64);
void slab_counts_async(const DevMat& X, int64_t* d_cnt64);
// Slab algebra (kernels.hip, last section): the vocabulary of the solver loops on matrices that stay in slab form -- real,
// unlabelled, square, one rank, FMA arithmetic.  Every function returns false and leaves its operands alone when it
// cannot take them (the caller packs and takes the general path).
bool slab_enter(DevMat& M);   // compressed columns -> slab form in place (false: not run-like, stored zeros, complex ...)
bool slab_multiply(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule, const SlabHalo* left = nullptr);
// the arithmetic modes the block path computes: the FMA chain (matrix cores) and, on request, unfused (vector units)
inline bool block_arithmetic_ok() { return options().spgemm_fma == 1 || (options().spgemm_fma == 0 && options().block_unfused != 0); }
bool slab_panels_ok();
void slab_allow_panels(bool on);   // a slab session across ranks: the operands of the slab algebra are column panels (rows != columns)
// complex operands in slab form (FMA arithmetic, option complex_tile; a session that allows them): runs of (re, im) pairs in
// slots aligned to 16 rows -- what the complex MFMA tile kernel reads and writes.  Each returns false when it does not take
// its operands (nothing changed): the caller packs and the compressed-column path does the work.
bool sa_operand_c(const DevMat& M);
bool slab_enter_c(DevMat& M);
bool slab_multiply_c(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, bool dense_rule);
bool slab_add_diagonal_c(DevMat& B, double alpha, int32_t col_offset);                      // slab_extra.hip
bool slab_norm_axpby_c(const DevMat& A, const DevMat& B, double alpha, double beta, double* out);   // slab_extra.hip
bool slab_axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold);   // B <- alpha A + beta B
bool slab_axpby_to(const DevMat& A, const DevMat& B, DevMat& Out, double alpha, double beta, double threshold);   // Out = alpha A + beta B
bool slab_clone(const DevMat& A, DevMat& Out);
bool slab_scale(DevMat& A, double c);
bool slab_dot(const DevMat& A, const DevMat& B, double out[2]);
bool slab_norm(const DevMat& A, double* out);   // max column abs-sum
// the same on complex matrices in slab form (complex sessions)
bool slab_axpby_c(const DevMat& A, DevMat& B, double alpha, double beta, double threshold);
bool slab_axpby_to_c(const DevMat& A, const DevMat& B, DevMat& Out, double alpha, double beta, double threshold);
bool slab_clone_c(const DevMat& A, DevMat& Out);
bool slab_scale_c(DevMat& A, double c);
bool slab_norm_c(const DevMat& A, double* out);
bool slab_gershgorin(const DevMat& A, int32_t col_offset, double* mn, double* mx);
bool slab_add_diagonal(DevMat& B, double alpha, int32_t col_offset);   // B <- B + alpha I in place (slab_extra.hip); false: not done
// TRS4's polynomial chain on slab-form X and X2 (slab_extra.hip): dot(X2, 4X - 3X2), dot(X2, I - 2X + X2); then the right
// operand (4X - 3X2) + sigma (I - 2X + X2) of the iteration's second product
bool slab_trs4_traces(const DevMat& X, const DevMat& X2, int32_t col_offset, double* trace_fx, double* trace_gx);
bool slab_trs4_operand(const DevMat& X, const DevMat& X2, double sigma, int32_t col_offset, DevMat& Out);
bool slab_norm_axpby(const DevMat& A, const DevMat& B, double alpha, double beta, double* out);   // MatrixNorm(alpha A + beta B), nothing built
bool slab_trace(const DevMat& A, int32_t col_offset, double* out);   // sum of the diagonal entries held by the local columns
int64_t slab_span_sum(const DevMat& M);   // rows covered by the runs (cached in the form)
long long slab_product_count(const DevMat& A, const DevMat& B);   // statistics (slab_extra.hip): intermediate products of A B
// compressed columns -> labelled slab form (SlabForm::lab; Xs = the matrix in the bandwidth-reducing order, lab[index] = the
// caller's index); false (nothing changed): its columns are not run-like
bool slab_from_csc(DevMat& Xs, DevBuf<int32_t>& lab);
// Operands that arrive relabelled (a band hidden by a symmetric permutation): relabel_enter finds a bandwidth-reducing
// order from the pattern of D (cached per operand, also when nothing was found), renames X into it and turns it into
// labelled slab form; relabelled_operand(D) = D in that order (the cache's copy) or nullptr.  The steps then follow
// the caller's labels (kernels.hip, SlabFuseArgs::lab), pack() renames back.  One rank, real operands.
bool relabel_enter(DevMat& X, const DevMat& D);
const DevMat* relabelled_operand(const DevMat& D);
void relabel_giveup(const DevMat& D);
void drop_thin_transposes();   // (spgemm_thin.hip: the transposes kept per thin left operand)
void drop_operand_caches();   // frees what the fused / relabelled TRS2 paths keep between solves
// band_scope.cpp: while a solve runs on operands redistributed in a recovered band order across ranks, the caller's label of every
// position (device, one per row of the matrices; nullptr outside such a solve).  The fused TRS2 panel steps then decide the
// "beyond the other column's last entry" cases of their merges on these labels, as the one-rank steps on a relabelled operand
// do (SlabForm::lab): the several-rank solve keeps the entries the one-rank solve keeps.
void set_scope_labels(const int32_t* lab);
const int32_t* scope_labels();
void drop_pending_exchange(); // psmatrix.cpp: the exchange layout a panel step prepared for a successor that never came
// relabel.hip: a bandwidth-reducing order of a symmetric pattern (Cuthill-McKee, breadth-first levels on the device):
// newpos[old index] = new index, *bandwidth = max |new row - new column|.  false: not a square packed matrix, or
// more components than the search is willing to chain
bool find_band_order(const DevMat& A, DevBuf<int32_t>& newpos, int64_t* bandwidth);
// sum over the columns of (last row - first row + 1) of a matrix in compressed columns (how run-like it is as it stands)
int64_t column_span_sum(const DevMat& A);
// order-independent fingerprint of a sparsity pattern (compressed columns)
unsigned long long pattern_fingerprint_of(const DevMat& A);
// since start: [0] steps computed with SlabFusion mode 1, [1] mode 2, [2] fused steps repeated on the unfused path
long long* fusion_counts();
// [0] multiplies done in the two-block geometry (spgemm_tile2.hip), [1] launches of it that did not fit and were repeated on k_spgemm_tile
long long* tile2_counts();
long long& band_searches();   // searches for a bandwidth-reducing order since start (one per sparsity PATTERN: relabel_enter)

// B <- alpha*A + B (AddSparseVectors semantics), same shape and scalar type.
void increment(const DevMat& A, DevMat& B, double alpha, double threshold);
// the same with the rule applied per block of `row_block` rows (the reference adds distributed matrices block by
// block: whether "the other column is exhausted" is decided inside each row block)
void increment_blocked(const DevMat& A, DevMat& B, double alpha, double threshold, int32_t row_block);
// B <- alpha*A + beta*B (B scaled first, then the same rules); if D and dot_out are given, also
// dot_out = sum conj(B_new) .* D, evaluated in the same pass
void axpby(const DevMat& A, DevMat& B, double alpha, double beta, double threshold, const DevMat* D, double* dot_out,
           double* trace_out = nullptr, int32_t trace_col_offset = 0);  // trace_out: also trace(B_new), same pass
// the same with the first operand taken from a loose product (its exact nnz is learned on the way and returned)
void axpby(const LooseProduct& A, DevMat& B, double alpha, double beta, double threshold, const DevMat* D, double* dot_out,
           double* trace_out, int32_t trace_col_offset, int64_t* a_nnz_out, bool keep_loose = false);
// keep_loose: B (packed or loose on entry) is left LOOSE -- its columns stay in the slots the merge kernels wrote them
// to and the compaction pass does not run.  Loose matrices (DevMat::loose()) are accepted by spgemm (both operands the
// same matrix: the register-slab path reads the slots directly; otherwise packed copies are made), by the second
// operand of axpby(LooseProduct, ...), by the first operand of dot_trace / trace and by clone(); everything else
// needs pack() first and says so loudly.
DevMat packed_copy(const DevMat& M);
void pack(DevMat& M);
// X <- X * X with the result left loose when the register-slab kernel computes it, out = dot(X_new, D), *trace_out =
// trace(X_new): the sigma < 0 step of TRS2 without a compaction pass.  false: operand types this path does not serve
// (nothing done).
bool square_keep_loose(DevMat& X, double threshold, bool dense_rule, const DevMat& D, double out[2], double* trace_out,
                       int32_t col_offset);
// C = A .* B on the intersection of the patterns (conj_a: conjugate A first)
void pairwise(const DevMat& A, const DevMat& B, DevMat& C, bool conj_a);
// out = sum conj(A) .* B  (out[1] = imaginary part, 0 for real)
void dot(const DevMat& A, const DevMat& B, double out[2]);
// the same, plus trace(A) from the same pass (diagonal = row col_offset + j of local column j)
void dot_trace(const DevMat& A, const DevMat& B, double out[2], double* trace_out, int32_t col_offset);
void grand_sum(const DevMat& A, double out[2]);
// sum of the real parts of entries with row == col + col_offset
double trace(const DevMat& A, int32_t col_offset);
// per-column sum |v| -> out[cols]
void column_abs_sums(const DevMat& A, DevBuf<double>& out);
double max_of(const DevBuf<double>& v, size_t n);
// per-column Gershgorin discs -> min over columns of (d - r), max of (d + r)
void gershgorin(const DevMat& A, int32_t col_offset, double* mn, double* mx);
void scale(DevMat& A, double c);
void conjugate(DevMat& A);
// values of column j *= factor[j] (cols scalars of the matrix' scalar type in device memory)
void scale_columns(DevMat& A, const double* d_factor);
DevMat to_complex(const DevMat& A);
DevMat to_real(const DevMat& A);
// identity restricted to local columns [col_offset, col_offset+cols) of an n x n matrix
DevMat identity(int32_t n, int32_t col_offset, int32_t cols, bool cplx);
// is it exactly the (local part of the) identity?  returns number of diagonal ones or -1
int64_t identity_check(const DevMat& A, int32_t col_offset);
DevMat transpose(const DevMat& A);
// general re-indexing: entry (i, j) -> (row_map[i], col_map[j]) (0-based device maps, may be null
// = identity); entries whose mapped column falls outside [col_lo, col_hi) are dropped, columns are
// stored relative to col_lo; the result is sorted.  drop_exact_zeros mimics a threshold-0 multiply
// with a permutation matrix (LoadBalancerModule.F90:38-47), which prunes stored zeros.
DevMat remap_general(const DevMat& A, const int32_t* d_row_map, const int32_t* d_col_map, int32_t new_rows,
                     int32_t col_lo, int32_t col_hi, bool drop_exact_zeros);
// columns [col_lo, col_hi) of A^T
DevMat transpose_slice(const DevMat& A, int32_t col_lo, int32_t col_hi);
// columns [c0, c1) of A as a new matrix
DevMat column_slice(const DevMat& A, int32_t c0, int32_t c1);
// A with only the columns of one K slice kept (the others empty): column j (global index col_offset + j) is kept when
// ((col_offset + j) / block) % slices == slice -- the share of the inner dimension one process slice of the reference
// multiplies (MatrixMultiply.f90:74-80, 102-110)
DevMat mask_columns(const DevMat& A, int32_t col_offset, int32_t block, int32_t slices, int32_t slice);
// concatenate column panels (all with the same rows / scalar type)
DevMat concat_columns(const std::vector<const DevMat*>& parts);

DevMat from_triplets(const HostTriplets& t, int32_t rows, int32_t cols, int32_t col_offset);
void to_triplets(const DevMat& A, int32_t col_offset, HostTriplets& out);

// dst[i] = src[i] + shift for i < count
void copy_shift_i64(const int64_t* d_src, int64_t* d_dst, int64_t count, int64_t shift);

// dst[i] = src[i] - src[0] + add ; dst[i] = v ; first/last stored row over all columns (INT_MAX / -1 if empty)
void rebase_i64(const int64_t* d_src, int64_t* d_dst, int64_t count, int64_t add);
void fill_i64(int64_t* d_dst, int64_t count, int64_t v);
void row_range(const DevMat& A, int32_t* lo, int32_t* hi);
// device-side pieces of the halo-exchange plan (comm.cpp gather_needed): no host round trip of their own
void halo_request_async(const DevMat& B, int64_t nnz_a, int64_t* d_out4);
// interior columns of a B panel (empty, or every row in [c0, c1)): d_out5 = {first interior column, last interior
// column, number of interior columns, entry offset of the first, entry offset one past the last}
void halo_interior_async(const DevMat& B, int32_t c0, int32_t c1, int64_t* d_out5);
// the P x P count matrix and my send bounds from the gathered requests and the gathered panel offsets (one kernel)
void halo_counts_async(const int64_t* d_req, const int64_t* d_outer_all, int pitch, int32_t dim, int P, int me, int64_t* d_cnt,
                       int64_t* d_bound);
void halo_bounds_async(const DevMat& A, int32_t c0, const int32_t* d_sa, const int32_t* d_sb, int P, int64_t* d_bound,
                       int64_t* d_cnt_row);

// C = alpha A B for a thin left operand (spgemm_thin.hip); false: not taken (C untouched).  dense_rule_bits: bit 0 the
// dense branch's order of threshold and alpha, bit 1 fma accumulation (real operands)
bool spgemm_thin_left(const DevMat& A, const DevMat& B, DevMat& C, double alpha, double threshold, int dense_rule_bits, int64_t* products,
                      hipEvent_t ev_begin, hipEvent_t ev_end);
// thin operands inside a slab session (spgemm_thin.hip; real, FMA arithmetic): the plan's block windows and output slots, B
// (and, thin right operand, A) as the runs of the slab form, a thin left operand as the compressed columns of its transpose
struct ThinSlabArgs {
  const int32_t *blk_lo = nullptr, *blk_w = nullptr;
  const int64_t* blk_toff = nullptr;
  const int32_t *bfirst = nullptr, *blast = nullptr;
  const int64_t* boff = nullptr;
  const double* bval = nullptr;
  const int32_t *afirst = nullptr, *alast = nullptr;
  const int64_t* aoff = nullptr;
  const double* aval = nullptr;
  const int64_t* at_outer = nullptr;
  const int32_t* at_inner = nullptr;
  const double* at_val = nullptr;
  double* out_val = nullptr;
  int32_t *count = nullptr, *ofirst = nullptr, *olast = nullptr;
  int64_t* ooff = nullptr;
  double alpha = 1.0, threshold = 0.0;
  int dense_rule = 0, ncols = 0, nrows = 0;
  int* flag = nullptr;   // thin right operand: raised when a column lists more non-zeros than the kernel holds
};
void launch_thin_slab(const ThinSlabArgs& a, bool left);
// counts the operations that change the values of a matrix in place (scale, conjugate, ...): together with the serial
// number of the value buffer's allocation it tells whether a cached derivative of a matrix is still that matrix
unsigned long long matrix_value_epoch();
void bump_matrix_value_epoch();
// compressed columns without the merge pass (column_fused.hip): B <- B + alpha I in place when every local column stores its
// diagonal entry (global columns c0 ...); max column abs-sum of alpha A + B without forming it.  false: not taken
bool add_identity_inplace(DevMat& B, double alpha, int32_t c0);
bool norm_axpy_columns(const DevMat& A, const DevMat& B, double alpha, double* norm);
// exclusive scan helper (device), out[n] = total; returns total (synchronises)
// dense side (dense.hip): entry filter, sparse <-> dense (column major), Hermitian eigendecomposition (parallel
// two-sided Jacobi), (pivoted) Cholesky on a dense copy
DevMat filter(const DevMat& A, double threshold);  // entries with |v| > threshold
void to_dense(const DevMat& A, double* d_dense, int64_t ld);
DevMat from_dense(const double* d_dense, int64_t ld, int32_t rows, int32_t c0, int32_t cols, bool cplx, double threshold);
void dense_zero_columns(double* d_dense, int64_t ld, int32_t rows, int32_t c_first, int32_t c_end, bool cplx);
void dense_eigh(double* d_A, int32_t n, bool cplx, double* d_W);
void dense_cholesky(const double* d_A, double* d_L, int32_t n, double threshold, int32_t rank);
int64_t exclusive_scan_i64(const int64_t* d_in, int64_t* d_out, int64_t n);
// the same without the read-back: out[0..n] = exclusive prefix sums, out[n] = total; asynchronous on the engine stream
void scan_i32_async(const int32_t* d_in, int64_t* d_out, int64_t n);
void scan_i64_async(const int64_t* d_in, int64_t* d_out, int64_t n);

}  // namespace ntp
