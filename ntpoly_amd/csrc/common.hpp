// Common host-side infrastructure of the MI355X engine: error handling, the HIP stream,
// a stream-ordered caching device allocator and the device-resident local matrix type.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <memory>
#include <vector>

namespace ntp {

// The reference has no status codes: fatal paths print and MPI_Abort
// (ErrorModule.F90:193-205).  Same contract here.
[[noreturn]] void fatal(const char* file, int line, const std::string& msg);
#define NTP_FATAL(msg) ::ntp::fatal(__FILE__, __LINE__, (msg))
#define HIP_CHECK(expr)                                                              \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess)                                                            \
      ::ntp::fatal(__FILE__, __LINE__, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// ---------------------------------------------------------------------------------
// Engine context: one process drives one GPU (LOCAL_RANK), one compute stream and one
// communication stream.
struct Context {
  int device = 0;
  hipStream_t stream = nullptr;       // all kernels
  hipStream_t comm_stream = nullptr;  // RCCL collectives (overlap with compute)
  int num_cus = 256;
  bool initialised = false;
};
Context& ctx();
void ensure_init();  // fails loudly when no GPU is present
inline hipStream_t stream() { return ctx().stream; }
void sync_stream();
long long& host_sync_count();   // host waits on the engine stream so far (counted in sync_stream itself)

// Batched read-back of small results: up to 8 device segments of 8-byte words are copied by ONE tiny kernel into
// host-mapped pinned memory and the stream is synchronised once (a hipMemcpyAsync per scalar costs a copy
// kernel and ~20 us of dispatch gap each).  fetch.add(dev_ptr, words, host_dst) ... fetch.run().
struct ScalarFetch {
  const void* src[16];
  int words[16];
  void* dst[16];
  int n = 0;
  void add(const void* dev, int words8, void* host_dst) {
    src[n] = dev; words[n] = words8; dst[n] = host_dst; ++n;
  }
  void run();  // defined in kernels.hip; synchronises the stream
};

// Caching allocator: sizes are rounded up to a small set of buckets and recycled.  All
// device work of the engine is ordered on ctx().stream, so a block freed by the host can be
// handed out again immediately (stream order protects it).
void* dev_alloc(size_t bytes);
void dev_free(void* p);
// serial number of the live allocation p was handed out by (0 if unknown): tells a block apart from an earlier one
// that lived at the same address
unsigned long long dev_alloc_serial(const void* p);
void dev_release_cache();
size_t dev_bytes_in_use();
size_t dev_bytes_cached();
void dev_malloc_stats(long long* calls, double* ms);  // hipMalloc calls behind the cache and the host time they took

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p = o.p; n = o.n; o.p = nullptr; o.n = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    n = count;
    p = static_cast<T*>(dev_alloc((count ? count : 1) * sizeof(T)));
  }
  void release() {
    if (p) dev_free(p);
    p = nullptr; n = 0;
  }
  void upload(const T* host, size_t count) {
    if (count) HIP_CHECK(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, stream()));
  }
  void download(T* host, size_t count) const {
    const size_t bytes = count * sizeof(T);
    if (bytes > 0 && bytes <= 1024 && bytes % 8 == 0) {
      ScalarFetch f;
      f.add(p, (int)(bytes / 8), host);
      f.run();
      return;
    }
    if (count) HIP_CHECK(hipMemcpyAsync(host, p, bytes, hipMemcpyDeviceToHost, stream()));
    sync_stream();
  }
  void zero() { if (n) HIP_CHECK(hipMemsetAsync(p, 0, n * sizeof(T), stream())); }
};

// ---------------------------------------------------------------------------------
// Local sparse matrix resident in HBM.  Same layout as the reference's Matrix_lsr /
// Matrix_lsc (SMatrixModule.F90:15-30): column-compressed, `outer` = cols+1 offsets (0-based,
// 64-bit here), `inner` = row ids ascending within a column (0-based here), `val` = nnz doubles
// (real) or 2*nnz doubles (complex, interleaved re/im).
// entries reserved past nnz in `inner` and `val` so that the SpGEMM kernels may over-read the tail of
// a column instead of clamping every lane (never dereferenced as data)
constexpr size_t kIndexSlack = 1024;

// A real square iterate of a purification loop kept in the form the register-slab SpGEMM kernel consumes (written by
// the kernel's fused epilogue, kernels.hpp SlabFusion; read by the next multiply X * X without any preparation pass):
//   column j: rows first[j] .. last[j] (first > last: empty) as a dense run at val[off[j] ...], zeros = no entry;
//   block b of 16 columns: the same entries as a row-major tile (k - kmin_b) x 16 at tiles[tile_off[b] ...], where
//   kmin_b .. kmax_b is the union of the blocks' column ranges.
// count[j] = entries of column j.  pack() turns it into compressed columns.
// block windows / k ranges of a step X * X on a slab-form X (k_slab_plan) and the three sizes a launch needs from them
struct SlabPlan {
  DevBuf<int32_t> blk_lo, blk_w, blk_kmin, blk_kn;
  DevBuf<int64_t> blk_toff;   // blocks + 1
  int64_t total = 0;          // output slots
  int max_w = 0, max_kn = 0;  // widest window, longest k range
  int align = 0;              // window alignment the plan was made for
};
struct SlabForm {
  DevBuf<int32_t> first, last, count;
  DevBuf<int64_t> off;        // cols + 1
  DevBuf<int64_t> tile_off;   // blocks + 1
  DevBuf<double> val, tiles;
  int64_t slots = 0;          // doubles addressable in val / tiles
  bool no_tile2 = false;      // a step on this iterate (or its predecessor) did not fit the two-block geometry of spgemm_tile2.hip: k_spgemm_tile from now on
  // the plan of the NEXT step on this iterate (kernels.hip slab_step, option plan_ahead): blocks' windows and k ranges
  // follow from the column extents alone, so the step that produced the iterate computes them right behind its
  // kernel and reads their sizes back together with its own results -- the next step launches without a read-back
  std::unique_ptr<SlabPlan> next_plan;
  // A matrix that STORES zero values (a Hamiltonian with a zero on its diagonal) cannot be told from its slab form
  // alone: slab_enter keeps the compressed columns it came from here, the slab form is then a read-only view for
  // products, dots and column norms (DevMat::zero_free = 0), and pack() hands the original back
  std::shared_ptr<struct DevMat> origin;
  mutable int64_t span_sum = -1;   // sum of the runs' lengths (last - first + 1), computed on first use (slab_extra.hip slab_span_sum)
  int row_pad = 1;            // > 1 (a multiple of 16): every column's slot holds row r at a position = r (mod row_pad) and reads as
                              // ZERO from the multiple of row_pad below its first row to the one above its last (results of the
                              // MFMA tile kernel, spgemm_tile.hpp, which reads several consecutive rows per lane)
  // Label-ordered form (optional): the indices are a bandwidth-reducing RELABELLING of the matrix the caller handed
  // in; lab[index] = the caller's index.  The arithmetic follows the caller's labels (order of the k steps, "last
  // row" of a column: plast[j] = largest label among the entries of column j, -1 if empty), so results are those of
  // the caller's matrix bit for bit; pack() renames them back.
  DevBuf<int32_t> lab, plast;
  bool labelled() const { return lab.p != nullptr; }
};

struct DevMat {
  int32_t rows = 0, cols = 0;
  bool cplx = false;
  int64_t nnz = 0;
  DevBuf<int64_t> outer;
  DevBuf<int32_t> inner;
  DevBuf<double> val;
  // "loose" storage (an iterate between two steps of a purification loop, kernels.hpp): column j holds the entries
  // outer[j] .. outer[j] + cnt[j] of inner / val, `slots` entries are addressable.  cnt.p == nullptr: packed, column j
  // ends at outer[j + 1].  Only the functions that say so accept a loose matrix; pack() converts.
  DevBuf<int32_t> cnt;
  int64_t slots = 0;
  bool loose() const { return cnt.p != nullptr; }
  // 1: no stored value is exactly zero (verified, or true by construction: every entry passed |v| > threshold or is a
  // scaled copy of one that did); 0: not known.  The fused purification steps (kernels.hpp, SlabFusion) read a zero of
  // the expanded columns as "no entry", which needs this.
  mutable int zero_free = 0;
  // set: the entries live in *slab (outer / inner / val are empty, nnz is valid); see SlabForm
  std::unique_ptr<SlabForm> slab;
  bool expanded() const { return slab != nullptr; }
  // set: the entries live in *blk as dense 16 x 16 tiles of a clustered index order (spgemm_block.hpp BlockForm; outer /
  // inner / val are empty, nnz is valid).  Left behind by the block path's products where the caller can take it: the
  // C ABI's MatrixMultiply (the next product of a caller's loop multiplies it as it is) and the TRS2 loop (its iterate).
  // Everything else sees compressed columns: pack() converts, view() refuses.
  std::shared_ptr<struct BlockForm> blk;
  bool blocked() const { return blk != nullptr; }
  // 1: a product with this matrix (in compressed columns) as an operand went through the block path -- the next one
  // goes there first, without the run statistics of the slab / tile kernels.  A property of THIS matrix, not of its
  // dimension: a banded matrix of the same size still finds its run-based kernel.
  mutable int block_hint = 0;
  // -1: slab_enter found this matrix (in compressed columns) not run-like -- asked again, it says no at once instead of
  // measuring the extents again (a caller's loop over the C ABI on a general sparse matrix asks at every call)
  mutable int slab_hint = 0;

  DevMat() = default;
  DevMat(int32_t r, int32_t c, bool z) { reset_empty(r, c, z); }
  DevMat(DevMat&&) noexcept = default;
  DevMat& operator=(DevMat&&) noexcept = default;
  void reset_empty(int32_t r, int32_t c, bool z);  // all-zero matrix of that shape
  void alloc(int32_t r, int32_t c, bool z, int64_t nz);
  DevMat clone() const;
  size_t wval() const { return cplx ? 2 : 1; }
};

DevMat packed_copy(const DevMat& M);   // kernels.hip: compressed columns from any storage form

// host-side triplets, NTPoly convention: 1-based (index_column, index_row, value)
struct HostTriplets {
  std::vector<int32_t> col, row;
  std::vector<double> val;  // nnz or 2*nnz
  bool cplx = false;
  size_t size() const { return col.size(); }
};

}  // namespace ntp
