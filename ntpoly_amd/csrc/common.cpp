#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "common.hpp"

#include <cstring>
#include <mutex>

namespace ntp {

void fatal(const char* file, int line, const std::string& msg) {
  std::fprintf(stderr, "[ntpoly_amd] FATAL %s:%d: %s\n", file, line, msg.c_str());
  std::fflush(stderr);
  std::abort();
}

// process-lifetime singletons are deliberately leaked: handles owned by the caller (e.g. Python
// objects) may be destroyed after static destructors have run
Context& ctx() {
  static Context* c = new Context();
  return *c;
}

void ensure_init() {
  Context& c = ctx();
  if (c.initialised) return;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    NTP_FATAL("no HIP device visible: the ntpoly_amd engine has no CPU fallback");
  int local_rank = 0;
  if (const char* lr = std::getenv("LOCAL_RANK")) local_rank = std::atoi(lr);
  c.device = local_rank % ndev;
  HIP_CHECK(hipSetDevice(c.device));
  HIP_CHECK(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&c.comm_stream, hipStreamNonBlocking));
  hipDeviceProp_t prop;
  HIP_CHECK(hipGetDeviceProperties(&prop, c.device));
  c.num_cus = prop.multiProcessorCount;
  c.initialised = true;
}

long long& host_sync_count() {
  static long long n = 0;
  return n;
}
namespace {
// NTPOLY_AMD_DEBUG_SYNC: where the host waits -- calls and milliseconds spent in sync_stream(), printed at exit
struct SyncClock {
  bool on = std::getenv("NTPOLY_AMD_DEBUG_SYNC") != nullptr;
  long long calls = 0, slow = 0;
  double ms = 0.0;
  ~SyncClock() {
    if (on) std::fprintf(stderr, "[sync_stream] %lld waits, %.1f ms in them, %lld longer than 10 ms\n", calls, ms, slow);
  }
};
SyncClock& sync_clock() {
  static SyncClock c;
  return c;
}
}  // namespace
void sync_stream() {
  host_sync_count() += 1;   // (every host wait on the engine stream goes through here: ScalarFetch::run, DevBuf::download)
  SyncClock& c = sync_clock();
  if (!c.on) {
    HIP_CHECK(hipStreamSynchronize(ctx().stream));
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  HIP_CHECK(hipStreamSynchronize(ctx().stream));
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  c.calls += 1;
  c.ms += ms;
  c.slow += ms > 10.0 ? 1 : 0;
}

// ---------------------------------------------------------------- caching allocator
namespace {
struct Pool {
  std::mutex mu;
  std::map<size_t, std::vector<void*>> free_lists;
  std::map<void*, size_t> live;
  std::map<const void*, unsigned long long> serial_of;
  unsigned long long next_serial = 1;
  size_t in_use = 0, cached = 0;
  long long n_malloc = 0;     // hipMalloc calls (cache misses)
  double ms_malloc = 0.0;     // host time spent in them
};
Pool& pool() {
  static Pool* p = new Pool();
  return *p;
}
// Size classes: powers of two up to 64 MiB; above that quarter-octave steps (1, 1.25, 1.5, 1.75 x 2^k) of the
// request plus 12.5 % headroom, so that operands growing from one solver iteration to the next (purification fills
// the band in over the first iterations) keep landing in blocks that already exist instead of forcing a hipMalloc
// inside the iteration.  With 288 GB of HBM the slack (< 1.45x) is cheap; the calls it saves are not.
size_t bucket(size_t bytes) {
  if (bytes < 512) return 512;
  const size_t big = (size_t)64 << 20;
  if (bytes > big) {
    const size_t want = bytes + bytes / 8;
    size_t p2 = big;
    while (p2 * 2 <= want) p2 <<= 1;
    const size_t step = p2 / 4;
    return (want + step - 1) / step * step;
  }
  size_t b = 512;
  while (b < bytes) b <<= 1;
  return b;
}
}  // namespace

void* dev_alloc(size_t bytes) {
  ensure_init();
  const size_t b = bucket(bytes);
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.free_lists.find(b);
  void* p = nullptr;
  size_t got = b;
  if (it == P.free_lists.end() || it->second.empty()) {
    // no block of this class: a cached block of a larger class (up to twice the size) serves as well -- buffers whose
    // sizes drift from one solver iteration to the next then keep circulating instead of forcing a hipMalloc
    it = P.free_lists.end();
    if (b >= ((size_t)1 << 20))
      for (auto jt = P.free_lists.upper_bound(b); jt != P.free_lists.end() && jt->first <= 2 * b; ++jt)
        if (!jt->second.empty()) { it = jt; got = jt->first; break; }
  }
  if (it != P.free_lists.end() && !it->second.empty()) {
    p = it->second.back();
    it->second.pop_back();
    P.cached -= got;
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, b);
    if (e != hipSuccess) {
      // give cached blocks back to the driver and retry once
      (void)hipGetLastError();
      for (auto& kv : P.free_lists) {
        for (void* q : kv.second) (void)hipFree(q);
        kv.second.clear();
      }
      P.cached = 0;
      HIP_CHECK(hipMalloc(&p, b));
    }
    P.n_malloc += 1;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    P.ms_malloc += ms;
    static const bool dbg = std::getenv("NTPOLY_AMD_DEBUG_ALLOC") != nullptr;
    if (dbg)
      std::fprintf(stderr, "[dev_alloc] hipMalloc %.1f MB (asked %.1f MB) took %.1f ms%s; in use %.1f GB, cached %.1f GB\n", b / 1048576.0,
                   bytes / 1048576.0, ms, e != hipSuccess ? " AFTER releasing the cache (out of memory)" : "", P.in_use / 1073741824.0,
                   P.cached / 1073741824.0);
  }
  P.live[p] = got;
  P.serial_of[p] = P.next_serial++;
  P.in_use += got;
  return p;
}

void dev_free(void* p) {
  if (!p) return;
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.live.find(p);
  if (it == P.live.end()) NTP_FATAL("dev_free of an unknown pointer");
  const size_t b = it->second;
  P.live.erase(it);
  P.serial_of.erase(p);
  P.in_use -= b;
  P.free_lists[b].push_back(p);
  P.cached += b;
}

unsigned long long dev_alloc_serial(const void* p) {
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  auto it = P.serial_of.find(p);
  return it == P.serial_of.end() ? 0ull : it->second;
}

void dev_release_cache() {
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  if (ctx().initialised) HIP_CHECK(hipStreamSynchronize(ctx().stream));
  for (auto& kv : P.free_lists) {
    for (void* q : kv.second) (void)hipFree(q);
    kv.second.clear();
  }
  P.cached = 0;
}
size_t dev_bytes_in_use() { return pool().in_use; }
size_t dev_bytes_cached() { return pool().cached; }
void dev_malloc_stats(long long* calls, double* ms) {
  *calls = pool().n_malloc;
  *ms = pool().ms_malloc;
}

// ---------------------------------------------------------------- DevMat
void DevMat::reset_empty(int32_t r, int32_t c, bool z) {
  rows = r;
  cols = c;
  cplx = z;
  nnz = 0;
  cnt.release();
  slab.reset();
  blk.reset();
  slots = 0;
  zero_free = 0;
  block_hint = 0;   // hints describe the PATTERN that was stored here, not this object
  slab_hint = 0;
  outer.alloc((size_t)c + 1);
  outer.zero();
  inner.alloc(kIndexSlack);
  val.alloc(kIndexSlack * (z ? 2 : 1));
}

void DevMat::alloc(int32_t r, int32_t c, bool z, int64_t nz) {
  rows = r;
  cols = c;
  cplx = z;
  nnz = nz;
  cnt.release();
  slab.reset();
  blk.reset();
  slots = 0;
  zero_free = 0;
  block_hint = 0;
  slab_hint = 0;
  outer.alloc((size_t)c + 1);
  inner.alloc((size_t)nz + kIndexSlack);
  val.alloc(((size_t)nz + kIndexSlack) * (z ? 2 : 1));
}

DevMat DevMat::clone() const {
  if (expanded() || blocked()) return packed_copy(*this);
  DevMat R;
  if (loose()) {  // the slots as they are
    R.rows = rows; R.cols = cols; R.cplx = cplx; R.nnz = nnz; R.slots = slots; R.zero_free = zero_free;
    R.outer.alloc((size_t)cols + 1);
    R.cnt.alloc((size_t)cols);
    R.inner.alloc((size_t)slots + kIndexSlack);
    R.val.alloc(((size_t)slots + kIndexSlack) * wval());
    HIP_CHECK(hipMemcpyAsync(R.outer.p, outer.p, sizeof(int64_t) * ((size_t)cols + 1), hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(R.cnt.p, cnt.p, sizeof(int32_t) * (size_t)cols, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(R.inner.p, inner.p, sizeof(int32_t) * (size_t)slots, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(R.val.p, val.p, sizeof(double) * (size_t)slots * wval(), hipMemcpyDeviceToDevice, stream()));
    return R;
  }
  R.alloc(rows, cols, cplx, nnz);
  R.block_hint = block_hint;
  R.slab_hint = slab_hint;
  HIP_CHECK(hipMemcpyAsync(R.outer.p, outer.p, sizeof(int64_t) * ((size_t)cols + 1), hipMemcpyDeviceToDevice, stream()));
  if (nnz) {
    HIP_CHECK(hipMemcpyAsync(R.inner.p, inner.p, sizeof(int32_t) * (size_t)nnz, hipMemcpyDeviceToDevice, stream()));
    HIP_CHECK(hipMemcpyAsync(R.val.p, val.p, sizeof(double) * (size_t)nnz * wval(), hipMemcpyDeviceToDevice, stream()));
  }
  return R;
}

}  // namespace ntp
