// RCCL plumbing: one communicator over all ranks (one process per MI355X, xGMI links).
// Replaces the reference's MPI communicators (ProcessGridModule.F90:186-262) for the calls on
// the hot path.  The unique id is exchanged by the launcher
// (bench.py / tests use torch.distributed for that) and handed in through comm_init().
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "engine.hpp"

namespace ntp {

#define NCCL_CHECK(expr)                                                                     \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess)                                                                   \
      ::ntp::fatal(__FILE__, __LINE__, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

Comm& world() {
  static Comm* c = new Comm();
  return *c;
}

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");

void comm_get_unique_id(char out[128]) {
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
}

void comm_init(const char idbytes[128], int rank, int nranks) {
  Comm& c = world();
  if (c.nccl) comm_finalize();
  c.rank = rank;
  c.nranks = nranks;
  const char* f = std::getenv("NTPOLY_AMD_FORCE_RCCL");
  c.force = f && f[0] == '1';
  if (nranks <= 1 && !c.force) {
    c.rank = 0;
    c.nranks = 1;
    return;
  }
  if (nranks <= 1) {
    c.rank = 0;
    c.nranks = 1;
  }
  ensure_init();
  ncclUniqueId id;
  std::memcpy(&id, idbytes, sizeof(id));
  if (c.force && nranks <= 1) NCCL_CHECK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  NCCL_CHECK(ncclCommInitRank(&comm, c.nranks, id, c.rank));
  c.nccl = comm;
}

void comm_finalize() {
  Comm& c = world();
  if (c.nccl) {
    sync_stream();
    (void)ncclCommDestroy(static_cast<ncclComm_t>(c.nccl));
    c.nccl = nullptr;
  }
  c.rank = 0;
  c.nranks = 1;
}

namespace {
void allreduce_f64(double* host_vals, int n, ncclRedOp_t op) {
  Comm& c = world();
  if (!c.active() || n == 0) return;
  DevBuf<double> d((size_t)n);
  d.upload(host_vals, (size_t)n);
  NCCL_CHECK(ncclAllReduce(d.p, d.p, (size_t)n, ncclDouble, op, static_cast<ncclComm_t>(c.nccl), stream()));
  d.download(host_vals, (size_t)n);
}
}  // namespace

void comm_allreduce_sum(double* v, int n) { allreduce_f64(v, n, ncclSum); }
void comm_allreduce_min(double* v, int n) { allreduce_f64(v, n, ncclMin); }
void comm_allreduce_max(double* v, int n) { allreduce_f64(v, n, ncclMax); }

void comm_allreduce_sum_i64(int64_t* v, int n) {
  Comm& c = world();
  if (!c.active() || n == 0) return;
  DevBuf<int64_t> d((size_t)n);
  d.upload(v, (size_t)n);
  NCCL_CHECK(ncclAllReduce(d.p, d.p, (size_t)n, ncclInt64, ncclSum, static_cast<ncclComm_t>(c.nccl), stream()));
  d.download(v, (size_t)n);
}

void comm_bcast_i32(int32_t* v, int n, int root) {
  Comm& c = world();
  if (!c.active() || n == 0) return;
  DevBuf<int32_t> d((size_t)n);
  d.upload(v, (size_t)n);
  NCCL_CHECK(ncclBroadcast(d.p, d.p, (size_t)n, ncclInt32, root, static_cast<ncclComm_t>(c.nccl), stream()));
  d.download(v, (size_t)n);
}

void comm_barrier() {
  double x = 0;
  comm_allreduce_sum(&x, 1);
}

// Panel all-gather (the reference's ReduceAndComposeMatrix{Sizes,Data,Cleanup}: M1-M3 of SURVEY 2c):
// every rank contributes its column panel; every rank receives all panels and concatenates them
// into the full matrix.  RCCL has no variable-count all-gather, so after the fixed-size size
// exchange each panel travels as a broadcast from its owner, all posted inside one group so the
// point-to-point xGMI links are driven concurrently.
DevMat gather_panels(const DevMat& loc, const std::vector<int32_t>& widths) {
  Comm& c = world();
  if (!c.active()) return loc.clone();
  ncclComm_t comm = static_cast<ncclComm_t>(c.nccl);
  const int P = c.nranks;
  if ((int)widths.size() != P || widths[(size_t)c.rank] != loc.cols) NTP_FATAL("gather_panels: inconsistent panel widths");
  // 1. sizes (M1)
  DevBuf<int64_t> d_sizes((size_t)P);
  int64_t mine = loc.nnz;
  HIP_CHECK(hipMemcpyAsync(d_sizes.p + c.rank, &mine, sizeof(int64_t), hipMemcpyHostToDevice, stream()));
  NCCL_CHECK(ncclAllGather(d_sizes.p + c.rank, d_sizes.p, 1, ncclInt64, comm, stream()));
  std::vector<int64_t> sizes((size_t)P);
  d_sizes.download(sizes.data(), (size_t)P);
  std::vector<int64_t> zoff((size_t)P + 1, 0), coff((size_t)P + 1, 0);
  for (int r = 0; r < P; ++r) {
    zoff[(size_t)r + 1] = zoff[(size_t)r] + sizes[(size_t)r];
    coff[(size_t)r + 1] = coff[(size_t)r] + widths[(size_t)r];
  }
  if (coff[(size_t)P] > 2147483647LL) NTP_FATAL("gather_panels: too many columns");
  // 2. data (M2, M3): column offsets go to a staging area (they need a per-panel shift), indices
  //    and values land directly in their final place
  DevMat full;
  full.alloc(loc.rows, (int32_t)coff[(size_t)P], loc.cplx, zoff[(size_t)P]);
  DevBuf<int64_t> stage((size_t)coff[(size_t)P] + (size_t)P);
  const size_t w = loc.wval();
  NCCL_CHECK(ncclGroupStart());
  for (int r = 0; r < P; ++r) {
    const size_t ncol = (size_t)widths[(size_t)r];
    int64_t* st = stage.p + (size_t)coff[(size_t)r] + (size_t)r;
    NCCL_CHECK(ncclBroadcast(loc.outer.p, st, ncol + 1, ncclInt64, r, comm, stream()));
    if (sizes[(size_t)r] > 0) {
      NCCL_CHECK(ncclBroadcast(loc.inner.p, full.inner.p + zoff[(size_t)r], (size_t)sizes[(size_t)r], ncclInt32, r,
                               comm, stream()));
      NCCL_CHECK(ncclBroadcast(loc.val.p, full.val.p + zoff[(size_t)r] * (int64_t)w, (size_t)sizes[(size_t)r] * w,
                               ncclDouble, r, comm, stream()));
    }
  }
  NCCL_CHECK(ncclGroupEnd());
  // 3. cleanup: every panel's offsets are shifted by the number of entries before it
  //    (ReduceAndComposeMatrixCleanup.f90:6-13); the last offset of panel r is the first of panel r+1
  for (int r = 0; r < P; ++r) {
    const int ncol = widths[(size_t)r];
    if (ncol > 0)
      copy_shift_i64(stage.p + (size_t)coff[(size_t)r] + (size_t)r, full.outer.p + coff[(size_t)r], ncol, zoff[(size_t)r]);
  }
  HIP_CHECK(hipMemcpyAsync(full.outer.p + coff[(size_t)P], &zoff[(size_t)P], sizeof(int64_t), hipMemcpyHostToDevice, stream()));
  sync_stream();
  return full;
}

// Which columns of rank s's panel does a requester with row range [kmin, kmax] need?  Pure host
// arithmetic, shared with the CPU tests through the C ABI (ntpoly_amd_halo_segment).
void halo_segment(int32_t dim, int P, int s, int32_t kmin, int32_t kmax, int32_t* a, int32_t* b) {
  int32_t c0, c1;
  panel_range(dim, P, s, &c0, &c1);
  const int32_t lo = std::max(c0, kmin), hi = std::min(c1, kmax + 1);
  if (kmax < kmin || hi <= lo) {
    *a = *b = c0;
  } else {
    *a = lo;
    *b = hi;
  }
}

// Range-restricted panel exchange ("halo"): rank q only needs the columns of A whose index appears
// as a row of its B panel, i.e. the contiguous range [kmin_q, kmax_q].  For banded operands that is
// its own panel plus a halo of one bandwidth on each side (KBs..MBs instead of the whole matrix);
// for permuted operands it degenerates to the full gather.  Protocol (all on the engine stream):
//   1. all-gather of the (kmin, kmax) pairs                                   [ncclAllGather, 2 ints]
//   2. every owner reads the column offsets at its segment boundaries -> entry counts per requester
//   3. all-gather of the P x P count matrix                                   [ncclAllGather, P int64]
//   4. one group of ncclSend / ncclRecv per (owner, requester) pair with a non-empty segment:
//      column offsets of the segment, row ids, values; the own segment is a device copy
//   5. segments are re-based into one dim-wide matrix whose other columns are empty, so the SpGEMM
//      kernels run unchanged.
DevMat gather_needed(const PSMatrix& m, int32_t kmin, int32_t kmax) {
  Comm& c = world();
  const int32_t dim = m.dim;
  if (!c.active()) NTP_FATAL("gather_needed without an active communicator");
  ncclComm_t comm = static_cast<ncclComm_t>(c.nccl);
  const int P = c.nranks, me = c.rank;
  // 1. ranges
  std::vector<int32_t> req((size_t)2 * P, 0);
  {
    DevBuf<int32_t> d((size_t)2 * P);
    int32_t mine[2] = {kmin, kmax};
    HIP_CHECK(hipMemcpyAsync(d.p + 2 * me, mine, sizeof(mine), hipMemcpyHostToDevice, stream()));
    NCCL_CHECK(ncclAllGather(d.p + 2 * me, d.p, 2, ncclInt32, comm, stream()));
    d.download(req.data(), (size_t)2 * P);
  }
  // 2. what I send to every requester
  std::vector<int32_t> sa((size_t)P), sb((size_t)P);
  std::vector<int64_t> bound((size_t)2 * P, 0);
  for (int q = 0; q < P; ++q) {
    halo_segment(dim, P, me, req[(size_t)2 * q], req[(size_t)2 * q + 1], &sa[(size_t)q], &sb[(size_t)q]);
    HIP_CHECK(hipMemcpyAsync(&bound[(size_t)2 * q], m.loc.outer.p + (sa[(size_t)q] - m.c0), sizeof(int64_t),
                             hipMemcpyDeviceToHost, stream()));
    HIP_CHECK(hipMemcpyAsync(&bound[(size_t)2 * q + 1], m.loc.outer.p + (sb[(size_t)q] - m.c0), sizeof(int64_t),
                             hipMemcpyDeviceToHost, stream()));
  }
  sync_stream();
  // 3. count matrix cnt[s*P + q] = entries rank s sends to rank q
  std::vector<int64_t> cnt((size_t)P * P, 0);
  {
    DevBuf<int64_t> d((size_t)P * P);
    std::vector<int64_t> row((size_t)P);
    for (int q = 0; q < P; ++q) row[(size_t)q] = bound[(size_t)2 * q + 1] - bound[(size_t)2 * q];
    HIP_CHECK(hipMemcpyAsync(d.p + (size_t)me * P, row.data(), sizeof(int64_t) * (size_t)P, hipMemcpyHostToDevice, stream()));
    NCCL_CHECK(ncclAllGather(d.p + (size_t)me * P, d.p, (size_t)P, ncclInt64, comm, stream()));
    d.download(cnt.data(), (size_t)P * P);
  }
  // 4. receive layout: sources in rank order (their segments tile [kmin, kmax] in ascending columns)
  std::vector<int32_t> ra((size_t)P), rb((size_t)P);
  std::vector<int64_t> zoff((size_t)P + 1, 0), soff((size_t)P + 1, 0);
  for (int s = 0; s < P; ++s) {
    halo_segment(dim, P, s, kmin, kmax, &ra[(size_t)s], &rb[(size_t)s]);
    zoff[(size_t)s + 1] = zoff[(size_t)s] + cnt[(size_t)s * P + me];
    soff[(size_t)s + 1] = soff[(size_t)s] + (rb[(size_t)s] - ra[(size_t)s] + 1);
  }
  const int64_t total = zoff[(size_t)P];
  DevMat full;
  full.alloc(dim, dim, m.cplx, total);
  DevBuf<int64_t> stage((size_t)soff[(size_t)P]);
  const size_t w = m.loc.wval();
  NCCL_CHECK(ncclGroupStart());
  for (int q = 0; q < P; ++q) {  // sends
    const int64_t n = cnt[(size_t)me * P + q];
    if (q == me || n == 0) continue;
    const int64_t first = bound[(size_t)2 * q];
    NCCL_CHECK(ncclSend(m.loc.outer.p + (sa[(size_t)q] - m.c0), (size_t)(sb[(size_t)q] - sa[(size_t)q] + 1), ncclInt64, q, comm, stream()));
    NCCL_CHECK(ncclSend(m.loc.inner.p + first, (size_t)n, ncclInt32, q, comm, stream()));
    NCCL_CHECK(ncclSend(m.loc.val.p + first * (int64_t)w, (size_t)n * w, ncclDouble, q, comm, stream()));
  }
  for (int s = 0; s < P; ++s) {  // receives
    const int64_t n = cnt[(size_t)s * P + me];
    if (s == me || n == 0) continue;
    NCCL_CHECK(ncclRecv(stage.p + soff[(size_t)s], (size_t)(rb[(size_t)s] - ra[(size_t)s] + 1), ncclInt64, s, comm, stream()));
    NCCL_CHECK(ncclRecv(full.inner.p + zoff[(size_t)s], (size_t)n, ncclInt32, s, comm, stream()));
    NCCL_CHECK(ncclRecv(full.val.p + zoff[(size_t)s] * (int64_t)w, (size_t)n * w, ncclDouble, s, comm, stream()));
  }
  NCCL_CHECK(ncclGroupEnd());
  {  // own segment
    const int64_t n = cnt[(size_t)me * P + me];
    if (n > 0) {
      const int64_t first = bound[(size_t)2 * me];
      HIP_CHECK(hipMemcpyAsync(stage.p + soff[(size_t)me], m.loc.outer.p + (sa[(size_t)me] - m.c0),
                               sizeof(int64_t) * (size_t)(sb[(size_t)me] - sa[(size_t)me] + 1), hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(full.inner.p + zoff[(size_t)me], m.loc.inner.p + first, sizeof(int32_t) * (size_t)n,
                               hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(full.val.p + zoff[(size_t)me] * (int64_t)w, m.loc.val.p + first * (int64_t)w,
                               sizeof(double) * (size_t)n * w, hipMemcpyDeviceToDevice, stream()));
    }
  }
  // 5. column offsets: 0 before the first needed column, re-based segments, `total` after the last
  int32_t pos = 0;
  for (int s = 0; s < P; ++s) {
    const int32_t a = ra[(size_t)s], b = rb[(size_t)s];
    const int64_t n = cnt[(size_t)s * P + me];
    if (b <= a) continue;
    if (a > pos) fill_i64(full.outer.p + pos, a - pos, zoff[(size_t)s]);
    if (n > 0) rebase_i64(stage.p + soff[(size_t)s], full.outer.p + a, b - a, zoff[(size_t)s]);
    else fill_i64(full.outer.p + a, b - a, zoff[(size_t)s]);
    pos = b;
  }
  fill_i64(full.outer.p + pos, (int64_t)dim + 1 - pos, total);
  sync_stream();
  return full;
}

DevMat ps_gather_full(const PSMatrix& m) {
  if (!world().active()) return m.loc.clone();
  const int P = world().nranks;
  std::vector<int32_t> widths((size_t)P);
  for (int r = 0; r < P; ++r) {
    int32_t a, b;
    panel_range(m.dim, P, r, &a, &b);
    widths[(size_t)r] = b - a;
  }
  return gather_panels(m.loc, widths);
}

}  // namespace ntp
