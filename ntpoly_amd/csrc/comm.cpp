// RCCL plumbing: one communicator over all ranks (one process per MI355X, xGMI links).
// Replaces the reference's MPI communicators (ProcessGridModule.F90:186-262) for the calls on
// the hot path.  The unique id is exchanged by the launcher
// (ntpoly_amd.host.init_comm_from_env: a file in /tmp, no torch) and handed in through comm_init().
#include <rccl/rccl.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "engine.hpp"

namespace ntp {

#define NCCL_CHECK(expr)                                                                     \
  do {                                                                                       \
    ncclResult_t r_ = (expr);                                                                \
    if (r_ != ncclSuccess)                                                                   \
      ::ntp::fatal(__FILE__, __LINE__, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
  } while (0)

Comm& base_world() {
  static Comm* c = new Comm();
  return *c;
}
namespace {
Comm* g_current_comm = nullptr;                  // nullptr: the communicator over all processes
std::vector<Comm*>& split_comms() {              // sub-communicators made by comm_split (kept until comm_finalize)
  static std::vector<Comm*>* v = new std::vector<Comm*>();
  return *v;
}
}  // namespace
Comm& world() { return g_current_comm ? *g_current_comm : base_world(); }
void use_comm(Comm* c) { g_current_comm = (c == &base_world()) ? nullptr : c; }
ExchangeStats& exchange_stats() {
  static ExchangeStats* e = new ExchangeStats();
  return *e;
}

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");

// ------------------------------------------------------------------ transport 1: RCCL (the product)
namespace {
struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  hipStream_t p2p = nullptr;  // stream of the next send / recv group (nullptr: the engine stream)
  hipStream_t p2p_stream() const { return p2p ? p2p : stream(); }
  void set_stream(hipStream_t st) override { p2p = st; }
  ~RcclTransport() override {
    if (comm) (void)ncclCommDestroy(comm);
  }
  void allgather(const void* send, void* recv, size_t bytes) override {
    NCCL_CHECK(ncclAllGather(send, recv, bytes, ncclInt8, comm, stream()));
  }
  void allreduce(void* buf, size_t count, bool is_f64, int op) override {
    const ncclRedOp_t o = op == 0 ? ncclSum : op == 1 ? ncclMin : ncclMax;
    NCCL_CHECK(ncclAllReduce(buf, buf, count, is_f64 ? ncclDouble : ncclInt64, o, comm, stream()));
  }
  void bcast(const void* send, void* recv, size_t bytes, int root) override {
    NCCL_CHECK(ncclBroadcast(send, recv, bytes, ncclInt8, root, comm, stream()));
  }
  void group_begin() override { NCCL_CHECK(ncclGroupStart()); }
  void send(const void* p, size_t bytes, int peer) override { NCCL_CHECK(ncclSend(p, bytes, ncclInt8, peer, comm, p2p_stream())); }
  void recv(void* p, size_t bytes, int peer) override { NCCL_CHECK(ncclRecv(p, bytes, ncclInt8, peer, comm, p2p_stream())); }
  void group_end() override { NCCL_CHECK(ncclGroupEnd()); }
  Transport* split(int color, int key, int, const std::vector<int>&) override {
    auto* t = new RcclTransport();
    NCCL_CHECK(ncclCommSplit(comm, color, key, &t->comm, nullptr));
    return t;
  }
};

// ------------------------------------------------------------------ transport 2: shared memory (tests only)
// Layout of the segment: a header with a sense-reversing barrier, then P*P mailboxes of `box` bytes
// (mailbox s*P + q carries rank s -> rank q).  Every operation synchronises the engine stream, stages through the
// host and uses two barriers; it is a correctness vehicle, not a fast path.
struct ShmHeader {
  volatile int count;
  volatile int sense;
  int nranks;
  int pad;
};
struct ShmTransport : Transport {
  int rank = 0, P = 1;
  size_t box = 0;
  char* base = nullptr;
  size_t total = 0;
  int local_sense = 0;
  // a sub-group (split): ranks of the segment's owner in new-rank order, a barrier slot of its own in the header page, the
  // owner's mailboxes addressed with the owner's rank numbers (the halves of a split work on disjoint pairs)
  std::vector<int> members;     // empty: all ranks of the segment
  int seg_P = 1;                // ranks of the segment (mailbox pitch)
  int slot = 0;                 // barrier slot (0: the segment's own)
  bool owner = true;
  struct Pending { const void* s; void* r; size_t bytes; int peer; };
  std::vector<Pending> sends, recvs;
  std::vector<char> host;

  // (the test transport moves bytes through host memory and has to wait for the stream where RCCL enqueues a collective:
  // its waits are not host synchronisations of the ENGINE and stay out of host_sync_count())
  // (NTPOLY_AMD_DEBUG_SYNC: time spent waiting for the stream and for the other ranks, printed when the transport closes)
  double wait_ms = 0.0, barrier_ms = 0.0, copy_ms = 0.0;
  long long waits = 0, slow_waits = 0;
  bool dbg_sync = std::getenv("NTPOLY_AMD_DEBUG_SYNC") != nullptr;
  void transport_wait() {
    if (!dbg_sync) { HIP_CHECK(hipStreamSynchronize(stream())); return; }
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipStreamSynchronize(stream()));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    wait_ms += ms; waits += 1; slow_waits += ms > 10.0 ? 1 : 0;
  }
  ShmHeader* hdr() { return reinterpret_cast<ShmHeader*>(base + 64 * slot); }
  int seg_rank(int r) const { return members.empty() ? r : members[(size_t)r]; }
  char* mailbox(int s, int q) { return base + 4096 + ((size_t)seg_rank(s) * seg_P + seg_rank(q)) * box; }
  Transport* split(int color, int, int new_rank, const std::vector<int>& mem) override {
    // barrier slots of the header page, numbered like the nodes of a binary tree: the two halves of the group on slot s get
    // the slots 2 s + 1 and 2 s + 2 (two colours: what SplitProcessGrid makes), so no two groups alive at the same time share
    // one; a later split of the same group reuses its children's slots and picks up the sense they were left in
    auto* t = new ShmTransport();
    t->rank = new_rank;
    t->P = (int)mem.size();
    t->box = box;
    t->base = base;
    t->total = total;
    t->seg_P = seg_P;
    t->owner = false;
    for (int m : mem) t->members.push_back(seg_rank(m));
    t->slot = 2 * slot + 1 + (color & 1);
    if (t->slot >= 64) NTP_FATAL("shm transport: communicators split more than five levels deep");
    t->local_sense = t->hdr()->sense;
    return t;
  }
  void barrier() {
    const auto tb0 = std::chrono::steady_clock::now();
    barrier_impl();
    if (dbg_sync) barrier_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count();
  }
  void barrier_impl() {
    local_sense = 1 - local_sense;
    if (__sync_add_and_fetch(&hdr()->count, 1) == P) {
      hdr()->count = 0;
      __sync_synchronize();
      hdr()->sense = local_sense;
    } else {
      while (hdr()->sense != local_sense) sched_yield();
    }
    __sync_synchronize();
  }
  void d2h(void* h, const void* d, size_t n) {
    if (!n) return;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    if (dbg_sync) copy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  void h2d(void* d, const void* h, size_t n) {
    if (!n) return;
    const auto t0 = std::chrono::steady_clock::now();
    HIP_CHECK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    if (dbg_sync) copy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  void check(size_t bytes) { if (bytes > box) NTP_FATAL("shm transport: message larger than the mailbox (NTPOLY_AMD_SHM_MB)"); }

  void allgather(const void* send, void* recv, size_t bytes) override {
    check(bytes);
    transport_wait();
    d2h(mailbox(rank, rank), send, bytes);
    barrier();
    for (int s = 0; s < P; ++s) h2d(static_cast<char*>(recv) + (size_t)s * bytes, mailbox(s, s), bytes);
    barrier();
  }
  void allreduce(void* buf, size_t count, bool is_f64, int op) override {
    const size_t bytes = count * 8;
    check(bytes);
    transport_wait();
    d2h(mailbox(rank, rank), buf, bytes);
    barrier();
    host.resize(bytes);
    std::memcpy(host.data(), mailbox(0, 0), bytes);
    for (int s = 1; s < P; ++s) {  // rank order: the same result on every rank
      for (size_t i = 0; i < count; ++i) {
        if (is_f64) {
          double& a = reinterpret_cast<double*>(host.data())[i];
          const double b = reinterpret_cast<const double*>(mailbox(s, s))[i];
          a = op == 0 ? a + b : op == 1 ? std::min(a, b) : std::max(a, b);
        } else {
          int64_t& a = reinterpret_cast<int64_t*>(host.data())[i];
          const int64_t b = reinterpret_cast<const int64_t*>(mailbox(s, s))[i];
          a = op == 0 ? a + b : op == 1 ? std::min(a, b) : std::max(a, b);
        }
      }
    }
    h2d(buf, host.data(), bytes);
    barrier();
  }
  void bcast(const void* send, void* recv, size_t bytes, int root) override {
    check(bytes);
    transport_wait();
    if (rank == root) d2h(mailbox(root, root), send, bytes);
    barrier();
    h2d(recv, mailbox(root, root), bytes);
    barrier();
  }
  void group_begin() override { sends.clear(); recvs.clear(); }
  void send(const void* p, size_t bytes, int peer) override { sends.push_back({p, nullptr, bytes, peer}); }
  void recv(void* p, size_t bytes, int peer) override { recvs.push_back({nullptr, p, bytes, peer}); }
  void group_end() override {
    transport_wait();
    std::vector<size_t> off((size_t)P, 0);
    for (const Pending& m : sends) {  // messages to one peer are appended in posting order
      if (off[(size_t)m.peer] + m.bytes > box) NTP_FATAL("shm transport: mailbox overflow (NTPOLY_AMD_SHM_MB)");
      d2h(mailbox(rank, m.peer) + off[(size_t)m.peer], m.s, m.bytes);
      off[(size_t)m.peer] += (m.bytes + 15) & ~(size_t)15;
    }
    barrier();
    std::fill(off.begin(), off.end(), 0);
    for (const Pending& m : recvs) {
      h2d(m.r, mailbox(m.peer, rank) + off[(size_t)m.peer], m.bytes);
      off[(size_t)m.peer] += (m.bytes + 15) & ~(size_t)15;
    }
    barrier();
    sends.clear();
    recvs.clear();
  }
  ~ShmTransport() override {
    if (dbg_sync)
      std::fprintf(stderr, "[shm transport] rank %d: %lld stream waits %.1f ms (%lld longer than 10 ms), barriers %.1f ms, staging copies %.1f ms\n",
                   rank, waits, wait_ms, slow_waits, barrier_ms, copy_ms);
    if (base && owner) munmap(base, total);
  }
};

Transport* open_shm(const std::string& name, int rank, int nranks) {
  auto* t = new ShmTransport();
  t->rank = rank;
  t->P = nranks;
  t->seg_P = nranks;
  const char* mb = std::getenv("NTPOLY_AMD_SHM_MB");
  t->box = (size_t)(mb ? std::atoi(mb) : 16) << 20;
  t->total = 4096 + (size_t)nranks * nranks * t->box;
  const std::string path = "/ntpoly_amd_" + name;
  int fd = shm_open(path.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0) NTP_FATAL("shm_open failed");
  if (ftruncate(fd, (off_t)t->total) != 0) NTP_FATAL("ftruncate failed");
  void* p = mmap(nullptr, t->total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) NTP_FATAL("mmap failed");
  t->base = static_cast<char*>(p);
  // the segment is created zero-filled by the launcher-unique name: count = 0, sense = 0
  return t;
}
}  // namespace

void comm_get_unique_id(char out[128]) {
  const char* mode = std::getenv("NTPOLY_AMD_COMM");
  if (mode && std::strncmp(mode, "shm:", 4) == 0) {  // nothing to exchange: the segment name is in the environment
    std::memset(out, 0, 128);
    return;
  }
  ncclUniqueId id;
  NCCL_CHECK(ncclGetUniqueId(&id));
  std::memcpy(out, &id, sizeof(id));
}

void comm_init(const char idbytes[128], int rank, int nranks) {
  Comm& c = base_world();
  if (c.tr) comm_finalize();
  c.user_init = true;
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
  const char* mode = std::getenv("NTPOLY_AMD_COMM");
  if (mode && std::strncmp(mode, "shm:", 4) == 0) {
    c.tr = open_shm(mode + 4, c.rank, c.nranks);
    return;
  }
  ncclUniqueId id;
  std::memcpy(&id, idbytes, sizeof(id));
  if (c.force && nranks <= 1) NCCL_CHECK(ncclGetUniqueId(&id));
  auto* t = new RcclTransport();
  NCCL_CHECK(ncclCommInitRank(&t->comm, c.nranks, id, c.rank));
  c.tr = t;
}

// ------------------------------------------------------------------ the caller's MPI communicator
// The reference binds the communicator handed to ConstructProcessGrid (ProcessGridModule.F90:130-197,
// Source/CPlusPlus/ProcessGrid.cc:12-48: a Fortran handle made with MPI_Comm_c2f).  A program written against the
// reference therefore runs under `mpiexec -n P` with MPI already initialised in the process and never calls this
// engine's own bootstrap.  The engine does not link MPI; when the program has loaded one, its entry points are found
// with dlsym and used ONCE, for rank / size and to broadcast the RCCL unique id -- the data plane stays RCCL.
// Two C ABIs are understood: the MPICH family (MPICH, Intel MPI, MVAPICH, Cray: handles are ints, Fortran and C
// handles coincide, MPI_BYTE = 0x4c00010d) and Open MPI (handles are pointers, MPI_Comm_f2c converts, MPI_BYTE is the
// address of ompi_mpi_byte).  Must run before the first HIP call of the process: the GPU is chosen from the rank.
namespace {
int env_int(const char* name, int dflt) {
  const char* v = std::getenv(name);
  return (v && *v) ? std::atoi(v) : dflt;
}
// size of the launch this process is a RANK of, according to the process managers' environment (1 when none says so).
// Only variables that identify this very process as one of several count -- its rank together with the size of its
// step: an allocation-wide SLURM_NTASKS or a stray WORLD_SIZE around a serial process does not.
int launcher_world_size() {
  const struct { const char* rank; const char* size; } pairs[] = {
      {"PMI_RANK", "PMI_SIZE"}, {"PMIX_RANK", "PMIX_SIZE"}, {"OMPI_COMM_WORLD_RANK", "OMPI_COMM_WORLD_SIZE"},
      {"SLURM_PROCID", "SLURM_STEP_NUM_TASKS"}, {"RANK", "WORLD_SIZE"}};
  for (const auto& p : pairs) {
    const char* r = std::getenv(p.rank);
    if (!r || !*r) continue;
    const int n = env_int(p.size, 0);
    if (n > 1) return n;
  }
  return 1;
}
int launcher_local_rank(int rank) {
  for (const char* name : {"LOCAL_RANK", "MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID"}) {
    const char* v = std::getenv(name);
    if (v && *v) return std::atoi(v);
  }
  return rank;  // one node: every rank is local
}
void refuse_replicas() {
  // A multi-process launch that hands the engine no communicator would run P identical single-rank solves on GPU 0 and
  // let all of them write the same files: refuse.
  const int n = launcher_world_size();
  if (n > 1 && !std::getenv("NTPOLY_AMD_ALLOW_REPLICAS"))
    NTP_FATAL("this process is one of " + std::to_string(n) + " launched together, but no communicator was given to the engine: "
              "initialise MPI before constructing the process grid (the communicator argument is then honoured), or call "
              "ntpoly_amd_init_comm (ntpoly_amd.host.init_comm_from_env); NTPOLY_AMD_ALLOW_REPLICAS=1 runs independent replicas");
}

// The MPI library the program has loaded, found once.  abi: 0 none / not initialised, 1 MPICH family, 2 Open MPI.
struct MpiLib {
  int abi = 0;
  // MPICH family: handles are ints
  int (*rank_i)(int, int*) = nullptr;
  int (*size_i)(int, int*) = nullptr;
  int (*bcast_i)(void*, int, int, int, int) = nullptr;
  // Open MPI: handles are pointers
  void* (*f2c)(int) = nullptr;
  int (*rank_p)(void*, int*) = nullptr;
  int (*size_p)(void*, int*) = nullptr;
  int (*bcast_p)(void*, int, void*, int, void*) = nullptr;
  void* byte_p = nullptr;
  void* null_p = nullptr;
};
const MpiLib& mpi_lib() {
  static MpiLib* lib = [] {
    auto* m = new MpiLib();
    using fn_initialized = int (*)(int*);
    auto initialized = reinterpret_cast<fn_initialized>(dlsym(RTLD_DEFAULT, "MPI_Initialized"));
    int inited = 0;
    if (initialized) initialized(&inited);
    if (!inited) return m;
    // which ABI?  Asked, not assumed: the library names itself
    using fn_version = int (*)(char*, int*);
    auto version = reinterpret_cast<fn_version>(dlsym(RTLD_DEFAULT, "MPI_Get_library_version"));
    std::string name;
    if (version) {
      std::vector<char> buf(8192 + 1, 0);   // (MPI_MAX_LIBRARY_VERSION_STRING is 8192 in MPICH, 256 in Open MPI)
      int len = 0;
      if (version(buf.data(), &len) == 0) name.assign(buf.data());
    }
    const bool is_ompi = name.find("Open MPI") != std::string::npos && dlsym(RTLD_DEFAULT, "ompi_mpi_comm_world") != nullptr;
    const bool is_mpich = !is_ompi && (name.find("MPICH") != std::string::npos || name.find("Intel(R) MPI") != std::string::npos ||
                                       name.find("MVAPICH") != std::string::npos || name.find("CRAY MPICH") != std::string::npos);
    if (is_ompi) {
      m->f2c = reinterpret_cast<void* (*)(int)>(dlsym(RTLD_DEFAULT, "MPI_Comm_f2c"));
      m->rank_p = reinterpret_cast<int (*)(void*, int*)>(dlsym(RTLD_DEFAULT, "MPI_Comm_rank"));
      m->size_p = reinterpret_cast<int (*)(void*, int*)>(dlsym(RTLD_DEFAULT, "MPI_Comm_size"));
      m->bcast_p = reinterpret_cast<int (*)(void*, int, void*, int, void*)>(dlsym(RTLD_DEFAULT, "MPI_Bcast"));
      m->byte_p = dlsym(RTLD_DEFAULT, "ompi_mpi_byte");
      m->null_p = dlsym(RTLD_DEFAULT, "ompi_mpi_comm_null");
      if (!m->f2c || !m->rank_p || !m->size_p || !m->bcast_p || !m->byte_p) NTP_FATAL("Open MPI is loaded but its entry points were not found");
      m->abi = 2;
    } else if (is_mpich) {
      m->rank_i = reinterpret_cast<int (*)(int, int*)>(dlsym(RTLD_DEFAULT, "MPI_Comm_rank"));
      m->size_i = reinterpret_cast<int (*)(int, int*)>(dlsym(RTLD_DEFAULT, "MPI_Comm_size"));
      m->bcast_i = reinterpret_cast<int (*)(void*, int, int, int, int)>(dlsym(RTLD_DEFAULT, "MPI_Bcast"));
      if (!m->rank_i || !m->size_i || !m->bcast_i) NTP_FATAL("an MPICH-family MPI library is loaded but its entry points were not found");
      m->abi = 1;
    } else {
      NTP_FATAL("MPI is initialised in this process, but the library (\"" + name.substr(0, 80) +
                "\") is neither of the MPICH family nor Open MPI: its handle ABI is not known to the engine; bootstrap with "
                "ntpoly_amd_init_comm instead");
    }
    return m;
  }();
  return *lib;
}
// rank and size of the communicator behind a Fortran handle; false: not a communicator (0, a null or a malformed handle
// -- "no communicator given": callers that never touch MPI pass 0)
bool mpi_rank_size(const MpiLib& m, int fcomm, int* rank, int* size) {
  if (m.abi == 1) {
    // MPICH handle layout: bits 30-31 kind (1 builtin, 2 direct, 3 indirect), bits 26-29 object type (1 = communicator)
    const unsigned h = (unsigned)fcomm;
    if ((h >> 30) == 0u || ((h >> 26) & 0xfu) != 1u) return false;
    return m.rank_i(fcomm, rank) == 0 && m.size_i(fcomm, size) == 0;
  }
  if (m.abi == 2) {
    if (fcomm < 0) return false;
    void* comm = m.f2c(fcomm);
    if (!comm || comm == m.null_p) return false;
    return m.rank_p(comm, rank) == 0 && m.size_p(comm, size) == 0;
  }
  return false;
}
}  // namespace

bool comm_bind_mpi(int fcomm) {
  Comm& c = base_world();
  const bool bound = c.tr || c.nranks > 1;       // the engine already has its communicator
  if (!bound && c.user_init) return false;       // an explicit single-rank bootstrap
  const MpiLib& m = mpi_lib();
  int rank = 0, size = 1;
  if (m.abi == 0 || !mpi_rank_size(m, fcomm, &rank, &size)) {
    // no MPI in this process, or no communicator in the argument (0 / null / malformed: a caller that does not use MPI)
    if (bound) return true;
    static bool checked = false;
    if (!checked) {
      checked = true;
      refuse_replicas();
    }
    return false;
  }
  if (bound) {
    // one communicator per process: a later grid on a communicator of another size cannot be served by the one the
    // engine is bound to (a sub-communicator of the same size, e.g. a duplicate, is the same set of ranks)
    if (size != c.nranks)
      NTP_FATAL("the engine is bound to a communicator of " + std::to_string(c.nranks) + " ranks; a process grid on a communicator of " +
                std::to_string(size) + " ranks is not supported (one communicator per process)");
    return true;
  }
  if (size <= 1) return false;
  char id[128];
  std::memset(id, 0, sizeof(id));
  if (rank == 0) comm_get_unique_id(id);
  if (m.abi == 1) {
    constexpr int kMpichByte = 0x4c00010d;  // MPI_BYTE
    if (m.bcast_i(id, 128, kMpichByte, 0, fcomm) != 0) NTP_FATAL("MPI_Bcast of the RCCL id failed");
  } else {
    if (m.bcast_p(id, 128, m.byte_p, 0, m.f2c(fcomm)) != 0) NTP_FATAL("MPI_Bcast of the RCCL id failed");
  }
  // one process per GPU: the device follows the node-local rank, and must be chosen before the runtime is touched
  Context& x = ctx();
  if (x.initialised)
    NTP_FATAL("the process grid must be constructed before any other engine call that touches the GPU when ranks come from MPI");
  if (!std::getenv("LOCAL_RANK")) setenv("LOCAL_RANK", std::to_string(launcher_local_rank(rank)).c_str(), 1);
  comm_init(id, rank, size);
  return true;
}

void comm_finalize() {
  use_comm(nullptr);
  for (Comm* sc : split_comms()) {   // (sub-communicators first: they lean on the transport of all processes)
    if (sc->tr) {
      sync_stream();
      delete sc->tr;
    }
    delete sc;
  }
  split_comms().clear();
  Comm& c = base_world();
  if (c.tr) {
    sync_stream();
    delete c.tr;
    c.tr = nullptr;
  }
  c.rank = 0;
  c.nranks = 1;
}

namespace {
void allreduce_host(void* host_vals, int n, bool is_f64, int op) {
  Comm& c = world();
  if (!c.active() || n == 0) return;
  DevBuf<double> d((size_t)n);
  HIP_CHECK(hipMemcpyAsync(d.p, host_vals, (size_t)n * 8, hipMemcpyHostToDevice, stream()));
  c.tr->allreduce(d.p, (size_t)n, is_f64, op);
  d.download(static_cast<double*>(host_vals), (size_t)n);
}
}  // namespace

void comm_allreduce_sum(double* v, int n) { allreduce_host(v, n, true, 0); }
void comm_allreduce_min(double* v, int n) { allreduce_host(v, n, true, 1); }
void comm_allreduce_max(double* v, int n) { allreduce_host(v, n, true, 2); }
void comm_allreduce_sum_i64(int64_t* v, int n) { allreduce_host(v, n, false, 0); }

void comm_bcast_i32(int32_t* v, int n, int root) {
  Comm& c = world();
  if (!c.active() || n == 0) return;
  DevBuf<int32_t> d((size_t)n);
  d.upload(v, (size_t)n);
  c.tr->bcast(d.p, d.p, (size_t)n * 4, root);
  HIP_CHECK(hipMemcpyAsync(v, d.p, (size_t)n * 4, hipMemcpyDeviceToHost, stream()));
  sync_stream();
}

void comm_allgather_i64(const int64_t* mine, int n, int64_t* all) {
  Comm& c = world();
  if (!c.active()) {
    std::memcpy(all, mine, sizeof(int64_t) * (size_t)n);
    return;
  }
  DevBuf<int64_t> d((size_t)n * c.nranks);
  HIP_CHECK(hipMemcpyAsync(d.p + (size_t)n * c.rank, mine, sizeof(int64_t) * (size_t)n, hipMemcpyHostToDevice, stream()));
  c.tr->allgather(d.p + (size_t)n * c.rank, d.p, sizeof(int64_t) * (size_t)n);
  HIP_CHECK(hipMemcpyAsync(all, d.p, sizeof(int64_t) * (size_t)n * c.nranks, hipMemcpyDeviceToHost, stream()));
  sync_stream();
}

void comm_barrier() {
  double x = 0;
  comm_allreduce_sum(&x, 1);
}

Comm* comm_split(int color, int key) {
  Comm& c = world();
  auto* n = new Comm();
  n->force = c.force;
  n->user_init = c.user_init;
  split_comms().push_back(n);
  if (!c.active()) {   // one process: the communicator of that process
    n->rank = 0;
    n->nranks = 1;
    return n;
  }
  // who has my colour, in (key, rank) order -- one all-gather of (colour, key) over the communicator being split
  std::vector<int64_t> mine = {(int64_t)color, (int64_t)key}, all((size_t)2 * c.nranks);
  comm_allgather_i64(mine.data(), 2, all.data());
  std::vector<int> members;
  for (int r = 0; r < c.nranks; ++r)
    if (all[(size_t)2 * r] == (int64_t)color) members.push_back(r);
  std::stable_sort(members.begin(), members.end(), [&](int x, int y) { return all[(size_t)2 * x + 1] < all[(size_t)2 * y + 1]; });
  int new_rank = -1;
  for (size_t i = 0; i < members.size(); ++i)
    if (members[i] == c.rank) new_rank = (int)i;
  if (new_rank < 0) NTP_FATAL("comm_split: this rank is not in its own colour");
  n->rank = new_rank;
  n->nranks = (int)members.size();
  n->tr = c.tr->split(color, key, new_rank, members);   // (collective: RCCL builds both halves in one call)
  if (n->nranks <= 1 && !n->force) {   // a half of one process needs no transport
    sync_stream();
    delete n->tr;
    n->tr = nullptr;
  }
  return n;
}

// Panel all-gather (the reference's ReduceAndComposeMatrix{Sizes,Data,Cleanup}: M1-M3 of SURVEY 2c):
// every rank contributes its column panel; every rank receives all panels and concatenates them
// into the full matrix.  RCCL has no variable-count all-gather, so after the fixed-size size
// exchange each panel travels as a broadcast from its owner, all posted inside one group so the
// point-to-point xGMI links are driven concurrently.
DevMat gather_panels(const DevMat& loc, const std::vector<int32_t>& widths) {
  Comm& c = world();
  if (!c.active()) return loc.clone();
  Transport& tr = *c.tr;
  const int P = c.nranks;
  if ((int)widths.size() != P || widths[(size_t)c.rank] != loc.cols) NTP_FATAL("gather_panels: inconsistent panel widths");
  // 1. sizes (M1)
  DevBuf<int64_t> d_sizes((size_t)P);
  int64_t mine = loc.nnz;
  HIP_CHECK(hipMemcpyAsync(d_sizes.p + c.rank, &mine, sizeof(int64_t), hipMemcpyHostToDevice, stream()));
  tr.allgather(d_sizes.p + c.rank, d_sizes.p, sizeof(int64_t));
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
  tr.group_begin();  // the owners' broadcasts travel concurrently over the point-to-point links
  for (int r = 0; r < P; ++r) {
    const size_t ncol = (size_t)widths[(size_t)r];
    int64_t* st = stage.p + (size_t)coff[(size_t)r] + (size_t)r;
    tr.bcast(loc.outer.p, st, (ncol + 1) * sizeof(int64_t), r);
    if (sizes[(size_t)r] > 0) {
      tr.bcast(loc.inner.p, full.inner.p + zoff[(size_t)r], (size_t)sizes[(size_t)r] * sizeof(int32_t), r);
      tr.bcast(loc.val.p, full.val.p + zoff[(size_t)r] * (int64_t)w, (size_t)sizes[(size_t)r] * w * sizeof(double), r);
    }
  }
  tr.group_end();
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
// for permuted operands it degenerates to the full gather.  Protocol (all on the engine stream, ONE host
// synchronisation):
//   1. every rank computes (kmin, kmax, nnz(A_loc), nnz(B_loc)) on the device; these records and the column offsets
//      of every panel (8 bytes per column: 2 MB at N = 262 144) are all-gathered back to back
//   2. a kernel derives the whole P x P count matrix and the owner's send bounds from the gathered arrays; one
//      read-back brings requests, counts and bounds (and the global nnz for the dense-branch rule) to the host
//   3. one group of send / recv per (owner, requester) pair with a non-empty segment: column offsets of the
//      segment, row ids, values; the own segment is a device copy
//   4. segments are re-based into one dim-wide matrix whose other columns are empty, so the SpGEMM
//      kernels run unchanged.  No synchronisation at the end: everything downstream is stream ordered.
namespace {
// events ordering the communication stream against the engine stream (overlapped halo exchange)
hipEvent_t halo_event(int which) {
  static hipEvent_t ev[2] = {nullptr, nullptr};
  if (!ev[which]) HIP_CHECK(hipEventCreateWithFlags(&ev[which], hipEventDisableTiming));
  return ev[which];
}
}  // namespace

void gather_needed_begin(HaloExchange& hx, const PSMatrix& m, const DevMat& Bloc, int64_t nnz_global[2], bool may_overlap) {
  const long long syncs_before = host_sync_count();   // (measured, not asserted: ExchangeStats::host_syncs)
  Comm& c = world();
  const int32_t dim = m.dim;
  if (!c.active()) NTP_FATAL("gather_needed without an active communicator");
  Transport& tr = *c.tr;
  const int P = c.nranks, me = c.rank;
  hx.dim = dim;
  hx.P = P;
  const int ov_opt = options().halo_overlap;
  const bool probe = may_overlap && ov_opt > 0;
  // 1. requests (+ the interior column range of my B panel when the caller can split its multiply) and the column
  //    offsets of every panel travel in two all-gathers enqueued back to back; the count matrix and my send bounds are
  //    then computed on the device, and ONE read-back brings everything the host needs to post the send / recv group
  int32_t maxw = 0;
  for (int q = 0; q < P; ++q) {
    int32_t a0, a1;
    panel_range(dim, P, q, &a0, &a1);
    maxw = std::max(maxw, a1 - a0);
  }
  const int pitch = maxw + 1;
  std::vector<int64_t> req((size_t)4 * P + 8, 0), bound((size_t)2 * P, 0), cnt((size_t)P * P, 0);
  {
    DevBuf<int64_t> d((size_t)4 * P + 8), d_outer_all((size_t)P * pitch), d_bound((size_t)2 * P), d_cnt((size_t)P * P);
    halo_request_async(Bloc, m.loc.nnz, d.p + 4 * me);
    if (probe) halo_interior_async(Bloc, m.c0, m.c1, d.p + 4 * P);
    HIP_CHECK(hipMemcpyAsync(d_outer_all.p + (size_t)me * pitch, m.loc.outer.p, sizeof(int64_t) * (size_t)(m.c1 - m.c0 + 1),
                             hipMemcpyDeviceToDevice, stream()));
    tr.allgather(d.p + 4 * me, d.p, 4 * sizeof(int64_t));
    tr.allgather(d_outer_all.p + (size_t)me * pitch, d_outer_all.p, (size_t)pitch * sizeof(int64_t));
    halo_counts_async(d.p, d_outer_all.p, pitch, dim, P, me, d_cnt.p, d_bound.p);
    const size_t nreq = (size_t)4 * P + (probe ? 8 : 0);
    if (nreq + (size_t)P * P + 2 * P <= 500) {
      ScalarFetch f;
      f.add(d.p, (int)nreq, req.data());
      f.add(d_bound.p, 2 * P, bound.data());
      f.add(d_cnt.p, P * P, cnt.data());
      f.run();
    } else {  // many ranks: plain copies, still one synchronisation
      HIP_CHECK(hipMemcpyAsync(req.data(), d.p, nreq * 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(bound.data(), d_bound.p, (size_t)2 * P * 8, hipMemcpyDeviceToHost, stream()));
      HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt.p, (size_t)P * P * 8, hipMemcpyDeviceToHost, stream()));
      sync_stream();
    }
  }
  exchange_stats().host_syncs += host_sync_count() - syncs_before;
  exchange_stats().exchanges += 1;
  nnz_global[0] = nnz_global[1] = 0;
  for (int q = 0; q < P; ++q) {
    nnz_global[0] += req[(size_t)4 * q + 2];
    nnz_global[1] += req[(size_t)4 * q + 3];
  }
  auto kmin_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? 0 : (int32_t)lo; };
  auto kmax_of = [&](int q) { int64_t lo = req[(size_t)4 * q], hi = req[(size_t)4 * q + 1]; return hi < lo ? -1 : (int32_t)hi; };
  const int32_t kmin = kmin_of(me), kmax = kmax_of(me);
  hx.kmin = kmin;
  hx.kmax = kmax;
  // 2. segment boundaries of what I send to every requester (host arithmetic, the same as the device's)
  std::vector<int32_t> sab((size_t)2 * P);
  for (int q = 0; q < P; ++q) halo_segment(dim, P, me, kmin_of(q), kmax_of(q), &sab[(size_t)q], &sab[(size_t)P + q]);
  const int32_t* sa = sab.data();
  const int32_t* sb = sab.data() + P;
  // 3. receive layout: sources in rank order (their segments tile [kmin, kmax] in ascending columns)
  hx.ra.assign((size_t)P, 0);
  hx.rb.assign((size_t)P, 0);
  hx.zoff.assign((size_t)P + 1, 0);
  hx.soff.assign((size_t)P + 1, 0);
  hx.cnt_from.assign((size_t)P, 0);
  int64_t remote = 0;
  for (int s = 0; s < P; ++s) {
    halo_segment(dim, P, s, kmin, kmax, &hx.ra[(size_t)s], &hx.rb[(size_t)s]);
    hx.cnt_from[(size_t)s] = cnt[(size_t)s * P + me];
    hx.zoff[(size_t)s + 1] = hx.zoff[(size_t)s] + hx.cnt_from[(size_t)s];
    hx.soff[(size_t)s + 1] = hx.soff[(size_t)s] + (hx.rb[(size_t)s] - hx.ra[(size_t)s] + 1);
    if (s != me) remote += hx.cnt_from[(size_t)s];
  }
  const int64_t total = hx.zoff[(size_t)P];
  hx.full.alloc(dim, dim, m.cplx, total);
  hx.stage.alloc((size_t)hx.soff[(size_t)P]);
  // overlap decision: the panel must have a clean interior (one contiguous run of interior columns with boundary
  // columns only on its two sides) and -- unless forced -- the halo must be worth hiding.  Every rank decides for
  // itself: the stream a rank enqueues its group on is invisible to its peers.
  hx.overlapped = false;
  if (probe) {
    const int64_t jl = req[(size_t)4 * P], jlast = req[(size_t)4 * P + 1], ncount = req[(size_t)4 * P + 2];
    const bool clean = jlast >= jl && ncount == jlast - jl + 1;
    const bool worth = ov_opt >= 2 || (remote * 4 >= m.loc.nnz && ncount * 2 >= Bloc.cols);
    if (clean && worth && (remote > 0 || ov_opt >= 3)) {  // (3: tests, also with nothing to receive)
      hx.overlapped = true;
      hx.jl = (int32_t)jl;
      hx.jr = (int32_t)jlast + 1;
      hx.off_l = req[(size_t)4 * P + 3];
      hx.off_r = req[(size_t)4 * P + 4];
    }
  }
  const size_t w = m.loc.wval();
  if (hx.overlapped) {  // the group goes to the communication stream, behind everything enqueued so far
    HIP_CHECK(hipEventRecord(halo_event(0), stream()));
    HIP_CHECK(hipStreamWaitEvent(ctx().comm_stream, halo_event(0), 0));
    tr.set_stream(ctx().comm_stream);
  }
  tr.group_begin();
  for (int q = 0; q < P; ++q) {  // sends
    const int64_t n = cnt[(size_t)me * P + q];
    if (q == me || n == 0) continue;
    const int64_t first = bound[(size_t)2 * q];
    tr.send(m.loc.outer.p + (sa[q] - m.c0), (size_t)(sb[q] - sa[q] + 1) * sizeof(int64_t), q);
    tr.send(m.loc.inner.p + first, (size_t)n * sizeof(int32_t), q);
    tr.send(m.loc.val.p + first * (int64_t)w, (size_t)n * w * sizeof(double), q);
  }
  for (int s = 0; s < P; ++s) {  // receives
    const int64_t n = hx.cnt_from[(size_t)s];
    if (s == me || n == 0) continue;
    tr.recv(hx.stage.p + hx.soff[(size_t)s], (size_t)(hx.rb[(size_t)s] - hx.ra[(size_t)s] + 1) * sizeof(int64_t), s);
    tr.recv(hx.full.inner.p + hx.zoff[(size_t)s], (size_t)n * sizeof(int32_t), s);
    tr.recv(hx.full.val.p + hx.zoff[(size_t)s] * (int64_t)w, (size_t)n * w * sizeof(double), s);
  }
  tr.group_end();
  if (hx.overlapped) {
    tr.set_stream(nullptr);
    HIP_CHECK(hipEventRecord(halo_event(1), ctx().comm_stream));
  }
  {  // own segment (engine stream; disjoint from the receive targets)
    const int64_t n = hx.cnt_from[(size_t)me];
    if (n > 0) {
      const int64_t first = bound[(size_t)2 * me];
      HIP_CHECK(hipMemcpyAsync(hx.stage.p + hx.soff[(size_t)me], m.loc.outer.p + (sa[me] - m.c0),
                               sizeof(int64_t) * (size_t)(sb[me] - sa[me] + 1), hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(hx.full.inner.p + hx.zoff[(size_t)me], m.loc.inner.p + first, sizeof(int32_t) * (size_t)n,
                               hipMemcpyDeviceToDevice, stream()));
      HIP_CHECK(hipMemcpyAsync(hx.full.val.p + hx.zoff[(size_t)me] * (int64_t)w, m.loc.val.p + first * (int64_t)w,
                               sizeof(double) * (size_t)n * w, hipMemcpyDeviceToDevice, stream()));
    }
  }
}

// 4. column offsets: 0 before the first needed column, re-based segments, `total` after the last
void HaloExchange::finish() {
  if (overlapped) HIP_CHECK(hipStreamWaitEvent(stream(), halo_event(1), 0));
  const int64_t total = zoff[(size_t)P];
  int32_t pos = 0;
  for (int s = 0; s < P; ++s) {
    const int32_t a = ra[(size_t)s], b = rb[(size_t)s];
    const int64_t n = cnt_from[(size_t)s];
    if (b <= a) continue;
    if (a > pos) fill_i64(full.outer.p + pos, a - pos, zoff[(size_t)s]);
    if (n > 0) rebase_i64(stage.p + soff[(size_t)s], full.outer.p + a, b - a, zoff[(size_t)s]);
    else fill_i64(full.outer.p + a, b - a, zoff[(size_t)s]);
    pos = b;
  }
  fill_i64(full.outer.p + pos, (int64_t)dim + 1 - pos, total);
  // `stage` is released by the caller afterwards: the allocator is stream ordered, later kernels run after the
  // re-base kernels above
}

DevMat gather_needed(const PSMatrix& m, const DevMat& Bloc, int64_t nnz_global[2]) {
  HaloExchange hx;
  gather_needed_begin(hx, m, Bloc, nnz_global, false);
  hx.finish();
  return std::move(hx.full);
}

DevMat ps_gather_full(const PSMatrix& m) {
  use_grid_comm(m.grid);
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
