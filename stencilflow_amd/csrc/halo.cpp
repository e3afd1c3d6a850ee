// halo.cpp — the multi-GPU half of the C ABI: stream-ordered flags in shared host
// memory, the peer-to-peer halo transport (sf_halo_*) and the deep-halo schedule of a
// slab-decomposed run (sf_plan_execute_decomposed).
#include "sf_internal.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <thread>
#include <memory>
#include <set>

using namespace sf;

// ---------------------------------------------------------------- flag kernels
// One lane each.  The flag lives in pinned host memory that several processes
// have mapped: system-scope atomics, so that neither side's caches hold it.
static __global__ void sf_flag_set_kernel(unsigned int* flag, unsigned int value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static __global__ void sf_flag_wait_kernel(const unsigned int* flag, unsigned int value,
                                           unsigned long long timeout_ticks, unsigned int* status,
                                           unsigned int* status2 = nullptr) {
  const unsigned long long t0 = wall_clock64();  // constant-rate counter (100 MHz)
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {  // never spin forever
      if (status) __hip_atomic_store(status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      if (status2) __hip_atomic_store(status2, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
  }
}

extern "C" {

int sf_host_register(void* ptr, size_t bytes, void** device_ptr) {
  SF_API_BEGIN
  if (!ptr || bytes == 0) throw sf::Error(SF_ERR_INVALID, "sf_host_register: null range");
  SF_HIP_CHECK(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
  if (device_ptr) {
    void* dev = nullptr;
    const hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) {
      (void)hipHostUnregister(ptr);
      throw sf::Error(SF_ERR_DEVICE, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    *device_ptr = dev;
  }
  return SF_OK;
  SF_API_END
}

int sf_host_unregister(void* ptr) {
  SF_API_BEGIN
  if (!ptr) return SF_OK;
  SF_HIP_CHECK(hipHostUnregister(ptr));
  return SF_OK;
  SF_API_END
}

int sf_copy_async(void* dst, const void* src, size_t bytes, void* stream) {
  SF_API_BEGIN
  if ((!dst || !src) && bytes) throw sf::Error(SF_ERR_INVALID, "sf_copy_async: null pointer");
  if (bytes) SF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, (hipStream_t)stream));
  return SF_OK;
  SF_API_END
}

int sf_flag_set(void* stream, unsigned int* flag, unsigned int value) {
  SF_API_BEGIN
  if (!flag) throw sf::Error(SF_ERR_INVALID, "sf_flag_set: null flag");
  hipLaunchKernelGGL(sf_flag_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
  SF_API_END
}

int sf_flag_wait(void* stream, const unsigned int* flag, unsigned int value, unsigned int timeout_ms,
                 unsigned int* status) {
  SF_API_BEGIN
  if (!flag) throw sf::Error(SF_ERR_INVALID, "sf_flag_wait: null flag");
  const unsigned long long ticks = (unsigned long long)std::max(1u, timeout_ms) * 100000ull;
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, ticks, status,
                     (unsigned int*)nullptr);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- sf_halo
// Halo transport of the slab decomposition, owned by the library.  Two rungs behind
// one handle:
//
//  * peer-to-peer pushes (default): a rank PUSHES the planes next to a slab boundary
//    straight into its neighbour's ghost planes -- device memory of the neighbour's
//    plan, mapped here through a HIP IPC handle -- with DMA copies (no compute units,
//    over xGMI between the GPUs of a node), ordered by the flag words of a small page of
//    host memory that the ranks of the node share (POSIX shared memory, pinned;
//    sf_flag_set / sf_flag_wait above);
//  * RCCL (sf_halo_use_rccl): grouped ncclSend / ncclRecv of the same planes on the
//    transport's stream -- the transport BASELINE.json's north_star names (the role of
//    the SMI remote streams of stencilflow/sdfg_generator.py:848-891).  librccl is
//    loaded with dlopen, so a machine without it still loads this library.
//
// Everything is enqueued on streams of the transport; the caller's compute stream only
// waits for their events.
//
// Flag page of rank r (unsigned words), per registered buffer `key` (< 60):
//   [16 key + d]      ready[d]  : r may receive exchange n from neighbour d (0 lower, 1 upper)
//   [16 key + 2 + d]  arrived[d]: neighbour d has delivered exchange n (written by that neighbour)
//   [1023]            status    : a wait of r, or a wait of a neighbour FOR r, timed out

namespace sf {

// ---- librccl through dlopen: the handful of entry points the exchange needs
struct Rccl {
  typedef int result_t;  // ncclResult_t (0 = ncclSuccess)
  typedef struct { char internal[SF_HALO_RCCL_ID_BYTES]; } unique_id;
  typedef void* comm_t;
  result_t (*GetUniqueId)(unique_id*) = nullptr;
  result_t (*CommInitRank)(comm_t*, int, unique_id, int) = nullptr;
  result_t (*CommDestroy)(comm_t) = nullptr;
  result_t (*CommAbort)(comm_t) = nullptr;  // optional: ends a communicator whose operations cannot complete
  result_t (*GroupStart)() = nullptr;
  result_t (*GroupEnd)() = nullptr;
  result_t (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  result_t (*CommGetAsyncError)(comm_t, result_t*) = nullptr;
  const char* (*GetErrorString)(result_t) = nullptr;
  std::string path;
  static const int kChar = 0;  // ncclInt8 / ncclChar
};

static Rccl& rccl() {
  static Rccl lib;
  static bool tried = false;
  if (tried) {
    if (!lib.Send) throw Error(SF_ERR_UNSUPPORTED, "librccl is not available: " + lib.path);
    return lib;
  }
  tried = true;
  void* h = nullptr;
  std::string tried_names;
  auto attempt = [&](const char* name, int flags) {
    if (h || !name || !*name) return;
    h = ::dlopen(name, flags);
    if (h) lib.path = name;
    else tried_names += std::string(tried_names.empty() ? "" : ", ") + name;
  };
  attempt(std::getenv("SF_RCCL_LIBRARY"), RTLD_NOW | RTLD_LOCAL);
  // a copy the process has loaded already (a PyTorch wheel brings its own) before another one
  for (const char* name : {"librccl.so", "librccl.so.1"}) attempt(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) attempt(name, RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    lib.path = "dlopen failed for " + tried_names;
    throw Error(SF_ERR_UNSUPPORTED, "librccl is not available: " + lib.path);
  }
  auto sym = [&](const char* name) {
    void* s = ::dlsym(h, name);
    if (!s) {
      lib.Send = nullptr;
      lib.path = std::string("symbol ") + name + " missing in " + lib.path;
      throw Error(SF_ERR_UNSUPPORTED, "librccl is not available: " + lib.path);
    }
    return s;
  };
  lib.GetUniqueId = reinterpret_cast<decltype(lib.GetUniqueId)>(sym("ncclGetUniqueId"));
  lib.CommInitRank = reinterpret_cast<decltype(lib.CommInitRank)>(sym("ncclCommInitRank"));
  lib.CommDestroy = reinterpret_cast<decltype(lib.CommDestroy)>(sym("ncclCommDestroy"));
  lib.GroupStart = reinterpret_cast<decltype(lib.GroupStart)>(sym("ncclGroupStart"));
  lib.GroupEnd = reinterpret_cast<decltype(lib.GroupEnd)>(sym("ncclGroupEnd"));
  lib.Recv = reinterpret_cast<decltype(lib.Recv)>(sym("ncclRecv"));
  lib.GetErrorString = reinterpret_cast<decltype(lib.GetErrorString)>(sym("ncclGetErrorString"));
  lib.CommGetAsyncError = reinterpret_cast<decltype(lib.CommGetAsyncError)>(::dlsym(h, "ncclCommGetAsyncError"));
  lib.CommAbort = reinterpret_cast<decltype(lib.CommAbort)>(::dlsym(h, "ncclCommAbort"));
  lib.Send = reinterpret_cast<decltype(lib.Send)>(sym("ncclSend"));  // (last: marks the table complete)
  return lib;
}

#define SF_RCCL_CHECK(expr)                                                                     \
  do {                                                                                          \
    const ::sf::Rccl::result_t r_ = (expr);                                                     \
    if (r_ != 0)                                                                                \
      throw ::sf::Error(SF_ERR_DEVICE, std::string(#expr) + ": " + ::sf::rccl().GetErrorString(r_)); \
  } while (0)

struct HaloBlob {  // what a rank tells its neighbours about one buffer (plain bytes)
  char magic[8];
  hipIpcMemHandle_t mem;
  unsigned long long plane_bytes;
  int n_local, halo, rank, device;
  char flags_name[96];
};
static_assert(sizeof(HaloBlob) <= SF_HALO_BLOB_BYTES, "SF_HALO_BLOB_BYTES too small");

struct FlagPage {
  std::string name;
  void* host = nullptr;
  unsigned* dev = nullptr;  // address kernels dereference
  bool owner = false;
};

struct HaloBuffer {
  char* base = nullptr;
  size_t plane_bytes = 0;
  int n_local = 0, halo = 0;
  unsigned count = 0;
  // neighbours (0 lower, 1 upper): their buffer mapped here and its geometry (peer-to-peer rung)
  char* peer[2] = {nullptr, nullptr};
  int peer_n_local[2] = {0, 0}, peer_halo[2] = {0, 0};
  bool connected = false;
  hipEvent_t sent = nullptr, received = nullptr;
  bool pending = false;
};

}  // namespace sf

struct sf_halo {
  int rank = 0, world = 1, device = 0;
  unsigned timeout_ms = 20000;
  // (RCCL rung) a word in pinned host memory the transport's stream ticks twice per exchange: sf_halo_check bounds the
  // time WITHOUT PROGRESS with it
  // exchange profile (sf_halo_set_profile): timing events on the transport's streams around every exchange -- from the
  // moment the launches it waits for are done to the moment its planes have arrived
  bool profile = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  size_t prof_used = 0;
  unsigned* progress = nullptr;       // device address
  unsigned* progress_host = nullptr;  // the same word on the host
  unsigned ticks = 0;
  std::string session;
  sf::FlagPage own, nb[2];
  std::map<int, sf::HaloBuffer> bufs;
  hipStream_t send = nullptr, recv = nullptr;
  hipEvent_t now = nullptr;
  // RCCL rung
  sf::Rccl::comm_t comm = nullptr;
  int comm_rank = 0, comm_size = 0;  // comm_size == 1 with world > 1: every halo comes back to the sender (tests)
  // an exchange that did not complete in time, an asynchronous RCCL error, or a communicator that did not form on
  // every rank (sf_halo_fail): the communicator is ended with ncclCommAbort, never waited for
  bool failed = false, aborted = false;
  // schedule refinements of sf_plan_execute_decomposed
  int reserved_cus = 0, early_exchange = 0;
  bool has_neighbour(int d) const { return d == 0 ? rank > 0 : rank < world - 1; }
};

namespace sf {

// End a communicator whose operations may never complete: its kernels leave the streams, so the events behind
// them fire and streams, plans and buffers can be released.  (Without ncclCommAbort in the library the
// communicator is left alone: better a leak than a wait without end.)
static void halo_abort_comm(sf_halo& h) {
  h.failed = true;
  if (!h.comm || h.aborted) return;
  // (SF_RCCL_NO_ABORT=1, tests: the stall is staged on this rank's own stream and resolves by itself;
  // the communicator is then left alone -- neither aborted nor destroyed)
  const char* no_abort = std::getenv("SF_RCCL_NO_ABORT");
  if (no_abort && no_abort[0] == '1') return;
  h.aborted = true;
  try {
    Rccl& nc = rccl();
    if (nc.CommAbort) (void)nc.CommAbort(h.comm);
  } catch (...) {
  }
}

static const size_t kFlagPageBytes = 4096;

static void map_flag_page(FlagPage& page, const std::string& name, bool create) {
  // (a page of the same session left behind by a run that died is stale by definition: session
  // names carry a token of the run that made them)
  if (create) ::shm_unlink(name.c_str());
  const int fd = ::shm_open(name.c_str(), O_RDWR | (create ? (O_CREAT | O_EXCL) : 0), 0600);
  if (fd < 0) throw Error(SF_ERR_DEVICE, "sf_halo: shm_open(" + name + ") failed: " + std::strerror(errno));
  if (create && ::ftruncate(fd, (off_t)kFlagPageBytes) != 0) {
    ::close(fd);
    ::shm_unlink(name.c_str());
    throw Error(SF_ERR_DEVICE, "sf_halo: ftruncate failed");
  }
  void* p = ::mmap(nullptr, kFlagPageBytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  ::close(fd);
  if (p == MAP_FAILED) throw Error(SF_ERR_DEVICE, "sf_halo: mmap of the flag page failed");
  if (create) std::memset(p, 0, kFlagPageBytes);
  hipError_t e = hipHostRegister(p, kFlagPageBytes, hipHostRegisterPortable | hipHostRegisterMapped);
  void* dev = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dev, p, 0);
  if (e != hipSuccess) {
    ::munmap(p, kFlagPageBytes);
    if (create) ::shm_unlink(name.c_str());
    throw Error(SF_ERR_DEVICE, std::string("sf_halo: pinning the flag page: ") + hipGetErrorString(e));
  }
  page.name = name;
  page.host = p;
  page.dev = static_cast<unsigned*>(dev);
  page.owner = create;
}

static void unmap_flag_page(FlagPage& page) {
  if (!page.host) return;
  (void)hipHostUnregister(page.host);
  ::munmap(page.host, kFlagPageBytes);
  if (page.owner) ::shm_unlink(page.name.c_str());
  page = FlagPage();
}

static void halo_flag_set(hipStream_t s, unsigned* flag, unsigned value) {
  hipLaunchKernelGGL(sf_flag_set_kernel, dim3(1), dim3(1), 0, s, flag, value);
  SF_HIP_CHECK(hipGetLastError());
}
// (a wait FOR neighbour `nb_status`'s page that times out marks both pages: the copy that
// follows on the stream cannot be taken back, so both ends must fail their check)
static void halo_flag_wait(sf_halo& h, hipStream_t s, const unsigned* flag, unsigned value, unsigned* nb_status = nullptr) {
  const unsigned long long ticks = (unsigned long long)std::max(1u, h.timeout_ms) * 100000ull;
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, s, flag, value, ticks, h.own.dev + 1023, nb_status);
  SF_HIP_CHECK(hipGetLastError());
}

// the planes rank `h` sends towards / receives from neighbour d
static const char* halo_send_planes(const HaloBuffer& b, int d, int depth) {
  return b.base + (size_t)(d == 0 ? b.halo : b.halo + b.n_local - depth) * b.plane_bytes;
}
static char* halo_ghost_planes(const HaloBuffer& b, int d, int depth) {
  return b.base + (size_t)(d == 0 ? b.halo - depth : b.halo + b.n_local) * b.plane_bytes;
}

}  // namespace sf

extern "C" {

int sf_halo_create(int rank, int world, const char* session, int device, unsigned int timeout_ms, sf_halo** out) {
  SF_API_BEGIN
  if (!out || !session || rank < 0 || world < 1 || rank >= world) throw Error(SF_ERR_INVALID, "sf_halo_create: bad argument");
  std::unique_ptr<sf_halo> h(new sf_halo);
  h->rank = rank;
  h->world = world;
  h->device = device;
  h->timeout_ms = timeout_ms ? timeout_ms : 20000;
  h->session = session;
  SF_HIP_CHECK(hipSetDevice(device));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&h->send, hipStreamNonBlocking));
  SF_HIP_CHECK(hipStreamCreateWithFlags(&h->recv, hipStreamNonBlocking));
  SF_HIP_CHECK(hipEventCreateWithFlags(&h->now, hipEventDisableTiming));
  try {
    sf::map_flag_page(h->own, "/sf_halo_" + h->session + "_" + std::to_string(rank), true);
  } catch (...) {
    (void)hipEventDestroy(h->now);
    (void)hipStreamDestroy(h->send);
    (void)hipStreamDestroy(h->recv);
    throw;
  }
  *out = h.release();
  return SF_OK;
  SF_API_END
}

int sf_halo_destroy(sf_halo* h) {
  SF_API_BEGIN
  if (!h) return SF_OK;
  (void)hipSetDevice(h->device);
  if (h->comm && h->failed) {
    // a communicator with an exchange that cannot complete: ended, not waited for
    sf::halo_abort_comm(*h);
    const bool can_abort = h->aborted && sf::rccl().CommAbort != nullptr;
    if (can_abort) {
      if (h->send) (void)hipStreamSynchronize(h->send);
      if (h->recv) (void)hipStreamSynchronize(h->recv);
    } else {
      //  the stuck ncclSend / ncclRecv kernels are still on the streams: destroying the streams or freeing
      // anything they use would block behind them for ever (hipFree synchronises the device) or pull memory from under
      // a live kernel.  Everything the transport owns is LEAKED, and said so; the process is expected to end.
      std::fprintf(stderr, "[sf_hip] sf_halo_destroy: rank %d: the failed RCCL communicator could not be aborted; its streams, events and "
                           "mappings are leaked\n", h->rank);
      h->bufs.clear();
      delete h;
      return SF_OK;
    }
  } else {
    if (h->send) (void)hipStreamSynchronize(h->send);
    if (h->recv) (void)hipStreamSynchronize(h->recv);
    if (h->comm) {
      try {
        (void)sf::rccl().CommDestroy(h->comm);
      } catch (...) {
      }
    }
  }
  for (auto& kv : h->bufs) {
    for (int d = 0; d < 2; ++d)
      if (kv.second.peer[d]) (void)hipIpcCloseMemHandle(kv.second.peer[d]);
    if (kv.second.sent) (void)hipEventDestroy(kv.second.sent);
    if (kv.second.received) (void)hipEventDestroy(kv.second.received);
  }
  sf::unmap_flag_page(h->nb[0]);
  sf::unmap_flag_page(h->nb[1]);
  sf::unmap_flag_page(h->own);
  if (h->now) (void)hipEventDestroy(h->now);
  for (auto& pe : h->prof_events) {
    (void)hipEventDestroy(pe.first);
    (void)hipEventDestroy(pe.second);
  }
  if (h->progress_host) (void)hipHostFree(h->progress_host);
  if (h->send) (void)hipStreamDestroy(h->send);
  if (h->recv) (void)hipStreamDestroy(h->recv);
  delete h;
  return SF_OK;
  SF_API_END
}

int sf_halo_fail(sf_halo* h) {
  SF_API_BEGIN
  if (!h) throw Error(SF_ERR_INVALID, "sf_halo_fail: null transport");
  // (the caller knows the communicator did not form on every rank, or gave the transport up: whatever RCCL
  // still holds is ended with ncclCommAbort at once and sf_halo_destroy will not wait for it)
  sf::halo_abort_comm(*h);
  return SF_OK;
  SF_API_END
}

int sf_halo_abandon(sf_halo* h) {
  SF_API_BEGIN
  if (!h) throw Error(SF_ERR_INVALID, "sf_halo_abandon: null transport");
  // A handle another thread is still inside (ncclCommInitRank that never returned) can be neither used nor
  // destroyed; what can be released without touching it is the NAME of its flag page in /dev/shm.
  if (h->own.owner && !h->own.name.empty()) ::shm_unlink(h->own.name.c_str());
  return SF_OK;
  SF_API_END
}

int sf_halo_rccl_id(void* id_out) {
  SF_API_BEGIN
  if (!id_out) throw Error(SF_ERR_INVALID, "sf_halo_rccl_id: null argument");
  sf::Rccl::unique_id id;
  SF_RCCL_CHECK(sf::rccl().GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return SF_OK;
  SF_API_END
}

int sf_halo_use_rccl(sf_halo* h, const void* id, int comm_rank, int comm_size) {
  SF_API_BEGIN
  if (!h || !id || comm_size < 1 || comm_rank < 0 || comm_rank >= comm_size)
    throw Error(SF_ERR_INVALID, "sf_halo_use_rccl: bad argument");
  if (h->comm) throw Error(SF_ERR_STATE, "sf_halo_use_rccl: the transport has a communicator already");
  if (!h->bufs.empty()) throw Error(SF_ERR_STATE, "sf_halo_use_rccl: call it before sf_halo_export");
  if (!(comm_size == 1 || (comm_size == h->world && comm_rank == h->rank)))
    throw Error(SF_ERR_INVALID, "sf_halo_use_rccl: the communicator spans the transport's ranks (or, for tests, "
                                "this rank alone: every halo then comes back to the sender)");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::Rccl::unique_id uid;
  std::memcpy(&uid, id, sizeof uid);
  sf::Rccl::comm_t comm = nullptr;
  SF_RCCL_CHECK(sf::rccl().CommInitRank(&comm, comm_size, uid, comm_rank));
  h->comm = comm;
  if (!h->progress_host) {
    void* word = nullptr;
    SF_HIP_CHECK(hipHostMalloc(&word, 64, hipHostMallocMapped));
    std::memset(word, 0, 64);
    void* dev = nullptr;
    SF_HIP_CHECK(hipHostGetDevicePointer(&dev, word, 0));
    h->progress_host = static_cast<unsigned*>(word);
    h->progress = static_cast<unsigned*>(dev);
  }
  h->comm_rank = comm_rank;
  h->comm_size = comm_size;
  return SF_OK;
  SF_API_END
}

const char* sf_halo_transport(const sf_halo* h) { return !h ? nullptr : h->comm ? "rccl" : "p2p"; }

int sf_halo_configure(sf_halo* h, int reserved_cus, int early_exchange) {
  SF_API_BEGIN
  if (!h || reserved_cus < 0 || reserved_cus >= 256) throw Error(SF_ERR_INVALID, "sf_halo_configure: bad argument");
  h->reserved_cus = reserved_cus;
  h->early_exchange = early_exchange != 0;
  return SF_OK;
  SF_API_END
}

int sf_halo_export(sf_halo* h, int key, void* device_base, size_t plane_bytes, int n_local, int halo, void* blob) {
  SF_API_BEGIN
  if (!h || !device_base || !blob || key < 0 || key >= 60 || plane_bytes == 0 || n_local < 1 || halo < 1)
    throw Error(SF_ERR_INVALID, "sf_halo_export: bad argument");
  if (h->bufs.count(key)) throw Error(SF_ERR_STATE, "sf_halo_export: buffer key already registered");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::HaloBuffer b;
  b.base = static_cast<char*>(device_base);
  b.plane_bytes = plane_bytes;
  b.n_local = n_local;
  b.halo = halo;
  sf::HaloBlob out;
  std::memset(&out, 0, sizeof out);
  std::memcpy(out.magic, h->comm ? "SFHALOR" : "SFHALO1", 8);
  if (h->world > 1 && !h->comm) SF_HIP_CHECK(hipIpcGetMemHandle(&out.mem, device_base));
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.sent, hipEventDisableTiming));
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.received, hipEventDisableTiming));

  out.plane_bytes = plane_bytes;
  out.n_local = n_local;
  out.halo = halo;
  out.rank = h->rank;
  out.device = h->device;
  std::snprintf(out.flags_name, sizeof out.flags_name, "%s", h->own.name.c_str());
  std::memset(blob, 0, SF_HALO_BLOB_BYTES);
  std::memcpy(blob, &out, sizeof out);
  // (RCCL: nothing of the neighbour is mapped here; a self-loop communicator needs no description at all)
  b.connected = h->comm && h->comm_size == 1;
  h->bufs[key] = b;
  return SF_OK;
  SF_API_END
}

int sf_halo_connect(sf_halo* h, int key, const void* lower_blob, const void* upper_blob) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_connect: unknown buffer key");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::HaloBuffer& b = h->bufs[key];
  if (h->comm && h->comm_size == 1) return SF_OK;  // (self-loop: the rank is its own neighbours)
  const void* blobs[2] = {lower_blob, upper_blob};
  for (int d = 0; d < 2; ++d) {
    if (!h->has_neighbour(d)) continue;
    if (!blobs[d]) throw Error(SF_ERR_INVALID, "sf_halo_connect: missing neighbour description");
    sf::HaloBlob in;
    std::memcpy(&in, blobs[d], sizeof in);
    if (std::memcmp(in.magic, h->comm ? "SFHALOR" : "SFHALO1", 8) != 0 || in.rank != h->rank + (d == 0 ? -1 : 1))
      throw Error(SF_ERR_INVALID, "sf_halo_connect: not the description of the neighbouring rank (or of another transport)");
    if (in.plane_bytes != b.plane_bytes || in.halo != b.halo)
      throw Error(SF_ERR_INVALID, "sf_halo_connect: the neighbour's buffer has another plane size or halo");
    b.peer_n_local[d] = in.n_local;
    b.peer_halo[d] = in.halo;
    if (h->comm) continue;  // RCCL: geometry checked, nothing to map
    void* mapped = nullptr;
    SF_HIP_CHECK(hipIpcOpenMemHandle(&mapped, in.mem, hipIpcMemLazyEnablePeerAccess));
    b.peer[d] = static_cast<char*>(mapped);
    if (!h->nb[d].host) sf::map_flag_page(h->nb[d], in.flags_name, false);
  }
  b.connected = true;
  return SF_OK;
  SF_API_END
}

int sf_halo_start(sf_halo* h, int key, int depth, void* compute_stream) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_start: unknown buffer key");
  sf::HaloBuffer& b = h->bufs[key];
  if (depth < 1 || depth > b.halo || depth > b.n_local) throw Error(SF_ERR_INVALID, "sf_halo_start: bad depth");
  if (b.pending) throw Error(SF_ERR_STATE, "sf_halo_start: the previous exchange of this buffer was not finished");
  if (h->world == 1) return SF_OK;
  if (!b.connected) throw Error(SF_ERR_STATE, "sf_halo_start: sf_halo_connect has not been called for this buffer");
  SF_HIP_CHECK(hipSetDevice(h->device));
  const unsigned n = ++b.count;
  const size_t bytes = (size_t)depth * b.plane_bytes;
  // the planes to send are final and the ghost planes no longer read once
  // everything queued on the compute stream so far has completed
  SF_HIP_CHECK(hipEventRecord(h->now, (hipStream_t)compute_stream));
  SF_HIP_CHECK(hipStreamWaitEvent(h->send, h->now, 0));
  hipEvent_t prof_end = nullptr;
  if (h->profile) {
    if (h->prof_used == h->prof_events.size()) {
      hipEvent_t a = nullptr, z = nullptr;
      SF_HIP_CHECK(hipEventCreate(&a));
      SF_HIP_CHECK(hipEventCreate(&z));
      h->prof_events.push_back({a, z});
    }
    SF_HIP_CHECK(hipEventRecord(h->prof_events[h->prof_used].first, h->send));
    prof_end = h->prof_events[h->prof_used++].second;
  }
  if (h->comm) {
    // RCCL: receives first, then sends, one pair per neighbour, in ONE group on the
    // transport's stream (ncclGroupStart; ncclSend/ncclRecv x <= 4; ncclGroupEnd)
    sf::Rccl& nc = sf::rccl();
    const bool self = h->comm_size == 1;
    // (progress word: ticks when the launches the exchange waits for are done, and again when it has arrived)
    if (h->progress) sf::halo_flag_set(h->send, h->progress, ++h->ticks);
    SF_RCCL_CHECK(nc.GroupStart());
    sf::Rccl::result_t r = 0;
    for (int d = 0; d < 2 && r == 0; ++d)
      if (self || h->has_neighbour(d))
        r = nc.Recv(sf::halo_ghost_planes(b, d, depth), bytes, sf::Rccl::kChar, self ? 0 : h->comm_rank + (d == 0 ? -1 : 1),
                    h->comm, h->send);
    for (int d = 0; d < 2 && r == 0; ++d)
      if (self || h->has_neighbour(d))
        r = nc.Send(sf::halo_send_planes(b, d, depth), bytes, sf::Rccl::kChar, self ? 0 : h->comm_rank + (d == 0 ? -1 : 1),
                    h->comm, h->send);
    const sf::Rccl::result_t e = nc.GroupEnd();
    SF_RCCL_CHECK(r);
    SF_RCCL_CHECK(e);
    if (h->progress) sf::halo_flag_set(h->send, h->progress, ++h->ticks);
    if (prof_end) SF_HIP_CHECK(hipEventRecord(prof_end, h->send));
    SF_HIP_CHECK(hipEventRecord(b.sent, h->send));
    SF_HIP_CHECK(hipEventRecord(b.received, h->send));
    b.pending = true;
    return SF_OK;
  }
  SF_HIP_CHECK(hipStreamWaitEvent(h->recv, h->now, 0));
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_set(h->recv, h->own.dev + 16 * key + d, n);
  for (int d = 0; d < 2; ++d) {
    if (!b.peer[d]) continue;
    const int their = 1 - d;  // which of the neighbour's sides we are on
    sf::halo_flag_wait(*h, h->send, h->nb[d].dev + 16 * key + their, n, h->nb[d].dev + 1023);
    char* dst = b.peer[d] + (size_t)(d == 0 ? b.peer_halo[d] + b.peer_n_local[d] : b.peer_halo[d] - depth) * b.plane_bytes;
    SF_HIP_CHECK(hipMemcpyAsync(dst, sf::halo_send_planes(b, d, depth), bytes, hipMemcpyDeviceToDevice, h->send));
    sf::halo_flag_set(h->send, h->nb[d].dev + 16 * key + 2 + their, n);
  }
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_wait(*h, h->recv, h->own.dev + 16 * key + 2 + d, n);
  if (prof_end) SF_HIP_CHECK(hipEventRecord(prof_end, h->recv));
  SF_HIP_CHECK(hipEventRecord(b.sent, h->send));
  SF_HIP_CHECK(hipEventRecord(b.received, h->recv));
  b.pending = true;
  return SF_OK;
  SF_API_END
}

int sf_halo_set_profile(sf_halo* h, int on) {
  SF_API_BEGIN
  if (!h) throw Error(SF_ERR_INVALID, "sf_halo_set_profile: null transport");
  h->profile = on != 0;
  h->prof_used = 0;
  return SF_OK;
  SF_API_END
}

int sf_halo_exchange_times(sf_halo* h, int* count, double* mean_ms, double* max_ms) {
  SF_API_BEGIN
  if (!h || !count || !mean_ms || !max_ms) throw Error(SF_ERR_INVALID, "sf_halo_exchange_times: null argument");
  SF_HIP_CHECK(hipSetDevice(h->device));
  if (h->send) SF_HIP_CHECK(hipStreamSynchronize(h->send));
  if (h->recv) SF_HIP_CHECK(hipStreamSynchronize(h->recv));
  double sum = 0, worst = 0;
  for (size_t i = 0; i < h->prof_used; ++i) {
    float ms = 0;
    SF_HIP_CHECK(hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second));
    sum += ms;
    worst = std::max(worst, (double)ms);
  }
  *count = (int)h->prof_used;
  *mean_ms = h->prof_used ? sum / (double)h->prof_used : 0.0;
  *max_ms = worst;
  return SF_OK;
  SF_API_END
}

int sf_halo_finish(sf_halo* h, int key, void* compute_stream) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_finish: unknown buffer key");
  sf::HaloBuffer& b = h->bufs[key];
  if (!b.pending) return SF_OK;
  SF_HIP_CHECK(hipSetDevice(h->device));
  SF_HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, b.sent, 0));
  SF_HIP_CHECK(hipStreamWaitEvent((hipStream_t)compute_stream, b.received, 0));
  b.pending = false;
  return SF_OK;
  SF_API_END
}

int sf_halo_check(sf_halo* h) {
  SF_API_BEGIN
  if (!h) throw Error(SF_ERR_INVALID, "sf_halo_check: null transport");
  if (h->comm) {
    if (h->failed) throw Error(SF_ERR_DEVICE, "sf_halo: the RCCL transport has failed earlier (communicator ended)");
    sf::Rccl& nc = sf::rccl();
    // Bounded on the host: ncclSend / ncclRecv have no time limit of their own, so a neighbour that
    // died or missed an exchange would hold every later synchronisation.  The limit bounds the time WITHOUT PROGRESS
    // (ADVICE r04): an exchange sits behind everything queued on the compute stream -- this rank's launches, and through
    // its neighbour's sends the neighbour's -- so a healthy run with more queued work than the limit must not lose its
    // communicator.  The transport's stream ticks a word in pinned host memory twice per exchange (when the launches it
    // waits for are done, when it has arrived); the clock starts again whenever that word or an event moves.  If
    // nothing moves for the limit -- or RCCL reports an asynchronous error -- the communicator is ended (ncclCommAbort:
    // its kernels leave the streams) and the transport stays failed.
    SF_HIP_CHECK(hipSetDevice(h->device));
    auto last_progress = std::chrono::steady_clock::now();
    unsigned seen = h->progress_host ? __atomic_load_n(h->progress_host, __ATOMIC_ACQUIRE) : 0u;
    auto fail = [&](const std::string& what) {
      sf::halo_abort_comm(*h);
      throw Error(SF_ERR_DEVICE, what);
    };
    for (auto& kv : h->bufs) {
      sf::HaloBuffer& b = kv.second;
      if (b.count == 0 || !b.received) continue;
      for (;;) {
        sf::Rccl::result_t async = 0;
        if (nc.CommGetAsyncError && nc.CommGetAsyncError(h->comm, &async) == 0 && async != 0)
          fail(std::string("sf_halo: RCCL reports ") + nc.GetErrorString(async));
        const hipError_t q = hipEventQuery(b.received);
        if (q == hipSuccess) {
          last_progress = std::chrono::steady_clock::now();
          break;
        }
        if (q != hipErrorNotReady) fail(std::string("sf_halo: ") + hipGetErrorString(q));
        const unsigned now_at = h->progress_host ? __atomic_load_n(h->progress_host, __ATOMIC_ACQUIRE) : seen;
        if (now_at != seen) {
          seen = now_at;
          last_progress = std::chrono::steady_clock::now();
        }
        const auto idle = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - last_progress).count();
        if (idle > (long long)h->timeout_ms)
          fail("sf_halo: rank " + std::to_string(h->rank) + ": the RCCL halo exchanges made no progress for " + std::to_string(h->timeout_ms) +
               " ms (communicator ended)");
        std::this_thread::sleep_for(std::chrono::microseconds(200));
      }
    }
    return SF_OK;
  }
  const unsigned status = __atomic_load_n(static_cast<unsigned*>(h->own.host) + 1023, __ATOMIC_ACQUIRE);
  if (status != 0)
    throw Error(SF_ERR_DEVICE, "sf_halo: rank " + std::to_string(h->rank) + " or a neighbour waiting for it gave up after " +
                                   std::to_string(h->timeout_ms) + " ms");
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- native slab schedule
// One rank's execution of the chain with the deep-halo schedule, the C
// twin of stencilflow_amd/distributed.py: SlabRunner -- including its two measured
// refinements (sf_halo_configure): the exchange started a launch ahead, and compute units
// left free beside an exchange for the copy kernels of a device-side transport.
extern "C" int sf_plan_execute_decomposed(sf_plan* plan, sf_halo* halo, int repetitions) {
  SF_API_BEGIN
  if (!plan || !halo || repetitions < 0) throw Error(SF_ERR_INVALID, "sf_plan_execute_decomposed: bad argument");
  sf_plan& pl = *plan;
  ensure_device(pl);
  autotune(pl);
  const Program& P = pl.P;
  const int n = (int)pl.n_local, H = pl.halo;
  const bool has_lower = pl.goff > 0, has_upper = pl.goff + pl.n_local < P.n[0];
  const bool alone = !has_lower && !has_upper;
  if (!alone && H < 1) throw Error(SF_ERR_STATE, "sf_plan_execute_decomposed: the plan has no halo (option slab=lo:hi:halo)");
  // a chain: every launch reads exactly the slab buffer the previous one wrote
  bool chain = true;
  for (size_t s = 0; s < pl.steps.size(); ++s) {
    const Step& st = pl.steps[s];
    if (st.in_bufs.size() != 1 || st.out_bufs.size() != 1 || (s > 0 && st.in_bufs[0] != pl.steps[s - 1].out_buf)) chain = false;
  }
  auto reach_of = [&](size_t s) { return pl.steps[s].halo_buf >= 0 ? pl.steps[s].halo_depth : 0; };
  auto exchange_all = [&](const std::vector<int>& bufs, int depth) {
    for (int b : bufs) {
      const int rc = sf_halo_start(halo, b, depth, (void*)pl.stream);
      if (rc != SF_OK) throw Error(rc, sf_last_error());
    }
  };
  auto finish_all = [&](const std::vector<int>& bufs) {
    for (int b : bufs) {
      const int rc = sf_halo_finish(halo, b, (void*)pl.stream);
      if (rc != SF_OK) throw Error(rc, sf_last_error());
    }
  };
  // the launch that runs beside a transfer leaves a few compute units to the copy kernels
  // of a device-side transport (its blocks run ~200 us and hold nearly all registers of
  // their unit: a copy kernel would queue behind them)
  auto launch_beside = [&](const Step& st, int i_begin, int i_end) {
    const int before = pl.reserved_cus;
    pl.reserved_cus = halo->reserved_cus;
    try {
      launch_ranges(pl, st, i_begin, i_end, 0, 0, pl.stream);
    } catch (...) {
      pl.reserved_cus = before;
      throw;
    }
    pl.reserved_cus = before;
  };
  // program inputs no launch writes (extra fields, auxiliary fields): their ghost planes are
  // filled once per call, to the full halo depth, at the first launch that reads them
  std::set<int> fixed, fresh;
  for (int i = 0; i < P.num_inputs; ++i) fixed.insert(pl.input_buf[i]);
  for (const Step& st : pl.steps)
    for (int ob : st.out_bufs) fixed.erase(ob);
  for (int rep = 0; rep < repetitions; ++rep) {
    int valid = 0;        // ghost planes of the chain's current field that are still good
    long long early = -1;  // step whose exchange was started a launch ahead
    for (size_t s = 0; s < pl.steps.size(); ++s) {
      const Step& st = pl.steps[s];
      const int d = reach_of(s);
      // (a zero-reach launch of a chain must still recompute the ghost planes that are good, or the next launch
      // reads planes nobody wrote: the chain case comes first, as in SlabRunner.step_begin -- ADVICE r03)
      if (alone || (d == 0 && !chain)) {
        launch_ranges(pl, st, 0, n, 0, 0, pl.stream);
        continue;
      }
      if (2 * std::max(d, chain ? H : d) > n) throw Error(SF_ERR_STATE, "slab too thin for its halo");
      if (chain && d <= valid) {
        const int ext = valid - d;
        const int lo_ext = has_lower ? ext : 0, hi_ext = has_upper ? ext : 0;
        if (halo->early_exchange && s + 1 < pl.steps.size() && reach_of(s + 1) > ext && n >= 2 * H) {
          // The NEXT launch needs fresh halos.  What it will send are this launch's planes
          // next to the slab boundaries: compute those first (one two-range launch), start
          // the exchange, and let it run beside the interior of this launch AND of the next.
          const int lo_cut = has_lower ? H : 0, hi_cut = has_upper ? n - H : n;
          launch_ranges(pl, st, -lo_ext, lo_cut, hi_cut, n + hi_ext, pl.stream);
          exchange_all({st.out_buf}, H);
          launch_beside(st, lo_cut, hi_cut);
          early = (long long)s + 1;
        } else {
          launch_ranges(pl, st, -lo_ext, n + hi_ext, 0, 0, pl.stream);
        }
        valid = ext;
        continue;
      }
      std::vector<int> bufs;
      if (chain) {
        bufs.push_back(st.in_bufs[0]);
      } else {
        for (int b : st.in_bufs)
          if (pl.buffers[b].slabbed && pl.buffers[b].planes > 1 && std::find(bufs.begin(), bufs.end(), b) == bufs.end())
            bufs.push_back(b);
      }
      const int depth = chain ? H : d;
      if (early == (long long)s) {
        early = -1;  // started beside the previous launch
      } else if (!chain) {
        std::vector<int> now, once;
        for (int b : bufs) {
          if (fresh.count(b)) continue;
          if (fixed.count(b)) {
            once.push_back(b);
            fresh.insert(b);
          } else {
            now.push_back(b);
          }
        }
        exchange_all(once, H);
        bufs = now;
        bufs.insert(bufs.end(), once.begin(), once.end());  // (finish_all below waits for both)
        exchange_all(now, depth);
      } else {
        exchange_all(bufs, depth);
      }
      launch_beside(st, has_lower ? d : 0, n - (has_upper ? d : 0));  // beside the transfer
      finish_all(bufs);
      const int ext = depth - d;
      launch_ranges(pl, st, has_lower ? -ext : 0, has_lower ? d : 0, has_upper ? n - d : 0, has_upper ? n + ext : 0,
                    pl.stream);
      valid = chain ? ext : 0;
    }
  }
  return SF_OK;
  SF_API_END
}
