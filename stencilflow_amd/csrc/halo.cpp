// halo.cpp — the multi-GPU half of the C ABI: stream-ordered flags in shared host
// memory, the peer-to-peer halo transport (sf_halo_*) and the deep-halo schedule of a
// slab-decomposed run (sf_plan_execute_decomposed).
#include "sf_internal.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
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
                                           unsigned long long timeout_ticks, unsigned int* status) {
  const unsigned long long t0 = wall_clock64();  // constant-rate counter (100 MHz)
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - value) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {  // never spin forever
      if (status) __hip_atomic_store(status, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, ticks, status);
  SF_HIP_CHECK(hipGetLastError());
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- sf_halo
// Peer-to-peer halo transport of the slab decomposition, owned by the library: a
// rank PUSHES the planes next to a slab boundary straight into its neighbour's
// ghost planes -- device memory of the neighbour's plan, mapped here through a HIP
// IPC handle -- with DMA copies (no compute units, over xGMI between the GPUs of a
// node), ordered by the flag words of a small page of host memory that the ranks of
// the node share (POSIX shared memory, pinned; sf_flag_set / sf_flag_wait above).
// Everything is enqueued on two streams of the transport; the caller's compute
// stream only waits for their events.
//
// Flag page of rank r (unsigned words), per registered buffer `key` (< 60):
//   [16 key + d]      ready[d]  : r may receive exchange n from neighbour d (0 lower, 1 upper)
//   [16 key + 2 + d]  arrived[d]: neighbour d has delivered exchange n (written by that neighbour)
//   [1023]            status    : a wait of r timed out
#include <fcntl.h>
#include <sys/mman.h>

namespace sf {

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
  // neighbours (0 lower, 1 upper): their buffer mapped here and its geometry
  char* peer[2] = {nullptr, nullptr};
  int peer_n_local[2] = {0, 0}, peer_halo[2] = {0, 0};
  hipEvent_t sent = nullptr, received = nullptr;
  bool pending = false;
};

}  // namespace sf

struct sf_halo {
  int rank = 0, world = 1, device = 0;
  unsigned timeout_ms = 20000;
  std::string session;
  sf::FlagPage own, nb[2];
  std::map<int, sf::HaloBuffer> bufs;
  hipStream_t send = nullptr, recv = nullptr;
  hipEvent_t now = nullptr;
};

namespace sf {

static const size_t kFlagPageBytes = 4096;

static void map_flag_page(FlagPage& page, const std::string& name, bool create) {
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
static void halo_flag_wait(sf_halo& h, hipStream_t s, const unsigned* flag, unsigned value) {
  const unsigned long long ticks = (unsigned long long)std::max(1u, h.timeout_ms) * 100000ull;
  hipLaunchKernelGGL(sf_flag_wait_kernel, dim3(1), dim3(1), 0, s, flag, value, ticks, h.own.dev + 1023);
  SF_HIP_CHECK(hipGetLastError());
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
  sf::map_flag_page(h->own, "/sf_halo_" + h->session + "_" + std::to_string(rank), true);
  *out = h.release();
  return SF_OK;
  SF_API_END
}

int sf_halo_destroy(sf_halo* h) {
  SF_API_BEGIN
  if (!h) return SF_OK;
  (void)hipSetDevice(h->device);
  if (h->send) (void)hipStreamSynchronize(h->send);
  if (h->recv) (void)hipStreamSynchronize(h->recv);
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
  if (h->send) (void)hipStreamDestroy(h->send);
  if (h->recv) (void)hipStreamDestroy(h->recv);
  delete h;
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
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.sent, hipEventDisableTiming));
  SF_HIP_CHECK(hipEventCreateWithFlags(&b.received, hipEventDisableTiming));
  sf::HaloBlob out;
  std::memset(&out, 0, sizeof out);
  std::memcpy(out.magic, "SFHALO1", 8);
  if (h->world > 1) SF_HIP_CHECK(hipIpcGetMemHandle(&out.mem, device_base));
  out.plane_bytes = plane_bytes;
  out.n_local = n_local;
  out.halo = halo;
  out.rank = h->rank;
  out.device = h->device;
  std::snprintf(out.flags_name, sizeof out.flags_name, "%s", h->own.name.c_str());
  std::memset(blob, 0, SF_HALO_BLOB_BYTES);
  std::memcpy(blob, &out, sizeof out);
  h->bufs[key] = b;
  return SF_OK;
  SF_API_END
}

int sf_halo_connect(sf_halo* h, int key, const void* lower_blob, const void* upper_blob) {
  SF_API_BEGIN
  if (!h || !h->bufs.count(key)) throw Error(SF_ERR_INVALID, "sf_halo_connect: unknown buffer key");
  SF_HIP_CHECK(hipSetDevice(h->device));
  sf::HaloBuffer& b = h->bufs[key];
  const void* blobs[2] = {h->rank > 0 ? lower_blob : nullptr, h->rank < h->world - 1 ? upper_blob : nullptr};
  for (int d = 0; d < 2; ++d) {
    const bool expected = d == 0 ? h->rank > 0 : h->rank < h->world - 1;
    if (!expected) continue;
    if (!blobs[d]) throw Error(SF_ERR_INVALID, "sf_halo_connect: missing neighbour description");
    sf::HaloBlob in;
    std::memcpy(&in, blobs[d], sizeof in);
    if (std::memcmp(in.magic, "SFHALO1", 8) != 0 || in.rank != h->rank + (d == 0 ? -1 : 1))
      throw Error(SF_ERR_INVALID, "sf_halo_connect: not the description of the neighbouring rank");
    if (in.plane_bytes != b.plane_bytes || in.halo != b.halo)
      throw Error(SF_ERR_INVALID, "sf_halo_connect: the neighbour's buffer has another plane size or halo");
    void* mapped = nullptr;
    SF_HIP_CHECK(hipIpcOpenMemHandle(&mapped, in.mem, hipIpcMemLazyEnablePeerAccess));
    b.peer[d] = static_cast<char*>(mapped);
    b.peer_n_local[d] = in.n_local;
    b.peer_halo[d] = in.halo;
    if (!h->nb[d].host) sf::map_flag_page(h->nb[d], in.flags_name, false);
  }
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
  SF_HIP_CHECK(hipSetDevice(h->device));
  const unsigned n = ++b.count;
  const size_t bytes = (size_t)depth * b.plane_bytes;
  // the planes to send are final and the ghost planes no longer read once
  // everything queued on the compute stream so far has completed
  SF_HIP_CHECK(hipEventRecord(h->now, (hipStream_t)compute_stream));
  SF_HIP_CHECK(hipStreamWaitEvent(h->send, h->now, 0));
  SF_HIP_CHECK(hipStreamWaitEvent(h->recv, h->now, 0));
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_set(h->recv, h->own.dev + 16 * key + d, n);
  for (int d = 0; d < 2; ++d) {
    if (!b.peer[d]) continue;
    const int their = 1 - d;  // which of the neighbour's sides we are on
    sf::halo_flag_wait(*h, h->send, h->nb[d].dev + 16 * key + their, n);
    const char* src = b.base + (size_t)(d == 0 ? b.halo : b.halo + b.n_local - depth) * b.plane_bytes;
    char* dst = b.peer[d] + (size_t)(d == 0 ? b.peer_halo[d] + b.peer_n_local[d] : b.peer_halo[d] - depth) * b.plane_bytes;
    SF_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->send));
    sf::halo_flag_set(h->send, h->nb[d].dev + 16 * key + 2 + their, n);
  }
  for (int d = 0; d < 2; ++d)
    if (b.peer[d]) sf::halo_flag_wait(*h, h->recv, h->own.dev + 16 * key + 2 + d, n);
  SF_HIP_CHECK(hipEventRecord(b.sent, h->send));
  SF_HIP_CHECK(hipEventRecord(b.received, h->recv));
  b.pending = true;
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
  const unsigned status = __atomic_load_n(static_cast<unsigned*>(h->own.host) + 1023, __ATOMIC_ACQUIRE);
  if (status != 0)
    throw Error(SF_ERR_DEVICE, "sf_halo: a neighbour of rank " + std::to_string(h->rank) + " did not answer within " +
                                   std::to_string(h->timeout_ms) + " ms");
  return SF_OK;
  SF_API_END
}

}  // extern "C"

// ---------------------------------------------------------------- native slab schedule
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
    if (st.in_bufs.size() != 1 || (s > 0 && st.in_bufs[0] != pl.steps[s - 1].out_buf)) chain = false;
  }
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
  // program inputs no launch writes (extra fields, auxiliary fields): their ghost planes are
  // filled once per call, to the full halo depth, at the first launch that reads them
  std::set<int> fixed, fresh;
  for (int i = 0; i < P.num_inputs; ++i) fixed.insert(pl.input_buf[i]);
  for (const Step& st : pl.steps) fixed.erase(st.out_buf);
  for (int rep = 0; rep < repetitions; ++rep) {
    int valid = 0;  // ghost planes of the chain's current field that are still good
    for (size_t s = 0; s < pl.steps.size(); ++s) {
      const Step& st = pl.steps[s];
      const int d = st.halo_buf >= 0 ? st.halo_depth : 0;
      if (alone || d == 0) {
        launch_ranges(pl, st, 0, n, 0, 0, pl.stream);
        continue;
      }
      if (2 * std::max(d, chain ? H : d) > n) throw Error(SF_ERR_STATE, "slab too thin for its halo");
      if (chain && d <= valid) {
        const int ext = valid - d;
        launch_ranges(pl, st, has_lower ? -ext : 0, n + (has_upper ? ext : 0), 0, 0, pl.stream);
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
      if (!chain) {
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
      } else
      exchange_all(bufs, depth);
      launch_ranges(pl, st, has_lower ? d : 0, n - (has_upper ? d : 0), 0, 0, pl.stream);  // beside the transfer
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
