// Does the Infinity Cache keep the tail of what a streaming kernel wrote, so that
// the next kernel gains from reading it back in the opposite direction?
// 256 blocks, each owning one contiguous chunk of a 512 MiB buffer (as the chunks
// of the plane-streaming stencil kernel do).  W writes its chunk front to back;
// R reads a chunk front to back (same direction) or back to front (opposite) and
// writes another buffer the same way.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ void __launch_bounds__(512) stream(const f4* __restrict__ in, f4* __restrict__ out, size_t vec_per_block,
                                              int reverse, float add) {
  const size_t base = (size_t)blockIdx.x * vec_per_block;
  const size_t steps = vec_per_block / 512;
  for (size_t s = 0; s < steps; ++s) {
    const size_t t = reverse ? steps - 1 - s : s;
    const size_t i = base + t * 512 + threadIdx.x;
    f4 v = in[i];
    v += add;
    if (NT) __builtin_nontemporal_store(v, &out[i]);
    else out[i] = v;
  }
}

// the same front-to-back stream with U loads in flight per thread before the stores
template <int U>
__global__ void __launch_bounds__(512) stream_deep(const f4* __restrict__ in, f4* __restrict__ out,
                                                   size_t vec_per_block, float add) {
  const size_t base = (size_t)blockIdx.x * vec_per_block;
  const size_t steps = vec_per_block / 512;
  for (size_t s = 0; s < steps; s += U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[base + (s + u) * 512 + threadIdx.x];
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_nontemporal_store(v[u] + add, &out[base + (s + u) * 512 + threadIdx.x]);
  }
}

// `streams` concurrent linear sweeps: the blocks of one stream cover consecutive 8 KiB
// pieces of it at every step (streams = 1: the whole chip sweeps the buffer front to
// back; streams = 8: the geometry of the stencil kernel's 8 chunks x 32 tiles)
__global__ void __launch_bounds__(512) stream_geo(const f4* __restrict__ in, f4* __restrict__ out, size_t nvec,
                                                  int streams, float add) {
  const int per = gridDim.x / streams;
  const int st = blockIdx.x / per, tile = blockIdx.x % per;
  const size_t span = nvec / streams;
  const size_t steps = span / ((size_t)per * 512);
  for (size_t s = 0; s < steps; ++s) {
    const size_t i = st * span + (s * per + tile) * 512 + threadIdx.x;
    __builtin_nontemporal_store(in[i] + add, &out[i]);
  }
}

static void run_geo(const f4* a, f4* b, size_t nvec, int blocks, int streams, hipEvent_t e0, hipEvent_t e1, size_t bytes) {
  std::vector<float> ms;
  for (int rep = 0; rep < 16; ++rep) {
    CK(hipEventRecord(e0));
    stream_geo<<<blocks, 512>>>((rep & 1) ? b : a, (rep & 1) ? (f4*)a : b, nvec, streams, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (rep >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2];
  std::printf("{\"blocks\": %d, \"concurrent linear sweeps\": %d, \"ms\": %.4f, \"GB/s\": %.1f}\n", blocks, streams, med,
              2.0 * bytes / med / 1e6);
}

template <int U>
static void run_deep(const f4* a, f4* b, size_t nvec, int blocks, hipEvent_t e0, hipEvent_t e1, size_t bytes) {
  std::vector<float> ms;
  for (int rep = 0; rep < 16; ++rep) {
    CK(hipEventRecord(e0));
    stream_deep<U><<<blocks, 512>>>((rep & 1) ? b : a, (rep & 1) ? (f4*)a : b, nvec / blocks, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t;
    CK(hipEventElapsedTime(&t, e0, e1));
    if (rep >= 4) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2];
  std::printf("{\"blocks\": %d, \"loads in flight per thread\": %d, \"ms\": %.4f, \"GB/s\": %.1f}\n", blocks, U, med,
              2.0 * bytes / med / 1e6);
}

int main() {
  const size_t bytes = 512ull << 20, nvec = bytes / 16;
  const int blocks = 256;
  f4 *a, *b;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes));
  CK(hipMemset(b, 0, bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int nt = 0; nt < 2; ++nt)
    for (int mode = 0; mode < 2; ++mode) {  // 0: every launch front to back; 1: alternate
      std::vector<float> ms;
      for (int rep = 0; rep < 24; ++rep) {
        const int rev = mode ? (rep & 1) : 0;
        const f4* src = (rep & 1) ? b : a;
        f4* dst = (rep & 1) ? a : b;
        CK(hipEventRecord(e0));
        if (nt) stream<true><<<blocks, 512>>>(src, dst, nvec / blocks, rev, 1.0f);
        else stream<false><<<blocks, 512>>>(src, dst, nvec / blocks, rev, 1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        if (rep >= 4) ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      const double med = ms[ms.size() / 2];
      std::printf("{\"stores\": \"%s\", \"direction\": \"%s\", \"ms\": %.4f, \"GB/s\": %.1f}\n", nt ? "nontemporal" : "plain",
                  mode ? "alternating" : "same", med, 2.0 * bytes / med / 1e6);
    }
  for (int blk : {256, 1024})
    for (int streams : {1, 2, 8, 32, 256}) run_geo(a, b, nvec, blk, streams, e0, e1, bytes);
  for (int blk : {256, 512, 1024}) {
    run_deep<1>(a, b, nvec, blk, e0, e1, bytes);
    run_deep<2>(a, b, nvec, blk, e0, e1, bytes);
    run_deep<4>(a, b, nvec, blk, e0, e1, bytes);
    run_deep<8>(a, b, nvec, blk, e0, e1, bytes);
  }
  return 0;
}
