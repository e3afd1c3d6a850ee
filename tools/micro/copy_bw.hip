// copy_bw.hip -- what a plain copy reaches on this chip, in the access shapes the
// guide quotes (MI355X_MICROARCH.md: 6.29 TB/s for a float4 copy) and in the
// shapes the stencil kernels use.  Bytes counted = read + written.
//   hipcc --offload-arch=gfx950 -O3 -o copy_bw copy_bw.hip && ./copy_bw [MiB per buffer, default 512]
// With 64 (the C2 field: two 64 MiB buffers, both resident in the 256 MiB Infinity Cache) every
// shape is also run PING-PONG -- a -> b, then b -> a, as a chain of launches reads what the previous
// one wrote -- which is the fabric / MALL ceiling the 2-D kernels work under.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("%s: %s\n", #x, hipGetErrorString(e_));                         \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

// one float4 per thread, as many blocks as it takes
__global__ void copy_flat(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
template <int NT>
__global__ void copy_flat_nt(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    f4 v = (NT & 2) ? __builtin_nontemporal_load(&in[i]) : in[i];
    if (NT & 1) __builtin_nontemporal_store(v, &out[i]);
    else out[i] = v;
  }
}
// grid-stride, U float4 in flight per thread
template <int U, int NT>
__global__ void copy_stride(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (NT & 2) ? __builtin_nontemporal_load(&in[i + u * stride]) : in[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT & 1) __builtin_nontemporal_store(v[u], &out[i + u * stride]);
      else out[i + u * stride] = v[u];
    }
  }
  for (; i < n; i += stride) out[i] = in[i];
}
// each block sweeps its own contiguous chunk (the stencil kernels' shape: a block
// marches through planes of its tile), U rows of 16 B per lane in flight
template <int U, int NT>
__global__ void copy_chunked(const f4* __restrict__ in, f4* __restrict__ out, size_t n) {
  const size_t per_block = n / gridDim.x;
  const size_t base = (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i + (size_t)(U - 1) * blockDim.x < per_block; i += (size_t)U * blockDim.x) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (NT & 2) ? __builtin_nontemporal_load(&in[base + i + u * blockDim.x]) : in[base + i + u * blockDim.x];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT & 1) __builtin_nontemporal_store(v[u], &out[base + i + u * blockDim.x]);
      else out[base + i + u * blockDim.x] = v[u];
    }
  }
}

template <typename L>
static double time_ms(L launch, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  double best = 1e30;
  for (int round = 0; round < 3; ++round) {
    hipEventRecord(a);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (ms / reps < best) best = ms / reps;
  }
  return best;
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? (size_t)std::atoll(argv[1]) : 512;  // 512 = the C3 field, 64 = the C2 field
  const size_t bytes = mib << 20;
  const bool pingpong = mib <= 96;
  const size_t n = bytes / sizeof(f4);
  f4 *in, *out;
  CHECK(hipMalloc(&in, bytes));
  CHECK(hipMalloc(&out, bytes));
  CHECK(hipMemset(in, 1, bytes));
  CHECK(hipMemset(out, 0, bytes));
  auto report = [&](const char* name, double ms) {
    std::printf("%-56s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, 2.0 * bytes / (ms * 1e-3) / 1e12);
  };
  report("flat, 256 thr, 1 float4/thread", time_ms([&] { copy_flat<<<(unsigned)((n + 255) / 256), 256>>>(in, out, n); }, 10));
  report("flat, 512 thr", time_ms([&] { copy_flat<<<(unsigned)((n + 511) / 512), 512>>>(in, out, n); }, 10));
  report("flat, 1024 thr", time_ms([&] { copy_flat<<<(unsigned)((n + 1023) / 1024), 1024>>>(in, out, n); }, 10));
  report("flat, 256 thr, nt store", time_ms([&] { copy_flat_nt<1><<<(unsigned)((n + 255) / 256), 256>>>(in, out, n); }, 10));
  report("flat, 256 thr, nt load+store", time_ms([&] { copy_flat_nt<3><<<(unsigned)((n + 255) / 256), 256>>>(in, out, n); }, 10));
  for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
    char name[128];
    std::snprintf(name, sizeof name, "grid-stride x4, 256 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_stride<4, 0><<<blocks, 256>>>(in, out, n); }, 10));
    std::snprintf(name, sizeof name, "grid-stride x4 nt store, 256 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_stride<4, 1><<<blocks, 256>>>(in, out, n); }, 10));
    std::snprintf(name, sizeof name, "grid-stride x8, 512 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_stride<8, 0><<<blocks, 512>>>(in, out, n); }, 10));
  }
  for (int blocks : {256, 512, 1024}) {
    char name[128];
    std::snprintf(name, sizeof name, "chunk per block x5, 512 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_chunked<5, 0><<<blocks, 512>>>(in, out, n); }, 10));
    std::snprintf(name, sizeof name, "chunk per block x5 nt store, 512 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_chunked<5, 1><<<blocks, 512>>>(in, out, n); }, 10));
    std::snprintf(name, sizeof name, "chunk per block x10 nt store, 512 thr, %d blocks", blocks);
    report(name, time_ms([&] { copy_chunked<10, 1><<<blocks, 512>>>(in, out, n); }, 10));
  }
  if (pingpong) {
    // a chain of launches, each reading what the previous one wrote (both buffers stay in the Infinity Cache)
    auto chain = [&](auto one, int reps) {
      return time_ms([&] { one(in, out); one(out, in); }, reps) / 2.0;
    };
    std::printf("-- ping-pong (a -> b, b -> a), %zu MiB per buffer\n", mib);
    report("pp flat, 256 thr, 1 float4/thread",
           chain([&](f4* a, f4* b) { copy_flat<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }, 20));
    for (int blocks : {1024, 2048, 3072, 4096, 8192}) {
      char name[128];
      std::snprintf(name, sizeof name, "pp grid-stride x4, 256 thr, %d blocks", blocks);
      report(name, chain([&](f4* a, f4* b) { copy_stride<4, 0><<<blocks, 256>>>(a, b, n); }, 20));
      std::snprintf(name, sizeof name, "pp chunk per block x4, 64 thr (one wave, the 2-D kernels' shape), %d blocks", blocks);
      report(name, chain([&](f4* a, f4* b) { copy_chunked<4, 0><<<blocks, 64>>>(a, b, n); }, 20));
      std::snprintf(name, sizeof name, "pp chunk per block x8, 64 thr, %d blocks", blocks);
      report(name, chain([&](f4* a, f4* b) { copy_chunked<8, 0><<<blocks, 64>>>(a, b, n); }, 20));
    }
  }
  return 0;
}
