// lds_dma_probe.hip — what `buffer_load_dwordx4 ... offen lds` does on gfx950 in the cases dense3d.h's plane staging
// relies on (round 5).  Build: hipcc --offload-arch=gfx950 -O3 lds_dma_probe.hip -o lds_dma_probe
//   case 0  plain: lane l of wave w writes LDS [M0 + 16 l], M0 = base + 1024 w
//   case 1  lanes whose offset lies outside the buffer resource (offset 0x80000000): zero written, or LDS left alone?
//   case 2  a resource of zero records: the same question for every lane
//   case 3  lanes switched off in EXEC: LDS left alone?
//   case 4  source 8 bytes off a 16-byte boundary
//   case 5  instruction offset field (offset:512) moves source AND destination? (prints what moved)
//   case 6  vmcnt: two DMAs in flight, `s_waitcnt vmcnt(1)` then barrier: the first one's data is there
// and a bandwidth figure: every block streams planes of (ROWS x 264 floats) through a three-slot ring two planes
// ahead, reading each plane once with ds_read_b128 -- DMA against register staging (buffer_load_dwordx2 + ds_write_b64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define RSRC_FLAGS 0x00020000

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}
__device__ __forceinline__ void dma16_off512(const __amdgpu_buffer_rsrc_t rs, const unsigned voff, const unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen offset:512 lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}

__global__ void probe(const float* in, float* out, int n, int which) {
  __shared__ float lds[2048];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 2048; i += blockDim.x) lds[i] = -7.0f;
  __syncthreads();
  const unsigned base = (unsigned)(size_t)lds;
  const unsigned wbase = __builtin_amdgcn_readfirstlane(base + wave * 1024);
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, n * 4, RSRC_FLAGS);
  unsigned voff = tid * 16;
  if (which == 0) dma16(rs, voff, wbase);
  if (which == 1) dma16(rs, (lane & 1) ? 0x80000000u : voff, wbase);
  if (which == 2) {
    __amdgpu_buffer_rsrc_t z = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0, RSRC_FLAGS);
    dma16(z, voff, wbase);
  }
  if (which == 3) {
    if (lane & 1) dma16(rs, voff, wbase);
  }
  if (which == 4) dma16(rs, voff + 8, wbase);
  if (which == 5) dma16_off512(rs, voff, wbase);
  if (which == 6) {
    if (wave == 0) {
      dma16(rs, voff, wbase);
      dma16(rs, voff + 1024, wbase + 1024);
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      out[2048 + lane] = 0;  // (nothing: keeps the shape)
    }
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = tid; i < 2048; i += blockDim.x) out[i] = lds[i];
}

// ---- bandwidth: planes through an LDS ring ----
#ifndef ROWS
#define ROWS 12
#endif
#define LS 264
#define SLOT (ROWS * LS)
#define CHUNKS (SLOT / 4)

template <int MODE, int THREADS>
__global__ void __launch_bounds__(THREADS, 2) stream(const float* in, float* out, int planes, int n1, int n2) {
  // MODE 0: register staging one plane ahead (pairs), two slots; MODE 1: LDS-DMA two planes ahead, three slots
  __shared__ float lds[3 * SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  constexpr int NW = THREADS / 64;
  const int jt = blockIdx.x % (n1 / (ROWS - 4)), ch = blockIdx.x / (n1 / (ROWS - 4));
  const int kt = ch & 1;
  const int chunk = ch >> 1;
  const int nchunks = gridDim.x / (n1 / (ROWS - 4)) / 2;
  const int pl = planes / nchunks, p0 = chunk * pl;
  const int tj0 = jt * (ROWS - 4) - 2, tk0 = kt * 256 - 4;
  const unsigned plane_bytes = (unsigned)(n1 * n2 * 4);
  f4 acc = {0, 0, 0, 0};
  if constexpr (MODE == 1) {
    constexpr int ND = (CHUNKS + THREADS - 1) / THREADS;
    unsigned off[ND];
    for (int n = 0; n < ND; ++n) {
      const int c = (n * NW + wave) * 64 + lane;
      const int row = c / (LS / 4), col = (c - row * (LS / 4)) * 4;
      const int j = tj0 + row, k = tk0 + col;
      off[n] = (c < CHUNKS && j >= 0 && j < n1 && k >= 0 && k < n2) ? (unsigned)((j * n2 + k) * 4) : 0x80000000u;
    }
    const unsigned base = (unsigned)(size_t)lds;
    auto issue = [&](int p) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(in) + (size_t)p * plane_bytes), 0, plane_bytes, RSRC_FLAGS);
      const unsigned sb = base + (unsigned)((p % 3) * SLOT * 4);
#pragma unroll
      for (int n = 0; n < ND; ++n) {
        const unsigned wb = __builtin_amdgcn_readfirstlane(sb + (unsigned)((n * NW + wave) * 1024));
        if ((n * NW + wave) * 64 < CHUNKS) dma16(rs, off[n], wb);
      }
    };
    issue(p0);
    issue(p0 + 1);
    const int ndw = ((CHUNKS + 63) / 64 - wave + NW - 1) / NW;  // DMAs this wave issues per plane
    for (int p = p0; p < p0 + pl; ++p) {
      // plane p has landed when at most one plane's DMAs are younger
      if (ndw == ND) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND - 1) : "memory");
      asm volatile("s_barrier" ::: "memory");
      if (p + 2 < p0 + pl) issue(p + 2);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float* sl = lds + (p % 3) * SLOT;
      for (int i = tid; i < CHUNKS; i += THREADS) acc += *reinterpret_cast<const f4*>(sl + 4 * i);
    }
  } else {
    constexpr int NP = (SLOT / 2 + THREADS - 1) / THREADS;
    unsigned off[NP];
    int dst[NP];
    for (int n = 0; n < NP; ++n) {
      const int c = n * THREADS + tid;
      const int row = c / (LS / 2), col = (c - row * (LS / 2)) * 2;
      const int j = tj0 + row, k = tk0 + col;
      off[n] = (c < SLOT / 2 && j >= 0 && j < n1 && k >= 0 && k < n2) ? (unsigned)((j * n2 + k) * 4) : 0x80000000u;
      dst[n] = c < SLOT / 2 ? 2 * c : -1;
    }
    f2 regs[NP];
    auto load = [&](int p) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(in) + (size_t)p * plane_bytes), 0, plane_bytes, RSRC_FLAGS);
#pragma unroll
      for (int n = 0; n < NP; ++n) regs[n] = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, off[n], 0, 0));
    };
    load(p0);
    for (int p = p0; p < p0 + pl; ++p) {
      float* sl = lds + (p & 1) * SLOT;
#pragma unroll
      for (int n = 0; n < NP; ++n)
        if (dst[n] >= 0) *reinterpret_cast<f2*>(sl + dst[n]) = regs[n];
      __syncthreads();
      if (p + 1 < p0 + pl) load(p + 1);
      for (int i = tid; i < CHUNKS; i += THREADS) acc += *reinterpret_cast<const f4*>(sl + 4 * i);
    }
  }
  out[(size_t)blockIdx.x * THREADS + tid] = acc[0] + acc[1] + acc[2] + acc[3];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int n = 1024;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)(i + 1);
  float *din, *dout;
  CK(hipMalloc(&din, n * 4 + 64));
  CK(hipMalloc(&dout, 4096 * 4));
  CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
  std::vector<float> o(2048);
  for (int which = 0; which <= 6; ++which) {
    probe<<<1, 128>>>(din, dout, which == 4 ? n - 2 : n, which);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dout, 2048 * 4, hipMemcpyDeviceToHost));
    std::printf("case %d:", which);
    for (int i = 0; i < 12; ++i) std::printf(" %g", o[i]);
    std::printf(" | [124..135]");
    for (int i = 124; i < 136; ++i) std::printf(" %g", o[i]);
    std::printf(" | [252..260]");
    for (int i = 252; i < 260; ++i) std::printf(" %g", o[i]);
    std::printf(" | [508..516]");
    for (int i = 508; i < 516; ++i) std::printf(" %g", o[i]);
    std::printf("\n");
  }
  // bandwidth
  const int n0 = 512, n1 = 512, n2 = 512;
  float *fin, *fout;
  CK(hipMalloc(&fin, (size_t)n0 * n1 * n2 * 4));
  CK(hipMalloc(&fout, (size_t)1 << 24));
  CK(hipMemset(fin, 0, (size_t)n0 * n1 * n2 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int jts = n1 / (ROWS - 4);
  for (int nch : {4, 8, 16}) {
    const int grid = jts * 2 * nch;
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        if (mode == 0) stream<0, 128><<<grid, 128>>>(fin, fout, n0, n1, n2);
        if (mode == 1) stream<1, 128><<<grid, 128>>>(fin, fout, n0, n1, n2);
        if (mode == 2) stream<0, 256><<<grid, 256>>>(fin, fout, n0, n1, n2);
        if (mode == 3) stream<1, 256><<<grid, 256>>>(fin, fout, n0, n1, n2);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      std::printf("stream rows %d chunks %d grid %d %s threads %d: %.1f us, %.0f GB/s of field bytes\n", ROWS, nch, grid,
                  (mode & 1) ? "LDS-DMA 2 ahead" : "registers 1 ahead", mode < 2 ? 128 : 256, best * 1e3,
                  (double)n0 * n1 * n2 * 4 / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
