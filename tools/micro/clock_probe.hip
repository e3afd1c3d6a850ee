// Shader clock under sustained vector load: every wave runs a long stream of independent v_add_f32 (or v_add_f64) and
// reads the shader clock counter (clock64: s_memtime) and the constant 100 MHz counter (wall_clock64: s_memrealtime)
// before and after; the ratio is the frequency the compute units actually ran at.  One block of 256 x W threads per
// compute unit (W waves per SIMD), all 256 compute units busy, ~20-40 ms per launch.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/clock_probe.hip -o tools/micro/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define X8(S) S S S S S S S S
template <int F64>
__global__ void __launch_bounds__(1024) burn(unsigned long long* out, int iters, float seed) {
  float a[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; d[i] = a[i]; }
  float x = seed * 0.5f;
  double xd = x;
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (F64)
      asm volatile(X8("v_add_f64 %0, %8, %0\n\tv_add_f64 %1, %8, %1\n\tv_add_f64 %2, %8, %2\n\tv_add_f64 %3, %8, %3\n\tv_add_f64 %4, %8, %4\n\tv_add_f64 %5, %8, %5\n\tv_add_f64 %6, %8, %6\n\tv_add_f64 %7, %8, %7\n\t")
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]) : "v"(xd));
    else
      asm volatile(X8("v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %8, %1\n\tv_add_f32 %2, %8, %2\n\tv_add_f32 %3, %8, %3\n\tv_add_f32 %4, %8, %4\n\tv_add_f32 %5, %8, %5\n\tv_add_f32 %6, %8, %6\n\tv_add_f32 %7, %8, %7\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(x));
  }
  const unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + (float)d[i];
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    out[2 * w] = c1 - c0;
    out[2 * w + 1] = (w1 - w0) + (s == 12345.678f ? 1 : 0);
  }
}
template <int F64>
static void run(int waves_per_simd, int iters, unsigned long long* d) {
  const int blocks = 256, threads = 256 * waves_per_simd, nw = blocks * threads / 64;
  hipFuncSetAttribute(reinterpret_cast<const void*>(burn<F64>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((burn<F64>), dim3(blocks), dim3(threads), 100 * 1024, 0, d, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nw);
    hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double fmin = 1e30, fmax = 0, fsum = 0;
    for (int w = 0; w < nw; ++w) {
      const double f = (double)h[2 * w] / ((double)h[2 * w + 1] / 100e6);
      fmin = f < fmin ? f : fmin;
      fmax = f > fmax ? f : fmax;
      fsum += f;
    }
    const double inst = (double)waves_per_simd * iters * 64.0;
    printf("%s, %d waves per SIMD, launch %d: %.2f ms; shader clock / 100 MHz clock: min %.0f mean %.0f max %.0f MHz; "
           "%.2f clocks per instruction and SIMD at the mean\n",
           F64 ? "v_add_f64" : "v_add_f32", waves_per_simd, rep, ms, fmin / 1e6, fsum / nw / 1e6, fmax / 1e6,
           ms * 1e-3 * (fsum / nw) / inst);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
}
int main() {
  unsigned long long* d;
  if (hipMalloc(&d, 2 * 256 * 16 * sizeof(unsigned long long)) != hipSuccess) return 1;
  run<0>(1, 200000, d);
  run<0>(2, 200000, d);
  run<0>(4, 100000, d);
  run<1>(2, 100000, d);
  run<1>(4, 50000, d);
  hipFree(d);
  return 0;
}
