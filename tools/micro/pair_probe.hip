// Does the overlap of two waves' v_add_f32 on a SIMD (0.53 quad-cycles per instruction in valu_rate.hip, where one
// source is the same register in every instruction) survive when every instruction reads TWO registers that change
// from instruction to instruction -- the shape of a stencil sum (acc += value)?
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/pair_probe.hip -o tools/micro/pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITER 4000
#define X8(S) S S S S S S S S
template <int K>
__global__ void __launch_bounds__(1024) rate(float* out, float seed) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * (i + 2); }
  for (int it = 0; it < ITER; ++it) {
    if (K == 0)  // one shared source
      asm volatile(X8("v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %8, %1\n\tv_add_f32 %2, %8, %2\n\tv_add_f32 %3, %8, %3\n\tv_add_f32 %4, %8, %4\n\tv_add_f32 %5, %8, %5\n\tv_add_f32 %6, %8, %6\n\tv_add_f32 %7, %8, %7\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    if (K == 1)  // a different second source per instruction
      asm volatile(X8("v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %9, %1\n\tv_add_f32 %2, %10, %2\n\tv_add_f32 %3, %11, %3\n\tv_add_f32 %4, %12, %4\n\tv_add_f32 %5, %13, %5\n\tv_add_f32 %6, %14, %6\n\tv_add_f32 %7, %15, %7\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    if (K == 2)  // the stencil shape: four accumulators, each value feeds them in turn (acc[v] += row[v + dk])
      asm volatile(X8("v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %9, %1\n\tv_add_f32 %2, %10, %2\n\tv_add_f32 %3, %11, %3\n\tv_add_f32 %0, %9, %0\n\tv_add_f32 %1, %10, %1\n\tv_add_f32 %2, %11, %2\n\tv_add_f32 %3, %12, %3\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    if (K == 3)  // e64 encoding (VOP3) of the same adds
      asm volatile(X8("v_add_f32_e64 %0, %8, %0\n\tv_add_f32_e64 %1, %9, %1\n\tv_add_f32_e64 %2, %10, %2\n\tv_add_f32_e64 %3, %11, %3\n\tv_add_f32_e64 %4, %12, %4\n\tv_add_f32_e64 %5, %13, %5\n\tv_add_f32_e64 %6, %14, %6\n\tv_add_f32_e64 %7, %15, %7\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    if (K == 4)  // destination differs from both sources (a three-address sum: t = x + y)
      asm volatile(X8("v_add_f32 %0, %8, %9\n\tv_add_f32 %1, %9, %10\n\tv_add_f32 %2, %10, %11\n\tv_add_f32 %3, %11, %12\n\tv_add_f32 %4, %12, %13\n\tv_add_f32 %5, %13, %14\n\tv_add_f32 %6, %14, %15\n\tv_add_f32 %7, %15, %8\n\t")
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}
template <int K>
static void run(const char* what, float* d, double hz) {
  printf("%-64s", what);
  for (int w : {1, 2, 4}) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(rate<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL((rate<K>), dim3(256), dim3(256 * w), 100 * 1024, 0, d, 1.0f);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
      hipEventDestroy(e0);
      hipEventDestroy(e1);
    }
    printf("  w%d: %5.2f", w, best * 1e-3 * hz / 4.0 / (double(w) * ITER * 64.0));
  }
  printf("\n");
}
int main() {
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  float* d;
  if (hipMalloc(&d, 4096) != hipSuccess) return 1;
  printf("quad-cycles per v_add_f32 and SIMD (peak clock %d MHz), 1 / 2 / 4 waves per SIMD\n", khz / 1000);
  run<0>("acc[k] += x           (one source shared by all instructions)", d, khz * 1e3);
  run<1>("acc[k] += b[k]        (eight value registers in turn)", d, khz * 1e3);
  run<2>("acc[v] += row[v + dk] (four accumulators, sliding values)", d, khz * 1e3);
  run<3>("acc[k] += b[k], VOP3 encoding", d, khz * 1e3);
  run<4>("t[k] = b[k] + b[k+1]  (destination is neither source)", d, khz * 1e3);
  hipFree(d);
  return 0;
}
