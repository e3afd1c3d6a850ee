// overlap_probe.hip -- what keeps two waves' v_add_f32 from overlapping on a SIMD (round 5).  valu_rate.hip /
// pair_probe.hip (round 4) found 0.53 quad-cycles per add with two or more waves per SIMD and that ONE DPP move
// switches the overlap off for ~2000 cycles; the dense kernels, which have no DPP, still run their adds at ~1.0
// (profiles/r05_box27_counters.log).  Here: N adds, then ONE other instruction -- LDS read, LDS write, LDS-DMA piece,
// buffer load, buffer store, barrier, wait, f64 multiply -- for N = 64 ... 4096, at 2 and 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/overlap_probe.hip -o tools/micro/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define X8(S) S S S S S S S S
#define ADD8 "v_add_f32 %0, %8, %0\n\tv_add_f32 %1, %9, %1\n\tv_add_f32 %2, %10, %2\n\tv_add_f32 %3, %11, %3\n\tv_add_f32 %4, %12, %4\n\tv_add_f32 %5, %13, %5\n\tv_add_f32 %6, %14, %6\n\tv_add_f32 %7, %15, %7\n\t"
typedef float f4 __attribute__((ext_vector_type(4)));

// K: the other instruction; G: groups of 64 adds between two of them
template <int K, int G>
__global__ void __launch_bounds__(1024) rate(float* out, const float* in, float seed, int iters) {
  extern __shared__ float lds[];
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; b[i] = seed * (i + 2); }
  const unsigned laddr = threadIdx.x * 16;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 20, 0x00020000);
  const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 1 << 20, 0x00020000);
  const unsigned voff = (blockIdx.x * 1024 + threadIdx.x) * 16 % (1 << 20);
  f4 t = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < G; ++g)
      asm volatile(X8(ADD8)
                   : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                   : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    if (K == 1) asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(laddr));                       // result never waited for in the loop
    if (K == 2) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(laddr));
    if (K == 3) asm volatile("ds_write_b128 %0, %1" : : "v"(laddr), "v"(t));
    if (K == 4) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" : : "v"(t), "v"(voff), "s"(rs) : "memory");
    if (K == 5) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(t) : "v"(voff), "s"(rz));
    if (K == 6) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rz), "s"(__builtin_amdgcn_readfirstlane((threadIdx.x >> 6) * 1024)) : "memory");
    if (K == 7) asm volatile("s_barrier");
    if (K == 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    if (K == 9) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&t)));
    if (K == 10) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1" : "+v"(t[0]) : "v"(a[0]));
    if (K == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t[0]) : "v"(a[0]));
    if (K == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(t[0]), "v"(a[0]) : "vcc");
    if (K == 13) asm volatile("ds_read_b32 %0, %1" : "=v"(t[0]) : "v"(laddr));
    if (K == 14) asm volatile("s_load_dword s20, %0, 0x0" : : "s"(in) : "s20");
    if (K == 15) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(*reinterpret_cast<double*>(&t)) : "v"(a[0]));
    if (K == 16) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(a[0]) : "s20");
    if (K == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(t[0]) : "v"(a[0]));
    if (K == 18) asm volatile("v_lshlrev_b32 %0, 2, %0" : "+v"(t[0]));
    if (K == 19) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(t[0]) : "v"(a[0]));
    if (K == 20) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(t[0]) : "v"(a[0]));
    if (K == 21) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(t[0]) : "v"(a[0]));
    // register allocation: the kernel owns registers up to v130 / v250 (the dense kernels hold 120-245)
    if (K == 22) asm volatile("v_add_u32 v130, v130, %0" : : "v"(a[0]) : "v130");
    if (K == 23) asm volatile("v_add_u32 v250, v250, %0" : : "v"(a[0]) : "v250");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  float s = t[0] + t[1] + t[2] + t[3];
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

static float* d_out;
static float* d_in;
static double hz;

template <int K, int G>
static double one(int w) {
  auto kern = rate<K, G>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  const int iters = 262144 / (64 * G);  // 4096 groups of 64 adds in all
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256 * w), 100 * 1024, 0, d_out, d_in, 1.0f, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
    hipEventDestroy(e0);
    hipEventDestroy(e1);
  }
  return best * 1e-3 * hz / 4.0 / (double(w) * iters * G * 64.0);
}

template <int K>
static void run(const char* what) {
  printf("%-46s", what);
  for (int w : {2, 3, 4}) {
    if ((K == 22 && w > 3) || (K == 23 && w > 2)) continue;
    printf(" | w%d:", w);
    printf(" %5.2f", one<K, 1>(w));
    printf(" %5.2f", one<K, 4>(w));
    printf(" %5.2f", one<K, 16>(w));
    printf(" %5.2f", one<K, 64>(w));
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  hz = khz * 1e3;
  if (hipMalloc(&d_out, 8 << 20) != hipSuccess || hipMalloc(&d_in, 8 << 20) != hipSuccess) return 1;
  hipMemset(d_in, 0, 8 << 20);
  printf("quad-cycles per v_add_f32 and SIMD (peak clock %d MHz); one other instruction every 64 / 256 / 1024 / 4096 adds; 2, 3 and 4 waves per SIMD\n", khz / 1000);
  run<0>("adds only");
  run<1>("ds_read_b128 (never waited for)");
  run<2>("ds_read_b128 + s_waitcnt lgkmcnt(0)");
  run<13>("ds_read_b32");
  run<3>("ds_write_b128");
  run<4>("buffer_store_dwordx4");
  run<5>("buffer_load_dwordx4");
  run<6>("buffer_load_dwordx4 ... lds");
  run<7>("s_barrier");
  run<8>("s_waitcnt vmcnt(0) lgkmcnt(0)");
  run<14>("s_load_dword");
  run<9>("v_mul_f64");
  run<15>("v_cvt_f64_f32");
  run<10>("v_mov_b32 DPP wave_shr");
  run<11>("v_cndmask_b32 (vcc)");
  run<12>("v_cmp_lt_f32");
  run<16>("v_readfirstlane_b32");
  run<17>("v_add_u32");
  run<18>("v_lshlrev_b32");
  run<19>("v_mad_u32_u24");
  run<20>("v_fma_f32");
  run<21>("v_mul_f32");
  run<22>("131 registers per wave (v_add_u32 v130)");
  run<23>("251 registers per wave (v_add_u32 v250)");
  return 0;
}
