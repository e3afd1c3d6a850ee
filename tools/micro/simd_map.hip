// Which SIMD of its compute unit does wave w of a 512-thread (8-wave) workgroup run on?  One workgroup per compute
// unit; every wave reads HW_REG_HW_ID (gfx9: WAVE_ID[3:0], SIMD_ID[5:4], PIPE_ID[7:6], CU_ID[11:8], SH_ID[12],
// SE_ID[15:13]).  Prints, per wave index, how often each SIMD was seen over all workgroups -- the input to
// star3d.h's SF_WMAP question (which two waves of a block share a SIMD).
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/simd_map.hip -o tools/micro/simd_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) probe(unsigned* out, int waves) {
  const int w = (threadIdx.y * blockDim.x + threadIdx.x) >> 6;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * waves + w] = id;
  // stay resident for a while so that every workgroup gets a compute unit of its own
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

int main() {
  for (int by : {4, 8}) {
    const dim3 block(by == 4 ? 128 : 64, by);
    const int waves = block.x * block.y / 64, blocks = 256;
    unsigned* d;
    if (hipMalloc(&d, blocks * waves * sizeof(unsigned)) != hipSuccess) return 1;
    hipLaunchKernelGGL(probe, dim3(blocks), block, 0, 0, d, waves);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    std::vector<unsigned> h(blocks * waves);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    printf("block %dx%d (%d waves), %d workgroups: SIMD seen per wave index [simd0 simd1 simd2 simd3]\n", block.x, block.y,
           waves, blocks);
    for (int w = 0; w < waves; ++w) {
      int c[4] = {0, 0, 0, 0};
      for (int b = 0; b < blocks; ++b) c[(h[b * waves + w] >> 4) & 3]++;
      printf("  wave %d: %4d %4d %4d %4d\n", w, c[0], c[1], c[2], c[3]);
    }
    for (int b = 0; b < 4; ++b) {
      printf("  workgroup %d: simd of waves 0..%d =", b, waves - 1);
      for (int w = 0; w < waves; ++w) printf(" %u", (h[b * waves + w] >> 4) & 3);
      printf("   (cu %u se %u)\n", (h[b * waves] >> 8) & 15, (h[b * waves] >> 13) & 7);
    }
    hipFree(d);
  }
  return 0;
}
