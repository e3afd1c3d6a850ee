// ds_swizzle_b32 rotate mode on gfx950: which lane does lane i read for swizzle(ROTATE, dir, 1)?
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/swizzle_probe.hip -o tools/micro/swizzle_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_ds_swizzle(lane, 0xC020);        // ROTATE, dir 0, by 1
  out[64 + lane] = __builtin_amdgcn_ds_swizzle(lane, 0xC420);   // ROTATE, dir 1, by 1
}
int main() {
  int* d;
  if (hipMalloc(&d, 128 * sizeof(int)) != hipSuccess) return 1;
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  int h[128];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  for (int dir = 0; dir < 2; ++dir) {
    printf("swizzle(ROTATE,%d,1): lane i reads lane", dir);
    for (int i = 0; i < 64; ++i) printf(" %d", h[dir * 64 + i]);
    printf("\n");
  }
  return 0;
}
