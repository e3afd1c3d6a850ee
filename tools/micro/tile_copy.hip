// tile_copy.hip -- the C3 launch's memory traffic without its arithmetic: 32 tiles x 8 chunks = 256 blocks of 512
// threads; block (tile t, chunk c) marches through the 64 planes of its chunk (+ 2 warm-up planes on either side),
// per plane reads the tile's 20 rows (16 own + 2 halo rows above and below: 40 KB contiguous, 5 x 16 B per thread,
// two planes in flight) and writes its 16 own rows (32 KB, non-temporal).  Plane stride = 1 MiB + `pad` bytes: does
// the 64 MiB distance between the eight chunk streams (same channels, same banks at the same moment?) cost anything?
//   hipcc --offload-arch=gfx950 -O3 -o tile_copy tile_copy.hip && ./tile_copy
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int NTL>
__global__ void __launch_bounds__(512) tile_copy(const char* __restrict__ in, char* __restrict__ out, size_t pitch, int planes,
                                                 int chunk, int rotate) {
  const int b = blockIdx.x, xcd = b & 7, L = xcd * (gridDim.x >> 3) + (b >> 3);  // XCD-aware order as in star3d.h
  const int t = L % 32, c = L / 32;
  const int cb = c * chunk, ce = cb + chunk < planes ? cb + chunk : planes;
  const int tid = threadIdx.x;
  // rows 16 t - 2 .. 16 t + 18 of a 512 x 512 float plane: byte offset of this thread's five vectors
  const long row0 = 16L * t - 2;
  size_t ld[5], st[5];
  bool ld_ok[5], st_ok[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const long row = row0 + (tid >> 7) * 5 + r;  // thread row tid / 128 owns five consecutive rows
    ld_ok[r] = row >= 0 && row < 512;
    st_ok[r] = ld_ok[r] && row >= 16L * t && row < 16L * t + 16;
    ld[r] = (size_t)(row < 0 ? 0 : row) * 2048 + (size_t)(tid & 127) * 16;
    st[r] = ld[r];
  }
  f4 w[3][5];
  auto load = [&](int slot, int p) {
    const int pp = rotate ? (p - cb + 8 * c) % chunk + cb : p;  // (rotate: every chunk starts at another plane of its range)
    const bool ok = p >= 0 && p < planes && pp >= 0 && pp < planes;
    const char* base = in + (size_t)(ok ? pp : 0) * pitch;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
      if (ok && ld_ok[r]) w[slot][r] = NTL && (((tid >> 7) * 5 + r) >= 4 && ((tid >> 7) * 5 + r) < 16)
                                           ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(base + ld[r]))
                                           : *reinterpret_cast<const f4*>(base + ld[r]);
    }
  };
  load(0, cb - 2);
  load(1, cb - 1);
  for (int p = cb - 2; p < ce + 2; p += 3) {
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
      const int q = p + ph;
      load((ph + 2) % 3, q + 2);
      const int qq = rotate ? (q - cb + 8 * c) % chunk + cb : q;
      if (q >= cb && q < ce) {
        char* base = out + (size_t)qq * pitch;
#pragma unroll
        for (int r = 0; r < 5; ++r)
          if (st_ok[r]) __builtin_nontemporal_store(w[ph][r], reinterpret_cast<f4*>(base + st[r]));
      }
    }
  }
}

int main() {
  const int planes = 512;
  const size_t plane = 512 * 512 * 4;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (size_t pad : {(size_t)0, (size_t)256, (size_t)4096, (size_t)8192, (size_t)65536, (size_t)(1 << 20) / 8 + 4096}) {
    const size_t pitch = plane + pad, bytes = pitch * planes;
    char *in, *out;
    if (hipMalloc(&in, bytes) != hipSuccess || hipMalloc(&out, bytes) != hipSuccess) return 1;
    hipMemset(in, 1, bytes);
    hipMemset(out, 0, bytes);
    for (int variant = 0; variant < 3; ++variant) {
      auto launch = [&](const char* src, char* dst) {
        if (variant == 0) tile_copy<0><<<256, 512>>>(src, dst, pitch, planes, 64, 0);
        else if (variant == 1) tile_copy<1><<<256, 512>>>(src, dst, pitch, planes, 64, 0);
        else tile_copy<0><<<256, 512>>>(src, dst, pitch, planes, 64, 1);
      };
      for (int i = 0; i < 20; ++i) launch(i & 1 ? out : in, i & 1 ? in : out);  // ping-pong, as the chain does
      hipDeviceSynchronize();
      float best = 1e30f;
      for (int round = 0; round < 3; ++round) {
        hipEventRecord(a);
        for (int i = 0; i < 100; ++i) launch(i & 1 ? out : in, i & 1 ? in : out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        if (ms / 100 < best) best = ms / 100;
      }
      // compulsory bytes: the field once in, once out (halo rows are re-read from L2)
      std::printf("pad %7zu B  %-34s %7.1f us  %5.2f TB/s (compulsory bytes)\n", pad,
                  variant == 0 ? "plain loads" : variant == 1 ? "own rows loaded non-temporally" : "chunks start at rotated planes", best * 1e3,
                  2.0 * plane * planes / (best * 1e-3) / 1e12);
    }
    hipFree(in);
    hipFree(out);
  }
  return 0;
}
