// Per-CU store-issue rate of the conv epilogue's access shapes (hipcc --offload-arch=gfx950 -O3 tools/store_probe.hip -o /tmp/store_probe).
// 256 blocks x 4 waves; every wave writes a 112-pixel x 64-channel bf16 sub-tile (14 KiB) of a [M][ldc] tensor, `reps` times
// (different tiles), as 7 fragments of 16 pixels:
//   shape 0 (current epilogue): lane (frow, g) writes 2 x 16 B at pixel frow, byte offset 32g and 32g+16   (4 x 16-B pieces per 128 B per instr)
//   shape 1: lane (frow, g) writes 16 B at byte offset 16g (instr 0) and 64 + 16g (instr 1)                  (64 contiguous B per pixel per instr)
//   shape 2: one pixel per 8 lanes: lane l writes 16 B at pixel (l>>3), offset 16*(l&7)                      (128 contiguous B per pixel, 8 pixels per instr)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void k(unsigned char* out, long long ldc_bytes, int reps, int shape, int tiles_per_rep) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, g = lane >> 4;
  uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  for (int r = 0; r < reps; ++r) {
    const long long tile = (long long)r * tiles_per_rep + blockIdx.x;
    unsigned char* base = out + tile * 224 * ldc_bytes + (wave >> 1) * 112 * ldc_bytes + (wave & 1) * 128;
    if (shape == 0) {
#pragma unroll
      for (int mi = 0; mi < 7; ++mi) {
        unsigned char* p = base + (long long)(mi * 16 + frow) * ldc_bytes + 32 * g;
        *reinterpret_cast<uint4*>(p) = v;
        *reinterpret_cast<uint4*>(p + 16) = v;
      }
    } else if (shape == 1) {
#pragma unroll
      for (int mi = 0; mi < 7; ++mi) {
        unsigned char* p = base + (long long)(mi * 16 + frow) * ldc_bytes + 16 * g;
        *reinterpret_cast<uint4*>(p) = v;
        *reinterpret_cast<uint4*>(p + 64) = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 14; ++j) {
        unsigned char* p = base + (long long)(j * 8 + (lane >> 3)) * ldc_bytes + 16 * (lane & 7);
        *reinterpret_cast<uint4*>(p) = v;
      }
    }
  }
}
int main() {
  const long long ldc_bytes = 1024;  // 512 channels bf16
  const int reps = 4;
  size_t bytes = (size_t)reps * 256 * 224 * ldc_bytes;
  unsigned char* d; hipMalloc(&d, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int tiles = 256; tiles >= 16; tiles /= 4)  // fewer blocks = fewer CUs storing at once: per-CU store path vs chip write bandwidth
  for (int shape = 0; shape < 3; ++shape) {
    float best = 1e9;
    for (int it = 0; it < 5; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 0, 0, d, ldc_bytes, reps, shape, tiles);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double written = (double)reps * tiles * 4 * 14336;
    printf("blocks %3d shape %d: %.1f us for %.1f MB -> %.2f TB/s, %.2f us per 56-KiB tile per CU\n", tiles, shape, best * 1e3, written / 1e6, written / best / 1e9, best * 1e3 / reps);
  }
  return 0;
}
