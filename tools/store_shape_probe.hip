// Chip-wide store rate of a [pixels][64 x bf16] (128-byte rows, contiguous) output written by grid-stride waves, 16 pixels (2 KiB) per
// wave and iteration, by the lane -> address mapping of the two 16-byte store instructions:
//   A: lane (g = l >> 4, col = l & 15) writes bytes [32g, 32g+16) and [32g+16, 32g+32) of pixel col     (what the MFMA layout gives)
//   B: instr 1 = pixels 0-7, instr 2 = pixels 8-15; lane writes chunk 2g + (col >> 3) of pixel col & 7   (whole rows, lanes strided)
//   C: instr i writes 1 KiB lane-linear: lane l -> byte 16 l of the i-th KiB                              (what a fill kernel does)
//   D: as A, one dwordx4 per lane and TWO pixels groups ... (not used)
// hipcc --offload-arch=gfx950 -O3 tools/store_shape_probe.hip -o /tmp/store_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SHAPE>
__global__ __launch_bounds__(256) void k(unsigned char* out, long long total_pix) {
  const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  uint4 v = make_uint4(lane, blockIdx.x, 3, 7);
  for (long long p0 = wave_id * 16; p0 + 16 <= total_pix; p0 += nwaves * 16) {
    unsigned char* base = out + p0 * 128;
    if (SHAPE == 0) {
      unsigned char* p = base + col * 128 + 32 * g;
      *reinterpret_cast<uint4*>(p) = v;
      *reinterpret_cast<uint4*>(p + 16) = v;
    } else if (SHAPE == 1) {
      unsigned char* p = base + (col & 7) * 128 + 32 * g + 16 * (col >> 3);
      *reinterpret_cast<uint4*>(p) = v;
      *reinterpret_cast<uint4*>(p + 1024) = v;
    } else {
      unsigned char* p = base + lane * 16;
      *reinterpret_cast<uint4*>(p) = v;
      *reinterpret_cast<uint4*>(p + 1024) = v;
    }
    v.x += 1;
  }
}
int main() {
  const long long pix = 64LL * 224 * 224;
  unsigned char* d; hipMalloc(&d, pix * 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid : {768, 1024, 1280, 2048})
    for (int shape = 0; shape < 3; ++shape) {
      float best = 1e9;
      for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, pix);
        else if (shape == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, pix);
        else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, pix);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("grid %4d shape %c: %.1f us for %.0f MB -> %.2f TB/s\n", grid, "ABC"[shape], best * 1e3, pix * 128 / 1e6, pix * 128 / best / 1e9);
    }
  return 0;
}
