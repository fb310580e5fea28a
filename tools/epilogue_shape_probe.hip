// In-kernel ISSUE time of the halo kernel's epilogue stores, per CU, by lane -> address shape (s_memtime around the instructions, no
// wait for completion: what a consumer wave is held up for before it can start the next tile's K-steps).
// One block = 4 waves = one 224-pixel x 128-channel bf16 tile of a [M][ldc] tensor (wave = 112 pixels x 64 channels = 14 KiB), `reps`
// different tiles back to back:
//   shape 0 (the epilogue today): fragment mi: lane (frow, g) writes 2 x 16 B of pixel frow at byte 32 g and 32 g + 16
//   shape 1: ... at byte 16 g and 64 + 16 g             (64 contiguous bytes per pixel and instruction)
//   shape 2: 8 lanes per pixel, 16 B each               (a pixel's whole 128 B per instruction, 8 pixels per instruction)
//   shape 3: as 2 through LDS: every lane first writes its MFMA-layout quads to a 2 KiB per-wave scratch and reads them back row-wise
//            (what the epilogue would have to do to get shape 2)
// hipcc --offload-arch=gfx950 -O3 tools/epilogue_shape_probe.hip -o /tmp/epilogue_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(unsigned char* out, int ldc_bytes, int reps, int shape, int tiles_per_rep, unsigned long long* cyc, unsigned total_bytes, int loads, int colmap) {
  __shared__ __attribute__((aligned(16))) unsigned char scratch[4][2][2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int frow = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)out, 0, (int)total_bytes, 0x00020000);
  u32x4 v[14];
#pragma unroll
  for (int i = 0; i < 14; ++i) v[i] = u32x4{(unsigned)lane, (unsigned)wave, blockIdx.x, (unsigned)i};
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    const int tile = r * tiles_per_rep + blockIdx.x;
    const int base = (tile * 224 + (wave >> 1) * 112) * ldc_bytes + (wave & 1) * 128;
    if (colmap) {
      // the halo kernel's tile: 8 rows x 28 columns of a 28-wide map (row stride 28 * ldc), wave = 14 columns x 64 channels; fragment mi = columns 2 mi, 2 mi + 1
      const int W = 28, cbase = (tile * 8 * W + (wave >> 1) * 14) * ldc_bytes + (wave & 1) * 128;
      if (shape == 0) {
#pragma unroll
        for (int mi = 0; mi < 7; ++mi) {
          const int p = cbase + ((frow & 7) * W + 2 * mi + (frow >> 3)) * ldc_bytes + 32 * g;
          if (loads) { v[2 * mi] += __builtin_amdgcn_raw_buffer_load_b128(rs, p + (int)(total_bytes / 2), 0, 0); v[2 * mi + 1] += __builtin_amdgcn_raw_buffer_load_b128(rs, p + 16 + (int)(total_bytes / 2), 0, 0); }
        }
#pragma unroll
        for (int mi = 0; mi < 7; ++mi) {
          const int p = cbase + ((frow & 7) * W + 2 * mi + (frow >> 3)) * ldc_bytes + 32 * g;
          __builtin_amdgcn_raw_buffer_store_b128(v[2 * mi], rs, p, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(v[2 * mi + 1], rs, p + 16, 0, 0);
        }
      } else {
        const int q = lane >> 3, c = lane & 7;
#pragma unroll
        for (int j = 0; j < 14; ++j)
          if (loads) v[j] += __builtin_amdgcn_raw_buffer_load_b128(rs, cbase + (q * W + j) * ldc_bytes + 16 * c + (int)(total_bytes / 2), 0, 0);
#pragma unroll
        for (int j = 0; j < 14; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, cbase + (q * W + j) * ldc_bytes + 16 * c, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 14; ++i) v[i][0] += 1;
      continue;
    }
    if (loads) {  // the residual: 14 loads in front of the stores (shape 0 / 1: MFMA layout; 2 / 3: a pixel's 128 B per 8 lanes), consumed before the first store
      if (shape <= 1) {
#pragma unroll
        for (int mi = 0; mi < 7; ++mi) {
          const int p = base + (mi * 16 + frow) * ldc_bytes + (shape == 0 ? 32 * g : 16 * g) + (int)(total_bytes / 2);
          v[2 * mi] += __builtin_amdgcn_raw_buffer_load_b128(rs, p, 0, 0);
          v[2 * mi + 1] += __builtin_amdgcn_raw_buffer_load_b128(rs, p + (shape == 0 ? 16 : 64), 0, 0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 14; ++j) v[j] += __builtin_amdgcn_raw_buffer_load_b128(rs, base + (j * 8 + (lane >> 3)) * ldc_bytes + 16 * (lane & 7) + (int)(total_bytes / 2), 0, 0);
      }
    }
    if (shape == 0 || shape == 1) {
#pragma unroll
      for (int mi = 0; mi < 7; ++mi) {
        const int p = base + (mi * 16 + frow) * ldc_bytes + (shape == 0 ? 32 * g : 16 * g);
        __builtin_amdgcn_raw_buffer_store_b128(v[2 * mi], rs, p, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(v[2 * mi + 1], rs, p + (shape == 0 ? 16 : 64), 0, 0);
      }
    } else if (shape == 2) {
#pragma unroll
      for (int j = 0; j < 14; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rs, base + (j * 8 + (lane >> 3)) * ldc_bytes + 16 * (lane & 7), 0, 0);
    } else {
#pragma unroll
      for (int mi = 0; mi < 7; ++mi) {
        unsigned char* sc = scratch[wave][mi & 1];
        // MFMA layout in: pixel frow, bytes [32 g, 32 g + 32) -> row-major scratch [16 pixels][128 B], 16-byte chunks XOR-swizzled by the pixel
        *reinterpret_cast<u32x4*>(sc + frow * 128 + (((2 * g) ^ (frow & 7)) << 4)) = v[2 * mi];
        *reinterpret_cast<u32x4*>(sc + frow * 128 + (((2 * g + 1) ^ (frow & 7)) << 4)) = v[2 * mi + 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int px = lane >> 3, ch = lane & 7;
        const u32x4 a = *reinterpret_cast<const u32x4*>(sc + px * 128 + ((ch ^ (px & 7)) << 4));
        const u32x4 b = *reinterpret_cast<const u32x4*>(sc + (px + 8) * 128 + ((ch ^ (px & 7)) << 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_raw_buffer_store_b128(a, rs, base + (mi * 16 + px) * ldc_bytes + 16 * ch, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(b, rs, base + (mi * 16 + 8 + px) * ldc_bytes + 16 * ch, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 14; ++i) v[i][0] += 1;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
    cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0;
  }
}
int main() {
  const int ldc_bytes = 1024;  // 512 channels bf16
  const int max_reps = 8;
  const size_t bytes = (size_t)2 * max_reps * 256 * 224 * ldc_bytes;  // second half: the tensor the loads read
  unsigned char* d; hipMalloc(&d, bytes);
  unsigned long long* c; hipMalloc(&c, 256 * 4 * 2 * 8);
  std::vector<unsigned long long> h(256 * 4 * 2);
  hipMemset(d, 1, bytes);
  for (int colmap = 0; colmap < 2; ++colmap)
  for (int loads = 0; loads < 2; ++loads)
  for (int reps : {8, 1})  // 8: a sustained stream (the chip's write bandwidth binds at 256 blocks); 1: ONE tile per CU, the burst a round of tiles ends in
  for (int tiles : {256, 64, 16})  // fewer blocks = fewer CUs storing at once: the CU's own store path vs the chip's
    for (int shape = 0; shape < (colmap ? 3 : 4); shape += (colmap ? 2 : 1)) {
      double best_issue = 1e30, best_done = 1e30;
      for (int it = 0; it < 5; ++it) {
        hipLaunchKernelGGL(k, dim3(tiles), dim3(256), 0, 0, d, ldc_bytes, reps, shape, tiles, c, (unsigned)bytes, loads, colmap);
        hipMemcpy(h.data(), c, tiles * 4 * 2 * 8, hipMemcpyDeviceToHost);
        double si = 0, sd = 0;
        for (int i = 0; i < tiles * 4; ++i) si += h[2 * i], sd += h[2 * i + 1];
        best_issue = std::min(best_issue, si / (tiles * 4) / reps);
        best_done = std::min(best_done, sd / (tiles * 4) / reps);
      }
      printf("%s%s reps %d blocks %3d shape %d: issue %.0f cycles per 14-KiB wave tile (%.0f per store instruction), to completion %.0f\n", colmap ? "[8 x 28 tile of a 28-wide map] " : "", loads ? "14 loads + 14 stores" : "14 stores", reps, tiles, shape, best_issue, best_issue / 14, best_done);
    }
  return 0;
}
