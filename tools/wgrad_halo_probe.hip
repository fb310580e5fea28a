// PROBE (timing only, results meaningless): the instruction mix of a "window" weight gradient -- one staged 2-D X window with its halo and one
// dY tile serve all NINE taps of a 3x3 layer -- against the mix of conv_wgrad_ws2_kernel (one tap per item: dY and X re-staged per tap).
//   block = 4 consumer + 4 loader waves (one of each per SIMD), one block per CU, as the production kernels.
//   window variant: tile 64 cout x 64 cin x 9 taps, K-tile = 8 rows x 28 columns = 224 pixels (7 K32 steps) per stage: dY 224 x 128 B = 28 KiB +
//     X window (8 + 2d) x (28 + 2d) pixels x 128 B <= 48 KiB -> 76 pieces per stage, 19 per loader wave; two stages (152 KiB);
//     consumer wave = 32 cout x 32 cin x 9 taps: per K32 step 4 transposed reads of dY + 36 of X (9 taps x 2 fragments x 2) and 36 MFMAs, ONE barrier per stage.
//   ws2 variant (today's kernel): tile 256 x 128 of one tap, K-step = 64 pixels: 48 pieces per step (12 per loader wave), consumer wave 128 x 64:
//     48 transposed reads + 64 MFMAs per step, one barrier per step, three stages of 48 KiB.
// Reads hit a conflict-free swizzled image, DMA pieces stream from an L2-resident buffer.  Reports TFLOP/s (wall) of the MFMA work.
//   hipcc --offload-arch=gfx950 -O3 tools/wgrad_halo_probe.hip -o /tmp/wgrad_halo_probe && /tmp/wgrad_halo_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define BLDS16(rsrc, lptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)

__device__ __forceinline__ u32x2 tr_read(const unsigned char* p) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p));
}
__device__ __forceinline__ void mfma(f32x4& c, const u32x4& a, const u32x4& b) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---------------- window variant ----------------
template <int PIECES, bool STREAM = false>  // LDS-DMA pieces per loader wave per stage (19 = d 2; 17 = d 1; 0 = consumers alone)
__global__ __launch_bounds__(512, 2) void probe_window(const unsigned char* __restrict__ src, unsigned src_bytes, float* __restrict__ out, int stages) {
  constexpr int DY_BYTES = 224 * 128, STAGE = DY_BYTES + 384 * 128;  // 28 + 48 KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= 4) {  // loader
    const int lw = wave - 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    unsigned voff = (unsigned)(lane * 16 + lw * 4096 + (blockIdx.x & 63) * 16384);
    // STREAM: the layer's real traffic pattern -- tensors [50176 pixels][512 channels] bf16 (dY at 0, X behind it); block = (cout tile, cin tile, pixel range):
    // a piece = 8 pixel rows x 128 B of this block's 64-channel column; blocks of one range share the rows (L2), ranges walk the pixels
    const int tile = blockIdx.x & 63, range = blockIdx.x >> 6;
    if constexpr (STREAM) voff = (unsigned)((lane >> 3) * 1024 + (lane & 7) * 16);
    int buf = 0;
    for (int s = 0; s < stages; ++s) {
      unsigned char* dst = smem + buf * STAGE;
      const int pix0 = (range * stages + s) * 224;
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        if constexpr (STREAM) {
          const int piece = j * 4 + lw;  // 0..27: dY rows 8 piece.., 28..75: X window rows
          const bool isx = piece >= 28;
          const unsigned so = (unsigned)(((isx ? 50176 : 0) + pix0 + (isx ? piece - 28 : piece) * 8) * 1024 + (isx ? (tile & 7) : (tile >> 3)) * 128);
          BLDS16(rs, dst + (piece & 75) * 1024, voff, so);
        } else {
          BLDS16(rs, dst + ((j * 4 + lw) & 75) * 1024, voff, (j * 1024 + s * 64) & 0xffff);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      buf ^= 1;
    }
    return;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
  // lane's pixel of a K32 step: row q4 + 4 h (image row inside the 8-row tile), column col0 + g; 128-byte LDS row per pixel, chunk pair swizzled by row & 3
  const int a_lane = (q4 * 28 + g) * 128 + (p4 & 1) * 8, b_lane = DY_BYTES + (q4 * 32 + g) * 128 + (p4 & 1) * 8;
  const int cpa = wr * 2 + 0, cpb = wc * 2 + 0;  // chunk pair of fragment 0 (fragment i: + i)
  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  int buf = 0;
  for (int s = 0; s < stages; ++s) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll 1
    for (int k32 = 0; k32 < 7; ++k32) {
      u32x4 af[2], bf[2][2];  // bf[parity][fragment]
      auto read_a = [&](int i) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int sw = ((q4 + 4 * h) & 3);
          const u32x2 t = tr_read(st + a_lane + (4 * h * 28 + 4 * k32) * 128 + ((((cpa + i) ^ sw) & 3) << 5) + ((p4 >> 1) << 4));
          af[i][2 * h] = t[0];
          af[i][2 * h + 1] = t[1];
        }
      };
      auto read_b = [&](int tap, int par) {
        const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int sw = ((q4 + 4 * h + 2 * ty) & 3);
            const u32x2 t = tr_read(st + b_lane + ((4 * h + 2 * ty) * 32 + 4 * k32 + 2 * tx) * 128 + ((((cpb + j) ^ sw) & 3) << 5) + ((p4 >> 1) << 4));
            bf[par][j][2 * h] = t[0];
            bf[par][j][2 * h + 1] = t[1];
          }
      };
      read_a(0);
      read_a(1);
      read_b(0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        if (tap + 1 < 9) read_b(tap + 1, (tap + 1) & 1);  // next tap's X fragments in flight behind this tap's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) mfma(acc[tap][i][j], af[i], bf[tap & 1][j]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();
    buf ^= 1;
  }
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) sum += acc[t][i][j][0] + acc[t][i][j][3];
  out[blockIdx.x * 256 + tid] = sum;
}

// ---------------- today's mix (conv_wgrad_ws2_kernel) ----------------
template <int PIECES, bool STREAM = false>  // 12, or 0 = consumers alone
__global__ __launch_bounds__(512, 2) void probe_ws2(const unsigned char* __restrict__ src, unsigned src_bytes, float* __restrict__ out, int steps) {
  constexpr int RBG = 512, RBX = 256, G_BYTES = 64 * RBG, STAGE = G_BYTES + 64 * RBX;  // 32 + 16 KiB
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= 4) {
    const int lw = wave - 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
    const unsigned voff = (unsigned)(lane * 16 + lw * 4096 + (blockIdx.x & 63) * 16384);
    // STREAM: block = (cout tile of 256 [2], cin tile of 128 [4], tap [8 of 9], pixel range [4]); dY piece = 2 pixel rows x 512 B, X piece = 4 rows x 256 B
    const int tt = blockIdx.x & 63, range = blockIdx.x >> 6, tco = tt & 1, tci = (tt >> 1) & 3, tap = tt >> 3;
    const unsigned vg = (unsigned)((lane >> 5) * 1024 + (lane & 31) * 16), vx = (unsigned)((lane >> 4) * 1024 + (lane & 15) * 16);
    int slot = 0;
    for (int s = 0; s < steps; ++s) {
      unsigned char* dst = smem + slot * STAGE;
      const int pix0 = (range * steps + s) * 64;
#pragma unroll
      for (int j = 0; j < PIECES; ++j) {
        if constexpr (STREAM) {
          const int piece = j * 4 + lw;  // 0..31 dY (2 rows each), 32..47 X (4 rows each)
          if (piece < 32) BLDS16(rs, dst + piece * 1024, vg, (unsigned)((pix0 + piece * 2) * 1024 + tco * 512));
          else BLDS16(rs, dst + piece * 1024, vx, (unsigned)((50176 + pix0 + (piece - 32) * 4 + tap * 3) * 1024 + tci * 256));
        } else {
          BLDS16(rs, dst + (j * 4 + lw) * 1024, voff, (j * 1024 + s * 64) & 0xffff);
        }
      }
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      slot = slot == 2 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
  const int s3 = ((8 * g + q4) & 3) | (((8 * g + q4) >> 3 & 1) << 2);
  const int a_base = (8 * g + q4) * RBG + (p4 & 1) * 8 + (((wr * 16) + (p4 >> 1)) << 4);
  const int b_base = G_BYTES + (8 * g + q4) * RBX + (p4 & 1) * 8 + ((((wc ^ (s3 >> 2)) << 3) + (p4 >> 1)) << 4);
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 af[2][8], bf[2][4];
#pragma unroll
  for (int i = 0; i < 8; ++i) af[1][i] = u32x4{0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < 4; ++j) bf[1][j] = u32x4{0, 0, 0, 0};
  int slot = 0;
  for (int s = 0; s < steps; ++s) {
    const unsigned char* st = smem + slot * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
        for (int r = 3 * grp; r < 3 * grp + 3; ++r) {
          if (r < 16) {
            const int i = r >> 1, h = r & 1;
            const u32x2 t = tr_read(st + a_base + (kk * 32 + 4 * h) * RBG + ((i ^ s3) << 5));
            af[kk][i][2 * h] = t[0];
            af[kk][i][2 * h + 1] = t[1];
          } else {
            const int j = (r - 16) >> 1, h = r & 1;
            const u32x2 t = tr_read(st + b_base + (kk * 32 + 4 * h) * RBX + ((j ^ (s3 & 3)) << 5));
            bf[kk][j][2 * h] = t[0];
            bf[kk][j][2 * h + 1] = t[1];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 4 * grp; q < 4 * grp + 4; ++q) mfma(acc[q >> 2][q & 3], af[kk ^ 1][q >> 2], bf[kk ^ 1][q & 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    slot = slot == 2 ? 0 : slot + 1;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  out[blockIdx.x * 256 + tid] = sum;
}

template <typename K>
double time_it(K launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 / reps;
}

int main() {
  const unsigned src_bytes = 2u << 20;
  std::vector<unsigned short> h(src_bytes / 2);
  srand(7);
  for (auto& v : h) {
    const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u;
    memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  unsigned char* src;
  float* out;
  hipMalloc(&src, src_bytes);
  hipMalloc(&out, 256 * 256 * 4);
  hipMemcpy(src, h.data(), src_bytes, hipMemcpyHostToDevice);
  const int grid = 256, reps = 6;
  const int stages = 600, steps = 2400;
  // the streaming runs: dY and X of a 512 -> 512 3x3 layer at bs 64 ([50176][512] bf16 each, + slack for the window's halo rows), random contents
  const unsigned big_bytes = 2u * 50176u * 1024u + (4u << 20);
  unsigned char* big;
  hipMalloc(&big, big_bytes);
  for (unsigned off = 0; off < big_bytes; off += src_bytes) hipMemcpy(big + off, h.data(), (big_bytes - off < src_bytes ? big_bytes - off : src_bytes), hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_window<19, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 76 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws2<12, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 48 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_window<19>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 76 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_window<17>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 76 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_window<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 76 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws2<12>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 48 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe_ws2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 48 * 1024);
  for (int round = 0; round < 2; ++round) {
    const double fw = 2.0 * 64 * 64 * 9 * 224 * (double)stages * grid;
    double t = time_it([&] { hipLaunchKernelGGL((probe_window<19>), dim3(grid), dim3(512), 2 * 76 * 1024, 0, src, src_bytes, out, stages); }, reps);
    printf("window  64x64x9 taps, 19 pieces/loader/stage : %7.1f TFLOP/s\n", fw / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL((probe_window<17>), dim3(grid), dim3(512), 2 * 76 * 1024, 0, src, src_bytes, out, stages); }, reps);
    printf("window  64x64x9 taps, 17 pieces/loader/stage : %7.1f TFLOP/s\n", fw / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL((probe_window<0>), dim3(grid), dim3(512), 2 * 76 * 1024, 0, src, src_bytes, out, stages); }, reps);
    printf("window  consumers alone                      : %7.1f TFLOP/s\n", fw / t / 1e12);
    const double f2 = 2.0 * 256 * 128 * 64 * (double)steps * grid;
    t = time_it([&] { hipLaunchKernelGGL((probe_ws2<12>), dim3(grid), dim3(512), 3 * 48 * 1024, 0, src, src_bytes, out, steps); }, reps);
    printf("ws2     256x128 one tap, 12 pieces/loader/step: %7.1f TFLOP/s\n", f2 / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL((probe_ws2<0>), dim3(grid), dim3(512), 3 * 48 * 1024, 0, src, src_bytes, out, steps); }, reps);
    printf("ws2     consumers alone                       : %7.1f TFLOP/s\n", f2 / t / 1e12);
    // streaming: 64 tiles x 4 pixel ranges x 56 stages of 224 pixels (the whole layer) / 64 tile-taps x 4 ranges x 196 steps of 64 pixels (8 of its 9 taps)
    const double fws = 2.0 * 64 * 64 * 9 * 224 * 56.0 * grid, f2s = 2.0 * 256 * 128 * 64 * 196.0 * grid;
    t = time_it([&] { hipLaunchKernelGGL((probe_window<19, true>), dim3(grid), dim3(512), 2 * 76 * 1024, 0, big, big_bytes, out, 56); }, 20);
    printf("window  STREAMING the layer's tensors, 56 stages: %7.1f us  %7.1f TFLOP/s\n", t * 1e6, fws / t / 1e12);
    t = time_it([&] { hipLaunchKernelGGL((probe_ws2<12, true>), dim3(grid), dim3(512), 3 * 48 * 1024, 0, big, big_bytes, out, 196); }, 20);
    printf("ws2     STREAMING the layer's tensors, 196 steps: %7.1f us  %7.1f TFLOP/s\n", t * 1e6, f2s / t / 1e12);
  }
  return 0;
}
