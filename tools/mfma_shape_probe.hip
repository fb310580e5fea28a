// MFMA shape A/B for the conv consumers' inner loop ("measure, don't cite"): the SAME wave tile (128 pixels x 64
// couts, f32 accumulate), the SAME LDS traffic (every operand fragment re-read from LDS by ds_read_b128 each K-step of 32: 8 + 4 reads for
// v_mfma_f32_16x16x32_bf16, 2 x (4 + 2) for v_mfma_f32_32x32x16_bf16 -- LDS bytes per wave tile and K are independent of the MFMA shape,
// the 32x32 form only HALVES THE NUMBER OF MFMA INSTRUCTIONS, 16 instead of 32 per K-step of 32), reads interleaved with the MFMAs as
// in conv_igemm_ws2_kernel, random bf16 data, no global memory traffic in the loop.  One or two waves per SIMD, every CU busy.
// Reports TFLOP/s (wall), cycles per K-step (s_memtime) and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-result tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// LDS image: [rows][128 B] (64 bf16 of K per row), 16-byte chunk c of row r stored at chunk c ^ (r & 7): the conv kernels' image.
// 192 rows per wave (128 pixel rows + 64 cout rows), two K-halves of 32 per row.
template <int SHAPE, int WAVES>  // SHAPE 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(64 * WAVES) void probe(const unsigned short* __restrict__ src, float* __restrict__ out, unsigned long long* __restrict__ stamps,
                                                     int ksteps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned char* mine = smem + (wave & 3) * 192 * 128;  // the two waves of a SIMD read the same (read-only) image: 4 x 24 KiB of LDS
  for (int i = lane; i < 192 * 8; i += 64) {  // fill this wave's image with random bf16 (16 bytes per i)
    const u32x4 v = reinterpret_cast<const u32x4*>(src)[(blockIdx.x * 7 + wave * 3 + i) & 4095];
    *reinterpret_cast<u32x4*>(mine + i * 16) = v;
  }
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  if constexpr (SHAPE == 0) {
    const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int coff = (((g + 4 * kh) ^ sw) << 4);
        u32x4 xf[8], wf[4];
#pragma unroll
        for (int r = 0; r < 12; ++r) {  // read, 2-3 MFMAs of the PREVIOUS operands would go here in the real loop; here: reads first, then MFMAs interleaved by hipcc
          if (r < 4) wf[r] = *reinterpret_cast<const u32x4*>(mine + (128 + r * 16 + frow) * 128 + coff);
          else xf[r - 4] = *reinterpret_cast<const u32x4*>(mine + ((r - 4) * 16 + frow) * 128 + coff);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][3];
  } else {
    // 32x32x16: lane l holds A[row l & 31][k = 8 (l >> 5) .. +7]: a 16-byte chunk of the row's K-quarter; per K-step of 32 two MFMA K-slices
    const int frow = lane & 31, g2 = lane >> 5, sw = lane & 7;
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int ks = 0; ks < ksteps; ++ks) {
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) {  // K-quarter of 16 within the 64-wide row: chunks 2 kq, 2 kq + 1
        const int coff = (((2 * kq + g2) ^ sw) << 4);
        u32x4 xf[4], wf[2];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
          if (r < 2) wf[r] = *reinterpret_cast<const u32x4*>(mine + (128 + r * 32 + frow) * 128 + coff);
          else xf[r - 2] = *reinterpret_cast<const u32x4*>(mine + ((r - 2) * 32 + frow) * 128 + coff);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[j]), __builtin_bit_cast(bf16x8, xf[i]), acc[i][j], 0, 0, 0);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) sum += acc[i][j][0] + acc[i][j][15];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + tid] = sum;
  if (lane == 0) {
    stamps[(blockIdx.x * WAVES + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * WAVES + wave) * 2 + 1] = r1 - r0;
  }
}

template <int SHAPE, int WAVES>
void run(const unsigned short* src, float* out, unsigned long long* stamps, int ksteps, const char* name) {
  const int grid = 256, reps = 8;
  const size_t lds = (size_t)4 * 192 * 128;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<SHAPE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, out, stamps, ksteps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<SHAPE, WAVES>), dim3(grid), dim3(64 * WAVES), lds, 0, src, out, stamps, ksteps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(grid * WAVES * 2);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0, ticks = 0;
  for (int i = 0; i < grid * WAVES; ++i) { cyc += (double)h[2 * i]; ticks += (double)h[2 * i + 1]; }
  const double flops = 2.0 * 128 * 64 * 64 * (double)ksteps * WAVES * grid * reps;  // a K-step here = 64 deep (two halves / four quarters)
  printf("%-14s %d wave(s)/SIMD: %7.1f TFLOP/s wall | %6.1f cycles per 64-deep K-step per wave (ideal %d) | in-kernel clock %.2f GHz\n", name, WAVES / 4,
         flops / (ms * 1e-3) / 1e12, cyc / (grid * WAVES) / ksteps, SHAPE == 0 ? 64 * 16 * (WAVES / 4) : 32 * 32 * (WAVES / 4), cyc / ticks * 0.1);
}

int main() {
  std::vector<unsigned short> h(4096 * 8);
  srand(7);
  for (auto& v : h) {  // uniform [-1, 1) as bf16
    const float f = (float)rand() / RAND_MAX * 2.f - 1.f;
    unsigned u; memcpy(&u, &f, 4);
    v = (unsigned short)(u >> 16);
  }
  unsigned short* src; float* out; unsigned long long* stamps;
  hipMalloc(&src, h.size() * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 8 * 2 * 8);
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  const int ksteps = 20000;
  for (int round = 0; round < 2; ++round) {  // interleaved rounds in one process
    run<0, 4>(src, out, stamps, ksteps, "16x16x32");
    run<1, 4>(src, out, stamps, ksteps, "32x32x16");
    run<0, 8>(src, out, stamps, ksteps, "16x16x32");
    run<1, 8>(src, out, stamps, ksteps, "32x32x16");
  }
  return 0;
}
