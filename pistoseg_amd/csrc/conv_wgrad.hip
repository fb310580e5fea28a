// Weight gradient of the 1x1 / 3x3 convolutions on gfx950 MFMA.
//
//   dW[cout][tap][cin] += sum_{pixel m} dY[m][cout] * X[m@tap][cin]
//
// GEMM view per tap: rows = cout, cols = cin, K = produced pixels.  Both operands are channels-last, i.e.
// K is the SLOW index of both tiles: pixel rows (contiguous channels) are staged HBM -> LDS with 16-byte
// LDS-DMA exactly as they lie in memory, and the K-major MFMA fragments are produced by the gfx950
// transposed LDS read `ds_read_b64_tr_b16` (bf16) or by one ds_read_b32 per k (exact-f32 MFMA 16x16x4).
// The chunk index of every LDS row is XOR-swizzled (on the DMA source address) so the transposed reads of
// a 32-lane half hit 8 distinct 32-byte bank slices.
// Block = 4 waves (2x2), tile BCO x BCI in {64,128}^2, one tap and one pixel range (split-K) per block;
// partial sums are added to the f32 gradient with global_atomic_add_f32 (64-byte row segments).
#include <algorithm>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

#include "ps_internal.h"

namespace {


struct WgradArgs {
  const unsigned char* x;   // forward input  [n,h,w,cin]
  const unsigned char* dy;  // output gradient [n,ho,wo,cout]
  float* dw;                // [cout][taps][cin]
  int H, W, Ho, Wo, M;      // M = n*ho*wo
  int stride, dil, taps, ctr;
  int cin, cout;
  long long x_pix_bytes, dy_pix_bytes;
  int tiles_co, tiles_ci, splits, ksteps;  // ksteps = ceil(M / KP)
  FastDiv div_hw, div_w;
  unsigned x_bytes, dy_bytes;  // buffer extents (num_records)
  int dq, dp, dn;              // how (q, p, n) of a pixel advance when its index grows by KP
  int raster;                  // block order (see kernel)
  int ablate;                  // timing experiments only (results WRONG): 1 = no atomics, 2 = plain stores instead of atomics
  int nb, tpb;                 // persistent kernel: blocks per batch (#CUs), items per block (0 = one batch), see ps_block_items
  int reserved;                // host side only (ps_conv_geom.cus_reserved): CUs left to a co-running kernel
  int use_queue;               // host side only (ps_conv_geom.tile_queue)
  unsigned* queue;             // conv_wgrad_ws2_kernel<.., Q = true>: this launch's ticket counters (ps_queue_slot)
  int cig, cog;                // item order: cin / cout tiles per group of consecutive items (see decode_item)
  float* part;                 // deterministic mode (DET kernels): workspace [pixel range][cout][taps][cin] f32 for the partial sums
  long long part_stride;       // elements per pixel range = cout * taps * cin
  // conv_wgrad_ws2_kernel<.., XM = 3>: padding validity of the X rows as a precomputed table of LANE MASKS (wgrad_valid_table):
  // vtab[((tap * vperiod + ks % vperiod) * 4 + loader wave) * 4 + j] = 64-bit mask of the lanes whose row is inside the image for that tap
  const unsigned long long* vtab;
  int vperiod;
  // split 16-bit operands (PS_BF16X3 / PS_F16X3: 32-channel blocks [hi(32) | lo(32)] per pixel): which half of every block this launch reads.
  // The weight gradient contracts over PIXELS, so hi / lo cannot share a K-line as in the forward kernels: dW = x_hi dy_hi + x_hi dy_lo + x_lo dy_hi is
  // three launches of the 16-bit kernels, whose loaders gather the chosen halves (a per-lane source offset, fixed per item)
  int x_split, x_lo, dy_split, dy_lo;
};

// byte offset, inside a pixel's row, of the 8 channels c0 + 8 chunk .. of a 16-bit tensor (plain, or one half of a split one)
__device__ __forceinline__ unsigned wg_chunk_off(int c0, int chunk, int split, int lo) {
  const int ch = c0 + 8 * chunk;
  return split ? (unsigned)((((ch >> 5) << 6) + (ch & 31) + (lo << 5)) * 2) : (unsigned)(ch * 2);
}

struct WTraitsBF16 {
  static constexpr int ES = 2, KP = 64;
  [[maybe_unused]] static constexpr bool F16 = false;
};
struct WTraitsF16 {
  static constexpr int ES = 2, KP = 64;
  static constexpr bool F16 = true;
};
struct WTraitsF32 {
  static constexpr int ES = 4, KP = 32;
  [[maybe_unused]] static constexpr bool F16 = false;
};

// XOR applied to the 16-byte chunk index of LDS row R (row = RB bytes); see file header.
template <int ES, int RB>
__device__ __forceinline__ int row_swz(int R) {
  if constexpr (ES == 4) return (R & 1) << 2;
  else if constexpr (RB >= 256) return ((R & 3) << 1) | (((R >> 3) & 1) << 3);  // rows of 256 or 512 bytes: whole bank rows
  else return (((R >> 1) & 1) << 1) | (((R >> 3) & 1) << 2);
}

// LDS-DMA through a buffer descriptor (32-bit lane offset + scalar offset); out-of-range lanes (padding pixels,
// the tail beyond the last pixel) are ZERO-filled by the DMA (tools/probe_oob.hip).
#define BLDS16(rsrc, lptr, voff, soff)                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)
constexpr unsigned PAD_ROW = 0x80000000u;

// WS (wave-specialised): 8 waves; waves 4-7 only issue the LDS-DMA (and track the tap-shifted pixel addresses), waves
// 0-3 only do transposed LDS reads + MFMA, <= 128 VGPRs so two blocks stay resident per CU (same idea and measurement
// as conv_igemm_ws_kernel: issuing the DMAs costs a wave more issue time than its MFMAs).
// DET (deterministic mode, ps_conv2d_wgrad_det): the block STORES its partial sums into its pixel range's slice of the workspace
// (every element of a slice is written by exactly one block) instead of adding them to dw with atomics; wgrad_reduce_kernel adds the
// slices up in range order afterwards.
template <typename Tr, int BCO, int BCI, bool WS, bool DET = false>
__global__ __launch_bounds__(WS ? 512 : 256, WS ? 4 : 1) void conv_wgrad_kernel(const WgradArgs a) {
  constexpr int ES = Tr::ES, KP = Tr::KP;
  constexpr int RBG = BCO * ES, RBX = BCI * ES;           // LDS row bytes of the dY / X tiles
  constexpr int G_BYTES = KP * RBG, X_BYTES = KP * RBX, STAGE = G_BYTES + X_BYTES;
  constexpr int NIG = BCO / 32, NIX = BCI / 32;           // LDS-DMA instructions per wave per stage
  constexpr int RPG = 1024 / RBG, RPX = 1024 / RBX;       // rows covered by one wave instruction
  constexpr int MI = BCO / 32, NI = BCI / 32;             // 16x16 fragments per wave (2x2 waves)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = WS && wave_all >= 4;
  const int wave = wave_all & 3;  // index within the role
  // block -> (co tile, ci tile, tap, split)
  // Raster: all (ci tile, co tile, tap) blocks of ONE pixel range are consecutive, so the blocks resident together
  // share their dY chunk across ci tiles / taps and their X chunk across co tiles through L2 (order 1: legacy,
  // split fastest -- neighbours share nothing).
  int bid = blockIdx.x, split, tci, tco, tap;
  if (a.raster == 1) {
    split = bid % a.splits; bid /= a.splits;
    tci = bid % a.tiles_ci; bid /= a.tiles_ci;
    tco = bid % a.tiles_co; bid /= a.tiles_co;
    tap = bid;
  } else {
    tci = bid % a.tiles_ci; bid /= a.tiles_ci;
    tco = bid % a.tiles_co; bid /= a.tiles_co;
    tap = bid % a.taps; bid /= a.taps;
    split = bid;
  }
  const int co0 = tco * BCO, ci0 = tci * BCI;
  const int ty = a.taps == 1 ? a.ctr : tap / 3, tx = a.taps == 1 ? a.ctr : tap - (tap / 3) * 3;
  const int dy_off = (ty - a.ctr) * a.dil, dx_off = (tx - a.ctr) * a.dil;
  const int per = (a.ksteps + a.splits - 1) / a.splits;
  const int ks0 = split * per, ks1 = min(a.ksteps, ks0 + per);
  if (ks0 >= ks1) return;

  const int wr = wave >> 1, wc = wave & 1;

  // ---- staging.  dY rows are linear in the pixel index: constant lane offset + a scalar offset per K-step; rows past
  // the last pixel fall outside the descriptor and arrive as zeros.  X rows follow the tap shift: the lane tracks
  // (n, p, q) of its NIX pixels incrementally (no division in the loop).
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)a.x_bytes, 0x00020000);
  const int g_rowin = lane / (RBG / 16), g_pos = lane % (RBG / 16);
  const int x_rowin = lane / (RBX / 16), x_pos = lane % (RBX / 16);
  unsigned goff[NIG];
#pragma unroll
  for (int j = 0; j < NIG; ++j) {
    const int R = (wave * NIG + j) * RPG + g_rowin;
    if constexpr (ES == 2) goff[j] = (unsigned)(R * (int)a.dy_pix_bytes) + wg_chunk_off(co0, g_pos ^ row_swz<ES, RBG>(R), a.dy_split, a.dy_lo);
    else goff[j] = (unsigned)(R * (int)a.dy_pix_bytes + co0 * ES + ((g_pos ^ row_swz<ES, RBG>(R)) << 4));
  }
  int xn[NIX], xp[NIX], xq[NIX];
  unsigned xchunk[NIX];
#pragma unroll
  for (int j = 0; j < NIX; ++j) {
    const int R = (wave * NIX + j) * RPX + x_rowin;
    const uint32_t m = (uint32_t)(ks0 * KP + R);
    const uint32_t n = fdiv(m, a.div_hw);
    const uint32_t rem = m - n * a.div_hw.d;
    const uint32_t p = fdiv(rem, a.div_w);
    xn[j] = (int)n; xp[j] = (int)p; xq[j] = (int)(rem - p * a.div_w.d);
    if constexpr (ES == 2) xchunk[j] = wg_chunk_off(ci0, x_pos ^ row_swz<ES, RBX>(R), a.x_split, a.x_lo);
    else xchunk[j] = (unsigned)(ci0 * ES + ((x_pos ^ row_swz<ES, RBX>(R)) << 4));
  }
  const int n_img = a.M / (a.Ho * a.Wo);

  auto stage = [&](int buf, int ks) {
    unsigned char* sg = smem + buf * STAGE;
    unsigned char* sx = sg + G_BYTES;
    const int gso = ks * KP * (int)a.dy_pix_bytes;
#pragma unroll
    for (int j = 0; j < NIG; ++j) BLDS16(rsG, sg + (wave * NIG + j) * 1024, goff[j], gso);
#pragma unroll
    for (int j = 0; j < NIX; ++j) {
      const int y = xp[j] * a.stride + dy_off, xx = xq[j] * a.stride + dx_off;
      const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W && xn[j] < n_img;
      const unsigned off = ok ? (unsigned)(((xn[j] * a.H + y) * a.W + xx) * (int)a.x_pix_bytes) + xchunk[j] : PAD_ROW;
      BLDS16(rsX, sx + (wave * NIX + j) * 1024, off, 0);
      // advance this lane's pixel by KP
      int q = xq[j] + a.dq;
      const int c1 = q >= a.Wo;
      q -= c1 ? a.Wo : 0;
      int p = xp[j] + a.dp + c1;
      const int c2 = p >= a.Ho;
      p -= c2 ? a.Ho : 0;
      xq[j] = q; xp[j] = p; xn[j] += a.dn + c2;
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, l16 = lane & 15;
  if constexpr (WS) {
    if (loader) {
      stage(0, ks0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      for (int ks = ks0; ks < ks1; ++ks) {
        if (ks + 1 < ks1) stage(((ks - ks0) & 1) ^ 1, ks + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
    __builtin_amdgcn_s_barrier();
  } else {
    stage(0, ks0);
    __syncthreads();
  }
  for (int ks = ks0; ks < ks1; ++ks) {
    const int cur = (ks - ks0) & 1;
    if constexpr (!WS) {
      if (ks + 1 < ks1) stage(cur ^ 1, ks + 1);
    }
    const unsigned char* sg = smem + cur * STAGE;
    const unsigned char* sx = sg + G_BYTES;
    if constexpr (ES == 2) {
      // bf16: K = 32 pixels per MFMA; lane group g owns k = 8g..8g+7, fetched by two transposed reads
      const int q4 = l16 >> 2, p4 = l16 & 3;
#pragma unroll
      for (int kk = 0; kk < KP / 32; ++kk) {
        u32x4 af[MI], bf[NI];  // raw dwords, typed at the MFMA only (see conv_wgrad_ws2_kernel: typed assembly costs the F16 instantiation v_bfi re-packing)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int R = kk * 32 + 8 * g + 4 * h + q4;
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int c0 = wr * (BCO / 2) + i * 16;  // channel base inside the tile
            const int chunk = (c0 >> 3) + (p4 >> 1);
            const unsigned char* ad = sg + R * RBG + ((chunk ^ row_swz<ES, RBG>(R)) << 4) + (p4 & 1) * 8;
            const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ad)));
            af[i][2 * h] = t[0];
            af[i][2 * h + 1] = t[1];
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int c0 = wc * (BCI / 2) + j * 16;
            const int chunk = (c0 >> 3) + (p4 >> 1);
            const unsigned char* ad = sx + R * RBX + ((chunk ^ row_swz<ES, RBX>(R)) << 4) + (p4 & 1) * 8;
            const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ad)));
            bf[j][2 * h] = t[0];
            bf[j][2 * h + 1] = t[1];
          }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            // the transposed read moves 16-bit words; fp16 reinterprets the same registers
            if constexpr (Tr::F16)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, bf[j]), acc[i][j], 0, 0, 0);
            else
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
          }
      }
    } else {
      // exact f32: K = 4 pixels per MFMA; lane group g owns pixel row 4*kk + g
#pragma unroll
      for (int kk = 0; kk < KP / 4; ++kk) {
        const int R = kk * 4 + g;
        float af[MI], bf[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int c = wr * (BCO / 2) + i * 16 + l16;
          af[i] = *reinterpret_cast<const float*>(sg + R * RBG + (((c >> 2) ^ row_swz<ES, RBG>(R)) << 4) + (c & 3) * 4);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int c = wc * (BCI / 2) + j * 16 + l16;
          bf[j] = *reinterpret_cast<const float*>(sx + R * RBX + (((c >> 2) ^ row_swz<ES, RBX>(R)) << 4) + (c & 3) * 4);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
    }
    if constexpr (WS) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
    }
  }

  // D[row = cout: 4g + r][col = cin: l16]
  const long long wrow = (long long)a.taps * a.cin;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = co0 + wr * (BCO / 2) + i * 16 + 4 * g + r;
        const int ci = ci0 + wc * (BCI / 2) + j * 16 + l16;
        if constexpr (DET) a.part[(long long)split * a.part_stride + co * wrow + (long long)tap * a.cin + ci] = acc[i][j][r];
        else atomicAdd(a.dw + co * wrow + (long long)tap * a.cin + ci, acc[i][j][r]);
      }
}

// Second pass of the deterministic mode: dw[i] += part[0][i] + part[1][i] + ... in pixel-range order (one thread owns an element:
// the summation order is fixed, so the result does not depend on scheduling).  HBM-bound: reads `live` slices, updates dw once.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n4, int live, long long stride4) {
  const float4* p4 = reinterpret_cast<const float4*>(part);
  float4* d4 = reinterpret_cast<float4*>(dw);
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 s = p4[i];
    for (int k = 1; k < live; ++k) {
      const float4 v = p4[k * stride4 + i];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 d = d4[i];
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    d4[i] = d;
  }
}

// ------------------------------------------------------------------------------------------------
// Large-tile wave-specialised weight gradient: 256 cout x 128 cin tile of one tap, 8 waves = 4 CONSUMERS (2x2, each
// 128 cout x 64 cin: 8 x 4 fragments) + 4 LOADERS, ONE persistent block per CU (<= 256 VGPRs), 3-stage LDS ring of
// 64-pixel K-steps -- the same pipeline as conv_igemm_ws2_kernel:
//   * a quarter less LDS traffic per MFMA than the 128x128 tile (12 fragments per 32 MFMAs instead of 8 per 16);
//   * the consumers keep the fragments of the NEXT half K-step in flight across the block barrier (the MFMAs of the
//     previous half step, operands already in registers, run right behind the barrier);
//   * the loaders run two K-steps ahead (counted vmcnt);
//   * work items (tile, tap, pixel range) form one flat K-step sequence per block, so the next item's first stages are
//     prefetched while the consumers issue the (no-return) f32 atomics of the finished one, and those drain behind the
//     next item's main loop.
// 16-bit operands only (transposed LDS reads); cout % 256 == 0 and cin % 128 == 0 (every trainable backbone conv).
// ------------------------------------------------------------------------------------------------
// XM: how the loaders address the X rows (compile time: the loader waves are issue-bound -- with the general form's per-row pixel
// tracking, 25 VALU instructions per X row and K-step, ablating it was worth +5..13 % on the kernel, profiles/r02_wgrad_loader_ablation.txt;
// and a run-time switch here is worse than no switch: hipcc triplicated the issue code and the kernel lost 12 %)
//   0: any stride: (n, p, q) of the lane's pixel tracked incrementally, address rebuilt per K-step
//   1: stride 1 (output map = input map): the tap's row is the pixel's row shifted by a constant, i.e. a per-lane constant offset plus a
//      SCALAR offset per K-step; only the padding test keeps per-lane state (the shifted coordinates, 11 instructions per row)
//   2: 1x1, stride 1: no padding either -- constant lane offset + scalar offset, as for dY (no VALU work per K-step at all)
//   3: as 1, with the padding test looked up instead of tracked: whether pixel m's row is inside the image for a tap depends on m mod (H W),
//      and a lane's pixel advances by 64 per K-step, so the pattern repeats every P = H W / gcd(H W, 64) K-steps (49 for 28 x 28 and
//      56 x 56 maps, 16 / 64 for 32 x 32 / 64 x 64).  A host-built table holds, per (tap, K-step mod P, loader wave, DMA instruction), the
//      64-bit mask of valid LANES; a K-step costs one 32-byte scalar load (issued in front of the dY pieces, which hide its latency) and ONE
//      v_cndmask per X piece instead of 11 VALU instructions per row (4 instead of 44 per wave and K-step: the loaders' issue stream is
//      part of the K-step's critical path, NOTES 7.19-7.21)
// Q: items behind the block's first one come from the launch's ticket queue (a.queue; ps_internal.h).  Consumer wave 0 draws item s + 1 when item
// s starts -- right behind item s - 1's atomics in its memory queue -- and publishes it in front of the barrier of the item's K-step len - 6: the
// loaders cross into item s + 1 when they stage item s's last K-step, two K-steps ahead of the consumers.  As late as that allows: collecting
// the ticket is a vmcnt(0), i.e. it also waits for those atomics, which take many microseconds to drain while other blocks add to the same rows.
// The host selects Q only where every item has at least PS_WGRAD_QMIN K-steps.
constexpr int PS_WGRAD_QMIN = 32;
template <bool F16, int XM, bool DET = false, bool Q = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad_ws2_kernel(const WgradArgs a) {
  constexpr int ES = 2, KP = 64, BCO = 256, BCI = 128;
  constexpr int RBG = BCO * ES, RBX = BCI * ES;             // 512 / 256 bytes per LDS row
  constexpr int G_BYTES = KP * RBG, X_BYTES = KP * RBX, STAGE = G_BYTES + X_BYTES;  // 32 + 16 KiB
  constexpr int NIG = G_BYTES / 1024 / 4, NIX = X_BYTES / 1024 / 4;  // 8 + 4 LDS-DMA instructions per loader wave per step
  constexpr int RPG = 1024 / RBG, RPX = 1024 / RBX;         // 2 / 4 rows per instruction
  constexpr int MI = 8, NI = 4, NLD = NIG + NIX;
  static_assert(NLD == 12, "vmcnt literal below");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks that share an XCD (= an L2) get CONSECUTIVE items: with the pixel range slowest in the item order they then stream the
  // same dY / X pixel rows (measured before the remap: every XCD pulled every pixel range through its L2, ~1 GB per launch
  // on the memory side of L2 for a 100 MB working set)
  const int per = (a.ksteps + a.splits - 1) / a.splits;           // K-steps per pixel range
  const int live = (a.ksteps + per - 1) / per;                    // ranges that get work
  int first, G, nitems;  // this block's items: first, first + G, ... < nitems  (Q: first, then whatever the queue hands out)
  [[maybe_unused]] unsigned q_tk = 0;   // Q, consumer wave 0: the ticket in flight
  [[maybe_unused]] bool q_peek = false;  // Q (wave-uniform): whether the next draw also looks at the other classes' counters
  if constexpr (Q) {
    // EVERY item comes from the queue, the first one too (see conv_igemm_halo_kernel): drawn here, collected behind the waves' item-independent set-up
    G = 0;
    first = -1;
    nitems = a.tiles_co * a.tiles_ci * a.taps * live;
    q_peek = ps_q_count(nitems, blockIdx.x & 7) <= 64;
#ifndef PS_Q_STATIC_TICKETS
    if (wave_all == 0) ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_peek, q_tk);
#endif
  } else {
    ps_block_items(blockIdx.x, gridDim.x, a.tiles_co * a.tiles_ci * a.taps * live, a.nb, a.tpb, first, G, nitems);
  }
  [[maybe_unused]] const unsigned mbox = ps_q_mbox_addr(smem + 3 * STAGE);  // Q: the mailbox (16 bytes behind the ring)
  // item -> (ci tile, co tile, tap, pixel range), pixel range slowest (co-running blocks share their dY / X chunks in L2).  Inside a
  // range the order is [cig cin tiles][taps][cog cout tiles][other cin groups][other cout groups] (plan_wgrad_ws2): the ~32 consecutive
  // items that run together on one XCD then need few DISTINCT dY / X rows per K-step -- the taps of a (cin, cout) tile pair read the same dY
  // rows, and X rows that coincide up to a skew of a few K-steps (r03: L2-miss traffic of the 3x3 layers, profiles/r03_pmc_hbm_traffic.json).
  // One branch-free formula for every layer (a run-time SWITCH between orders made hipcc triplicate the loaders' issue code: -12 %, NOTES 7.19).
  auto decode = [&](int item, int& tci, int& tco, int& tap, int& ks0, int& ks1) {
    const int t_lo = item % a.cig; item /= a.cig;
    tap = item % a.taps; item /= a.taps;
    const int u_lo = item % a.cog; item /= a.cog;
    const int nci = a.tiles_ci / a.cig, nco = a.tiles_co / a.cog;
    tci = (item % nci) * a.cig + t_lo; item /= nci;
    tco = (item % nco) * a.cog + u_lo; item /= nco;
    ks0 = item * per;
    ks1 = min(a.ksteps, ks0 + per);
  };
  // total K-steps of this block's items (ranges are equal except the last); Q: not known in advance
  int total_steps = 0;
  if constexpr (!Q) {
    for (int it = first; it < nitems; it += G) {
      const int sp = it / (a.tiles_ci * a.tiles_co * a.taps);
      total_steps += min(a.ksteps, (sp + 1) * per) - sp * per;
    }
  }

  if (wave_all >= 4) {
    // ================= loader =================
    PS_LOADER_SETPRIO();
    const int wave = wave_all - 4;
    const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
    // XM >= 1: the descriptor starts `xpad` bytes BEFORE the tensor (the largest negative tap shift), so that scalar + lane offsets of
    // every row are non-negative; nothing in front of the tensor is ever read (those rows fail the padding test)
    const int xpad = (XM == 1 || XM == 3) ? (a.dil * a.W + a.dil) * (int)a.x_pix_bytes : 0;
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - xpad), 0, (int)a.x_bytes + xpad, 0x00020000);
    const int g_rowin = lane / (RBG / 16), g_pos = lane % (RBG / 16);
    const int x_rowin = lane / (RBX / 16), x_pos = lane % (RBX / 16);
    const int n_img = a.M / (a.Ho * a.Wo);
    unsigned goff[NIG], xchunk[NIX];
    int xn[XM == 3 ? 1 : NIX], xp[XM == 3 ? 1 : NIX], xq[XM == 3 ? 1 : NIX];
    int item = first, ks = 0, ks_end = 0, dy_off = 0, dx_off = 0;
    [[maybe_unused]] int vkk = 0;                                 // XM 3: ks mod vperiod
    [[maybe_unused]] const unsigned long long* vrow = nullptr;    // XM 3: this (tap, wave)'s masks of K-step 0; 16 masks (128 bytes) per K-step
    [[maybe_unused]] const unsigned pad_row = PAD_ROW;
    auto item_setup = [&](int it) {
      int tci, tco, tap;
      decode(it, tci, tco, tap, ks, ks_end);
      const int ty = a.taps == 1 ? a.ctr : tap / 3, tx = a.taps == 1 ? a.ctr : tap - (tap / 3) * 3;
      dy_off = (ty - a.ctr) * a.dil;
      dx_off = (tx - a.ctr) * a.dil;
#pragma unroll
      for (int j = 0; j < NIG; ++j) {
        const int R = (wave * NIG + j) * RPG + g_rowin;
        goff[j] = (unsigned)(R * (int)a.dy_pix_bytes) + wg_chunk_off(tco * BCO, g_pos ^ row_swz<ES, RBG>(R), a.dy_split, a.dy_lo);
      }
#pragma unroll
      for (int j = 0; j < NIX; ++j) {
        const int R = (wave * NIX + j) * RPX + x_rowin;
        if constexpr (XM != 2 && XM != 3) {
          const uint32_t m = (uint32_t)(ks * KP + R);
          const uint32_t n = fdiv(m, a.div_hw);
          const uint32_t rem = m - n * a.div_hw.d;
          const uint32_t p = fdiv(rem, a.div_w);
          xn[j] = (int)n; xp[j] = (int)p; xq[j] = (int)(rem - p * a.div_w.d);
          if constexpr (XM == 1) { xp[j] += dy_off; xq[j] += dx_off; }  // the SHIFTED coordinates are what is tracked and tested
        }
        xchunk[j] = wg_chunk_off(tci * BCI, x_pos ^ row_swz<ES, RBX>(R), a.x_split, a.x_lo);
        if constexpr (XM >= 1) xchunk[j] += (unsigned)(R * (int)a.x_pix_bytes);  // the lane's constant row offset
      }
      if constexpr (XM == 3) {
        vkk = ks % a.vperiod;
        vrow = a.vtab + ((long long)tap * a.vperiod * 4 + wave) * 4;
      }
    };
    if constexpr (Q) {  // the block's first item (published by consumer wave 0 in front of this barrier)
      __builtin_amdgcn_s_barrier();
      item = ps_q_mbox_read(mbox, 0);
      if (item < 0) return;
    }
    item_setup(item);
    int slot = 0, issued = 0;
    [[maybe_unused]] int l_seq = 0;        // Q: items this cursor has left behind
    [[maybe_unused]] bool l_done = false;  // Q: the last K-step of the block's last item has been staged
    // stages the next K-step of the flat sequence: exactly NLD loads per wave.  The caller guarantees issued < total_steps (the loaders'
    // issue stream is on the K-step's critical path, ~0.1 % of the kernel per scalar instruction: no per-step end test, no selector test in
    // the product build, NOTES 7.20)
    auto issue_next = [&]() {
      unsigned char* sg = smem + slot * STAGE;
      unsigned char* sx = sg + G_BYTES;
      const int gso = ks * KP * (int)a.dy_pix_bytes;
      const bool stage = !(PS_ABLATE(a.ablate) == 3 && issued >= 3);  // timing experiment (results WRONG): consumers run on stale LDS contents
      // XM 3: this K-step's four lane masks -- a uniform address, i.e. one scalar load, whose latency the dY pieces below cover
      unsigned long long vm[XM == 3 ? NIX : 1];
      if constexpr (XM == 3) {
        // (address made visibly wave-uniform and read through the constant address space: an SMEM load, counted by lgkmcnt -- a VECTOR load
        // here would sit in the loaders' in-order vmcnt queue in front of the dY pieces, and waiting for it would drain the previous step's)
        const unsigned long long addr = reinterpret_cast<unsigned long long>(vrow + (long long)vkk * 16);
        const unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)addr), ahi = __builtin_amdgcn_readfirstlane((unsigned)(addr >> 32));
        typedef const __attribute__((address_space(4))) unsigned long long* cptr;
        const cptr vp = (cptr)(((unsigned long long)ahi << 32) | alo);
#pragma unroll
        for (int j = 0; j < NIX; ++j) vm[j] = vp[j];
        vkk = (vkk + 1 == a.vperiod) ? 0 : vkk + 1;
      }
#pragma unroll
      for (int j = 0; j < NIG; ++j)
        if (stage) BLDS16(rsG, sg + (wave * NIG + j) * 1024, goff[j], gso);
      if constexpr (XM == 3) __builtin_amdgcn_sched_barrier(0);  // keep the masks' first use (and its lgkmcnt wait) behind all the dY pieces
      // scalar part of an X row's offset (XM >= 1): (first pixel of the K-step + the tap's shift) rows, from the padded descriptor base
      const int xso = XM >= 1 ? (ks * KP + dy_off * a.W + dx_off) * (int)a.x_pix_bytes + xpad : 0;
#pragma unroll
      for (int j = 0; j < NIX; ++j) {
        if constexpr (XM == 2) {
          if (stage) BLDS16(rsX, sx + (wave * NIX + j) * 1024, xchunk[j], xso);  // (rows past the last pixel: outside the descriptor -> zeros)
        } else if constexpr (XM == 3) {
          // lane offset where the lane's bit is set, the padding marker elsewhere: the uniform mask is used as the select's lane mask directly
          const unsigned off = __builtin_amdgcn_inverse_ballot_w64(vm[j]) ? xchunk[j] : pad_row;
          if (stage) BLDS16(rsX, sx + (wave * NIX + j) * 1024, off, xso);
        } else if constexpr (XM == 1) {
          const bool ok = (unsigned)xp[j] < (unsigned)a.H && (unsigned)xq[j] < (unsigned)a.W;  // (past the last image: outside the descriptor)
          if (stage) BLDS16(rsX, sx + (wave * NIX + j) * 1024, ok ? xchunk[j] : PAD_ROW, xso);
          int q = xq[j] + a.dq;  // advance the lane's (shifted) pixel by KP: wrap where the UNSHIFTED coordinate leaves the map
          const int c1 = q >= a.W + dx_off;
          q -= c1 ? a.W : 0;
          int p = xp[j] + a.dp + c1;
          const int c2 = p >= a.H + dy_off;
          p -= c2 ? a.H : 0;
          xq[j] = q; xp[j] = p;
        } else {
          const int y = xp[j] * a.stride + dy_off, xx = xq[j] * a.stride + dx_off;
          const bool ok = (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W && xn[j] < n_img;
          const unsigned off = ok ? (unsigned)(((xn[j] * a.H + y) * a.W + xx) * (int)a.x_pix_bytes) + xchunk[j] : PAD_ROW;
          if (stage) BLDS16(rsX, sx + (wave * NIX + j) * 1024, off, 0);
          int q = xq[j] + a.dq;  // advance this lane's pixel by KP
          const int c1 = q >= a.Wo;
          q -= c1 ? a.Wo : 0;
          int p = xp[j] + a.dp + c1;
          const int c2 = p >= a.Ho;
          p -= c2 ? a.Ho : 0;
          xq[j] = q; xp[j] = p; xn[j] += a.dn + c2;
        }
      }
      ++issued;
      slot = (slot == 2) ? 0 : slot + 1;
      if constexpr (Q) {
        if (++ks == ks_end) {
          item = ps_q_mbox_read(mbox, ++l_seq);
          if (item >= 0) item_setup(item);
          else l_done = true;
        }
      } else {
        if (++ks == ks_end && issued < total_steps) {
          item += G;
          item_setup(item);
        }
      }
    };
    // Steps 0 and 1, then one step per barrier while there are steps left to stage (the wait leaves the newest step's NLD pieces in
    // flight), then the drain: nothing to stage, everything must have landed.
    if constexpr (Q) {  // every block has an item, every item more than two K-steps
      issue_next();
      issue_next();
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // step 0 visible
      while (!l_done) {
        issue_next();
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_barrier();
      return;
    }
    if (total_steps > 0) issue_next();  // (a block without work stages nothing; it still meets the consumers' first barrier)
    if (total_steps > 1) {
      issue_next();
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // step 0 visible
    int gs = 0;
    for (; gs + 2 < total_steps; ++gs) {
      issue_next();  // step gs + 2; ring slot (gs+2)%3 was released by the previous barrier
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // step gs + 1 landed
      __builtin_amdgcn_s_barrier();
    }
    for (; gs < total_steps; ++gs) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================= consumer =================
  const int wave = wave_all;
  const int wr = wave >> 1, wc = wave & 1;
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
  // byte offsets of this lane's transposed reads inside a stage, for K-half kk and half-fragment h: row R = 32kk + 8g + 4h + q4.
  // row_swz only depends on R & 3 (= q4) and (R >> 3) & 1 (= g & 1), so one swizzle value serves every (kk, h).
  const int swz = row_swz<ES, RBG>(8 * g + q4);  // identical formula for both row widths
  // A fragment's 16-byte chunk index is (chunk ^ swz) with chunk = wr*16 + 2i + c resp. wc*8 + 2j + c (c = p4 >> 1) and swz confined to
  // bits 1..3, so (chunk ^ swz) << 4 = lane constant + ((i ^ s) << 5) resp. ((j ^ (s & 3)) << 5), s = swz >> 1: the fragment addresses are
  // rebuilt from two lane constants at every read (one xor + one shift-add each) instead of living in 12 registers -- the kernel
  // sits at the 256-VGPR limit, and ANY spill reload in this wave carries a vmcnt(0) that waits for the previous item's atomics.
  static_assert(ES == 2 && MI == 8 && NI == 4, "fragment address algebra below");
  const int s3 = swz >> 1;
  const int a_base = (8 * g + q4) * RBG + (p4 & 1) * 8 + (((wr * 16) + (p4 >> 1)) << 4);
  const int b_base = G_BYTES + (8 * g + q4) * RBX + (p4 & 1) * 8 + ((((wc ^ (s3 >> 2)) << 3) + (p4 >> 1)) << 4);

  // Fragments live in registers as raw dwords (u32x4 = eight 16-bit K values), typed only at the MFMA: assembled element by element as bf16
  // vectors and re-typed for the fp16 MFMA, hipcc re-packed every transposed read with v_bfi_b32 and waited for it right behind the read
  // (48 extra VALU + 33 extra s_waitcnt per K-step in the F16 instantiation: its weight gradients ran 18-30 % behind the bf16 ones)
  auto mfma1 = [&](f32x4& c, const u32x4& x, const u32x4& y) {
    if constexpr (F16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, x), __builtin_bit_cast(bf16x8, y), c, 0, 0, 0);
  };
  // One K-half (32 pixels): the 2 x (MI + NI) = 24 transposed reads of (stage st, K-half kk) into (afn, bfn), INTERLEAVED with the
  // MI x NI = 32 MFMAs on (afo, bfo), whose operands are already in registers: 3 reads, 4 MFMAs, 3 reads, 4 MFMAs, ...  A burst of
  // 24 LDS instructions in front of the MFMAs (the first version) left the matrix pipe idle for their issue time twice per
  // K-step -- this wave is its SIMD's only MFMA source: the consumers alone (loads ablated) ran at 1200-1360 TFLOP/s.  The order
  // is pinned with scheduling barriers (hipcc otherwise hoists every read to the top).
  auto half = [&](auto do_mma, const unsigned char* st, int kk, u32x4 (&afn)[MI], u32x4 (&bfn)[NI], const u32x4 (&afo)[MI],
                  const u32x4 (&bfo)[NI], f32x4 (&acc)[MI][NI]) {  // do_mma: std::true_type / std::false_type (an item's first half step)
    int z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));  // opaque zero: keeps the address arithmetic inside the loop (see above)
    const int sd = s3 + z;
    constexpr int NR = 2 * (MI + NI), NM = MI * NI;  // 24 reads, 32 MFMAs: groups of 3 reads + 4 MFMAs
    auto read1 = [&](int r) {  // r: 0..15 = A fragment r >> 1, half r & 1; 16..23 = B fragment (r - 16) >> 1, half (r - 16) & 1
      if (r < 2 * MI) {
        const int i = r >> 1, h = r & 1, roff = kk * 32 + 4 * h;
        const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(st + a_base + roff * RBG + ((i ^ sd) << 5))));
        afn[i][2 * h] = t[0];
        afn[i][2 * h + 1] = t[1];
      } else {
        const int j = (r - 2 * MI) >> 1, h = r & 1, roff = kk * 32 + 4 * h;
        const u32x2 t = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(st + b_base + roff * RBX + ((j ^ (sd & 3)) << 5))));
        bfn[j][2 * h] = t[0];
        bfn[j][2 * h + 1] = t[1];
      }
    };
#pragma unroll
    for (int grp = 0; grp < 8; ++grp) {
#pragma unroll
      for (int r = 3 * grp; r < 3 * grp + 3; ++r) read1(r);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(do_mma)::value) {
#pragma unroll
        for (int q = 4 * grp; q < 4 * grp + 4; ++q) mfma1(acc[q >> 2][q & 3], afo[q >> 2], bfo[q & 3]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    static_assert(NR == 24 && NM == 32, "interleave pattern");
  };
  auto mma = [&](f32x4 (&acc)[MI][NI], const u32x4 (&af)[MI], const u32x4 (&bf)[NI]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) mfma1(acc[i][j], af[i], bf[j]);
  };

  if constexpr (Q) {
    if (wave == 0) {
#ifdef PS_Q_STATIC_TICKETS
      ps_q_mbox_write(mbox, 0, (int)blockIdx.x < nitems ? ps_xcd_remap(blockIdx.x, gridDim.x) : -1);
#else
      ps_q_mbox_write(mbox, 0, ps_q_resolve(a.queue, blockIdx.x & 7, lane, 0, nitems, q_tk, q_peek));
#endif
    }
    __builtin_amdgcn_s_barrier();
    first = ps_q_mbox_read(mbox, 0);
    if (first < 0) {
#ifndef PS_Q_STATIC_TICKETS
      if (wave == 0) ps_q_block_done(a.queue, lane, gridDim.x);
#endif
      return;
    }
  }
  __builtin_amdgcn_s_barrier();  // step 0 visible
  int cur = 0;
  const long long wrow = (long long)a.taps * a.cin;
  [[maybe_unused]] int q_seq = 0;
  for (int item = first; Q ? item >= 0 : item < nitems; item = Q ? ps_q_mbox_read(mbox, ++q_seq) : item + G) {
    int tci, tco, tap, ks0, ks1;
    decode(item, tci, tco, tap, ks0, ks1);
    if constexpr (Q) {
#ifndef PS_Q_STATIC_TICKETS  // (diagnostic build: the queue's code paths fed with the static schedule, no atomics)
      if (wave == 0) ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_peek, q_tk);  // the ticket of this block's item q_seq + 1: in flight until K-step ks1 - 6
#endif
    }
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 af0[MI], bf0[NI], af1[MI], bf1[NI];
    {  // first K-step of the item (ks1 > ks0 always): nothing to overlap its first reads with
      const unsigned char* st = smem + cur * STAGE;
      half(std::false_type{}, st, 0, af0, bf0, af1, bf1, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      half(std::true_type{}, st, 1, af1, bf1, af0, bf0, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
    }
    for (int ks = ks0 + 1; ks < ks1; ++ks) {
      if constexpr (Q) {
#ifdef PS_Q_STATIC_TICKETS
        if (ks == ks1 - 6 && wave == 0) ps_q_mbox_write(mbox, q_seq + 1, item + (int)gridDim.x < nitems ? item + (int)gridDim.x : -1);
#else
        if (ks == ks1 - 6 && wave == 0) ps_q_mbox_write(mbox, q_seq + 1, ps_q_resolve(a.queue, blockIdx.x & 7, lane, G, nitems, q_tk, q_peek));
#endif
      }
      const unsigned char* st = smem + cur * STAGE;
      half(std::true_type{}, st, 0, af0, bf0, af1, bf1, acc);  // this step's first half is read behind the previous step's second-half MFMAs
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      half(std::true_type{}, st, 1, af1, bf1, af0, bf0, acc);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this ring slot may be refilled after the barrier
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
    }
    mma(acc, af1, bf1);
    // D fragment (i, j): lane (g, l16) holds rows 4g + r, column l16.  A 4x4 transpose between the lane-group index g and the
    // fragment index j (two half-wave swaps + two row swaps per register quartet) leaves register j' of lane (g', l16) with
    // row 4j' + r, column 16g' + l16 = the lane id: every atomic wave-instruction then adds ONE 256-byte contiguous row
    // segment (the shape the memory-side atomic units take at full rate) instead of four 64-byte pieces.
    // Buffer atomics: descriptor + scalar item offset + a 32-bit lane offset.  (No 64-bit VGPR address is live across the K loop: a
    // spilled one used to be reloaded at the top of every item, and the vmcnt(0) of that reload made the wave wait for all of the
    // previous item's atomics before it could start the next item's MFMAs.)
    // deterministic mode: the item's pixel range owns one slice of the workspace; plain stores (each element written by one block)
    float* const dst_base = DET ? a.part + (long long)(ks0 / per) * a.part_stride : a.dw;
    const __amdgpu_buffer_rsrc_t rs_dw = __builtin_amdgcn_make_buffer_rsrc((void*)dst_base, 0, (int)((long long)a.cout * wrow * 4), 0x00020000);
    const unsigned row_bytes = (unsigned)wrow * 4u;
    const unsigned item_off = (unsigned)(((long long)(tco * BCO + wr * 128) * wrow + (long long)tap * a.cin + tci * BCI + wc * 64) * 4);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uint32_t v0 = __float_as_uint(acc[i][0][r]), v1 = __float_as_uint(acc[i][1][r]);
        uint32_t v2 = __float_as_uint(acc[i][2][r]), v3 = __float_as_uint(acc[i][3][r]);
        auto s02 = __builtin_amdgcn_permlane32_swap(v0, v2, false, false);  // g bit 1 <-> j bit 1
        auto s13 = __builtin_amdgcn_permlane32_swap(v1, v3, false, false);
        v0 = s02[0]; v2 = s02[1]; v1 = s13[0]; v3 = s13[1];
        auto s01 = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);  // g bit 0 <-> j bit 0
        auto s23 = __builtin_amdgcn_permlane16_swap(v2, v3, false, false);
        const float o[4] = {__uint_as_float(s01[0]), __uint_as_float(s01[1]), __uint_as_float(s23[0]), __uint_as_float(s23[1])};
#pragma unroll
        for (int jp = 0; jp < 4; ++jp) {
          // (the item offset is part of the lane offset on purpose: an item-invariant offset would be hoisted out of the item
          // loop, 128 live registers)
          const unsigned voff = item_off + (unsigned)(i * 16 + 4 * jp + r) * row_bytes + (unsigned)lane * 4u;
          if constexpr (DET) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[jp]), rs_dw, voff, 0, 0);
          else if (PS_ABLATE(a.ablate) == 0) __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(o[jp], rs_dw, voff, 0, 0);
          else if (PS_ABLATE(a.ablate) == 2) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[jp]), rs_dw, voff, 0, 0);
          else asm volatile("" ::"v"(o[jp]));
        }
      }
  }
  if constexpr (Q) {
#ifndef PS_Q_STATIC_TICKETS
    if (wave == 0) ps_q_block_done(a.queue, lane, gridDim.x);  // (every draw of this block has returned: the last one was resolved in its last item)
#endif
  }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT, debug library only (r03; built, parity-green, measured SLOWER than the wave-specialised kernel above and therefore not
// selected: profiles/r03_wgrad256_vs_ws2.txt).  256 cout x 256 cin tile, all eight waves MFMA waves: the weight-gradient counterpart
// of conv_gemm256_kernel.
//   A SIMD's instruction issue -- MFMAs (8 cycles each), transposed LDS reads, LDS-DMA pieces (60-100 cycles each) -- is what
//   paces the wave-specialised kernel above: per 64-pixel K-step its consumer wave issues 64 MFMAs + 48 reads and its loader
//   partner 12 DMA pieces + the pixel tracking, ~2100 issue cycles for 1024 matrix-pipe cycles.  A 256 x 256 tile stages
//   64 KiB per K-step for twice the MACs: 8 pieces per wave, and both waves of a SIMD are MFMA waves.
//   * waves: grp = wave >> 2 picks the 128-cout half, wc = wave & 3 the 64-cin quarter: 8 x 4 fragments per wave, the ws2 consumers'
//     shape, fragment reads (ds_read_b64_tr_b16 on the K(pixel)-major images), swizzle and atomic epilogue.
//   * LDS: two K-step buffers of [64 pixel rows x 512 B of dY | 64 x 512 B of X].
//   * schedule: exactly conv_gemm256_kernel's -- two wave groups one barrier apart, per K-step and group four phases
//     L(q) | barrier | C(q) | barrier over the quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0) with A = dY fragments, B = X fragments;
//     a group stages row quarters {grp, grp + 2} (16 pixel rows = 8 pieces, two per wave) of dY(t+1) in L(q0), L(q1) and of X(t+2)
//     in L(q2), L(q3); every L ends with vmcnt(4) + lgkmcnt(0) before its barrier.  (All 64 rows of dY(t+1) are first read in
//     interval 8t + 8; the last quarter is staged in 8t + 3 and waited for in 8t + 7.)
//   * one work item (tile, tap, pixel range) per block, item order as decode_item; f32 atomics or, DET, plain stores into the range's
//     workspace slice.
// ------------------------------------------------------------------------------------------------
template <bool F16, int XM, bool DET = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad256_kernel(const WgradArgs a) {
  constexpr int ES = 2, KP = 64, RB = 512;                    // both LDS images: 64 pixel rows x 512 B (256 channels)
  constexpr int IMG = KP * RB, BUF = 2 * IMG;                 // 32 KiB per operand, 64 KiB per K-step
  constexpr int MI = 8, NI = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  const int per = (a.ksteps + a.splits - 1) / a.splits;
  // item -> (cin tile, cout tile, tap, pixel range); see launch_wgrad256 for the order
  int item = ps_xcd_remap(blockIdx.x, gridDim.x);
  const int t_lo = item % a.cig; item /= a.cig;
  const int tap = item % a.taps; item /= a.taps;
  const int u_lo = item % a.cog; item /= a.cog;
  const int ci_hi = item % (a.tiles_ci / a.cig); item /= (a.tiles_ci / a.cig);
  const int co_hi = item % (a.tiles_co / a.cog); item /= (a.tiles_co / a.cog);
  const int tci = ci_hi * a.cig + t_lo, tco = co_hi * a.cog + u_lo;
  const int ks0 = item * per, ks1 = min(a.ksteps, ks0 + per);
  const int NT = ks1 - ks0;
  if (NT <= 0) return;
  const int ty = a.taps == 1 ? a.ctr : tap / 3, tx = a.taps == 1 ? a.ctr : tap - (tap / 3) * 3;
  const int dy_off = (ty - a.ctr) * a.dil, dx_off = (tx - a.ctr) * a.dil;

  // ---- staging: this group's row quarters {grp, grp + 2} of both images; wave wc owns pieces 2 wc, 2 wc + 1 (two pixel rows each)
  const __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc((void*)a.dy, 0, (int)a.dy_bytes, 0x00020000);
  const int xpad = XM == 1 ? (a.dil * a.W + a.dil) * (int)a.x_pix_bytes : 0;  // see conv_wgrad_ws2_kernel
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.x - xpad), 0, (int)a.x_bytes + xpad, 0x00020000);
  const int rowin = lane >> 5, pos = lane & 31;
  // Pixel row of piece (h, p): R = 16 grp + 32 h + 4 wc + 2 p + rowin.  Its swizzle only depends on p (R & 3 = 2 p + rowin,
  // (R >> 3) & 1 = wc >> 1), so one lane offset per p serves both quarters; the quarter's 32 rows go into the SCALAR offset.
  const int R0 = 16 * grp + 4 * wc + rowin;
  unsigned goff[2], xoff[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int R = R0 + 2 * p;
    const unsigned sw = (unsigned)((pos ^ row_swz<ES, RB>(R)) << 4);
    static_assert(XM >= 1, "stride-1 layers only (the lane offset of an X row is a constant)");
    goff[p] = (unsigned)(R * (int)a.dy_pix_bytes + tco * 256 * ES) + sw;
    xoff[p] = (unsigned)(R * (int)a.x_pix_bytes + tci * 256 * ES) + sw;
  }
  auto stage_g = [&](int buf, int h, int ks) {  // dY rows of K-step ks (absolute), quarter grp + 2 h
    unsigned char* dst = smem + buf * BUF + ((grp + 2 * h) * 8 + 2 * wc) * 1024;
    const int gso = (ks * KP + 32 * h) * (int)a.dy_pix_bytes;
    BLDS16(rsG, dst, goff[0], gso);
    BLDS16(rsG, dst + 1024, goff[1], gso);
  };
  // X rows of K-step ks: the tap's row is the pixel's row shifted by a constant, i.e. lane constant + scalar offset from a descriptor that
  // starts (d W + d) pixels before the tensor; 3x3: rows whose shifted coordinates leave the image are padding (zero-filled) -- the test
  // is recomputed from the pixel index (two magic-number divisions per row: no per-lane state, this kernel has no registers to spare)
  auto stage_x = [&](int buf, int h, int ks) {
    unsigned char* dst = smem + buf * BUF + IMG + ((grp + 2 * h) * 8 + 2 * wc) * 1024;
    const int xso = (ks * KP + 32 * h + dy_off * a.W + dx_off) * (int)a.x_pix_bytes + xpad;
    bool ok[2] = {true, true};
    if constexpr (XM == 1) {  // ONE division pair per call: the second piece's rows are two pixels further along the image row
      const uint32_t m = (uint32_t)(ks * KP + 32 * h + R0);
      const uint32_t n = fdiv(m, a.div_hw);
      const uint32_t rem = m - n * a.div_hw.d;
      const uint32_t pp = fdiv(rem, a.div_w);
      int y = (int)pp + dy_off, xx = (int)(rem - pp * a.div_w.d) + dx_off;
      ok[0] = (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;  // (past the last image: outside the descriptor)
      xx += 2;
      const bool wrap = xx >= a.W + dx_off;  // the unshifted column left the row: first columns of the next row (of the next image: row 0)
      xx -= wrap ? a.W : 0;
      y += wrap ? 1 : 0;
      y -= (y >= a.H + dy_off) ? a.H : 0;
      ok[1] = (unsigned)y < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if constexpr (XM == 2) {
        BLDS16(rsX, dst + p * 1024, xoff[p], xso);  // (rows past the last pixel: outside the descriptor -> zeros)
      } else {
        BLDS16(rsX, dst + p * 1024, ok[p] ? xoff[p] : PAD_ROW, xso);
      }
    }
  };

  // ---- fragments (see conv_wgrad_ws2_kernel for the address algebra)
  const int g = lane >> 4, l16 = lane & 15, q4 = l16 >> 2, p4 = l16 & 3;
  const int s3 = row_swz<ES, RB>(8 * g + q4) >> 1;
  const int a_base = (8 * g + q4) * RB + (p4 & 1) * 8 + (((grp * 16) + (p4 >> 1)) << 4);
  const int b_base = IMG + (8 * g + q4) * RB + (p4 & 1) * 8 + ((((wc ^ (s3 >> 2)) << 3) + (p4 >> 1)) << 4);
  f32x4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[4][2];      // A0 or A1: 4 cout fragments x 2 K-halves
  bf16x8 bf0[2][2][2];  // B0 (cin fragments 0, 1) of the current and of the next K-step: [set][fragment][K-half]
  bf16x8 bf1[2][2];     // B1 (cin fragments 2, 3)
  // The transposed reads are INLINE ASM on purpose: through the builtin, hipcc cannot tell them from the LDS-DMA destinations of this
  // same wave and puts `s_waitcnt vmcnt(0)` in front of the first read after every barrier -- draining the two K-steps of prefetch
  // this schedule keeps in flight (measured r03: 937-1025 TFLOP/s with the builtin).  Ordering is this kernel's own: every load
  // phase ends with lgkmcnt(0) before its barrier (end_load), and no MFMA is scheduled above that point (sched_barrier).
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  // fragment i of A / j of B: lane address = base + ((i ^ s) << 5), rebuilt per fragment (one xor + one add per four reads) instead of
  // living in 12 registers -- the kernel sits at the 256-VGPR limit
  const unsigned a_lane = lds0 + (unsigned)a_base, b_lane = lds0 + (unsigned)b_base;
  const unsigned a_sw = (unsigned)s3 << 5, b_sw = (unsigned)(s3 & 3) << 5;
#define PS_TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
  auto read_a = [&](auto bufc, int sub) {
    constexpr int BO = decltype(bufc)::value * BUF;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 t[4];
      const unsigned ad = a_lane + (a_sw ^ (unsigned)((4 * sub + j) << 5)) + BO;  // (the 16-bit offset field cannot hold the second buffer's 64 KiB)
      PS_TR_READ(t[0], ad, 0 * RB);
      PS_TR_READ(t[1], ad, 4 * RB);
      PS_TR_READ(t[2], ad, 32 * RB);
      PS_TR_READ(t[3], ad, 36 * RB);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        af[j][0][e] = t[0][e]; af[j][0][4 + e] = t[1][e];
        af[j][1][e] = t[2][e]; af[j][1][4 + e] = t[3][e];
      }
    }
  };
  auto read_b = [&](auto bufc, int sub, bf16x8 (&b)[2][2]) {
    constexpr int BO = decltype(bufc)::value * BUF;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x4 t[4];
      const unsigned ad = b_lane + (b_sw ^ (unsigned)((2 * sub + j) << 5)) + BO;
      PS_TR_READ(t[0], ad, 0 * RB);
      PS_TR_READ(t[1], ad, 4 * RB);
      PS_TR_READ(t[2], ad, 32 * RB);
      PS_TR_READ(t[3], ad, 36 * RB);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        b[j][0][e] = t[0][e]; b[j][0][4 + e] = t[1][e];
        b[j][1][e] = t[2][e]; b[j][1][4 + e] = t[3][e];
      }
    }
  };
#undef PS_TR_READ
  auto mfma1 = [&](f32x4& c, const bf16x8& x, const bf16x8& y) {
    if constexpr (F16) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
  };
  auto mma_quadrant = [&](int msub, int nsub, const bf16x8 (&b)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) mfma1(acc[4 * msub + j][2 * nsub + i], af[j][kk], b[i][kk]);
    __builtin_amdgcn_s_setprio(0);
  };
  // End of a load phase.  The phase's LDS reads retire AFTER the barrier (lgkmcnt(0) in front of the first MFMA), so their latency
  // overlaps the barrier -- except where the NEXT interval already re-stages what was just read: G1's B1 reads of L(q1) (interval
  // 8t + 3) and G0's LDS-DMA of L(q2) (interval 8t + 4) into the same rows; there the reads are drained before the barrier.
  auto end_load = [&](bool staged, bool drain_reads) {
    if (staged) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (drain_reads) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto end_compute = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: dY(0), X(0), X(1)
  stage_g(0, 0, ks0);
  stage_g(0, 1, ks0);
  stage_x(0, 0, ks0);
  stage_x(0, 1, ks0);
  if (NT > 1) {
    stage_x(1, 0, ks0 + 1);
    stage_x(1, 1, ks0 + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_b(std::integral_constant<int, 0>{}, 0, bf0[0]);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: G1 runs one barrier behind G0

  auto kstep = [&](auto parity, int t) {
    constexpr int P = decltype(parity)::value;
    constexpr std::integral_constant<int, P> cb{};
    constexpr std::integral_constant<int, P ^ 1> nb{};
    const bool s1 = t + 1 < NT, s2 = t + 2 < NT;
    read_a(cb, 0);
    if (s1) stage_g(P ^ 1, 0, ks0 + t + 1);
    end_load(s1, false);
    mma_quadrant(0, 0, bf0[P]);
    end_compute();
    read_b(cb, 1, bf1);
    if (s1) stage_g(P ^ 1, 1, ks0 + t + 1);
    end_load(s1, grp == 1);
    mma_quadrant(0, 1, bf1);
    end_compute();
    read_a(cb, 1);
    if (s2) stage_x(P, 0, ks0 + t + 2);
    end_load(s2, false);
    mma_quadrant(1, 1, bf1);
    end_compute();
    if (s1) read_b(nb, 0, bf0[P ^ 1]);
    if (s2) stage_x(P, 1, ks0 + t + 2);
    end_load(s2, false);
    mma_quadrant(1, 0, bf0[P]);
    end_compute();
  };
  int t = 0;
  for (; t + 1 < NT; t += 2) {
    kstep(std::integral_constant<int, 0>{}, t);
    kstep(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t < NT) kstep(std::integral_constant<int, 0>{}, t);
  if (grp == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue: conv_wgrad_ws2_kernel's (4 x 4 lane <-> fragment transpose, one 256-byte row segment per wave instruction)
  const long long wrow = (long long)a.taps * a.cin;
  float* const dst_base = DET ? a.part + (long long)(ks0 / per) * a.part_stride : a.dw;
  const __amdgpu_buffer_rsrc_t rs_dw = __builtin_amdgcn_make_buffer_rsrc((void*)dst_base, 0, (int)((long long)a.cout * wrow * 4), 0x00020000);
  const unsigned row_bytes = (unsigned)wrow * 4u;
  const unsigned item_off = (unsigned)(((long long)(tco * 256 + grp * 128) * wrow + (long long)tap * a.cin + tci * 256 + wc * 64) * 4);
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t v0 = __float_as_uint(acc[i][0][r]), v1 = __float_as_uint(acc[i][1][r]);
      uint32_t v2 = __float_as_uint(acc[i][2][r]), v3 = __float_as_uint(acc[i][3][r]);
      auto s02 = __builtin_amdgcn_permlane32_swap(v0, v2, false, false);
      auto s13 = __builtin_amdgcn_permlane32_swap(v1, v3, false, false);
      v0 = s02[0]; v2 = s02[1]; v1 = s13[0]; v3 = s13[1];
      auto s01 = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);
      auto s23 = __builtin_amdgcn_permlane16_swap(v2, v3, false, false);
      const float o[4] = {__uint_as_float(s01[0]), __uint_as_float(s01[1]), __uint_as_float(s23[0]), __uint_as_float(s23[1])};
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {
        const unsigned voff = item_off + (unsigned)(i * 16 + 4 * jp + r) * row_bytes + (unsigned)lane * 4u;
        if constexpr (DET) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[jp]), rs_dw, voff, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(o[jp], rs_dw, voff, 0, 0);
      }
    }
}

#ifdef PS_DEBUG_HOOKS
constexpr bool kWgradDebugBuild = true;
#else
constexpr bool kWgradDebugBuild = false;
#endif
PS_TUNABLE g_wgrad_ws = 1;
PS_TUNABLE g_wgrad_ablate = 0;
PS_TUNABLE g_wgrad_ovh = 16;  // per-item overhead of the persistent kernel in K-step units (atomics + pipeline refill)
PS_TUNABLE g_wgrad_ws2 = 1;  // large-tile persistent kernel for 16-bit operands with cout % 256 == 0, cin % 128 == 0

PS_TUNABLE g_wgrad256 = 0;   // EXPERIMENT (debug library only; measured r03, profiles/r03_wgrad256_vs_ws2.txt: 3-15 % SLOWER than ws2 on every
                              // layer but 4096->4096): 256 x 256 tile kernel (cout, cin % 256 == 0, 16-bit): 0 off, 1 by shape, 2 whenever legal
PS_TUNABLE g_wgrad_raster = -1;  // -1: by shape (measured r01: pixel-range-slowest wins for 3x3 layers with >= 64 tiles)

// Geometry-derived fields + split-K choice of conv_wgrad_kernel; returns the number of pixel ranges that get work.
template <typename Tr, int BCO, int BCI>
long long plan_wgrad(WgradArgs& a) {
  a.tiles_co = a.cout / BCO;
  a.tiles_ci = a.cin / BCI;
  a.ksteps = (a.M + Tr::KP - 1) / Tr::KP;
  a.dq = Tr::KP % a.Wo;
  a.dp = (Tr::KP / a.Wo) % a.Ho;
  a.dn = Tr::KP / (a.Wo * a.Ho);
  a.raster = g_wgrad_raster >= 0 ? g_wgrad_raster : ((a.taps == 9 && (long long)a.tiles_co * a.tiles_ci * a.taps >= 64) ? 0 : 1);
  const long long tiles = (long long)a.tiles_co * a.tiles_ci * a.taps;
  // Split-K choice.  Two blocks are resident per CU; the launch takes as long as the busiest CU needs for its
  // ceil(blocks / 256) blocks of (K-steps per block + epilogue).  Pick the split count minimising that estimate
  // (ties go to fewer splits = fewer f32 atomics).
  const long long max_splits = (a.ksteps + 7) / 8 > 0 ? (a.ksteps + 7) / 8 : 1;  // at least 8 K-steps per block
  long long splits = 1, best = -1;
  for (long long sp = 1; sp <= max_splits && tiles * sp <= 16384; ++sp) {
    const long long per = (a.ksteps + sp - 1) / sp;
    const long long live = (a.ksteps + per - 1) / per;  // splits that actually get work
    const long long per_cu = (tiles * live + 255) / 256;  // blocks the busiest CU runs, two at a time
    // two co-resident blocks take ~1.4x the time of one alone (measured: they hide each other's latencies)
    const long long cost = ((per_cu / 2) * 14 + (per_cu % 2) * 10) * (per + 8);
    if (best < 0 || cost < best) { best = cost; splits = sp; }
  }
  a.splits = (int)splits;
  const long long per = (a.ksteps + splits - 1) / splits;
  return (a.ksteps + per - 1) / per;
}

template <typename Tr, int BCO, int BCI, bool DET = false>
int launch_wgrad(WgradArgs a, hipStream_t s) {
  plan_wgrad<Tr, BCO, BCI>(a);
  const long long tiles = (long long)a.tiles_co * a.tiles_ci * a.taps;
  const size_t lds = 2 * (size_t)Tr::KP * (BCO + BCI) * Tr::ES;
  if (g_wgrad_ws && BCO == 128 && BCI == 128) {
    hipLaunchKernelGGL((conv_wgrad_kernel<Tr, BCO, BCI, true, DET>), dim3((unsigned)(tiles * a.splits)), dim3(512), lds, s, a);
  } else {
    hipLaunchKernelGGL((conv_wgrad_kernel<Tr, BCO, BCI, false, DET>), dim3((unsigned)(tiles * a.splits)), dim3(256), lds, s, a);
  }
  PS_CHECK_LAUNCH("conv_wgrad");
  return PS_OK;
}

PS_TUNABLE g_wgrad_vtab = 1;  // 3x3 stride-1 layers: padding validity from the precomputed lane-mask table (XM = 3) instead of per-row tracking (XM = 1)

// The lane-mask table of conv_wgrad_ws2_kernel<.., XM = 3> for an H x W map and a dilation (device memory, built once per device and
// geometry -- a handful per network -- and kept for the life of the process: the per-device kernel / tuning cache the C-ABI allows; the
// one-time build copies synchronously).  nullptr when the validity pattern's period exceeds 128 K-steps (the kernel then tracks the
// rows itself, XM = 1).  Entry [tap][kk][wave][j]: bit l = the X row staged by lane l of loader `wave`'s j-th piece of K-step kk -- row
// R = (wave * 4 + j) * 4 + l / 16 of the step, pixel m = 64 kk + R, (p, q) = divmod(m mod (H W), W) -- is inside the image for `tap`.
static const unsigned long long* wgrad_valid_table(int H, int W, int dil, int* period) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int>, std::pair<const unsigned long long*, int>> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_tuple(dev, H, W, dil);
  auto it = cache.find(key);
  if (it != cache.end()) { *period = it->second.second; return it->second.first; }
  const long long hw = (long long)H * W;
  long long gcd = hw, b = 64;
  while (b) { const long long t = gcd % b; gcd = b; b = t; }
  const long long P = hw / gcd;
  const unsigned long long* dptr = nullptr;
  if (P <= 128) {
    std::vector<unsigned long long> host((size_t)9 * P * 16, 0ull);
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = (tap / 3 - 1) * dil, dx = (tap % 3 - 1) * dil;
      for (long long kk = 0; kk < P; ++kk)
        for (int w = 0; w < 4; ++w)
          for (int j = 0; j < 4; ++j) {
            unsigned long long mask = 0;
            for (int r = 0; r < 4; ++r) {
              const long long m = kk * 64 + (w * 4 + j) * 4 + r, rem = m % hw, p = rem / W, q = rem % W;
              if (p + dy >= 0 && p + dy < H && q + dx >= 0 && q + dx < W) mask |= 0xFFFFull << (16 * r);
            }
            host[(((size_t)tap * P + kk) * 4 + w) * 4 + j] = mask;
          }
    }
    void* d = nullptr;
    if (hipMalloc(&d, host.size() * 8) == hipSuccess) {
      if (hipMemcpy(d, host.data(), host.size() * 8, hipMemcpyHostToDevice) == hipSuccess) dptr = static_cast<const unsigned long long*>(d);
      else (void)hipFree(d);
    }
    (void)hipGetLastError();
  }
  cache[key] = {dptr, (int)P};
  *period = (int)P;
  return dptr;
}

// Geometry-derived fields + pixel-range count of conv_wgrad_ws2_kernel; returns the number of ranges that get work.
static long long plan_wgrad_ws2(WgradArgs& a) {
  a.tiles_co = a.cout / 256;
  a.tiles_ci = a.cin / 128;
  a.ksteps = (a.M + 63) / 64;
  a.dq = 64 % a.Wo;
  a.dp = (64 / a.Wo) % a.Ho;
  a.dn = 64 / (a.Wo * a.Ho);
  a.raster = 0;
  a.ablate = g_wgrad_ablate;
  const long long tiles = (long long)a.tiles_co * a.tiles_ci * a.taps;
  const int all_cus = ps_num_cus();
  const int ncu = a.reserved <= 0 ? all_cus : std::max(all_cus - a.reserved, all_cus / 4);  // (the split-K plan is made for the CUs the launch may count on)
  // Pixel-range count: one block per CU works through ceil(items / CUs) items of `per` K-steps each (+ ~6 K-steps' worth of
  // atomics and pipeline refill per item); ties go to fewer ranges (fewer atomics).
  long long splits = 1, best = -1;
  const long long max_splits = std::max<long long>(1, a.ksteps / 8);
  for (long long sp = 1; sp <= max_splits && tiles * sp <= 65536; ++sp) {
    const long long per = (a.ksteps + sp - 1) / sp, live = (a.ksteps + per - 1) / per;
    const long long rounds = (tiles * live + ncu - 1) / ncu;
    const long long cost = rounds * (per + g_wgrad_ovh);
    if (best < 0 || cost < best) { best = cost; splits = sp; }
  }
  a.splits = (int)splits;
  a.nb = ncu;
  // item order inside a pixel range (see the kernel's decode): an item stages 16 KiB of X (128 cins) and 32 KiB of dY (256 couts) per K-step;
  // 3x3: 4 cin tiles x 9 taps of ONE cout tile = 36 items sharing 64 + 32 KiB; 1x1: 8 cin x 4 cout tiles = 128 + 128 KiB
  auto pow2_div = [](int n, int cap) { int g = 1; while (g * 2 <= cap && n % (g * 2) == 0) g *= 2; return g; };
  a.cig = pow2_div(a.tiles_ci, a.taps == 9 ? 4 : 8);
  a.cog = pow2_div(a.tiles_co, std::max(1, 32 / (a.cig * a.taps)));
  const long long per = (a.ksteps + splits - 1) / splits;
  return (a.ksteps + per - 1) / per;
}

template <typename Tr, bool DET = false>
int launch_wgrad_ws2(WgradArgs a, hipStream_t s) {
  const long long live = plan_wgrad_ws2(a);
  const long long items = (long long)a.tiles_co * a.tiles_ci * a.taps * live;
  const unsigned grid = ps_persistent_grid(items, a.nb, a.tpb);
  const size_t lds = 3 * 64 * (256 + 128) * 2;
  // tile_queue: the stride-1 layers (XM 2 / 3), where there is something to hand out and every item (the last pixel range is the shortest) is
  // long enough to draw one item ahead
  if (a.use_queue && a.stride == 1 && items > a.nb) {
    const long long per = (a.ksteps + a.splits - 1) / a.splits, last = a.ksteps - (live - 1) * per;
    const bool vt = a.taps == 9 && g_wgrad_vtab && (a.vtab = wgrad_valid_table(a.H, a.W, a.dil, &a.vperiod)) != nullptr;
    if (std::min(per, last) >= PS_WGRAD_QMIN && (a.taps == 1 || vt)) {
      a.queue = ps_queue_slot(s);
      PS_REQUIRE(a.queue != nullptr, "wgrad: no ticket counters (hipMalloc failed)");
      if (a.taps == 1) hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 2, DET, true>), dim3((unsigned)a.nb), dim3(512), lds + 16, s, a);
      else hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 3, DET, true>), dim3((unsigned)a.nb), dim3(512), lds + 16, s, a);
      PS_CHECK_LAUNCH("conv_wgrad_ws2<queue>");
      return PS_OK;
    }
  }
  if (a.stride == 1 && a.taps == 1) hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 2, DET>), dim3(grid), dim3(512), lds, s, a);
  else if (a.stride == 1 && a.taps == 9 && g_wgrad_vtab && (a.vtab = wgrad_valid_table(a.H, a.W, a.dil, &a.vperiod)) != nullptr)
    hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 3, DET>), dim3(grid), dim3(512), lds, s, a);
  else if (a.stride == 1) hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 1, DET>), dim3(grid), dim3(512), lds, s, a);
  else hipLaunchKernelGGL((conv_wgrad_ws2_kernel<Tr::F16, 0, DET>), dim3(grid), dim3(512), lds, s, a);
  PS_CHECK_LAUNCH("conv_wgrad_ws2");
  return PS_OK;
}

// Geometry-derived fields + pixel-range count + item order of conv_wgrad256_kernel; returns the number of ranges that get work.
static long long plan_wgrad256(WgradArgs& a) {
  a.tiles_co = a.cout / 256;
  a.tiles_ci = a.cin / 256;
  a.ksteps = (a.M + 63) / 64;
  a.dq = 64 % a.Wo;
  a.dp = (64 / a.Wo) % a.Ho;
  a.dn = 64 / (a.Wo * a.Ho);
  a.raster = 0;
  a.ablate = 0;
  const long long tiles = (long long)a.tiles_co * a.tiles_ci * a.taps;
  const int ncu = ps_num_cus();
  // one block per item, one resident block per CU: rounds x (K-steps per item + prologue / epilogue in K-step units)
  long long splits = 1, best = -1;
  const long long max_splits = std::max<long long>(1, a.ksteps / 8);
  for (long long sp = 1; sp <= max_splits && tiles * sp <= 65536; ++sp) {
    const long long per = (a.ksteps + sp - 1) / sp, live = (a.ksteps + per - 1) / per;
    const long long rounds = (tiles * live + ncu - 1) / ncu;
    const long long cost = rounds * (per + g_wgrad_ovh);
    if (best < 0 || cost < best) { best = cost; splits = sp; }
  }
  a.splits = (int)splits;
  a.nb = ncu;
  // Item order: consecutive items = the blocks that run together on one XCD and stream a pixel range through its L2.  An item stages its
  // 256 couts of dY and 256 cins of X per K-step; the taps of one (cin, cout) tile pair read the SAME dY rows and X rows that coincide up
  // to a skew of a few K-steps, so taps vary fastest after a small group of cin tiles, then a group of cout tiles:
  //   [cig cin tiles][taps][cog cout tiles][remaining cin groups][remaining cout groups][pixel range]
  a.cig = a.taps == 9 ? (a.tiles_ci % 2 == 0 ? 2 : 1) : (a.tiles_ci % 4 == 0 ? 4 : (a.tiles_ci % 2 == 0 ? 2 : 1));
  const int want_co = std::max(1, 32 / (a.cig * a.taps));
  a.cog = 1;
  while (a.cog * 2 <= want_co && a.tiles_co % (a.cog * 2) == 0) a.cog *= 2;
  const long long per = (a.ksteps + splits - 1) / splits;
  return (a.ksteps + per - 1) / per;
}

template <typename Tr, bool DET = false>
int launch_wgrad256(WgradArgs a, hipStream_t s) {
#ifndef PS_DEBUG_HOOKS
  (void)a; (void)s;
  return PS_ERR_ARG;  // unreachable: use_wgrad256 is false in the product build
#else
  const long long live = plan_wgrad256(a);
  const long long items = (long long)a.tiles_co * a.tiles_ci * a.taps * live;
  const size_t lds = 2 * 2 * 64 * 512;
  if (a.taps == 1) hipLaunchKernelGGL((conv_wgrad256_kernel<Tr::F16, 2, DET>), dim3((unsigned)items), dim3(512), lds, s, a);
  else hipLaunchKernelGGL((conv_wgrad256_kernel<Tr::F16, 1, DET>), dim3((unsigned)items), dim3(512), lds, s, a);
  PS_CHECK_LAUNCH("conv_wgrad256");
  return PS_OK;
#endif
}

static bool use_wgrad256(int esize, long long M, int cout, int cin, int taps, int tpb, int stride, int W) {
  if (!kWgradDebugBuild) return false;  // the product library carries no kernel it cannot reach
  if (!g_wgrad256 || esize != 2 || cout % 256 != 0 || cin % 256 != 0 || tpb != 0 || stride != 1 || W < 2) return false;
  return g_wgrad256 > 1 || M * cout * cin * taps >= (1LL << 31);
}

static bool use_wgrad_ws2(int esize, long long M, int cout, int cin, int taps) {
  return g_wgrad_ws2 && esize == 2 && cout % 256 == 0 && cin % 128 == 0 && (g_wgrad_ws2 > 1 || M * cout * cin * taps >= (1LL << 31));
}

template <typename Tr, bool DET = false>
int dispatch_wgrad(const WgradArgs& a, hipStream_t s) {
  if constexpr (Tr::ES == 2) {
    if (!a.x_split && use_wgrad256(Tr::ES, a.M, a.cout, a.cin, a.taps, a.tpb, a.stride, a.W)) return launch_wgrad256<Tr, DET>(a, s);
    if (use_wgrad_ws2(Tr::ES, a.M, a.cout, a.cin, a.taps)) return launch_wgrad_ws2<Tr, DET>(a, s);
  }
  const bool co128 = a.cout % 128 == 0, ci128 = a.cin % 128 == 0;
  if (co128 && ci128) return launch_wgrad<Tr, 128, 128, DET>(a, s);
  if (co128) return launch_wgrad<Tr, 128, 64, DET>(a, s);
  if (ci128) return launch_wgrad<Tr, 64, 128, DET>(a, s);
  return launch_wgrad<Tr, 64, 64, DET>(a, s);
}

// Pixel ranges (split-K parts) the dispatcher will cut this problem into: the same plan functions the launchers use.
template <typename Tr>
long long wgrad_live_ranges(WgradArgs a) {
  if (!a.x_split && use_wgrad256(Tr::ES, a.M, a.cout, a.cin, a.taps, a.tpb, a.stride, a.W)) return plan_wgrad256(a);
  if (use_wgrad_ws2(Tr::ES, a.M, a.cout, a.cin, a.taps)) return plan_wgrad_ws2(a);
  const bool co128 = a.cout % 128 == 0, ci128 = a.cin % 128 == 0;
  if (co128 && ci128) return plan_wgrad<Tr, 128, 128>(a);
  if (co128) return plan_wgrad<Tr, 128, 64>(a);
  if (ci128) return plan_wgrad<Tr, 64, 128>(a);
  return plan_wgrad<Tr, 64, 64>(a);
}

int fill_wgrad_args(const ps_conv_geom* g, const void* x, const void* dy, float* dw, WgradArgs& a, const char* who) {
  PS_REQUIRE(g != nullptr, "%s: null geometry", who);
  PS_REQUIRE(ps_conv_supported(g), "%s: unsupported geometry (%s)", who, ps_last_error());
  const int es = ps_esize(g->dtype);
  a.x = static_cast<const unsigned char*>(x);
  a.dy = static_cast<const unsigned char*>(dy);
  a.dw = dw;
  a.H = g->h; a.W = g->w;
  a.Ho = (g->h - 1) / g->stride + 1; a.Wo = (g->w - 1) / g->stride + 1;
  a.M = g->n * a.Ho * a.Wo;
  a.stride = g->stride; a.dil = g->dilation;
  a.taps = g->ksize * g->ksize; a.ctr = g->ksize / 2;
  a.cin = g->cin; a.cout = g->cout;
  a.tpb = g->tiles_per_block;
  a.reserved = g->cus_reserved;
  a.use_queue = g->tile_queue;
  a.queue = nullptr;
  a.x_pix_bytes = (long long)g->ldc_x * es;
  a.dy_pix_bytes = (long long)g->ldc_y * es;
  a.div_hw = make_fastdiv((uint32_t)(a.Ho * a.Wo));
  a.div_w = make_fastdiv((uint32_t)a.Wo);
  const long long xb = (long long)g->n * g->h * g->w * a.x_pix_bytes, gb = (long long)a.M * a.dy_pix_bytes;
  PS_REQUIRE(xb < (1LL << 31) && gb < (1LL << 31), "%s: tensor larger than 2 GiB", who);
  a.x_bytes = (unsigned)xb;
  a.dy_bytes = (unsigned)gb;
  a.part = nullptr;
  a.part_stride = (long long)a.cout * a.taps * a.cin;
  a.vtab = nullptr;
  a.vperiod = 1;
  a.x_split = a.dy_split = ps_planes(g->dtype) == 2;
  a.x_lo = a.dy_lo = 0;
  return PS_OK;
}

// the 16-bit type whose kernels serve a geometry (split formats: their plane type)
int base_dtype(int dtype) { return dtype == PS_BF16X3 ? PS_BF16 : dtype == PS_F16X3 ? PS_F16 : dtype; }

long long live_ranges_of(const ps_conv_geom* g, const WgradArgs& a) {
  const int dt = base_dtype(g->dtype);
  if (dt == PS_BF16) return wgrad_live_ranges<WTraitsBF16>(a);
  if (dt == PS_F16) return wgrad_live_ranges<WTraitsF16>(a);
  return wgrad_live_ranges<WTraitsF32>(a);
}

// the launches of one weight gradient: one for the plain types; for the split ones x_hi dy_hi and, with wgrad_terms = 3, x_hi dy_lo and x_lo dy_hi
// (all accumulate into dw).  Default: the hi halves only.  A lo term is 2^-8 (bf16) / 2^-11 (fp16) of its product and their sum over the pixels
// is incoherent: measured on the CPU oracle's gradients, the two lo launches move a flip-free step's weight gradient by 1.8e-4 (fp16x3; bf16x3:
// 3.2e-3 -> 3.5e-3, inside its own noise) and a real-size step's not at all (n = 24: 3.287e-4 vs 3.296e-4, the ReLU-boundary flips dominate) --
// for 3 x the weight-gradient time (and the halves are gathered in 64-byte segments: 1.6 x slower per launch than contiguous rows).
template <typename F>
int for_each_plane_pair(const ps_conv_geom* g, WgradArgs& a, F&& launch) {
  if (ps_planes(g->dtype) == 1) return launch(a);
  static const int pairs[3][2] = {{0, 0}, {0, 1}, {1, 0}};
  const int terms = g->wgrad_terms == 3 ? 3 : 1;  // ps_conv_geom.wgrad_terms: 0 (default) | 1 = x_hi dy_hi only, 3 = + x_hi dy_lo + x_lo dy_hi
  int k = 0;
  for (const auto& pr : pairs) {
    if (k++ >= terms) break;
    a.x_lo = pr[0];
    a.dy_lo = pr[1];
    if (int rc = launch(a)) return rc;
  }
  return PS_OK;
}

}  // namespace

#ifdef PS_DEBUG_HOOKS
extern "C" void ps_debug_set_wgrad_ws(int v) { g_wgrad_ws = v; }
extern "C" void ps_debug_set_wgrad_ws2(int v) { g_wgrad_ws2 = v; }
extern "C" void ps_debug_set_wgrad256(int v) { g_wgrad256 = v; }
extern "C" void ps_debug_set_wgrad_ablate(int v) { g_wgrad_ablate = v; }
extern "C" void ps_debug_set_wgrad_ovh(int v) { g_wgrad_ovh = v; }
extern "C" void ps_debug_set_wgrad_raster(int v) { g_wgrad_raster = v; }
extern "C" void ps_debug_set_wgrad_vtab(int v) { g_wgrad_vtab = v; }
void ps_debug_reset_wgrad(void) {
  g_wgrad_ws = 1; g_wgrad_ablate = 0; g_wgrad_ovh = 16; g_wgrad_ws2 = 1; g_wgrad256 = 0; g_wgrad_raster = -1; g_wgrad_vtab = 1;
}
#endif

extern "C" int ps_conv_wgrad_variant(const ps_conv_geom* g) {
  if (!g || !ps_conv_supported(g)) return -1;
  const long long ho = (g->h - 1) / g->stride + 1, wo = (g->w - 1) / g->stride + 1;
  if (ps_planes(g->dtype) == 1 &&
      use_wgrad256(ps_esize(g->dtype), (long long)g->n * ho * wo, g->cout, g->cin, g->ksize * g->ksize, g->tiles_per_block, g->stride, g->w))
    return 2;
  return use_wgrad_ws2(ps_esize(g->dtype), (long long)g->n * ho * wo, g->cout, g->cin, g->ksize * g->ksize) ? 1 : 0;
}

extern "C" int ps_conv2d_wgrad(const ps_conv_geom* g, const void* x, const void* dy, float* dw, void* stream) {
  PS_REQUIRE(g && x && dy && dw, "conv2d_wgrad: null argument");
  PS_REQUIRE(ps_aligned16(x) && ps_aligned16(dy) && ps_aligned16(dw), "conv2d_wgrad: misaligned pointer");
  WgradArgs a{};
  if (int rc = fill_wgrad_args(g, x, dy, dw, a, "conv2d_wgrad")) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int dt = base_dtype(g->dtype);
  return for_each_plane_pair(g, a, [&](const WgradArgs& b) -> int {
    if (dt == PS_BF16) return dispatch_wgrad<WTraitsBF16>(b, s);
    if (dt == PS_F16) return dispatch_wgrad<WTraitsF16>(b, s);
    return dispatch_wgrad<WTraitsF32>(b, s);
  });
}

extern "C" int64_t ps_conv2d_wgrad_det_workspace_bytes(const ps_conv_geom* g) {
  WgradArgs a{};
  if (fill_wgrad_args(g, nullptr, nullptr, nullptr, a, "conv2d_wgrad_det_workspace_bytes") != PS_OK) return -1;
  const long long live = live_ranges_of(g, a);
  return live <= 1 ? 0 : (int64_t)(live * a.part_stride * 4);
}

extern "C" int ps_conv2d_wgrad_det(const ps_conv_geom* g, const void* x, const void* dy, float* dw, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  PS_REQUIRE(g && x && dy && dw, "conv2d_wgrad_det: null argument");
  PS_REQUIRE(ps_aligned16(x) && ps_aligned16(dy) && ps_aligned16(dw), "conv2d_wgrad_det: misaligned pointer");
  WgradArgs a{};
  if (int rc = fill_wgrad_args(g, x, dy, dw, a, "conv2d_wgrad_det")) return rc;
  const long long live = live_ranges_of(g, a);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int dt = base_dtype(g->dtype);
  if (live <= 1) {
    // one pixel range: every element of dw receives exactly ONE addition per launch, in launch order -- the atomic kernels as they are
    return for_each_plane_pair(g, a, [&](const WgradArgs& b) -> int {
      if (dt == PS_BF16) return dispatch_wgrad<WTraitsBF16>(b, s);
      if (dt == PS_F16) return dispatch_wgrad<WTraitsF16>(b, s);
      return dispatch_wgrad<WTraitsF32>(b, s);
    });
  }
  const long long need = live * a.part_stride * 4;
  PS_REQUIRE(workspace && ps_aligned16(workspace) && workspace_bytes >= need,
             "conv2d_wgrad_det: workspace of %lld bytes needed (ps_conv2d_wgrad_det_workspace_bytes), got %lld at %p", need,
             (long long)workspace_bytes, workspace);
  PS_REQUIRE(a.part_stride * 4 < (1LL << 31), "conv2d_wgrad_det: weight tensor larger than 2 GiB");
  a.part = static_cast<float*>(workspace);
  // (split types: each of the three launches fills the workspace and is reduced into dw before the next one reuses it)
  return for_each_plane_pair(g, a, [&](const WgradArgs& b) -> int {
    int rc;
    if (dt == PS_BF16) rc = dispatch_wgrad<WTraitsBF16, true>(b, s);
    else if (dt == PS_F16) rc = dispatch_wgrad<WTraitsF16, true>(b, s);
    else rc = dispatch_wgrad<WTraitsF32, true>(b, s);
    if (rc != PS_OK) return rc;
    const long long n4 = b.part_stride / 4;  // cin is a multiple of 32: whole float4s
    const unsigned blocks = (unsigned)std::min<long long>((n4 + 255) / 256, 8LL * ps_num_cus());
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, s, b.part, dw, n4, (int)live, n4);
    PS_CHECK_LAUNCH("wgrad_reduce");
    return (int)PS_OK;
  });
}
