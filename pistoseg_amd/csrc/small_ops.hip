// HBM-bound kernels of the segmentation hot path: conv1a (K = 27 direct conv), the fc8 class head and its
// backward, weight layout transforms, and the flat-arena optimisers.  wave64 everywhere; 16-byte accesses.
#include <math.h>

#include <algorithm>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

#include "ps_internal.h"

namespace {

// ------------------------------------------------------------------------------------------------
// conv1a: NCHW f32 image -> channels-last activation, on the exact-f32 MFMA (16x16x4): the 27-tap im2col row of a
// pixel is the B operand (lane (g = lane>>4, col = lane&15) fetches taps k = 4s + g of pixel col straight from the
// image, zero outside), the [64][27] weights are the A operand held in registers for the whole kernel; 7 k-steps x 4
// cout fragments per 16 pixels.  Same f32 arithmetic for both output dtypes; store-bound (64 channels per pixel).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1a_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     T* __restrict__ out_act, T* __restrict__ out_raw, int n, int h, int wd) {
  const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
  // A operand: lane supplies W[cout(f, col)][k = 4s + g]; cout(f, rho) = 16*(rho>>2) + 4f + (rho&3), so accumulator
  // register r of fragment f in lane group g is channel 16g + 4f + r (16 contiguous channels per lane)
  float wa[7][4];
#pragma unroll
  for (int sidx = 0; sidx < 7; ++sidx)
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const int k = 4 * sidx + g, co = 16 * (col >> 2) + 4 * f + (col & 3);
      wa[sidx][f] = k < 27 ? w[co * 27 + k] : 0.f;
    }
  float sc[16], sh[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    sc[i] = scale ? scale[16 * g + i] : 1.f;
    sh[i] = shift ? shift[16 * g + i] : 0.f;
  }
  // this lane's 7 taps: k = 4s + g -> (c, ky, kx)
  int toff[7], tdy[7], tdx[7];
#pragma unroll
  for (int sidx = 0; sidx < 7; ++sidx) {
    const int k = 4 * sidx + g, c = k / 9, r = k - 9 * c, ky = r / 3, kx = r - 3 * ky;
    tdy[sidx] = k < 27 ? ky - 1 : (1 << 20);  // out-of-range tap -> never valid
    tdx[sidx] = kx - 1;
    toff[sidx] = c * h * wd;
  }
  const long long total = (long long)n * h * wd;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  for (long long p0 = wave_id * 16; p0 < total; p0 += nwaves * 16) {
    const long long pix = p0 + col;
    const bool live = pix < total;
    const int img = live ? (int)(pix / ((long long)h * wd)) : 0;
    const int rem = live ? (int)(pix - (long long)img * h * wd) : 0;
    const int y = rem / wd, xx = rem - y * wd;
    const float* xb = x + (long long)img * 3 * h * wd;
    f32x4 acc[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int sidx = 0; sidx < 7; ++sidx) {
      const int yy = y + tdy[sidx], xs = xx + tdx[sidx];
      float v = 0.f;
      if (live && yy >= 0 && yy < h && xs >= 0 && xs < wd) v = xb[toff[sidx] + yy * wd + xs];
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[sidx][f], v, acc[f], 0, 0, 0);
    }
    if (!live) continue;
    float v[16];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * f + r] = acc[f][r];
    if (out_raw) {
      ps_store8<T>(out_raw + pix * 64 + 16 * g, v);
      ps_store8<T>(out_raw + pix * 64 + 16 * g + 8, v + 8);
    }
    if (out_act) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i] * sc[i] + sh[i], 0.f);
      ps_store8<T>(out_act + pix * 64 + 16 * g, v);
      ps_store8<T>(out_act + pix * 64 + 16 * g + 8, v + 8);
    }
  }
}

// conv1a for the 16-bit paths: ONE 16x16x32 MFMA per 16 pixels x 16 couts (K = 27 taps padded to 32) with operands rounded to the
// storage type, instead of 7 exact-f32 16x16x4 MFMAs: the f32 form is matrix-pipe-bound (28 x 32 cycles per 16 pixels, 330 us at
// bs = 64), this one leaves only the 64-channel store stream.  Lane (g, col) supplies k = 8g .. 8g+7 of weight row cout(f, col)
// and of pixel col's im2col row.
template <typename T, bool F16, bool RAW, bool ACT>  // RAW / ACT: which outputs exist (compile time: see the loop)
__global__ __launch_bounds__(256) void conv1a_lowp_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          T* __restrict__ out_act, T* __restrict__ out_raw, int n, int h, int wd,
                                                          const FastDiv div_hw, const FastDiv div_w) {
  const int lane = threadIdx.x & 63, g = lane >> 4, col = lane & 15;
  typedef T t8 __attribute__((ext_vector_type(8)));
  t8 wa[4];
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    const int co = 16 * (col >> 2) + 4 * f + (col & 3);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * g + e;
      wa[f][e] = static_cast<T>(k < 27 ? w[co * 27 + k] : 0.f);
    }
  }
  // Registers decide how many waves hide this kernel's memory latency (it was 140 -> 3 waves per SIMD): the BN scale / shift rows
  // (32 values per lane) live in LDS and are re-read in every group's epilogue, and a tap's validity is 6 flag bits, not two shifts.
  __shared__ __attribute__((aligned(16))) float s_aff[2][64];
  if (threadIdx.x < 64) {
    s_aff[0][threadIdx.x] = scale ? scale[threadIdx.x] : 1.f;
    s_aff[1][threadIdx.x] = shift ? shift[threadIdx.x] : 0.f;
  }
  __syncthreads();
  // this lane's 8 taps k = 8g + e -> (c, ky, kx): byte offset relative to the centre pixel of channel 0, and flags naming the pixel
  // conditions under which the tap is padding: bit 0 top row, 1 bottom row, 2 first column, 3 last column, 4 always (k >= 27),
  // 5 pixel past the end of the batch
  int tapoff[8];
  unsigned long long tapflags = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * g + e, c = k / 9, r = k - 9 * c, ky = r / 3, kx = r - 3 * ky;
    tapoff[e] = (c * h * wd + (ky - 1) * wd + (kx - 1)) * 4;
    const unsigned f = (ky == 0 ? 1u : 0u) | (ky == 2 ? 2u : 0u) | (kx == 0 ? 4u : 0u) | (kx == 2 ? 8u : 0u) | (k >= 27 ? 16u : 0u) | 32u;
    tapflags |= (unsigned long long)f << (8 * e);
  }
  // (pixel indices and image byte offsets fit 31 bits: checked by the launcher.  The decomposition pixel -> (image, y, x) is two
  // magic-number divisions, and the taps are fetched through a buffer descriptor with 32-bit offsets: an invalid tap -- zero padding,
  // k >= 27, a pixel past the end -- gets an offset outside the descriptor and arrives as 0.0 without a select.  Written with `/` on a
  // 64-bit index, 64-bit addresses and a validity mask the gather was ~35 VALU instructions per tap.)
  const int total = n * h * wd;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, total * 12, 0x00020000);
  // The NEXT 16 pixels' taps are loaded before this group's stores are issued: vmcnt is one in-order queue, a load issued behind the
  // stores would wait for them.
  float raw[8];
  auto gather = [&](int p0) {
    const int pix = p0 + col;
    const bool live = pix < total;
    const int img = (int)fdiv((uint32_t)pix, div_hw);
    const int rem = pix - img * h * wd;
    const int y = (int)fdiv((uint32_t)rem, div_w), xx = rem - y * wd;
    const int centre = (img * 3 * h * wd + rem) * 4;
    const unsigned pm = (y == 0 ? 1u : 0u) | (y == h - 1 ? 2u : 0u) | (xx == 0 ? 4u : 0u) | (xx == wd - 1 ? 8u : 0u) | 16u | (live ? 0u : 32u);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned bad = (unsigned)(tapflags >> (8 * e)) & pm;  // branch-free: a padding tap gets an offset outside the descriptor
      raw[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsx, (centre + tapoff[e]) | (bad ? (int)0x80000000 : 0), 0, 0));
    }
  };
  // One group of 16 pixels.  FULL groups store unconditionally and the outputs are template flags, so that the number of stores behind
  // the next group's loads is a compile-time constant and the wait at the top of the next iteration is a counted `vmcnt(stores)`:
  // with run-time `if (out_raw)` / `if (live)` around the stores hipcc could only write `vmcnt(0)` there, i.e. every iteration
  // waited for its own stores to be acknowledged.  Only the last group of the launch can be partial; it takes the predicated copy.
  // (Store shape: a lane writes the two 16-byte chunks it holds of its pixel's 128-byte row.  Swapping chunks between lanes so that an
  // instruction writes eight WHOLE rows -- 1 KiB contiguous -- was measured twice and is 50 % slower: 237 vs 156 us.)
  auto group = [&](auto full, int p0, int pnext) {
    t8 bv;
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = static_cast<T>(raw[e]);
    gather(pnext);
    f32x4 acc[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      acc[f] = f32x4{0, 0, 0, 0};
      if constexpr (F16) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wa[f]), __builtin_bit_cast(f16x8, bv), acc[f], 0, 0, 0);
      else acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[f]), __builtin_bit_cast(bf16x8, bv), acc[f], 0, 0, 0);
    }
    const long long pix = (long long)p0 + col;
    const bool live = decltype(full)::value || pix < total;
    float v[16];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * f + r] = acc[f][r];
    if constexpr (RAW) {
      if (live) {
        ps_store8<T>(out_raw + pix * 64 + 16 * g, v);
        ps_store8<T>(out_raw + pix * 64 + 16 * g + 8, v + 8);
      }
    }
    if constexpr (ACT) {
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));  // opaque zero: keeps the 32 affine values out of registers across the loop
      const float4* af = reinterpret_cast<const float4*>(&s_aff[0][16 * g + z]);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 a = af[i], b = af[16 + i];
        v[4 * i + 0] = fmaxf(v[4 * i + 0] * a.x + b.x, 0.f);
        v[4 * i + 1] = fmaxf(v[4 * i + 1] * a.y + b.y, 0.f);
        v[4 * i + 2] = fmaxf(v[4 * i + 2] * a.z + b.z, 0.f);
        v[4 * i + 3] = fmaxf(v[4 * i + 3] * a.w + b.w, 0.f);
      }
      if (live) {
        ps_store8<T>(out_act + pix * 64 + 16 * g, v);
        ps_store8<T>(out_act + pix * 64 + 16 * g + 8, v + 8);
      }
    }
  };
  const int stride = nwaves * 16;
  int p0 = wave_id * 16;
  gather(p0);
  auto next_of = [&](int p) { return p + stride < total ? p + stride : p; };  // (the last group re-reads itself)
  if (p0 + 16 <= total) {
    // first group peeled: the loop header then sees the same queue (8 loads, then this group's stores) from the entry as from the
    // back edge -- merged with an entry that has no stores behind the loads, the wait would again be vmcnt(0)
    group(std::true_type{}, p0, next_of(p0));
    for (p0 += stride; p0 + 16 <= total; p0 += stride) group(std::true_type{}, p0, next_of(p0));
  }
  if (p0 < total) group(std::false_type{}, p0, p0);
}

// ------------------------------------------------------------------------------------------------
// fc8 forward: cam[m,c] = sum_k x[m,k]*drop[n,k]*w[c,k].  One wave per 8 pixels; a lane owns 8 channels
// of every 512-channel chunk (weights for the chunk live in registers across the 8 pixels).
// ------------------------------------------------------------------------------------------------
constexpr int FC8_MAXC = 8;
template <typename T, int C>  // C = number of classes (compile time: the per-class accumulators and weights live in registers)
__global__ __launch_bounds__(256) void fc8_fwd_kernel(const T* __restrict__ x, int ldc, const float* __restrict__ w, int ldw,
                                                      const float* __restrict__ bias, const float* __restrict__ drop,
                                                      float* __restrict__ cam, int accumulate, int M, int ppi, int K, int ppw) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int mw = (blockIdx.x * 4 + wave) * ppw, mwe = min(M, mw + ppw);  // this wave's pixels, eight at a time
  for (int m0 = mw; m0 < mwe; m0 += 8) {
    float acc[8][C];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[p][c] = 0.f;
    // the waves start at different 512-channel chunks (and wrap): in lockstep they would all read the same 1-KiB slice of every
    // 8-KiB pixel row at the same time, i.e. hammer a subset of the HBM channels
    const int nchunks = K / 512, rot = (blockIdx.x * 4 + wave) % nchunks;
    for (int j = 0; j < nchunks; ++j) {
      const int kc = j + rot < nchunks ? j + rot : j + rot - nchunks;
      const int k0 = kc * 512 + lane * 8;
      float wv[C][8], dva[8], dvb[8];
#pragma unroll
      for (int c = 0; c < C; ++c) ps_load8<float>(w + (long long)c * ldw + k0, wv[c]);
      // the eight pixel rows are loaded back to back (rows past the end of the range re-read its last row: their sums are never
      // written); the dropout multipliers change per image only: those of the first and of the last pixel's image cover the group
      // unless the maps are smaller than a group (then per pixel)
      const int na = m0 / ppi, nb = min(m0 + 7, mwe - 1) / ppi;
      if (drop) {
        ps_load8<float>(drop + (long long)na * K + k0, dva);
        ps_load8<float>(drop + (long long)nb * K + k0, dvb);
      }
      PsRaw8<T> xr[8];
#pragma unroll
      for (int p = 0; p < 8; ++p) xr[p].load(x + (long long)min(m0 + p, mwe - 1) * ldc + k0);
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        float xv[8];
        xr[p].unpack(xv);
        if (drop) {
          const int n = min(m0 + p, mwe - 1) / ppi;
          if (nb - na > 1 && n != na && n != nb) {  // wave-uniform, maps smaller than 8 pixels only
            float dv[8];
            ps_load8<float>(drop + (long long)n * K + k0, dv);
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] *= dv[i];
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] *= (n == na ? dva[i] : dvb[i]);
          }
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[p][c] = fmaf(xv[i], wv[c][i], acc[p][c]);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float s = ps_wave_sum(acc[p][c]);
        if (lane == 0 && m0 + p < mwe) {
          float* o = cam + (long long)(m0 + p) * C + c;
          *o = (accumulate ? *o : 0.f) + s + (bias ? bias[c] : 0.f);
        }
      }
    }
  }
}

// Resident 256-thread blocks per CU of a kernel (from its register use), queried once per instantiation: the fc8 kernels size
// their per-wave / per-block pixel ranges so that the grid is a whole number of full rounds (at the training shape the fixed
// 8 / 64 pixels gave 2.04 rounds, i.e. three).
// Cached per KERNEL POINTER (instantiations that share a function-pointer type share one generic-lambda body, so a `static` inside the
// lambda would hand the first variant's occupancy to the others) -- a handful of entries, looked up linearly under a mutex.
template <typename Kern>
static int ps_blocks_per_cu(Kern kernel) {
  static std::mutex mu;
  static std::vector<std::pair<const void*, int>> cache;
  const void* key = reinterpret_cast<const void*>(kernel);
  std::lock_guard<std::mutex> lock(mu);
  for (const auto& e : cache)
    if (e.first == key) return e.second;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess || n < 1) n = 1;
  cache.emplace_back(key, n);
  return n;
}

// fc8 backward (+ ReLU(bn7) mask): a thread owns 8 channels and walks a pixel range, four pixels per group (the kernel is a pure
// stream: read x, write dx).
//   * Groups alternate between two register sets and the NEXT group's loads are issued before the current group's stores: vmcnt
//     counts loads and stores in one in-order queue, so loads issued behind stores could not be consumed before those stores were
//     acknowledged by L2.  The loop body is branch-free (rows past the end are clamped: they re-store the last row and contribute
//     nothing to dw), so the compiler's vmcnt bookkeeping stays exact.
//   * A block's pixel range spans at most two images (the host picks pix_per_block <= pixels per image): their dropout multipliers
//     are loaded once and selected per pixel.
//   * dw: every block holds a [C][2048] partial sum.  With a workspace the partials are stored ([pixel block][C][K], plain stores)
//     and summed by fc8_dw_reduce_kernel; without one they go to dw with atomics -- 9.6 M float atomics on 12 K addresses at the
//     training shape, measured r01: 270 of the kernel's 455 us.
template <typename T, int C, bool DROP>
__global__ __launch_bounds__(256) void fc8_bwd_kernel(const T* __restrict__ x, int ldc, const float* __restrict__ w,
                                                      const float* __restrict__ drop, const float* __restrict__ scale7,
                                                      const float* __restrict__ dcam, T* __restrict__ dx, int ldc_dx,
                                                      float* __restrict__ dw, float* __restrict__ partial, int M, int ppi, int K,
                                                      int pix_per_block) {
  const int kblocks = K / 2048;
  const int kb = blockIdx.x % kblocks, mb = blockIdx.x / kblocks;
  const int k0 = kb * 2048 + threadIdx.x * 8;
  const int ma = mb * pix_per_block, me = min(M, ma + pix_per_block);
  float wv[C][8], gw[C][8], s7[8], dva[8], dvb[8];
#pragma unroll
  for (int c = 0; c < C; ++c) {
#pragma unroll
    for (int i = 0; i < 8; ++i) gw[c][i] = 0.f;
    ps_load8<float>(w + (long long)c * K + k0, wv[c]);
  }
  ps_load8<float>(scale7 + k0, s7);
  const int na = ma / ppi;  // image of the first pixel; pixels of the block lie in image na or na + 1
  if constexpr (DROP) {
    ps_load8<float>(drop + (long long)na * K + k0, dva);
    ps_load8<float>(drop + (long long)((me - 1) / ppi) * K + k0, dvb);
  }
  constexpr int U = 4;
  struct Group {
    PsRaw8<T> x[U];
    float dc[U][C];  // the pixels' dcam rows (same address in every lane; loaded with the group so that they share its wait)
  };
  auto issue = [&](Group& gr, int m0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(m0 + u, me - 1);
      gr.x[u].load(x + (long long)m * ldc + k0);
#pragma unroll
      for (int c = 0; c < C; ++c) gr.dc[u][c] = dcam[(long long)m * C + c];
    }
  };
  auto process = [&](const Group& gr, int m0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int m = min(m0 + u, me - 1);
      const float live = (m0 + u < me) ? 1.f : 0.f;
      float xv[8], dv[8], g[8];
      gr.x[u].unpack(xv);
      const bool first = m / ppi == na;
#pragma unroll
      for (int i = 0; i < 8; ++i) dv[i] = DROP ? (first ? dva[i] : dvb[i]) : 1.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float d = gr.dc[u][c];
        const float dl = d * live;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          g[i] = fmaf(d, wv[c][i], g[i]);
          gw[c][i] = fmaf(dl, xv[i] * dv[i], gw[c][i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) g[i] = xv[i] > 0.f ? g[i] * dv[i] * s7[i] : 0.f;
      ps_store8<T>(dx + (long long)m * ldc_dx + k0, g);
    }
  };
  // (loads past the end of the range are clamped to its last row.  The first group is peeled so that the loop header sees the same
  // queue shape -- loads, stores, loads -- from the entry and from the back edge, and the waits stay vmcnt(N > 0).)
  Group ga, gb;
  issue(ga, ma);
  issue(gb, ma + U);
  process(ga, ma);
  issue(ga, ma + 2 * U);
  for (int m0 = ma + U; m0 < me; m0 += 2 * U) {
    process(gb, m0);
    issue(gb, m0 + 2 * U);
    if (m0 + U >= me) break;
    process(ga, m0 + U);
    issue(ga, m0 + 3 * U);
  }
  if (partial) {
#pragma unroll
    for (int c = 0; c < C; ++c) ps_store8<float>(partial + ((long long)mb * C + c) * K + k0, gw[c]);
  } else {
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) atomicAdd(dw + (long long)c * K + k0 + i, gw[c][i]);
    }
  }
}

// dw[j] += sum over the pixel blocks b of partial[b][j], j < C*K, in a FIXED order (deterministic: no atomics).  A block of 256 threads
// owns 16 groups of four consecutive j; thread (slice, jq) sums slice `slice` of the pixel blocks for group jq in block order, the 16
// slice sums of a group meet in LDS and one thread adds them up in slice order and updates dw (the only writer of those four elements).
__global__ __launch_bounds__(256) void fc8_dw_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int ck, int nblocks,
                                                            int per_slice) {
  __shared__ float4 part[16][16];
  const int jq = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int j = (blockIdx.x * 16 + jq) * 4;
  const int b0 = slice * per_slice, b1 = min(nblocks, b0 + per_slice);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (j < ck) {
    for (int b = b0; b < b1; ++b) {
      const float4 v = *reinterpret_cast<const float4*>(partial + (long long)b * ck + j);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[slice][jq] = acc;
  __syncthreads();
  if (slice == 0 && j < ck) {
    float4 t = part[0][jq];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float4 v = part[k][jq];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    float4 d = *reinterpret_cast<float4*>(dw + j);
    d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
    *reinterpret_cast<float4*>(dw + j) = d;
  }
}

// ------------------------------------------------------------------------------------------------
// weight layout: dst[cin][tap][cout] = src[cout][tap][cin]  (32x32 LDS tile transpose per tap)
// ------------------------------------------------------------------------------------------------
template <typename S>
__device__ __forceinline__ float to_f32(S v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return ps_bf16_to_f32(v); }
template <typename D>
__device__ __forceinline__ D from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ uint16_t from_f32<uint16_t>(float v) { return ps_f32_to_bf16(v); }

struct h16 { uint16_t bits; };  // fp16 storage tag (uint16_t is taken by bf16 here)
template <>
__device__ __forceinline__ float to_f32<h16>(h16 v) { return ps_f16_to_f32(v.bits); }
template <>
__device__ __forceinline__ h16 from_f32<h16>(float v) { return h16{ps_f32_to_f16(v)}; }

template <typename S, typename D>
__global__ __launch_bounds__(256) void weight_transpose_kernel(const S* __restrict__ src, D* __restrict__ dst, int cout, int taps,
                                                               int cin) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < cout && ci < cin) ? to_f32<S>(src[((long long)co * taps + tap) * cin + ci]) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout) dst[((long long)ci * taps + tap) * cout + co] = from_f32<D>(tile[tx][r]);
  }
}

// Many weight tensors in one launch (the per-step refresh of every trainable conv's data-gradient layout was 33 launches of a
// few microseconds of work each): the items ride in the kernel arguments, a block finds its item by scanning the tile prefix sums.
// dst rows may be wider than cout (dst_ld): two tensors transposed side by side form the K-concatenated weights of a fused unit.
constexpr int WT_MAX_ITEMS = PS_WT_MAX_ITEMS;
struct WtBatch {
  ps_wt_item it[WT_MAX_ITEMS];
  int tile_end[WT_MAX_ITEMS];
  int n;
};
// 64 x 64 tiles with 16-byte global accesses when both sides are the same 16-bit type and the shapes allow (every backbone conv);
// 32 x 32 element-wise tiles otherwise.  The host's tile prefix sums use the same rule (wt_fast).
__host__ __device__ static inline bool wt_fast(const ps_wt_item& it, bool same16) {
  return same16 && it.cin % 64 == 0 && it.cout % 64 == 0 && it.dst_ld % 8 == 0 && (reinterpret_cast<uintptr_t>(it.src) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(it.dst) & 15) == 0;
}
template <typename S, typename D>
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const WtBatch b) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 66 * 2 > 32 * 33 * 4 ? 64 * 66 * 2 : 32 * 33 * 4];
  int k = 0;
  while (k + 1 < b.n && (int)blockIdx.x >= b.tile_end[k]) ++k;
  const ps_wt_item it = b.it[k];
  const int t = blockIdx.x - (k ? b.tile_end[k - 1] : 0);
  constexpr bool same16 = sizeof(S) == 2 && sizeof(D) == 2;
  if constexpr (same16) {
    if (wt_fast(it, true)) {
      uint16_t (*tile)[66] = reinterpret_cast<uint16_t (*)[66]>(lds);  // [co][ci], 132-byte rows: column reads hit distinct banks
      const int nx = it.cin / 64, ny = it.cout / 64;
      const int tap = t / (nx * ny), r2 = t - tap * nx * ny;
      const int ci0 = (r2 % nx) * 64, co0 = (r2 / nx) * 64;
      const uint16_t* src = static_cast<const uint16_t*>(it.src);
      uint16_t* dst = static_cast<uint16_t*>(it.dst);
      const int row = threadIdx.x >> 3, ch = threadIdx.x & 7;  // 32 rows x 8 chunks of 8 elements per pass
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int co = row + 32 * p;
        const uint4 q = *reinterpret_cast<const uint4*>(src + ((long long)(co0 + co) * it.taps + tap) * it.cin + ci0 + ch * 8);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<uint32_t*>(&tile[co][ch * 8 + 2 * e]) = w[e];
      }
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int ci = row + 32 * p;
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)tile[ch * 8 + 2 * e][ci] | ((uint32_t)tile[ch * 8 + 2 * e + 1][ci] << 16);
        *reinterpret_cast<uint4*>(dst + ((long long)(ci0 + ci) * it.taps + tap) * it.dst_ld + co0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
      }
      return;
    }
  }
  float (*tile)[33] = reinterpret_cast<float (*)[33]>(lds);
  const int nx = (it.cin + 31) / 32, ny = (it.cout + 31) / 32;
  const int tap = t / (nx * ny), r2 = t - tap * nx * ny;
  const int ci0 = (r2 % nx) * 32, co0 = (r2 / nx) * 32;
  const S* src = static_cast<const S*>(it.src);
  D* dst = static_cast<D*>(it.dst);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < it.cout && ci < it.cin) ? to_f32<S>(src[((long long)co * it.taps + tap) * it.cin + ci]) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < it.cin && co < it.cout) dst[((long long)ci * it.taps + tap) * it.dst_ld + co] = from_f32<D>(tile[tx][r]);
  }
}

template <typename D>
__global__ __launch_bounds__(256) void cast_f32_lowp_kernel(const float* __restrict__ src, D* __restrict__ dst, long long n) {
  const long long stride = (long long)gridDim.x * 256 * 8;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      float v[8];
      ps_load8<float>(src + i, v);
      ps_store8<D>(dst + i, v);
    } else {
      for (long long j = i; j < n; ++j) dst[j] = static_cast<D>(src[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// optimisers over a flat arena
// ------------------------------------------------------------------------------------------------
// state (guarded form, ps_adamw_step_guarded): state[0] = optimiser steps applied so far, state[1] = non-finite gradient elements of THIS
// step (ps_nonfinite_count).  The launch does nothing when state[1] != 0 -- the overflow check of dynamic loss scaling without a host round
// trip -- and otherwise takes its bias corrections from t = state[0] + 1 (a skipped step must not advance Adam's step count).
template <typename SH>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, SH* __restrict__ pb, long long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt, float ginv,
                                                    const int* __restrict__ state) {
  if (state) {
    if (state[1] != 0) return;
    const float t = (float)(state[0] + 1);
    bc1 = 1.f - powf(b1, t);
    bc2_sqrt = sqrtf(1.f - powf(b2, t));
  }
  const long long stride = (long long)gridDim.x * 256 * 4;
  const float step_size = lr / bc1, decay = 1.f - lr * wd;
  auto upd = [&](float& pk, float gk, float& mk, float& vk) {
    gk *= ginv;  // 1 unless the gradient carries an fp16 loss scale (x * 1.0f is exact)
    pk *= decay;
    mk = mk + (gk - mk) * (1.f - b1);          // torch: exp_avg.lerp_(grad, 1 - beta1)
    vk = b2 * vk + (1.f - b2) * gk * gk;        // exp_avg_sq.mul_(b2).addcmul_(g, g, 1-b2)
    pk -= step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
  };
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {  // 16-byte accesses (the arenas are 16-byte aligned)
      float4 pk = *reinterpret_cast<const float4*>(p + i), mk = *reinterpret_cast<const float4*>(m + i), vk = *reinterpret_cast<const float4*>(v + i);
      const float4 gk = *reinterpret_cast<const float4*>(g + i);
      upd(pk.x, gk.x, mk.x, vk.x); upd(pk.y, gk.y, mk.y, vk.y); upd(pk.z, gk.z, mk.z, vk.z); upd(pk.w, gk.w, mk.w, vk.w);
      *reinterpret_cast<float4*>(p + i) = pk;
      *reinterpret_cast<float4*>(m + i) = mk;
      *reinterpret_cast<float4*>(v + i) = vk;
      if (pb) {
        typedef SH sh4 __attribute__((ext_vector_type(4)));
        sh4 o;
        o[0] = static_cast<SH>(pk.x); o[1] = static_cast<SH>(pk.y); o[2] = static_cast<SH>(pk.z); o[3] = static_cast<SH>(pk.w);
        *reinterpret_cast<sh4*>(pb + i) = o;
      }
    } else {
      for (long long k = i; k < n; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        upd(pk, g[k], mk, vk);
        p[k] = pk; m[k] = mk; v[k] = vk;
        if (pb) pb[k] = static_cast<SH>(pk);
      }
    }
  }
}

// guarded form (ps_sgd_step_guarded): state as for AdamW above; the launch does nothing when state[1] != 0, takes `first` (momentum buffer = gradient)
// from state[0] == 0 and scales lr by utils.PolyOptimizer's (1 - t / max_step) ** power with t = state[0] (held at max_step - 1 beyond the schedule).
template <typename SH, bool VEC>
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  SH* __restrict__ pb, long long n, float lr, float mom, float wd, int first, float ginv,
                                                  const int* __restrict__ state, int poly_max_step, float poly_power) {
  if (state) {
    if (state[1] != 0) return;
    const int t = state[0];
    first = t == 0;
    if (poly_max_step > 0) lr *= powf(1.f - (float)(t < poly_max_step ? t : poly_max_step - 1) / (float)poly_max_step, poly_power);
  }
  auto upd = [&](float& pk, float gk, float& bk) {
    gk *= ginv;
    if (wd != 0.f) gk = fmaf(wd, pk, gk);
    if (mom != 0.f) {
      bk = first ? gk : mom * bk + gk;
      gk = bk;
    }
    pk -= lr * gk;
  };
  // one element group per thread (no loop: a second iteration's loads would queue behind the first one's stores)
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
  if (i >= n) return;
  if constexpr (VEC) {
    if (i + 4 <= n) {  // 16-byte accesses (the host checks the alignment)
      float4 pk = *reinterpret_cast<const float4*>(p + i), bk = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 gk = *reinterpret_cast<const float4*>(g + i);
      if (mom != 0.f && !first) bk = *reinterpret_cast<const float4*>(buf + i);
      upd(pk.x, gk.x, bk.x); upd(pk.y, gk.y, bk.y); upd(pk.z, gk.z, bk.z); upd(pk.w, gk.w, bk.w);
      *reinterpret_cast<float4*>(p + i) = pk;
      if (mom != 0.f) *reinterpret_cast<float4*>(buf + i) = bk;
      if (pb) {
        typedef SH sh4 __attribute__((ext_vector_type(4)));
        sh4 o;
        o[0] = static_cast<SH>(pk.x); o[1] = static_cast<SH>(pk.y); o[2] = static_cast<SH>(pk.z); o[3] = static_cast<SH>(pk.w);
        *reinterpret_cast<sh4*>(pb + i) = o;
      }
      return;
    }
  }
  for (long long k = i; k < n && k < i + (VEC ? 4 : 1); ++k) {
    float pk = p[k], bk = (mom != 0.f && !first) ? buf[k] : 0.f;
    upd(pk, g[k], bk);
    p[k] = pk;
    if (mom != 0.f) buf[k] = bk;
    if (pb) pb[k] = static_cast<SH>(pk);
  }
}

// count of non-finite elements (fp16 dynamic loss scaling: an overflowed activation gradient reaches the arena as inf/nan)
__global__ __launch_bounds__(256) void nonfinite_kernel(const float* __restrict__ g, long long n, int* __restrict__ out) {
  const long long stride = (long long)gridDim.x * 256 * 4;
  int bad = 0;
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      const uint4 u = *reinterpret_cast<const uint4*>(g + i);
      bad += ((u.x & 0x7f800000u) == 0x7f800000u) + ((u.y & 0x7f800000u) == 0x7f800000u) + ((u.z & 0x7f800000u) == 0x7f800000u) +
             ((u.w & 0x7f800000u) == 0x7f800000u);
    } else {
      for (long long k = i; k < n; ++k) bad += (__float_as_uint(g[k]) & 0x7f800000u) == 0x7f800000u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out, bad);
}

static inline int grid_for(long long work_items, int per_block, int cap = 256 * 8) {
  long long b = (work_items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" int ps_conv1a_fwd(int32_t out_dtype, const float* x, const float* w, const float* scale, const float* shift,
                             void* out_act, void* out_raw, int32_t n, int32_t h, int32_t wd, void* stream) {
  PS_REQUIRE(x && w && (out_act || out_raw), "conv1a: null argument");
  PS_REQUIRE(n > 0 && h > 0 && wd > 0, "conv1a: empty input");
  PS_REQUIRE((!out_act || ps_aligned16(out_act)) && (!out_raw || ps_aligned16(out_raw)), "conv1a: misaligned output");
  const long long pix = (long long)n * h * wd;
  PS_REQUIRE((long long)3 * h * wd < (1LL << 31), "conv1a: image too large");
  PS_REQUIRE(pix * 12 < (1LL << 31) - 64, "conv1a: image batch of 2 GiB or more in one launch");
  const FastDiv div_hw = make_fastdiv((uint32_t)(h * wd)), div_w = make_fastdiv((uint32_t)wd);
  const int grid = grid_for(pix, 64, 256 * 8);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the 16-bit kernels walk the pixels with a grid stride: exactly the resident blocks (one round, equal shares) -- the fixed cap of
  // 2048 blocks was 2.67 rounds of the 3 blocks a CU held
  auto lowp = [&](auto kernel, auto* oa, auto* orw) {
    const int bpc = ps_blocks_per_cu(kernel);
    hipLaunchKernelGGL(kernel, dim3(std::min(grid, ps_num_cus() * bpc)), dim3(256), 0, s, x, w, scale, shift, oa, orw, n, h, wd, div_hw, div_w);
  };
  if (out_dtype == PS_BF16) {
    __bf16 *oa = (__bf16*)out_act, *orw = (__bf16*)out_raw;
    if (oa && orw) lowp(conv1a_lowp_kernel<__bf16, false, true, true>, oa, orw);
    else if (oa) lowp(conv1a_lowp_kernel<__bf16, false, false, true>, oa, orw);
    else lowp(conv1a_lowp_kernel<__bf16, false, true, false>, oa, orw);
  } else if (out_dtype == PS_F16) {
    _Float16 *oa = (_Float16*)out_act, *orw = (_Float16*)out_raw;
    if (oa && orw) lowp(conv1a_lowp_kernel<_Float16, true, true, true>, oa, orw);
    else if (oa) lowp(conv1a_lowp_kernel<_Float16, true, false, true>, oa, orw);
    else lowp(conv1a_lowp_kernel<_Float16, true, true, false>, oa, orw);
  } else if (out_dtype == PS_F32)
    hipLaunchKernelGGL(conv1a_kernel<float>, dim3(grid), dim3(256), 0, s, x, w, scale, shift, (float*)out_act, (float*)out_raw, n, h, wd);
  else
    PS_REQUIRE(false, "conv1a: dtype %d unsupported", out_dtype);
  PS_CHECK_LAUNCH("conv1a");
  return PS_OK;
}

extern "C" int ps_fc_head_fwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, int32_t ldw, const float* bias, const float* drop,
                              float* cam, int32_t accumulate, int32_t m_total, int32_t ppi, int32_t k, int32_t c, void* stream) {
  PS_REQUIRE(x && w && cam, "fc_head_fwd: null argument");
  PS_REQUIRE(c >= 1 && c <= FC8_MAXC, "fc_head_fwd: C=%d unsupported (1..%d)", c, FC8_MAXC);
  PS_REQUIRE(k % 512 == 0 && m_total > 0 && ppi > 0, "fc_head_fwd: K=%d must be a multiple of 512", k);
  PS_REQUIRE(ldw >= k && ldw % 4 == 0, "fc_head_fwd: weight row stride %d must be >= K and a multiple of 4", ldw);
  PS_REQUIRE(ps_aligned16(x) && ps_aligned16(w) && (ldc_x * ps_esize(dtype)) % 16 == 0, "fc_head_fwd: misaligned input");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PS_REQUIRE(ps_dtype_ok(dtype), "fc_head_fwd: dtype %d unsupported", dtype);
  // pixels per wave: one (or a whole number of) full round(s) of resident waves, at most ~24 pixels each
  auto launch = [&](auto kernel, auto xp) {
    const int bpc = ps_blocks_per_cu(kernel);
    const long long cap = (long long)ps_num_cus() * bpc * 4;
    const long long rounds = (m_total + cap * 24 - 1) / (cap * 24);
    const int ppw = (int)((m_total + cap * rounds - 1) / (cap * rounds));
    const int grid = (int)(((m_total + ppw - 1) / ppw + 3) / 4);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, xp, ldc_x, w, ldw, bias, drop, cam, accumulate, m_total, ppi, k, ppw);
  };
#define PS_FC8_FWD(CC)                                                                            \
  case CC:                                                                                        \
    if (dtype == PS_BF16) launch(fc8_fwd_kernel<__bf16, CC>, (const __bf16*)x);                   \
    else if (dtype == PS_F16) launch(fc8_fwd_kernel<_Float16, CC>, (const _Float16*)x);           \
    else launch(fc8_fwd_kernel<float, CC>, (const float*)x);                                      \
    break;
  switch (c) { PS_FC8_FWD(1) PS_FC8_FWD(2) PS_FC8_FWD(3) PS_FC8_FWD(4) PS_FC8_FWD(5) PS_FC8_FWD(6) PS_FC8_FWD(7) PS_FC8_FWD(8) }
#undef PS_FC8_FWD
  PS_CHECK_LAUNCH("fc_head_fwd");
  return PS_OK;
}

extern "C" int ps_fc8_fwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, float* cam,
                          int32_t m_total, int32_t ppi, int32_t k, int32_t c, void* stream) {
  return ps_fc_head_fwd(dtype, x, ldc_x, w, k, nullptr, drop, cam, 0, m_total, ppi, k, c, stream);
}

// Pixels per block of fc8_bwd_kernel: at most one image's worth (a block's pixels span <= two images), at least min(32, ppi),
// and -- given the kernel's resident blocks per CU -- such that the grid is a whole number of full rounds.
static int fc8_bwd_ppb(int m_total, int ppi, int kblocks, int bpc) {
  const long long cap = (long long)ps_num_cus() * bpc;
  const long long rounds = ((long long)m_total * kblocks + cap * 96 - 1) / (cap * 96);
  const long long mblocks = cap * rounds / kblocks > 0 ? cap * rounds / kblocks : 1;
  long long ppb = (m_total + mblocks - 1) / mblocks;
  const int lo = ppi < 32 ? ppi : 32;
  if (ppb < lo) ppb = lo;
  if (ppb > ppi) ppb = ppi;
  return (int)ppb;
}

extern "C" int64_t ps_fc8_bwd_workspace_floats(int32_t m_total, int32_t ppi, int32_t k, int32_t c) {
  if (m_total <= 0 || ppi <= 0 || k <= 0 || c <= 0) return 0;
  const int lo = ppi < 32 ? ppi : 32;  // smallest block any launch uses
  return (int64_t)((m_total + lo - 1) / lo) * c * k;
}

extern "C" int ps_fc8_bwd_ws(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, const float* scale7,
                             const float* dcam, void* dx, int32_t ldc_dx, float* dw, int32_t m_total, int32_t ppi, int32_t k,
                             int32_t c, float* workspace, int64_t workspace_floats, void* stream) {
  PS_REQUIRE(x && w && scale7 && dcam && dx && dw, "fc8_bwd: null argument");
  PS_REQUIRE(c >= 1 && c <= FC8_MAXC, "fc8_bwd: C=%d unsupported (1..%d)", c, FC8_MAXC);
  PS_REQUIRE(k % 2048 == 0 && m_total > 0 && ppi > 0, "fc8_bwd: K=%d must be a multiple of 2048", k);
  const int es = ps_esize(dtype);
  PS_REQUIRE(ps_aligned16(x) && ps_aligned16(dx) && (ldc_x * es) % 16 == 0 && (ldc_dx * es) % 16 == 0, "fc8_bwd: misaligned tensor");
  PS_REQUIRE(!workspace || ps_aligned16(workspace), "fc8_bwd: workspace must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  PS_REQUIRE(ps_dtype_ok(dtype), "fc8_bwd: dtype %d unsupported", dtype);
  int mblocks = 0, rc = PS_OK;
  auto launch = [&](auto kernel, auto xp, auto dxp) {
    const int bpc = ps_blocks_per_cu(kernel);
    const int ppb = fc8_bwd_ppb(m_total, ppi, k / 2048, bpc);
    mblocks = (m_total + ppb - 1) / ppb;
    if (workspace && workspace_floats < (int64_t)mblocks * c * k) {
      ps_set_error("fc8_bwd: workspace of %lld floats, need %lld", (long long)workspace_floats, (long long)mblocks * c * k);
      rc = PS_ERR_ARG;
      return;
    }
    hipLaunchKernelGGL(kernel, dim3((k / 2048) * mblocks), dim3(256), 0, s, xp, ldc_x, w, drop, scale7, dcam, dxp, ldc_dx, dw, workspace, m_total,
                       ppi, k, ppb);
  };
#define PS_FC8_BWD_T(TT, CC)                                                          \
  if (drop) launch(fc8_bwd_kernel<TT, CC, true>, (const TT*)x, (TT*)dx);              \
  else launch(fc8_bwd_kernel<TT, CC, false>, (const TT*)x, (TT*)dx);
#define PS_FC8_BWD(CC)                                       \
  case CC:                                                   \
    if (dtype == PS_BF16) { PS_FC8_BWD_T(__bf16, CC) }       \
    else if (dtype == PS_F16) { PS_FC8_BWD_T(_Float16, CC) } \
    else { PS_FC8_BWD_T(float, CC) }                         \
    break;
  switch (c) { PS_FC8_BWD(1) PS_FC8_BWD(2) PS_FC8_BWD(3) PS_FC8_BWD(4) PS_FC8_BWD(5) PS_FC8_BWD(6) PS_FC8_BWD(7) PS_FC8_BWD(8) }
#undef PS_FC8_BWD_T
#undef PS_FC8_BWD
  if (rc != PS_OK) return rc;
  PS_CHECK_LAUNCH("fc8_bwd");
  if (workspace) {
    const int ck = c * k;
    const int per_slice = (mblocks + 15) / 16;  // 16 slices of the pixel blocks per group of four outputs, combined in slice order
    hipLaunchKernelGGL(fc8_dw_reduce_kernel, dim3((ck / 4 + 15) / 16), dim3(256), 0, s, workspace, dw, ck, mblocks, per_slice);
    PS_CHECK_LAUNCH("fc8_dw_reduce");
  }
  return PS_OK;
}

extern "C" int ps_fc8_bwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, const float* scale7,
                          const float* dcam, void* dx, int32_t ldc_dx, float* dw, int32_t m_total, int32_t ppi, int32_t k,
                          int32_t c, void* stream) {
  return ps_fc8_bwd_ws(dtype, x, ldc_x, w, drop, scale7, dcam, dx, ldc_dx, dw, m_total, ppi, k, c, nullptr, 0, stream);
}

extern "C" int ps_weight_transpose(int32_t sdt, int32_t ddt, const void* src, void* dst, int32_t cout, int32_t taps, int32_t cin,
                                   void* stream) {
  PS_REQUIRE(src && dst && cout > 0 && taps > 0 && cin > 0, "weight_transpose: bad argument");
  dim3 grid((cin + 31) / 32, (cout + 31) / 32, taps);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (sdt == PS_F32 && ddt == PS_F32)
    hipLaunchKernelGGL((weight_transpose_kernel<float, float>), grid, dim3(256), 0, s, (const float*)src, (float*)dst, cout, taps, cin);
  else if (sdt == PS_F32 && ddt == PS_BF16)
    hipLaunchKernelGGL((weight_transpose_kernel<float, uint16_t>), grid, dim3(256), 0, s, (const float*)src, (uint16_t*)dst, cout, taps, cin);
  else if (sdt == PS_F32 && ddt == PS_F16)
    hipLaunchKernelGGL((weight_transpose_kernel<float, h16>), grid, dim3(256), 0, s, (const float*)src, (h16*)dst, cout, taps, cin);
  else if ((sdt == PS_BF16 && ddt == PS_BF16) || (sdt == PS_F16 && ddt == PS_F16))  // same 16-bit type: bits move unchanged (bf16 <-> f32 is exact)
    hipLaunchKernelGGL((weight_transpose_kernel<uint16_t, uint16_t>), grid, dim3(256), 0, s, (const uint16_t*)src, (uint16_t*)dst, cout, taps, cin);
  else
    PS_REQUIRE(false, "weight_transpose: dtype pair (%d,%d) unsupported", sdt, ddt);
  PS_CHECK_LAUNCH("weight_transpose");
  return PS_OK;
}

extern "C" int ps_weight_transpose_batched(int32_t sdt, int32_t ddt, int32_t n_items, const ps_wt_item* items, void* stream) {
  PS_REQUIRE(n_items >= 0 && (n_items == 0 || items), "weight_transpose_batched: bad argument");
  hipStream_t s = static_cast<hipStream_t>(stream);
  for (int base = 0; base < n_items; base += WT_MAX_ITEMS) {
    WtBatch b{};
    b.n = n_items - base < WT_MAX_ITEMS ? n_items - base : WT_MAX_ITEMS;
    long long tiles = 0;
    for (int k = 0; k < b.n; ++k) {
      const ps_wt_item& it = items[base + k];
      PS_REQUIRE(it.src && it.dst && it.cout > 0 && it.taps > 0 && it.cin > 0 && it.dst_ld >= it.cout, "weight_transpose_batched: bad item %d", base + k);
      b.it[k] = it;
      const bool same16 = (sdt == PS_BF16 && ddt == PS_BF16) || (sdt == PS_F16 && ddt == PS_F16);
      const int ts = wt_fast(it, same16) ? 64 : 32;
      tiles += (long long)((it.cin + ts - 1) / ts) * ((it.cout + ts - 1) / ts) * it.taps;
      PS_REQUIRE(tiles < (1LL << 31), "weight_transpose_batched: too many tiles");
      b.tile_end[k] = (int)tiles;
    }
    const dim3 grid((unsigned)tiles);
    if (sdt == PS_F32 && ddt == PS_F32)
      hipLaunchKernelGGL((weight_transpose_batched_kernel<float, float>), grid, dim3(256), 0, s, b);
    else if (sdt == PS_F32 && ddt == PS_BF16)
      hipLaunchKernelGGL((weight_transpose_batched_kernel<float, uint16_t>), grid, dim3(256), 0, s, b);
    else if (sdt == PS_F32 && ddt == PS_F16)
      hipLaunchKernelGGL((weight_transpose_batched_kernel<float, h16>), grid, dim3(256), 0, s, b);
    else if ((sdt == PS_BF16 && ddt == PS_BF16) || (sdt == PS_F16 && ddt == PS_F16))
      hipLaunchKernelGGL((weight_transpose_batched_kernel<uint16_t, uint16_t>), grid, dim3(256), 0, s, b);
    else
      PS_REQUIRE(false, "weight_transpose_batched: dtype pair (%d,%d) unsupported", sdt, ddt);
    PS_CHECK_LAUNCH("weight_transpose_batched");
  }
  return PS_OK;
}

extern "C" int ps_cast_f32_lowp(const float* src, void* dst, int32_t dst_dtype, int64_t n, void* stream) {
  PS_REQUIRE(src && dst && n >= 0, "cast_f32_lowp: bad argument");
  PS_REQUIRE(dst_dtype == PS_BF16 || dst_dtype == PS_F16, "cast_f32_lowp: dtype %d unsupported", dst_dtype);
  if (n == 0) return PS_OK;
  PS_REQUIRE(ps_aligned16(src) && ps_aligned16(dst), "cast_f32_lowp: misaligned pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dst_dtype == PS_BF16)
    hipLaunchKernelGGL(cast_f32_lowp_kernel<__bf16>, dim3(grid_for(n, 256 * 8)), dim3(256), 0, s, src, (__bf16*)dst, (long long)n);
  else
    hipLaunchKernelGGL(cast_f32_lowp_kernel<_Float16>, dim3(grid_for(n, 256 * 8)), dim3(256), 0, s, src, (_Float16*)dst, (long long)n);
  PS_CHECK_LAUNCH("cast_f32_lowp");
  return PS_OK;
}

extern "C" int ps_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  return ps_cast_f32_lowp(src, dst, PS_BF16, n, stream);
}

extern "C" int ps_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* p_shadow, int32_t shadow_dtype, int64_t n,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                    float grad_inv_scale, void* stream) {
  PS_REQUIRE(p && g && m && v && n >= 0 && step >= 1, "adamw_step: bad argument");
  PS_REQUIRE(!p_shadow || shadow_dtype == PS_BF16 || shadow_dtype == PS_F16, "adamw_step: shadow dtype %d unsupported", shadow_dtype);
  if (n == 0) return PS_OK;
  PS_REQUIRE(ps_aligned16(p) && ps_aligned16(g) && ps_aligned16(m) && ps_aligned16(v) && (!p_shadow || (reinterpret_cast<uintptr_t>(p_shadow) & 7u) == 0),
             "adamw_step: arenas must be 16-byte aligned");
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p_shadow && shadow_dtype == PS_F16)
    hipLaunchKernelGGL(adamw_kernel<_Float16>, dim3(grid_for(n, 256 * 4, 1 << 30)), dim3(256), 0, s, p, g, m, v, (_Float16*)p_shadow, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_inv_scale, (const int*)nullptr);
  else
    hipLaunchKernelGGL(adamw_kernel<__bf16>, dim3(grid_for(n, 256 * 4, 1 << 30)), dim3(256), 0, s, p, g, m, v, (__bf16*)p_shadow, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_inv_scale, (const int*)nullptr);
  PS_CHECK_LAUNCH("adamw_step");
  return PS_OK;
}

namespace {
__global__ void adamw_state_advance_kernel(int* state) {
  if (state[1] == 0) state[0] += 1;
}
}  // namespace

extern "C" int ps_adamw_step_guarded(float* p, const float* g, float* m, float* v, void* p_shadow, int32_t shadow_dtype, int64_t n,
                                     float lr, float beta1, float beta2, float eps, float weight_decay, int32_t* state,
                                     float grad_inv_scale, void* stream) {
  PS_REQUIRE(p && g && m && v && state && n >= 0, "adamw_step_guarded: bad argument");
  PS_REQUIRE(!p_shadow || shadow_dtype == PS_BF16 || shadow_dtype == PS_F16, "adamw_step_guarded: shadow dtype %d unsupported", shadow_dtype);
  if (n == 0) return PS_OK;
  PS_REQUIRE(ps_aligned16(p) && ps_aligned16(g) && ps_aligned16(m) && ps_aligned16(v) && (!p_shadow || (reinterpret_cast<uintptr_t>(p_shadow) & 7u) == 0),
             "adamw_step_guarded: arenas must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (p_shadow && shadow_dtype == PS_F16)
    hipLaunchKernelGGL(adamw_kernel<_Float16>, dim3(grid_for(n, 256 * 4, 1 << 30)), dim3(256), 0, s, p, g, m, v, (_Float16*)p_shadow, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_inv_scale, (const int*)state);
  else
    hipLaunchKernelGGL(adamw_kernel<__bf16>, dim3(grid_for(n, 256 * 4, 1 << 30)), dim3(256), 0, s, p, g, m, v, (__bf16*)p_shadow, (long long)n, lr,
                       beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_inv_scale, (const int*)state);
  hipLaunchKernelGGL(adamw_state_advance_kernel, dim3(1), dim3(1), 0, s, state);  // behind the update: every block of it read the old count
  PS_CHECK_LAUNCH("adamw_step_guarded");
  return PS_OK;
}

extern "C" int ps_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int32_t step, void* stream) {
  return ps_adamw_step_scaled(p, g, m, v, p_bf16, PS_BF16, n, lr, beta1, beta2, eps, weight_decay, step, 1.0f, stream);
}

extern "C" int ps_nonfinite_count(const float* g, int64_t n, int32_t* count, void* stream) {
  PS_REQUIRE(g && count && n >= 0, "nonfinite_count: bad argument");
  if (n == 0) return PS_OK;
  PS_REQUIRE(ps_aligned16(g), "nonfinite_count: misaligned pointer");
  hipLaunchKernelGGL(nonfinite_kernel, dim3(grid_for(n, 256 * 4)), dim3(256), 0, static_cast<hipStream_t>(stream), g, (long long)n, count);
  PS_CHECK_LAUNCH("nonfinite_count");
  return PS_OK;
}

extern "C" int ps_sgd_step_scaled(float* p, const float* g, float* buf, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr,
                                  float momentum, float weight_decay, int32_t first_step, float grad_inv_scale, void* stream) {
  PS_REQUIRE(p && g && n >= 0 && (momentum == 0.f || buf), "sgd_step: bad argument");
  PS_REQUIRE(!p_shadow || shadow_dtype == PS_BF16 || shadow_dtype == PS_F16, "sgd_step: shadow dtype %d unsupported", shadow_dtype);
  if (n == 0) return PS_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = ps_aligned16(p) && ps_aligned16(g) && (!buf || ps_aligned16(buf)) && (!p_shadow || (reinterpret_cast<uintptr_t>(p_shadow) & 7u) == 0);
  const dim3 grid((unsigned)((n + (vec ? 1024 : 256) - 1) / (vec ? 1024 : 256)));
#define PS_SGD(SH)                                                                                                                              \
  if (vec) hipLaunchKernelGGL((sgd_kernel<SH, true>), grid, dim3(256), 0, s, p, g, buf, (SH*)p_shadow, (long long)n, lr, momentum, weight_decay, \
                              first_step, grad_inv_scale, (const int*)nullptr, 0, 0.f);                                                        \
  else hipLaunchKernelGGL((sgd_kernel<SH, false>), grid, dim3(256), 0, s, p, g, buf, (SH*)p_shadow, (long long)n, lr, momentum, weight_decay,    \
                          first_step, grad_inv_scale, (const int*)nullptr, 0, 0.f);
  if (p_shadow && shadow_dtype == PS_F16) { PS_SGD(_Float16) } else { PS_SGD(__bf16) }
#undef PS_SGD
  PS_CHECK_LAUNCH("sgd_step");
  return PS_OK;
}

extern "C" int ps_sgd_step_guarded(float* p, const float* g, float* buf, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr, float momentum,
                                   float weight_decay, int32_t* state, int32_t advance, int32_t poly_max_step, float poly_power, float grad_inv_scale,
                                   void* stream) {
  PS_REQUIRE(p && g && state && n >= 0 && (momentum == 0.f || buf), "sgd_step_guarded: bad argument");
  PS_REQUIRE(!p_shadow || shadow_dtype == PS_BF16 || shadow_dtype == PS_F16, "sgd_step_guarded: shadow dtype %d unsupported", shadow_dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n > 0) {
    const bool vec = ps_aligned16(p) && ps_aligned16(g) && (!buf || ps_aligned16(buf)) && (!p_shadow || (reinterpret_cast<uintptr_t>(p_shadow) & 7u) == 0);
    const dim3 grid((unsigned)((n + (vec ? 1024 : 256) - 1) / (vec ? 1024 : 256)));
#define PS_SGD(SH)                                                                                                                              \
  if (vec) hipLaunchKernelGGL((sgd_kernel<SH, true>), grid, dim3(256), 0, s, p, g, buf, (SH*)p_shadow, (long long)n, lr, momentum, weight_decay, \
                              0, grad_inv_scale, (const int*)state, poly_max_step, poly_power);                                                \
  else hipLaunchKernelGGL((sgd_kernel<SH, false>), grid, dim3(256), 0, s, p, g, buf, (SH*)p_shadow, (long long)n, lr, momentum, weight_decay,    \
                          0, grad_inv_scale, (const int*)state, poly_max_step, poly_power);
    if (p_shadow && shadow_dtype == PS_F16) { PS_SGD(_Float16) } else { PS_SGD(__bf16) }
#undef PS_SGD
  }
  if (advance) hipLaunchKernelGGL(adamw_state_advance_kernel, dim3(1), dim3(1), 0, s, state);  // behind the update(s): every block read the old count
  PS_CHECK_LAUNCH("sgd_step_guarded");
  return PS_OK;
}

extern "C" int ps_sgd_step(float* p, const float* g, float* buf, void* p_bf16, int64_t n, float lr, float momentum,
                           float weight_decay, int32_t first_step, void* stream) {
  return ps_sgd_step_scaled(p, g, buf, p_bf16, PS_BF16, n, lr, momentum, weight_decay, first_step, 1.0f, stream);
}

// dst[r][0 .. row_bytes) = src[r][0 .. row_bytes) for rows r with independent pitches (16-byte vectors): assembling K-concatenated weights
namespace {
__global__ __launch_bounds__(256) void copy_rows_kernel(const unsigned char* __restrict__ src, long long src_ld, unsigned char* __restrict__ dst,
                                                        long long dst_ld, long long rows, int vec_per_row) {
  const long long total = rows * vec_per_row;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / vec_per_row;
    const int v = (int)(i - r * vec_per_row);
    *reinterpret_cast<uint4*>(dst + r * dst_ld + v * 16) = *reinterpret_cast<const uint4*>(src + r * src_ld + v * 16);
  }
}
}  // namespace

namespace {
// ------------------------------------------------------------------------------------------------
// Row-wise format conversion between f32 and the activation / weight storage formats: rows of c logical channels, 8 per thread,
// arbitrary row pitches (channel slices of wider buffers).  f32 -> {bf16, f16, split}; {bf16, f16, split} -> f32.
// Split (PS_BF16X3 / PS_F16X3): 2 c stored 16-bit channels per row in blocks of 32 logical channels, [hi(32) | lo(32)]; hi = round16(v) RNE,
// lo = round16(v - hi); reading back: hi + lo.  Activations and weights share the layout (`pattern` is accepted and ignored).
// ------------------------------------------------------------------------------------------------
template <int SRC, int DST>
__global__ __launch_bounds__(256) void convert_rows_kernel(const unsigned char* __restrict__ src, long long ld_src, unsigned char* __restrict__ dst,
                                                            long long ld_dst, long long rows, int c8, int pattern) {
  const long long total = rows * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c8;
    const int c = (int)(i - r * c8) * 8;
    float v[8];
    if constexpr (SRC == PS_F32) {
      ps_load8<float>(reinterpret_cast<const float*>(src) + r * ld_src + c, v);
    } else if constexpr (SRC == PS_BF16) {
      ps_load8<__bf16>(reinterpret_cast<const __bf16*>(src) + r * ld_src + c, v);
    } else if constexpr (SRC == PS_F16) {
      ps_load8<_Float16>(reinterpret_cast<const _Float16*>(src) + r * ld_src + c, v);
    } else {
      typedef typename std::conditional<SRC == PS_F16X3, _Float16, __bf16>::type PT;
      const PT* p = reinterpret_cast<const PT*>(src) + r * ld_src + ((c >> 5) << 6) + (c & 31);
      float lo[8];
      ps_load8<PT>(p, v);
      ps_load8<PT>(p + 32, lo);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += lo[k];
    }
    if constexpr (DST == PS_F32) {
      ps_store8<float>(reinterpret_cast<float*>(dst) + r * ld_dst + c, v);
    } else if constexpr (DST == PS_BF16) {
      ps_store8<__bf16>(reinterpret_cast<__bf16*>(dst) + r * ld_dst + c, v);
    } else if constexpr (DST == PS_F16) {
      ps_store8<_Float16>(reinterpret_cast<_Float16*>(dst) + r * ld_dst + c, v);
    } else {
      typedef typename std::conditional<DST == PS_F16X3, _Float16, __bf16>::type PT;
      PT* p = reinterpret_cast<PT*>(dst) + r * ld_dst + ((c >> 5) << 6) + (c & 31);
      float hi[8], lo[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        hi[k] = static_cast<float>(static_cast<PT>(v[k]));
        lo[k] = v[k] - hi[k];
      }
      ps_store8<PT>(p, hi);
      ps_store8<PT>(p + 32, lo);
    }
  }
}
}  // namespace

extern "C" int ps_convert_rows(const void* src, int32_t src_fmt, int64_t ld_src, void* dst, int32_t dst_fmt, int64_t ld_dst, int64_t rows, int32_t c,
                               int32_t pattern, void* stream) {
  PS_REQUIRE(src && dst && rows >= 0 && c > 0, "convert_rows: bad argument");
  PS_REQUIRE((src_fmt == PS_F32) != (dst_fmt == PS_F32), "convert_rows: exactly one side must be f32 (got %d -> %d)", src_fmt, dst_fmt);
  PS_REQUIRE(ps_conv_dtype_ok(src_fmt) && ps_conv_dtype_ok(dst_fmt) && (pattern == 0 || pattern == 1), "convert_rows: bad format / pattern");
  PS_REQUIRE(c % 8 == 0 && ld_src >= ps_planes(src_fmt) * (int64_t)c && ld_dst >= ps_planes(dst_fmt) * (int64_t)c, "convert_rows: c=%d must be a multiple of 8 and fit the pitches", c);
  PS_REQUIRE((ps_planes(src_fmt) == 1 && ps_planes(dst_fmt) == 1) || c % 32 == 0, "convert_rows: split formats store blocks of 32 channels (c=%d)", c);
  PS_REQUIRE(ps_aligned16(src) && ps_aligned16(dst) && (ld_src * ps_esize(src_fmt)) % 16 == 0 && (ld_dst * ps_esize(dst_fmt)) % 16 == 0, "convert_rows: misaligned");
  if (rows == 0) return PS_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(grid_for(rows * (c / 8), 256)), block(256);
  const unsigned char* sp = static_cast<const unsigned char*>(src);
  unsigned char* dp = static_cast<unsigned char*>(dst);
#define PS_CVT(S, D) hipLaunchKernelGGL((convert_rows_kernel<S, D>), grid, block, 0, s, sp, (long long)ld_src, dp, (long long)ld_dst, (long long)rows, c / 8, pattern)
  if (src_fmt == PS_F32) {
    if (dst_fmt == PS_BF16) PS_CVT(PS_F32, PS_BF16);
    else if (dst_fmt == PS_F16) PS_CVT(PS_F32, PS_F16);
    else if (dst_fmt == PS_F16X3) PS_CVT(PS_F32, PS_F16X3);
    else PS_CVT(PS_F32, PS_BF16X3);
  } else {
    if (src_fmt == PS_BF16) PS_CVT(PS_BF16, PS_F32);
    else if (src_fmt == PS_F16) PS_CVT(PS_F16, PS_F32);
    else if (src_fmt == PS_F16X3) PS_CVT(PS_F16X3, PS_F32);
    else PS_CVT(PS_BF16X3, PS_F32);
  }
#undef PS_CVT
  PS_CHECK_LAUNCH("convert_rows");
  return PS_OK;
}

extern "C" int ps_copy_rows(const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes, int64_t rows, int64_t row_bytes, void* stream) {
  PS_REQUIRE(src && dst && rows > 0 && row_bytes > 0, "copy_rows: bad argument");
  PS_REQUIRE(row_bytes % 16 == 0 && src_ld_bytes % 16 == 0 && dst_ld_bytes % 16 == 0 && ps_aligned16(src) && ps_aligned16(dst),
             "copy_rows: rows, pitches and pointers must be 16-byte multiples");
  PS_REQUIRE(row_bytes <= src_ld_bytes && row_bytes <= dst_ld_bytes && row_bytes / 16 < (1LL << 31), "copy_rows: row longer than a pitch");
  hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(rows * (row_bytes / 16), 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const unsigned char*)src, (long long)src_ld_bytes, (unsigned char*)dst, (long long)dst_ld_bytes, (long long)rows,
                     (int)(row_bytes / 16));
  PS_CHECK_LAUNCH("copy_rows");
  return PS_OK;
}

namespace {
// ------------------------------------------------------------------------------------------------
// Dropout2d multipliers for a whole training step in ONE launch: a flat f32 buffer of up to 8 segments (one per Dropout2d of the
// net: [n, channels] each), element e of segment k = (u >= p_k) / (1 - p_k) with u uniform from Philox4x32-10 keyed by the seed,
// counter = (e / 4, offset).  replaces: the Bernoulli draw of nn.Dropout2d (resnet38d.py:63,67,85,90; revise_net.py:11,50) --
// the multiply itself is fused into the conv epilogues.  (No RNG parity with torch is possible or claimed: tests inject masks.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
__global__ __launch_bounds__(256) void dropout2d_masks_kernel(float* __restrict__ out, ps_dropout_plan plan, uint64_t seed, uint64_t offset) {
  const long long total = plan.end[plan.nseg - 1];
  const long long q = (long long)blockIdx.x * 256 + threadIdx.x;  // one Philox block = 4 elements
  if (q * 4 >= total) return;
  uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long e = q * 4 + i;
    if (e >= total) break;
    int k = 0;
    while (k < plan.nseg - 1 && e >= plan.end[k]) ++k;
    const float u = (float)(c[i] >> 8) * (1.0f / 16777216.0f);  // [0, 1) on a 24-bit grid
    const float p = plan.p[k];
    out[e] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
  }
}
}  // namespace

extern "C" int ps_dropout2d_masks(float* out, const ps_dropout_plan* plan, uint64_t seed, uint64_t offset, void* stream) {
  PS_REQUIRE(out && plan && plan->nseg >= 1 && plan->nseg <= 8, "dropout2d_masks: bad argument");
  long long prev = 0;
  for (int k = 0; k < plan->nseg; ++k) {
    PS_REQUIRE(plan->end[k] >= prev && plan->p[k] >= 0.f && plan->p[k] < 1.f, "dropout2d_masks: segment %d: end %lld, p %f", k, (long long)plan->end[k],
               (double)plan->p[k]);
    prev = plan->end[k];
  }
  if (prev == 0) return PS_OK;
  hipLaunchKernelGGL(dropout2d_masks_kernel, dim3((unsigned)((prev + 1023) / 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), out, *plan, seed, offset);
  PS_CHECK_LAUNCH("dropout2d_masks");
  return PS_OK;
}

namespace {

// Testing hook: `blocks` workgroups of 256 threads that each hold a CU slot for `usec` microseconds (spin on the 100 MHz real-time
// counter) -- stands in for a communication kernel running beside the persistent conv kernels (tools/hog_probe.py).
__global__ __launch_bounds__(256) void hog_kernel(long long ticks, int* sink) {
  extern __shared__ int hog_lds[];  // dynamic LDS only serves to keep other workgroups off this CU
  if (sink) hog_lds[threadIdx.x] = 0;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  int n = 0;
  while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    __builtin_amdgcn_s_sleep(32);
    ++n;
  }
  if (sink && n < 0) *sink = n;
}
}  // namespace

#ifdef PS_DEBUG_HOOKS
extern "C" int ps_debug_hog(int32_t blocks, int32_t usec, int32_t lds_bytes, void* stream) {
  PS_REQUIRE(blocks > 0 && usec > 0 && usec <= 2000000 && lds_bytes >= 0 && lds_bytes <= 160 * 1024, "debug_hog: bad argument");
  hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, static_cast<hipStream_t>(stream), (long long)usec * 100, (int*)nullptr);
  PS_CHECK_LAUNCH("debug_hog");
  return PS_OK;
}
#endif
