// Implicit-GEMM convolution on gfx950 MFMA: forward and data-gradient of every 1x1 / 3x3 (stride 1|2,
// dilation 1|2|4) convolution of ResNet38-d, channels-last, with the BN+ReLU(+Dropout2d)/residual
// epilogues fused.  One kernel template serves both directions: a "produced" pixel grid gathers rows of a
// "source" tensor through per-tap offsets (forward: y = p*stride + (ty-c)*dil; dgrad: y = (p - (ty-c)*dil)/stride
// when divisible), so dilation is nothing but a tap offset and padding is a zero row.
//
// GEMM view   D[cout][pixel] = sum_{tap, cin} W[cout][tap][cin] * X[pixel@tap][cin]
//   * K is walked in 128-byte "K-lines" (64 bf16 / 32 f32 channels of one tap): both operands are rows of
//     contiguous K, staged HBM -> LDS with 16-byte `global_load_lds` (LDS-DMA, per-lane gather address,
//     padding rows read a zero page) into an XOR-swizzled [row][128 B] image, double buffered.
//   * 256 threads = 4 waves; block tile 128 pixels x BN couts (BN = 128: waves 2x2 of 64x64; BN = 64:
//     waves 4x1 of 32x64); MFMA 16x16x32 bf16 (or 4 x 16x16x4 exact-f32) with the weights as the A
//     operand, so a lane's 4 accumulator registers are 4 consecutive couts of ONE pixel; the cout rows
//     are permuted while staging so that each lane ends up with 16 contiguous couts -> 16-byte stores.
#include "ps_internal.h"
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#ifndef PS_READ_PER
#define PS_READ_PER 2  // MFMAs between two fragment reads of the consumers' software pipeline (A/B builds: tools/ab_build.py)
#endif

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_zero_line[256];  // source of padding rows

struct IgemmArgs {
  const unsigned char* src;   // gathered activation (forward: x, dgrad: dy)
  const unsigned char* wgt;   // [Cd][taps][Cs] rows of contiguous K
  int Hs, Ws, Ho, Wo, M;      // source dims, produced grid, produced pixels
  int epi_M;                  // = M; the epilogue skips pixels >= epi_M (0 under ps_debug_set_ablate(3): timing without any epilogue traffic)
  int mul, dstep, div_shift;  // gather arithmetic (see file header)
  int taps, klines;           // 1|9, Cs*esize/128
  int ctr;                    // centre tap coordinate (0 for 1x1, 1 for 3x3)
  long long pix_bytes;        // source channel stride in bytes
  long long wrow_bytes;       // taps*Cs*esize
  int ntn;                    // number of cout tiles
  int ntm;                    // number of pixel tiles
  int supertile;              // cout tiles per super-column of the block raster (0 = plain row-major)
  int Cd;                     // produced channels
  unsigned src_bytes;         // extent of the gathered tensor / of the weights (buffer descriptors' num_records)
  unsigned wgt_bytes;
  int ablate;                 // timing experiments only (results WRONG): 1 = stage the first two K-steps only; 2 = also no per-step barriers (ws kernel)
  int nb, tpb;                // persistent kernels: blocks per batch (#CUs), tiles per block (0 = one batch), see ps_block_items
  int shared;                 // host side only (ps_conv_geom.gpu_shared): another stream fills this launch's partial last round -- no tail launch
  int reserved;               // host side only (ps_conv_geom.cus_reserved): CUs left to a co-running kernel; the persistent grids use the rest
  int use_queue;              // host side only (ps_conv_geom.tile_queue): the persistent kernels draw their tiles from a ticket queue
  unsigned* queue;            // <.., Q = true> kernels: this launch's ticket counters (ps_queue_slot)
  // stride-2 data gradient, one launch per output parity class (conv_igemm_ws2_kernel<.., SPLIT>): the produced grid Ho x Wo is the
  // class's sub-grid, pixel (p', q') of it is pixel (oy + 2p', ox + 2q') of the full Hf x Wf gradient, and only the taps in tap_mask
  // (those that hit dy at integer positions for this parity) are staged and multiplied.  epi_M = rows of the full tensor.
  int tap_mask, oy, ox, Hf, Wf;
  // conv_gemm256_kernel<.., HEAD>: the narrow 1x1 head fused behind BN + ReLU (inference): per-block partial sums of
  // cam[m][c] = sum_ch round(max(acc * scale + shift, 0))[m][ch] * head_w[c][ch] go to head_part[(tn * 4 + wave column)][m][c]
  const float* head_w;
  float* head_part;
  int head_c;
  int m_off;                  // rows of the layer that precede this launch's row 0 (a gemm256 launch's tail rows go to a second launch on the
                              // gathered-tile kernels, with every tensor pointer advanced): only the dropout epilogue's image index needs it
  int tm0;                    // halo kernel: first pixel tile of this launch (a layer's tail tiles go to a second launch as 64-cout half tiles)
  // conv_igemm_halo_kernel<.., SK>: the launch's partial last round is cut along K ("stream-K").  Every block first takes its sk_dp whole
  // tiles of the static schedule (tiles [0, sk_tile0) = sk_dp whole rounds), then a contiguous share of the sk_lines = (ntiles - sk_tile0) * klines
  // K-lines of the remaining tiles: block s (XCD-remapped id) owns lines [sk_lines * s / nb, sk_lines * (s + 1) / nb).  A tile whose lines are spread
  // over several blocks is finished by the block whose part arrives LAST: parts go to f32 slabs (write-through stores), arrivals are counted per
  // (tile, consumer wave), the last arriver adds the parts up in part order -- bit-identical from run to run -- and runs the epilogue.
  int sk_dp, sk_tile0, sk_lines, sk_maxparts;
  float* sk_slabs;            // [tile - sk_tile0][part < sk_maxparts][consumer wave][MI * WI fragments][64 lanes] float4
  unsigned* sk_cnt;           // [tile - sk_tile0][consumer wave] arrival counters, zeroed by the host in front of the launch
  unsigned sk_slab_bytes;     // extent of sk_slabs (buffer descriptor)
  int stagger, stagger_phases; // halo kernel: block b starts its first tile ((b / 8) % stagger_phases) * stagger * 2048 shader cycles late (0 = off), see g_halo_stagger
  ps_epilogue epi;
};

// Block id -> (pixel tile, cout tile).  The blocks resident on one XCD at a time are a contiguous id range (see
// ps_xcd_remap), so ids are laid out in super-columns of G (=4) cout tiles: 64 consecutive blocks then cover 16 x 4 tiles
// (4 weight tiles + 16 activation tiles stream through that XCD's 4 MiB L2) instead of 2 x 32 (all of a wide layer's
// weights per 2 pixel tiles: 16 MB for the 2048->4096 1x1, re-streamed from the Infinity Cache 25 times per XCD).
__device__ __forceinline__ void ps_tile_of_block(int bid, int ntn, int ntm, int& tm, int& tn, int G) {
  if (G > 0 && ntn > G && ntn % G == 0) {
    const int per = G * ntm, sc = bid / per, r = bid - sc * per;
    tm = r / G;
    tn = sc * G + (r - tm * G);
  } else {
    tn = bid % ntn;
    tm = bid / ntn;
  }
}

// Epilogue storage tag of the split-bf16 path (PS_BF16X3).  A value is hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits); a
// tensor of C logical channels stores 2 C bf16 channels per pixel in blocks of 32 logical channels: [hi(32) | lo(32)] = one 128-byte
// K-line.  Weights are laid out the same way along their K axis, so every kernel STAGES a split tensor exactly like a plain 16-bit one (the
// loaders do not know the difference) and a staged K-line's two 64-byte halves are the hi and the lo fragments of the same 32 channels.
// The consumers multiply each K-line THREE times -- hi.hi, hi.lo, lo.hi on the fragments they hold anyway (Tr::split) -- with f32
// accumulation: x w up to the dropped lo.lo term at 3 MFMAs per product, with the LDS-DMA pieces and fragment reads of 2.  (Round 4's
// first version duplicated the hi plane -- [hi | lo | hi] against [hi | hi | lo] -- so that the K loops ran unchanged over 3 C channels:
// 6 bytes per value, 3 x the pieces and reads.)  sizeof == 1 so that the `sizeof(T) == 2` 16-bit fast paths do not take it.
struct bf16x3_t {};
struct f16x3_t {};  // the same on fp16 planes (PS_F16X3): 11 + 11 significant bits, fp16's range
template <typename T> struct is_split_t { static constexpr bool value = std::is_same<T, bf16x3_t>::value || std::is_same<T, f16x3_t>::value; };
template <typename T> struct plane_of { typedef T type; };               // the 16-bit element type of a split tensor's planes
template <> struct plane_of<bf16x3_t> { typedef __bf16 type; };
template <> struct plane_of<f16x3_t> { typedef _Float16 type; };

struct TraitsBF16 {
  typedef __bf16 elem;
  typedef __bf16 epi;
  static constexpr bool split = false;
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  }
};
struct TraitsBF16X3 {  // bf16 staging and MFMA; each K-line multiplied three times (hi.hi, hi.lo, lo.hi); split epilogue
  typedef __bf16 elem;
  typedef bf16x3_t epi;
  static constexpr bool split = true;
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) { TraitsBF16::mma(w, x, acc); }
};
struct TraitsF16X3 {
  typedef _Float16 elem;
  typedef f16x3_t epi;
  static constexpr bool split = true;
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  }
};
struct TraitsF16 {
  typedef _Float16 elem;
  typedef _Float16 epi;
  static constexpr bool split = false;
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  }
};
struct TraitsF32 {
  typedef float elem;
  typedef float epi;
  static constexpr bool split = false;
  // a 16-byte chunk holds 4 consecutive k of this lane's row; MFMA j consumes element j of both operands,
  // i.e. k = 4*(chunk) + j for lane group (lane>>4): every k of the K-line is visited exactly once.
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) {
    const f32x4 wf = __builtin_bit_cast(f32x4, w), xf = __builtin_bit_cast(f32x4, x);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], xf[j], acc, 0, 0, 0);
  }
};

template <typename T>
__device__ __forceinline__ void load16(const T* p, float* v) {
  ps_load8<T>(p, v);
  ps_load8<T>(p + 8, v + 8);
}
template <typename T>
__device__ __forceinline__ void store16(T* p, const float* v) {
  ps_store8<T>(p, v);
  ps_store8<T>(p + 8, v + 8);
}

#define GLDS16(gptr, lptr)                                                                              \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),               \
                                   (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// LDS-DMA through a buffer descriptor: 32-bit per-lane offset + scalar offset; an out-of-range lane (padding row,
// offset 0x80000000) makes the DMA write ZEROS into LDS (checked on gfx950: tools/probe_oob.hip).
#define BLDS16(rsrc, lptr, voff, soff)                                                                  \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lptr), 16, (int)(voff), (int)(soff), 0, 0)
constexpr unsigned PAD_ROW = 0x80000000u;

// Epilogue shared by the conv kernels: lane (frow = lane&15, g = lane>>4) owns pixels mbase + mi*16 + frow and the
// CH = 4*WI contiguous produced channels cbase + CH*g .. (acc[mi][i][r] = channel 4*i + r of that group).
// COLMAP (halo kernel): fragment mi holds an 8-row x 2-column patch of the tile instead of 16 consecutive pixels:
// pixel = mbase + (frow & 7) * Wo + 2 * mi + (frow >> 3).
// Eight consecutive tensor elements as loaded through a buffer descriptor (no conversion until they are needed).
template <typename T>
struct Raw8 {
  static constexpr bool X3 = is_split_t<T>::value, XH = std::is_same<T, f16x3_t>::value;
  static constexpr int NQ = X3 ? 2 : sizeof(T) / 2;  // 16-byte quads (split: the hi and the lo plane's)
  u32x4 q[NQ];
  // split tensors: voff addresses 8 hi values; their lo partners sit 64 bytes further (the other half of the 32-channel block), which the
  // compiler folds into the instruction's immediate offset
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int voff) {
    if constexpr (X3) {
      q[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
      q[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 64, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < NQ; ++k) q[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 16 * k, 0, 0);
    }
  }
  __device__ __forceinline__ void unpack(float* v) const {
    if constexpr (XH) {
      const f16x8 h = __builtin_bit_cast(f16x8, q[0]), l = __builtin_bit_cast(f16x8, q[1]);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]) + static_cast<float>(l[i]);
    } else if constexpr (X3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(q[0][i] << 16) + __uint_as_float(q[1][i] << 16);
        v[2 * i + 1] = __uint_as_float(q[0][i] & 0xffff0000u) + __uint_as_float(q[1][i] & 0xffff0000u);
      }
    } else if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = __uint_as_float(q[i >> 2][i & 3]);
    } else if constexpr (std::is_same<T, __bf16>::value) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(q[0][i] << 16);
        v[2 * i + 1] = __uint_as_float(q[0][i] & 0xffff0000u);
      }
    } else {
      const f16x8 h = __builtin_bit_cast(f16x8, q[0]);
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]);
    }
  }
  static __device__ __forceinline__ void store(__amdgpu_buffer_rsrc_t rs, int voff, const float* v) {
    if constexpr (X3) {
      u32x4 hi, lo;
      if constexpr (XH) {
        f16x8 h, l;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          h[i] = static_cast<_Float16>(v[i]);
          l[i] = static_cast<_Float16>(v[i] - static_cast<float>(h[i]));
        }
        hi = __builtin_bit_cast(u32x4, h);
        lo = __builtin_bit_cast(u32x4, l);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const uint32_t h0 = ps_f32_to_bf16(v[2 * i]), h1 = ps_f32_to_bf16(v[2 * i + 1]);
          const uint32_t l0 = ps_f32_to_bf16(v[2 * i] - __uint_as_float(h0 << 16)), l1 = ps_f32_to_bf16(v[2 * i + 1] - __uint_as_float(h1 << 16));
          hi[i] = h0 | (h1 << 16);
          lo[i] = l0 | (l1 << 16);
        }
      }
      // (Offsets of split stores never go through an SGPR scalar offset: a 128-bit buffer store whose SOFFSET is an SGPR reads its data registers
      // late, and hipcc (ROCm 7.2) only guards the immediate-SOFFSET form of that hazard -- the first version of this epilogue addressed its
      // planes that way, the compiler scheduled the next row's first v_mov straight behind the store and one dword came out as garbage in a few
      // rows per launch; found by tests/test_split_gpu.py on conv_gemm256_kernel, gfx950.)
      __builtin_amdgcn_raw_buffer_store_b128(hi, rs, voff, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b128(lo, rs, voff + 64, 0, 0);
    } else if constexpr (sizeof(T) == 4) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const u32x4 d = {__float_as_uint(v[4 * k]), __float_as_uint(v[4 * k + 1]), __float_as_uint(v[4 * k + 2]), __float_as_uint(v[4 * k + 3])};
        __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff + 16 * k, 0, 0);
      }
    } else if constexpr (std::is_same<T, __bf16>::value) {
      u32x4 d;
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = ps_f32_to_bf16(v[2 * i]) | (static_cast<uint32_t>(ps_f32_to_bf16(v[2 * i + 1])) << 16);
      __builtin_amdgcn_raw_buffer_store_b128(d, rs, voff, 0, 0);
    } else {
      f16x8 h;
#pragma unroll
      for (int i = 0; i < 8; ++i) h[i] = static_cast<_Float16>(v[i]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs, voff, 0, 0);
    }
  }
};

// Tile epilogue on a wave's MI x WI accumulator fragments: [+ add0] [-> out_raw], then BN+ReLU[+dropout] or the ReLU-backward mask
// [+ add1] -> out.  Every global LOAD of the MI rows is issued before the first STORE and the arithmetic is done in place on the
// accumulators: vmcnt counts loads and stores in one in-order queue, so a load issued behind a store cannot be consumed before that
// store has been acknowledged by L2 -- with one row's loads behind the previous row's stores the epilogue took MI store round trips
// per tile.  All tensors are addressed through buffer descriptors of epi_M rows: a lane's row offset is one VGPR + a scalar step,
// rows past the end are dropped by the range check (no predication, no 64-bit address arithmetic), and no load is left pending on
// any path, so the next tile's first MFMAs need no vmcnt wait.  Layers with dropout (per-row multipliers; b6 / b7 only) keep the
// row-by-row order.  (f32 tensors: 8 channels at a time to bound the registers.)
template <typename T, int MI, int WI, int MAP = 0, int CW = (sizeof(T) == 2 ? 4 * WI : 8)>  // MAP: 0 rows = pixels in order, 1 halo kernel's 8 x 2 patches, 2 parity sub-grid
__device__ __forceinline__ void conv_epilogue(const IgemmArgs& a, f32x4 (&acc)[MI][WI], int mbase, int cbase, int lane) {
  constexpr bool COLMAP = MAP == 1;
  constexpr int CH = 4 * WI;  // 16 or 8 channels per lane, handled CW at a time
  constexpr bool X3 = is_split_t<T>::value;  // split tensors: blocks of 32 logical channels stored [hi(32) | lo(32)] (see bf16x3_t)
  typedef typename plane_of<T>::type PT;
  auto sch = [](int c) { return X3 ? ((c >> 5) << 6) + (c & 31) : c; };  // logical channel -> stored channel (of its hi half)
  constexpr int NO = CW / 8, ES = X3 ? 2 : (int)sizeof(T);
  static_assert(CW % 8 == 0 && CH % CW == 0, "channel chunk");
  const int frow = lane & 15, g = lane >> 4;
  const ps_epilogue& e = a.epi;
  const int row0 = COLMAP ? mbase + (frow & 7) * a.Wo + (frow >> 3) : mbase + frow;  // row of fragment mi: row0 + mi * RSTEP
  constexpr int RSTEP = COLMAP ? 2 : 16;
  // MAP 2: fragment row mi is pixel m = row0 + 16 mi of the class's sub-grid -> row rrow[mi] of the full tensor (epi_M = dropped)
  int rrow[MAP == 2 ? MI : 1];
  if constexpr (MAP == 2) {
    const int hw = a.Ho * a.Wo;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int m = row0 + mi * RSTEP;
      const int n = m / hw, rem = m - n * hw, pp = rem / a.Wo, qq = rem - pp * a.Wo;
      rrow[mi] = m < a.M ? (n * a.Hf + a.oy + 2 * pp) * a.Wf + a.ox + 2 * qq : a.epi_M;
    }
  }
  auto roff = [&](int mi, int ldc, int cb) {  // byte offset of fragment row mi, channel cb, in a tensor of row stride ldc
    if constexpr (MAP == 2) return (rrow[mi] * ldc + sch(cb)) * ES;
    else return ((row0 + mi * RSTEP) * ldc + sch(cb)) * ES;
  };
  auto rsrc = [&](const void* base, int ldc) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, a.epi_M * ldc * ES, 0x00020000);
  };
#pragma unroll
  for (int c0 = 0; c0 < CH; c0 += CW) {
    const int cb = cbase + CH * g + c0;
    float sc[CW], sh[CW];
    auto load_vec = [&](const float* p, float* v) {  // CW consecutive floats (16-byte aligned: cb is a multiple of 8)
#pragma unroll
      for (int i = 0; i < CW; i += 4) {
        const float4 q = *reinterpret_cast<const float4*>(p + i);
        v[i] = q.x; v[i + 1] = q.y; v[i + 2] = q.z; v[i + 3] = q.w;
      }
    };
#pragma unroll
    for (int i = 0; i < CW; ++i) sc[i] = 1.f, sh[i] = 0.f;
    if (e.scale && e.mode != PS_EPI_NONE) load_vec(e.scale + cb, sc);
    if (e.shift && e.mode == PS_EPI_BNRELU) load_vec(e.shift + cb, sh);
    // Dropout (b6 / b7): the multipliers are per (image, channel) and >= 0, so when every row of this wave lies in ONE image they fold
    // into the per-channel affine -- max(x*sc + sh, 0) * dm = max(x*(sc*dm) + sh*dm, 0) -- and the wave takes the batched path below at
    // no extra cost.  Waves that straddle two images keep the row-by-row path.
    float dm[CW];
    bool fold = false;
    if (e.drop && e.mode != PS_EPI_NONE && MAP != 2) {
      const int hw = a.Ho * a.Wo;
      const int mf = row0 < a.epi_M ? row0 : 0, ml0 = row0 + (MI - 1) * RSTEP, ml = ml0 < a.epi_M ? ml0 : mf;
      const int nf = (mf + a.m_off) / hw, nl = (ml + a.m_off) / hw, n0 = __builtin_amdgcn_readfirstlane(nf);
      fold = __builtin_amdgcn_ballot_w64(nf != n0 || nl != n0) == 0;
      if (fold) load_vec(e.drop + (long long)n0 * a.Cd + cb, dm);
    }
    // Row loads run LW rows ahead of their use (a row's registers are refilled with row + LW as soon as it has been consumed):
    // every load still precedes every store, with LW instead of MI rows of operands live.
    // (split tensors: two quads per row and hi / lo temporaries at the stores.  A window of all MI = 7 rows -- every load of the tile in flight at once --
    // was measured on the halo kernel: 12.9 k instead of 13.1 k epilogue cycles, +-0.5 % per layer: the epilogue is throughput-, not latency-bound; NOTES R5.7)
    constexpr int LW = X3 ? (MI < 2 ? MI : 2) : (MI < 4 ? MI : 4);
    auto add_rows = [&](const void* base, int ldc) {  // acc += tensor rows
      const __amdgpu_buffer_rsrc_t rs = rsrc(base, ldc);
      Raw8<T> t[LW][NO];
#pragma unroll
      for (int mi = 0; mi < LW; ++mi)
#pragma unroll
        for (int o = 0; o < NO; ++o) t[mi][o].load(rs, roff(mi, ldc, cb) + 8 * o * ES);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          float f[8];
          t[mi % LW][o].unpack(f);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[mi][(c0 + 8 * o + i) / 4][i & 3] += f[i];
          if (mi + LW < MI) t[mi % LW][o].load(rs, roff(mi + LW, ldc, cb) + 8 * o * ES);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // hi_base (split types, `out` only): ALSO the hi halves as a plain 16-bit tensor of row stride hi_ldc -- what the weight gradient, which
    // contracts over pixels and wants contiguous hi rows, reads instead of gathering every other 64 bytes of the split tensor (ps_epilogue.out_hi)
    auto store_rows = [&](void* base, int ldc, bool bnrelu, void* hi_base = nullptr, int hi_ldc = 0) {  // bnrelu: max(acc * sc + sh, 0) on the way out (a select, not a branch)
      const __amdgpu_buffer_rsrc_t rs = rsrc(base, ldc);
      [[maybe_unused]] const __amdgpu_buffer_rsrc_t rs_hi =
          __builtin_amdgcn_make_buffer_rsrc(X3 && hi_base ? hi_base : base, 0, X3 && hi_base ? a.epi_M * hi_ldc * 2 : 0, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          float f[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = acc[mi][(c0 + 8 * o + i) / 4][i & 3];
            const float y = fmaxf(x * sc[8 * o + i] + sh[8 * o + i], 0.f);
            f[i] = bnrelu ? y : x;
          }
          Raw8<T>::store(rs, roff(mi, ldc, cb) + 8 * o * ES, f);
          if constexpr (X3) {
            if (hi_base) {
              const int row = MAP == 2 ? rrow[MAP == 2 ? mi : 0] : row0 + mi * RSTEP;
              Raw8<PT>::store(rs_hi, (row * hi_ldc + cb + 8 * o) * 2, f);  // RNE cast = the split tensor's hi half
            }
          }
        }
    };
    if (e.add0) add_rows(e.add0, e.ldc_add0);
    __builtin_amdgcn_sched_barrier(0);  // keep the phases apart: a hoisted second batch of loads would double the live registers
    // scale / shift have arrived by now; "use" them before the first store so that the compiler's wait for them does not become
    // a vmcnt(0) behind the stores below
    if (fold) {
#pragma unroll
      for (int i = 0; i < CW; ++i) sc[i] *= dm[i], sh[i] *= dm[i];
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) asm volatile("" ::"v"(sc[i]), "v"(sh[i]));
    if (e.out_raw) store_rows(e.out_raw, e.ldc_raw, false);
    if (e.mode == PS_EPI_NONE) continue;  // next channel chunk
    if (e.drop && !fold) {
      // dropout, rows of more than one image: per-row multipliers, row by row
      const int hw = a.Ho * a.Wo;
      auto ld8 = [&](const void* base, int ldc, int m, int c, float* v) {  // 8 channels c.. of row m
        if constexpr (X3) {
          const PT* p = reinterpret_cast<const PT*>(base) + (long long)m * ldc + sch(c);
          float lo[8];
          ps_load8<PT>(p, v);
          ps_load8<PT>(p + 32, lo);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += lo[i];
        } else {
          ps_load8<T>(reinterpret_cast<const T*>(base) + (long long)m * ldc + c, v);
        }
      };
      auto st8 = [&](void* base, int ldc, int m, int c, const float* v) {
        if constexpr (X3) {
          PT* p = reinterpret_cast<PT*>(base) + (long long)m * ldc + sch(c);
          float hi[8], lo[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            hi[i] = static_cast<float>(static_cast<PT>(v[i]));
            lo[i] = v[i] - hi[i];
          }
          ps_store8<PT>(p, hi);
          ps_store8<PT>(p + 32, lo);
          if (e.out_hi) ps_store8<PT>(reinterpret_cast<PT*>(e.out_hi) + (long long)m * e.ldc_hi + c, hi);
        } else {
          ps_store8<T>(reinterpret_cast<T*>(base) + (long long)m * ldc + c, v);
        }
      };
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int m = row0 + mi * RSTEP;  // (MAP 2 never comes with dropout: the host keeps such launches on the plain path)
        if (m >= a.epi_M) continue;
        const float* d = e.drop + (long long)((m + a.m_off) / hw) * a.Cd + cb;
        float v[CW];
#pragma unroll
        for (int i = 0; i < CW; ++i) v[i] = acc[mi][(c0 + i) / 4][i & 3];
        if (e.mode == PS_EPI_BNRELU) {
#pragma unroll
          for (int i = 0; i < CW; ++i) v[i] = fmaxf(v[i] * sc[i] + sh[i], 0.f) * d[i];
        } else {
          float ms[CW];
#pragma unroll
          for (int o = 0; o < NO; ++o) ld8(e.mask_src, e.ldc_mask, m, cb + 8 * o, ms + 8 * o);
#pragma unroll
          for (int i = 0; i < CW; ++i) v[i] = ms[i] > 0.f ? v[i] * sc[i] * d[i] : 0.f;
          if (e.add1) {
            float t[CW];
#pragma unroll
            for (int o = 0; o < NO; ++o) ld8(e.add1, e.ldc_add1, m, cb + 8 * o, t + 8 * o);
#pragma unroll
            for (int i = 0; i < CW; ++i) v[i] += t[i];
          }
        }
#pragma unroll
        for (int o = 0; o < NO; ++o) st8(e.out, e.ldc_out, m, cb + 8 * o, v + 8 * o);
      }
      continue;
    }
    if (e.mode == PS_EPI_RELUBWD) {  // in place: acc = mask > 0 ? acc * sc : 0 [+ add1]
      const __amdgpu_buffer_rsrc_t rs = rsrc(e.mask_src, e.ldc_mask);
      Raw8<T> t[LW][NO];
#pragma unroll
      for (int mi = 0; mi < LW; ++mi)
#pragma unroll
        for (int o = 0; o < NO; ++o) t[mi][o].load(rs, roff(mi, e.ldc_mask, cb) + 8 * o * ES);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          float ms[8];
          t[mi % LW][o].unpack(ms);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float x = acc[mi][(c0 + 8 * o + i) / 4][i & 3] * sc[8 * o + i];
            acc[mi][(c0 + 8 * o + i) / 4][i & 3] = ms[i] > 0.f ? x : 0.f;
          }
          if (mi + LW < MI) t[mi % LW][o].load(rs, roff(mi + LW, e.ldc_mask, cb) + 8 * o * ES);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (e.add1) add_rows(e.add1, e.ldc_add1);
    }
    __builtin_amdgcn_sched_barrier(0);
    store_rows(e.out, e.ldc_out, e.mode == PS_EPI_BNRELU, e.out_hi, e.ldc_hi);
  }
}

// Two-stage kernel.  Block tile BM pixels x BN couts, 4 waves arranged WMW x WNW; each wave owns MI = BM/(16*WMW)
// pixel fragments and WI = BN/(16*WNW) cout fragments.  Instantiated as
//   128 x 128, waves 2x2 (64x64 per wave)   -- general large problems
//   112 x 128, waves 1x4 (112x32 per wave)  -- 7-fragment pixel tile: with 28x28 / 56x56 / 112x112 feature maps the
//                                              pixel count is a multiple of 49*16, so 112-pixel tiles give every CU
//                                              the same number of blocks (a 128-pixel tiling leaves a ~15% tail)
//   128 x 64,  waves 4x1 (32x64 per wave)   -- narrow layers / small problems
// STG: 0 registers, 1 global_load_lds, 2 buffer_load ... lds (default)
template <typename Tr, int BM, int BN, int WMW, int WNW, int STG>
__global__ __launch_bounds__(64 * WMW * WNW) void conv_igemm_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  constexpr int NW = WMW * WNW;  // waves per block (4 or 8)
  constexpr int MI = BM / (16 * WMW);        // pixel fragments per wave
  constexpr int WI = BN / (16 * WNW);        // cout fragments per wave
  constexpr int WM = 16 * MI, WN = 16 * WI;  // wave tile
  constexpr int AINS = BM / 8, BINS = BN / 8;  // 8-row LDS-DMA instructions per stage (dealt round-robin to the waves)
  constexpr int AJ = (AINS + NW - 1) / NW, BJ = (BINS + NW - 1) / NW;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = ps_xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  ps_tile_of_block(bid, a.ntn, a.ntm, tm, tn, a.supertile);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave / WNW, wn = wave % WNW;

  // ---------------- staging state ----------------
  // LDS row R = t*8 + (lane>>3) of instruction t = j*NW + wave; the lane moves source chunk (lane&7)^(R&7) to position lane&7
  const int srow = lane >> 3;
  const int chunk_off = ((lane & 7) ^ srow) << 4;
  int py[AJ], px[AJ], nb[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + (j * NW + wave) * 8 + srow;
    if (m < a.M && j * NW + wave < AINS) {
      const int hw = a.Ho * a.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int p = rem / a.Wo, q = rem - p * a.Wo;
      py[j] = p * a.mul;
      px[j] = q * a.mul;
      nb[j] = n * a.Hs * a.Ws;
    } else {
      py[j] = -(1 << 20);  // never valid
      px[j] = 0;
      nb[j] = 0;
    }
  }
  // weight rows: LDS row Rb of the wave column wg = Rb / WN holds cout n0 + wg*WN + perm(Rb % WN), where
  // perm(16*i + rho) = 4*WI*(rho>>2) + 4*i + (rho&3)  (so that accumulator register r of fragment i, lane group g
  // is cout 4*WI*g + 4*i + r: 4*WI contiguous couts per lane)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
  unsigned woff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int rb = (j * NW + wave) * 8 + srow;
    const int wg = rb / WN, within = rb % WN, fi = within >> 4, rho = within & 15;
    const int cout = n0 + wg * WN + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
    woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
  }
  unsigned aoff[AJ];  // byte offset of each pixel row for the current tap, PAD_ROW if the tap falls outside
  int tap = 0, kl = 0;
  int wk = 0;  // running K byte offset in a weight row

  auto tap_offsets = [&](int t) {
    const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
    const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
    const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int yn = py[j] + dy, xn = px[j] + dx;
      const int y = yn >> a.div_shift, x = xn >> a.div_shift;
      const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
      aoff[j] = ok ? (unsigned)((nb[j] + y * a.Ws + x) * (int)a.pix_bytes) + chunk_off : PAD_ROW;
    }
  };
  tap_offsets(0);

  u32x4 ra[AJ], rb_[BJ];  // register staging (STG == 0)
  auto stage_issue = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sb = smem + buf * STAGE + A_BYTES;
    const int ko = kl * 128;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int t = j * NW + wave;
      if (AINS % NW != 0 && t >= AINS) continue;
      if constexpr (STG == 2) BLDS16(rsA, sa + t * 1024, aoff[j], ko);
      else {
        const unsigned char* g = aoff[j] != PAD_ROW ? a.src + aoff[j] + ko : g_zero_line + (lane & 7) * 16;
        if constexpr (STG == 1) GLDS16(g, sa + t * 1024);
        else ra[j] = *reinterpret_cast<const u32x4*>(g);
      }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      const int t = j * NW + wave;
      if (BINS % NW != 0 && t >= BINS) continue;
      if constexpr (STG == 2) BLDS16(rsB, sb + t * 1024, woff[j], wk);
      else {
        const unsigned char* g = a.wgt + woff[j] + wk;
        if constexpr (STG == 1) GLDS16(g, sb + t * 1024);
        else rb_[j] = *reinterpret_cast<const u32x4*>(g);
      }
    }
    wk += 128;
    if (++kl == a.klines) {
      kl = 0;
      if (++tap < a.taps) tap_offsets(tap);
    }
  };
  auto stage_commit = [&](int buf) {  // register staging only: registers -> LDS
    if constexpr (STG == 0) {
      unsigned char* sa = smem + buf * STAGE + lane * 16;
      unsigned char* sb = smem + buf * STAGE + A_BYTES + lane * 16;
#pragma unroll
      for (int j = 0; j < AJ; ++j)
        if (j * NW + wave < AINS) *reinterpret_cast<u32x4*>(sa + (j * NW + wave) * 1024) = ra[j];
#pragma unroll
      for (int j = 0; j < BJ; ++j)
        if (j * NW + wave < BINS) *reinterpret_cast<u32x4*>(sb + (j * NW + wave) * 1024) = rb_[j];
    }
  };

  // ---------------- accumulate ----------------
  f32x4 acc[MI][WI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4;
  const int nsteps = a.taps * a.klines;
  // fragment byte offsets inside a stage (row&7 == lane&7 for both operands)
  const int xfrag = (wm * WM + frow) * 128, wfrag = A_BYTES + (wn * WN + frow) * 128;
  const int sw = lane & 7;

  stage_issue(0);
  stage_commit(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps && !(PS_ABLATE(a.ablate) == 1 && s >= 1)) stage_issue(cur ^ 1);
    const unsigned char* st = smem + cur * STAGE;
    u32x4 wf[2][WI], xf[2][MI];  // all fragments of the K-line first: the reads overlap the MFMAs of the first half
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((g + 4 * kk) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < WI; ++i) wf[kk][i] = *reinterpret_cast<const u32x4*>(st + wfrag + i * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[kk][mi] = *reinterpret_cast<const u32x4*>(st + xfrag + mi * 2048 + coff);
    }
    if constexpr (Tr::split) {  // the K-line's halves are the hi and the lo fragments of 32 channels: hi.hi + hi.lo + lo.hi
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < WI; ++i) Tr::mma(wf[t == 2 ? 1 : 0][i], xf[t == 1 ? 1 : 0][mi], acc[mi][i]);
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < WI; ++i) Tr::mma(wf[kk][i], xf[kk][mi], acc[mi][i]);
    }
    if (s + 1 < nsteps) stage_commit(cur ^ 1);
    __syncthreads();
  }

  conv_epilogue<typename Tr::epi, MI, WI>(a, acc, m0 + wm * WM, n0 + wn * WN, lane);
}

// ------------------------------------------------------------------------------------------------
// 256-pixel x 128-cout tile, 8 waves (4 x 2 of 64x64), THREE LDS stages of 48 KiB: the LDS-DMA of K-step s+2 is
// issued before the MFMAs of step s and stays in flight across the (raw) barrier; a counted `s_waitcnt vmcnt(6)`
// retires only step s+1's six loads.  Same operand images, fragment reads and epilogue as the 2-stage kernel.
// ------------------------------------------------------------------------------------------------
template <typename Tr>
__global__ __launch_bounds__(512) void conv_igemm3_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  constexpr int BM = 256, BN = 128, WM = 64, MI = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = ps_xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  ps_tile_of_block(bid, a.ntn, a.ntm, tm, tn, a.supertile);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;

  const int srow = lane >> 3;
  const int chunk_off = ((lane & 7) ^ srow) << 4;
  int py[4], px[4], nb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + wave * 32 + j * 8 + srow;
    if (m < a.M) {
      const int hw = a.Ho * a.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int p = rem / a.Wo, q = rem - p * a.Wo;
      py[j] = p * a.mul;
      px[j] = q * a.mul;
      nb[j] = n * a.Hs * a.Ws;
    } else {
      py[j] = -(1 << 20);
      px[j] = 0;
      nb[j] = 0;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
  unsigned woff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rb = wave * 16 + j * 8 + srow;
    const int within = rb & 63, fi = within >> 4, rho = within & 15;
    const int cout = n0 + (rb & ~63) + 16 * (rho >> 2) + 4 * fi + (rho & 3);
    woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
  }
  unsigned aoff[4];
  int tap = 0, kl = 0;
  int wk = 0;

  auto tap_offsets = [&](int t) {
    const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
    const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
    const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yn = py[j] + dy, xn = px[j] + dx;
      const int y = yn >> a.div_shift, x = xn >> a.div_shift;
      const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
      aoff[j] = ok ? (unsigned)((nb[j] + y * a.Ws + x) * (int)a.pix_bytes) + chunk_off : PAD_ROW;
    }
  };
  tap_offsets(0);

  auto stage_issue = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE + (wave * 32) * 128;
    unsigned char* sb = smem + buf * STAGE + A_BYTES + (wave * 16) * 128;
    const int ko = kl * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) BLDS16(rsA, sa + j * 1024, aoff[j], ko);
#pragma unroll
    for (int j = 0; j < 2; ++j) BLDS16(rsB, sb + j * 1024, woff[j], wk);
    wk += 128;
    if (++kl == a.klines) {
      kl = 0;
      if (++tap < a.taps) tap_offsets(tap);
    }
  };

  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4;
  const int nsteps = a.taps * a.klines;
  const int xfrag = (wm * WM + frow) * 128, wfrag = A_BYTES + (wn * 64 + frow) * 128;
  const int sw = lane & 7;

  stage_issue(0);
  if (nsteps > 1) {
    stage_issue(1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  int cur = 0, nxt2 = 2;  // buffer of step s, buffer for step s+2
  for (int s = 0; s < nsteps; ++s) {
    if (s + 2 < nsteps) stage_issue(nxt2);
    const unsigned char* st = smem + cur * STAGE;
    u32x4 wf[2][4], xf[2][MI];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((g + 4 * kk) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[kk][i] = *reinterpret_cast<const u32x4*>(st + wfrag + i * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[kk][mi] = *reinterpret_cast<const u32x4*>(st + xfrag + mi * 2048 + coff);
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) Tr::mma(wf[kk][i], xf[kk][mi], acc[mi][i]);
    // step s+1 must have landed (its loads are older than step s+2's six) before anyone reads it
    if (s + 2 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    cur = (cur == 2) ? 0 : cur + 1;
    nxt2 = (nxt2 == 2) ? 0 : nxt2 + 1;
  }
  conv_epilogue<typename Tr::epi, MI, 4>(a, acc, m0 + wm * WM, n0 + wn * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// Ping-pong kernel: 8 waves = two groups of four; group g computes the pixel rows [g*GM, (g+1)*GM) of a (2*GM) x 128
// tile against the SAME weight tile.  The groups run half a K-step apart, separated by block barriers:
//     phase 2k   : group 0 issues the LDS-DMA of step k+2 and reads the fragments of step k | group 1 runs the MFMAs of step k-1
//     phase 2k+1 : group 0 runs the MFMAs of step k                                        | group 1 issues step k+2, reads step k
// so each SIMD always has one wave feeding the matrix pipe while its partner fetches, fragments are complete in
// registers before a wave's MFMA burst starts, and the DMA of a step has two phases to land (3-stage LDS ring, counted
// vmcnt, raw s_barrier).  GM = 128: groups are 2x2 waves of 64x64; GM = 112: 1x4 waves of 112x32 (balanced tiling for
// 28x28-derived pixel counts, see conv_igemm_kernel).
// ------------------------------------------------------------------------------------------------
template <typename Tr, int GM>
__global__ __launch_bounds__(512) void conv_igemm_pp_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  constexpr int BM = 2 * GM, BN = 128;
  constexpr int WMW = (GM == 128) ? 2 : 1, WNW = 4 / WMW;
  constexpr int MI = GM / (16 * WMW), WI = BN / (16 * WNW);
  constexpr int WM = 16 * MI, WN = 16 * WI;
  constexpr int AINS = BM / 8, BINS = BN / 8, AJ = (AINS + 7) / 8, BJ = BINS / 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wl = wave & 3;
  const int bid = ps_xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  ps_tile_of_block(bid, a.ntn, a.ntm, tm, tn, a.supertile);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wl / WNW, wn = wl % WNW;

  const int srow = lane >> 3;
  const int chunk_off = ((lane & 7) ^ srow) << 4;
  int py[AJ], px[AJ], nb[AJ];
#pragma unroll
  for (int j = 0; j < AJ; ++j) {
    const int m = m0 + (j * 8 + wave) * 8 + srow;
    if (m < a.M && j * 8 + wave < AINS) {
      const int hw = a.Ho * a.Wo;
      const int n = m / hw, rem = m - n * hw;
      const int p = rem / a.Wo, q = rem - p * a.Wo;
      py[j] = p * a.mul;
      px[j] = q * a.mul;
      nb[j] = n * a.Hs * a.Ws;
    } else {
      py[j] = -(1 << 20);
      px[j] = 0;
      nb[j] = 0;
    }
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
  unsigned woff[BJ];
#pragma unroll
  for (int j = 0; j < BJ; ++j) {
    const int rb = (j * 8 + wave) * 8 + srow;
    const int wg = rb / WN, within = rb % WN, fi = within >> 4, rho = within & 15;
    const int cout = n0 + wg * WN + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
    woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
  }
  unsigned aoff[AJ];
  int tap = 0, kl = 0, wk = 0;
  auto tap_offsets = [&](int t) {
    const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
    const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
    const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int yn = py[j] + dy, xn = px[j] + dx;
      const int y = yn >> a.div_shift, x = xn >> a.div_shift;
      const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
      aoff[j] = ok ? (unsigned)((nb[j] + y * a.Ws + x) * (int)a.pix_bytes) + chunk_off : PAD_ROW;
    }
  };
  tap_offsets(0);
  // every wave issues exactly NLOAD loads per step (rows beyond the tile are issued as zero-filled padding so that
  // the vmcnt bookkeeping is the same literal for all waves)
  constexpr int NLOAD = AJ + BJ;
  auto stage_issue = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE;
    unsigned char* sb = smem + buf * STAGE + A_BYTES;
    const int ko = kl * 128;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
      const int t = j * 8 + wave;
      if (AINS % 8 != 0 && t >= AINS) BLDS16(rsA, smem + 3 * STAGE, PAD_ROW, 0);  // dummy (scratch KiB after the ring)
      else BLDS16(rsA, sa + t * 1024, aoff[j], ko);
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) BLDS16(rsB, sb + (j * 8 + wave) * 1024, woff[j], wk);
    wk += 128;
    if (++kl == a.klines) {
      kl = 0;
      if (++tap < a.taps) tap_offsets(tap);
    }
  };

  f32x4 acc[MI][WI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 wf[2][WI], xf[2][MI];

  const int frow = lane & 15, g = lane >> 4;
  const int nsteps = a.taps * a.klines;
  const int xfrag = (grp * GM + wm * WM + frow) * 128, wfrag = A_BYTES + (wn * WN + frow) * 128;
  const int sw = lane & 7;

  auto read_frags = [&](int buf) {
    const unsigned char* st = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((g + 4 * kk) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < WI; ++i) wf[kk][i] = *reinterpret_cast<const u32x4*>(st + wfrag + i * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[kk][mi] = *reinterpret_cast<const u32x4*>(st + xfrag + mi * 2048 + coff);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  auto mma_all = [&]() {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int i = 0; i < WI; ++i) Tr::mma(wf[kk][i], xf[kk][mi], acc[mi][i]);
    __builtin_amdgcn_s_setprio(0);
  };
  auto wait_older = [&](bool more_in_flight) {  // all but the newest step's loads of this wave have landed
    if (more_in_flight) {
      if constexpr (NLOAD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  stage_issue(0);
  if (nsteps > 1) stage_issue(1);
  wait_older(nsteps > 1);
  __builtin_amdgcn_s_barrier();
  int cur = 0, nx2 = 2;
  for (int k = 0; k < nsteps; ++k) {
    // ---- phase 2k
    if (grp == 0) {
      if (k + 2 < nsteps) stage_issue(nx2);
      read_frags(cur);
    } else if (k > 0) {
      mma_all();
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2k+1
    if (grp == 0) {
      mma_all();
    } else {
      if (k + 2 < nsteps) stage_issue(nx2);
      read_frags(cur);
    }
    wait_older(k + 2 < nsteps);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    cur = (cur == 2) ? 0 : cur + 1;
    nx2 = (nx2 == 2) ? 0 : nx2 + 1;
  }
  if (grp == 1) mma_all();
  conv_epilogue<typename Tr::epi, MI, WI>(a, acc, m0 + grp * GM + wm * WM, n0 + wn * WN, lane);
}

// Epilogue of the wave-specialised kernel (bf16): the per-channel scale/shift and the tile-shaped operands (residual
// addend / ReLU-mask source / second addend) were brought into LDS by the LOADER waves while the consumers were still
// in their last K-line, so the consumers' epilogue has no global-load latency in it (measured: ~11 us of the ~48 us a
// 512-cout block lives were serialized epilogue/prologue latency).  Tile image: [row][256 B], 16-byte chunk c of row R
// at position c ^ (R & 15) (conflict-free 16-byte reads of 16 different rows).
template <typename T, int MI, int WI>
__device__ __forceinline__ void conv_epilogue_lds(const IgemmArgs& a, f32x4 (&acc)[MI][WI], int mbase, int rbase, int cbase, int clocal,
                                                  int lane, const float* prm, const unsigned char* t1, const unsigned char* t2) {
  constexpr int CH = 4 * WI, NV = CH / 8;
  const int frow = lane & 15, g = lane >> 4;
  const ps_epilogue& e = a.epi;
  const int cb = cbase + CH * g, cl = clocal + CH * g;  // global / tile-local first channel of this lane
  const bool bwd = e.mode == PS_EPI_RELUBWD;
  auto tile8 = [&](const unsigned char* t, int R, int o, float* v) {  // 8 channels cl + 8o.. of tile row R
    const int c = (cl + 8 * o) >> 3;
    ps_load8<T>(reinterpret_cast<const T*>(t + R * 256 + ((c ^ (R & 15)) << 4)), v);
  };
  // which operand sits in which tile (the loader uses the same rule)
  const unsigned char* t_add0 = nullptr; const unsigned char* t_mask = nullptr; const unsigned char* t_add1 = nullptr;
  if (!bwd) t_add0 = e.add0 ? t1 : nullptr;
  else { t_mask = t1; if (e.add0) t_add0 = t2; else if (e.add1) t_add1 = t2; }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int R = rbase + mi * 16 + frow, m = mbase + mi * 16 + frow;
    if (m >= a.epi_M) continue;
#pragma unroll
    for (int o = 0; o < NV; ++o) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = acc[mi][(8 * o + i) >> 2][(8 * o + i) & 3];
      if (e.add0) {
        float t[8];
        if (t_add0) tile8(t_add0, R, o, t);
        else ps_load8<T>(reinterpret_cast<const T*>(e.add0) + (long long)m * e.ldc_add0 + cb + 8 * o, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += t[i];
      }
      if (e.out_raw) ps_store8<T>(reinterpret_cast<T*>(e.out_raw) + (long long)m * e.ldc_raw + cb + 8 * o, v);
      if (e.mode == PS_EPI_NONE) continue;
      float sc[8], dm[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) sc[i] = prm[cl + 8 * o + i];
      if (e.drop) {
        const float* d = e.drop + (long long)(m / (a.Ho * a.Wo)) * a.Cd + cb + 8 * o;
#pragma unroll
        for (int i = 0; i < 8; ++i) dm[i] = d[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) dm[i] = 1.f;
      }
      if (!bwd) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i] * sc[i] + prm[128 + cl + 8 * o + i], 0.f) * dm[i];
      } else {
        float ms[8];
        tile8(t_mask, R, o, ms);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ms[i] > 0.f ? v[i] * sc[i] * dm[i] : 0.f;
        if (e.add1) {
          float t[8];
          if (t_add1) tile8(t_add1, R, o, t);
          else ps_load8<T>(reinterpret_cast<const T*>(e.add1) + (long long)m * e.ldc_add1 + cb + 8 * o, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] += t[i];
        }
      }
      ps_store8<T>(reinterpret_cast<T*>(e.out) + (long long)m * e.ldc_out + cb + 8 * o, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised kernel: 128 x 128 tile, 8 waves = 4 CONSUMER waves (2x2 of 64x64: ds_read + MFMA only) + 4 LOADER
// waves (LDS-DMA issue only).  Issuing one 1-KiB LDS-DMA costs a wave ~100-180 cycles of its in-order issue stream
// (measured: the 4-wave kernel spends more issue time on its 8 DMAs per K-step than on its 32 MFMAs); moving the
// DMAs to partner waves on the same SIMD takes them out of the MFMA waves' streams.  <= 128 VGPRs so that two blocks
// (2 consumers + 2 loaders per SIMD) stay resident per CU; two LDS stages, one block barrier per K-step.
// ------------------------------------------------------------------------------------------------
template <typename Tr, int BM>  // BM = 128: consumers 2x2 of 64x64; BM = 112: consumers 1x4 of 112x32 (balanced tiling, see above)
__global__ __launch_bounds__(512, 4) void conv_igemm_ws_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  constexpr int BN = 128, WMW = (BM == 128) ? 2 : 1, WNW = 4 / WMW;
  constexpr int MI = BM / (16 * WMW), WI = BN / (16 * WNW), WM = 16 * MI, WN = 16 * WI;
  constexpr int AINS = BM / 8;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = ps_xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  ps_tile_of_block(bid, a.ntn, a.ntm, tm, tn, a.supertile);
  const int m0 = tm * BM, n0 = tn * BN;
  const int nsteps = a.taps * a.klines;
  // bf16: epilogue operands travel through LDS (see conv_epilogue_lds); which tensor goes to which tile:
  constexpr bool LDS_EPI = sizeof(T) == 2 && std::is_same<T, typename Tr::epi>::value;  // (not for split tensors)
  const ps_epilogue& ep = a.epi;
  const bool ep_bwd = ep.mode == PS_EPI_RELUBWD;
  const void* tile1_src = ep_bwd ? ep.mask_src : ep.add0;
  const int tile1_ldc = ep_bwd ? ep.ldc_mask : ep.ldc_add0;
  const void* tile2_src = ep_bwd ? (ep.add0 ? ep.add0 : ep.add1) : nullptr;
  const int tile2_ldc = ep_bwd ? (ep.add0 ? ep.ldc_add0 : ep.ldc_add1) : 0;
  float* prm = reinterpret_cast<float*>(smem + 2 * STAGE);  // scale[128] | shift[128]

  if (wave >= 4) {
    // ================= loader =================
    PS_LOADER_SETPRIO();
    const int lw = wave - 4;
    const int srow = lane >> 3;
    const int chunk_off = ((lane & 7) ^ srow) << 4;
    int py[4], px[4], nb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + (j * 4 + lw) * 8 + srow;
      if (m < a.M && j * 4 + lw < AINS) {
        const int hw = a.Ho * a.Wo;
        const int n = m / hw, rem = m - n * hw;
        const int p = rem / a.Wo, q = rem - p * a.Wo;
        py[j] = p * a.mul;
        px[j] = q * a.mul;
        nb[j] = n * a.Hs * a.Ws;
      } else {
        py[j] = -(1 << 20);
        px[j] = 0;
        nb[j] = 0;
      }
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
    unsigned woff[4], aoff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = (j * 4 + lw) * 8 + srow;
      const int wg = rb / WN, within = rb % WN, fi = within >> 4, rho = within & 15;
      const int cout = n0 + wg * WN + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
      woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
    }
    int tap = 0, kl = 0, wk = 0;
    auto tap_offsets = [&](int t) {
      const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
      const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
      const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int yn = py[j] + dy, xn = px[j] + dx;
        const int y = yn >> a.div_shift, x = xn >> a.div_shift;
        const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
        aoff[j] = ok ? (unsigned)((nb[j] + y * a.Ws + x) * (int)a.pix_bytes) + chunk_off : PAD_ROW;
      }
    };
    tap_offsets(0);
    auto stage_issue = [&](int buf) {
      unsigned char* sa = smem + buf * STAGE;
      unsigned char* sb = smem + buf * STAGE + A_BYTES;
      const int ko = kl * 128;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (AINS % 4 == 0 || j * 4 + lw < AINS) BLDS16(rsA, sa + (j * 4 + lw) * 1024, aoff[j], ko);
#pragma unroll
      for (int j = 0; j < 4; ++j) BLDS16(rsB, sb + (j * 4 + lw) * 1024, woff[j], wk);
      wk += 128;
      if (++kl == a.klines) {
        kl = 0;
        if (++tap < a.taps) tap_offsets(tap);
      }
    };
    if (LDS_EPI && lw == 0) {  // per-channel epilogue parameters of this block's 128 couts
      for (int i = lane; i < 128; i += 64) {
        prm[i] = ep.scale ? ep.scale[n0 + i] : 1.f;
        prm[128 + i] = (ep.shift && ep.mode == PS_EPI_BNRELU) ? ep.shift[n0 + i] : 0.f;
      }
    }
    // DMA of a [BM][128-channel] bf16 tile of an epilogue operand into a free stage buffer (4 rows per instruction)
    auto tile_issue = [&](const void* src, int ldc, unsigned char* dst) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(a.M * ldc * 2), 0x00020000);
      const int rin = lane >> 4, pos = lane & 15;
#pragma unroll
      for (int j = 0; j < BM / 16; ++j) {
        const int t = j * 4 + lw, R = t * 4 + rin;
        const unsigned off = (m0 + R < a.M) ? (unsigned)((m0 + R) * ldc * 2 + n0 * 2 + ((pos ^ (R & 15)) << 4)) : PAD_ROW;
        BLDS16(rs, dst + t * 1024, off, 0);
      }
    };
    stage_issue(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps && !(PS_ABLATE(a.ablate) >= 1 && s >= 1)) stage_issue((s + 1) & 1);  // the buffer the consumers finished before the last barrier
      if (LDS_EPI && s + 1 == nsteps && tile1_src) tile_issue(tile1_src, tile1_ldc, smem + ((s + 1) & 1) * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (PS_ABLATE(a.ablate) != 2) __builtin_amdgcn_s_barrier();
    }
    if (LDS_EPI && tile2_src) {  // second operand: into the buffer the consumers have just finished
      tile_issue(tile2_src, tile2_ldc, smem + ((nsteps - 1) & 1) * STAGE);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================= consumer =================
  const int wm = wave / WNW, wn = wave % WNW;
  f32x4 acc[MI][WI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, g = lane >> 4;
  const int xfrag = (wm * WM + frow) * 128, wfrag = A_BYTES + (wn * WN + frow) * 128;
  const int sw = lane & 7;
  __builtin_amdgcn_s_barrier();  // step 0 staged
  for (int s = 0; s < nsteps; ++s) {
    const unsigned char* st = smem + (s & 1) * STAGE;
    if constexpr (Tr::split) {  // hi.hi + hi.lo + lo.hi on the two halves of the K-line (see bf16x3_t)
      u32x4 wf[2][WI], xf[2][MI];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int coff = ((g + 4 * kk) ^ sw) << 4;
#pragma unroll
        for (int i = 0; i < WI; ++i) wf[kk][i] = *reinterpret_cast<const u32x4*>(st + wfrag + i * 2048 + coff);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) xf[kk][mi] = *reinterpret_cast<const u32x4*>(st + xfrag + mi * 2048 + coff);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < WI; ++i) Tr::mma(wf[t == 2 ? 1 : 0][i], xf[t == 1 ? 1 : 0][mi], acc[mi][i]);
    } else {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((g + 4 * kk) ^ sw) << 4;
      u32x4 wf[WI], xf[MI];
#pragma unroll
      for (int i = 0; i < WI; ++i) wf[i] = *reinterpret_cast<const u32x4*>(st + wfrag + i * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const u32x4*>(st + xfrag + mi * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int i = 0; i < WI; ++i) Tr::mma(wf[i], xf[mi], acc[mi][i]);
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (PS_ABLATE(a.ablate) != 2) __builtin_amdgcn_s_barrier();
  }
  if constexpr (LDS_EPI) {
    if (tile2_src) __builtin_amdgcn_s_barrier();
    conv_epilogue_lds<T, MI, WI>(a, acc, m0 + wm * WM, wm * WM, n0 + wn * WN, wn * WN, lane, prm, smem + (nsteps & 1) * STAGE,
                              smem + ((nsteps - 1) & 1) * STAGE);
  } else {
    conv_epilogue<typename Tr::epi, MI, WI>(a, acc, m0 + wm * WM, n0 + wn * WN, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised kernel, large tile: BM x 128 (BM = 256 or 224 pixels), 8 waves = 4 CONSUMER waves (2x2, each
// (BM/2) x 64: 8 or 7 pixel fragments x 4 cout fragments) + 4 LOADER waves, ONE block per CU (<= 256 VGPRs), 3-stage
// LDS ring.  Against the 128x128 / two-blocks-per-CU kernel above:
//   * LDS traffic per MFMA falls by a quarter (a 128x64 wave tile reads 12 fragments per 32 MFMAs instead of 8 per 16,
//     and the block stages 48 KiB per 256 MFMAs/wave-step instead of 32 KiB per 128) -- the LDS array (fragment reads +
//     DMA writes) is what the small tile saturates;
//   * the consumers software-pipeline the fragments ACROSS the block barrier: after barrier s a wave issues the reads of
//     step s+1's first K-half and immediately has the 32 MFMAs of step s's second K-half to run (operands already in
//     registers), so no LDS latency is exposed behind a barrier;
//   * the loaders run two K-steps ahead (counted vmcnt), so a DMA has two barrier intervals (~2 x 1024 cycles) to land.
// BM = 224 (7-fragment waves) makes 28x28-derived pixel counts tile exactly: 64 x 784 = 224 x 224.
// ------------------------------------------------------------------------------------------------
// Q: every tile comes from the launch's ticket queue (a.queue; ps_internal.h, conv_igemm_halo_kernel).  Tiles here can be as short as three
// K-steps, so consumer wave 0 draws TWO tiles ahead: the tickets of tiles 0 and 1 at the top of the kernel, the ticket of tile s + 2 when tile s
// starts, collected and published in front of tile s's LAST barrier, number (s + 1) n of the block (n = K-steps per tile).  The loaders stage two
// K-steps ahead: they cross into tile s + 2 in their iteration (s + 2) n - 3, behind barrier (s + 2) n - 3 -- not earlier than the publication
// iff n >= 3 (the dispatcher's condition; with n = 2 they would read the mailbox one barrier early).  Four mailbox entries: s + 1, s + 2 and the
// one being read.
template <typename Tr, int BM, bool SPLIT = false, bool Q = false>  // SPLIT: one parity class of a stride-2 data gradient (IgemmArgs::tap_mask ...)
__global__ __launch_bounds__(512, 2) void conv_igemm_ws2_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  constexpr int BN = 128, MI = BM / 32, WI = 4, WM = 16 * MI, WN = 64;
  constexpr int AINS = BM / 8, AJ = AINS / 4, BJ = 4, NLD = AJ + BJ;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  static_assert(BM == 256 || BM == 224, "pixel tile");
  static_assert(NLD == 12 || NLD == 11, "vmcnt literals below");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // PERSISTENT over tiles: block b works on tiles b, b + G, b + 2G, ... of the raster (G = gridDim.x <= #CUs).  The K-steps of
  // all its tiles form ONE flat sequence through the LDS ring, so the loaders prefetch the next tile's first two steps while
  // the consumers are still in the epilogue, and the epilogue's global stores drain behind the next tile's main loop (with
  // one block per CU and equal tiles, every CU reaches its epilogue at the same time: as separate blocks the stores of a
  // whole round -- tens of MB -- were exposed at HBM speed before any CU could start its next tile).
  int first, G, ntiles;  // this block's tiles: first, first + G, ... < ntiles  (Q: whatever the queue hands out)
  [[maybe_unused]] unsigned q_tk = 0, q_tk1 = 0;  // Q, consumer wave 0: the ticket in flight (two at the top of the kernel)
  [[maybe_unused]] bool q_pk = false;             // ... and whether that draw also looks at the other classes' counters (wave-uniform)
  [[maybe_unused]] const unsigned mbox = ps_q_mbox_addr(smem + 3 * STAGE);
  if constexpr (Q) {
    static_assert(!SPLIT, "the parity-class launches keep the static schedule");
    G = 0;
    first = -1;
    ntiles = a.ntm * a.ntn;
    q_pk = ps_q_count(ntiles, blockIdx.x & 7) <= 96;
    if (wave == 0) {
      ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_pk, q_tk);
      ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_pk, q_tk1);
    }
  } else {
    ps_block_items(blockIdx.x, gridDim.x, a.ntm * a.ntn, a.nb, a.tpb, first, G, ntiles);
  }
  const int ntap = SPLIT ? __builtin_popcount(a.tap_mask) : a.taps;
  const int nsteps = ntap * a.klines;
  const int my_tiles = Q ? 1 : (ntiles - first + G - 1) / G;  // >= 1
  const int total_steps = my_tiles * nsteps;

  if (wave >= 4) {
    // ================= loader =================
    PS_LOADER_SETPRIO();
    const int lw = wave - 4;
    const int srow = lane >> 3;
    const int chunk_off = ((lane & 7) ^ srow) << 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
    int py[AJ], px[AJ], nb[AJ];
    unsigned woff[BJ], aoff[AJ];
    int tap = 0, kl = 0, wk = 0;
    auto tap_id = [&](int t) {  // SPLIT: the t-th tap of tap_mask
      if constexpr (!SPLIT) return t;
      int mk = a.tap_mask;
      for (int i = 0; i < t; ++i) mk &= mk - 1;
      return __builtin_ctz(mk);
    };
    auto tap_offsets = [&](int tt) {
      const int t = tap_id(tt);
      const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
      const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
      const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int yn = py[j] + dy, xn = px[j] + dx;
        const int y = yn >> a.div_shift, x = xn >> a.div_shift;
        const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
        aoff[j] = ok ? (unsigned)((nb[j] + y * a.Ws + x) * (int)a.pix_bytes) + chunk_off : PAD_ROW;
      }
    };
    auto tile_setup = [&](int tile) {
      int tm, tn;
      ps_tile_of_block(tile, a.ntn, a.ntm, tm, tn, a.supertile);
      const int m0 = tm * BM, n0 = tn * BN;
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        const int m = m0 + (j * 4 + lw) * 8 + srow;
        if (m < a.M) {
          const int hw = a.Ho * a.Wo;
          const int n = m / hw, rem = m - n * hw;
          const int p = rem / a.Wo, q = rem - p * a.Wo;
          py[j] = (SPLIT ? a.oy + 2 * p : p) * a.mul;
          px[j] = (SPLIT ? a.ox + 2 * q : q) * a.mul;
          nb[j] = n * a.Hs * a.Ws;
        } else {
          py[j] = -(1 << 20);
          px[j] = 0;
          nb[j] = 0;
        }
      }
#pragma unroll
      for (int j = 0; j < BJ; ++j) {
        const int rb = (j * 4 + lw) * 8 + srow;
        const int wg = rb / WN, within = rb % WN, fi = within >> 4, rho = within & 15;
        const int cout = n0 + wg * WN + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
        woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
      }
      tap = 0; kl = 0; wk = SPLIT ? tap_id(0) * a.klines * 128 : 0;
      tap_offsets(0);
    };
    if constexpr (Q) {  // the block's first tile (published by consumer wave 0 in front of this barrier)
      __builtin_amdgcn_s_barrier();
      first = ps_q_mbox_read(mbox, 0);
      if (first < 0) return;
    }
    int tile = first, slot = 0, issued = 0;
    [[maybe_unused]] int l_seq = 0;        // Q: tiles this cursor has left behind
    [[maybe_unused]] bool l_done = false;  // Q: the last K-step of the block's last tile has been staged
    tile_setup(tile);
    // stages the next K-step of the flat sequence (exactly NLD loads per wave); the caller guarantees issued < total_steps (no end test and,
    // in the product build, no selector test in the issue path: it is on the K-step's critical path, NOTES 7.20)
    auto issue_next = [&]() {
      if (PS_ABLATE(a.ablate) == 1 && issued >= 3) { ++issued; return; }  // timing experiment: consumers run on stale LDS contents
      unsigned char* sa = smem + slot * STAGE;
      unsigned char* sb = sa + A_BYTES;
      const int ko = kl * 128;
#pragma unroll
      for (int j = 0; j < AJ; ++j) BLDS16(rsA, sa + (j * 4 + lw) * 1024, aoff[j], ko);
#pragma unroll
      for (int j = 0; j < BJ; ++j) BLDS16(rsB, sb + (j * 4 + lw) * 1024, woff[j], wk);
      ++issued;
      slot = (slot == 2) ? 0 : slot + 1;
      wk += 128;
      if (++kl == a.klines) {
        kl = 0;
        if (++tap < ntap) {
          tap_offsets(tap);
          if constexpr (SPLIT) wk = tap_id(tap) * a.klines * 128;
        } else if constexpr (Q) {
          tile = ps_q_mbox_read(mbox, ++l_seq);
          if (tile >= 0) tile_setup(tile);
          else l_done = true;
        } else if (issued < total_steps) {
          tile += G;
          tile_setup(tile);
        }
      }
    };
    auto wait_newest_in_flight = [&]() {  // everything but the newest step's loads has landed
      if constexpr (NLD == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
    };
    // Steps 0 and 1, then one step per barrier while there are steps left to stage, then the drain (nothing to stage: everything landed).
    if constexpr (Q) {  // (every tile has at least three K-steps: the dispatcher's condition)
      issue_next();
      issue_next();
      wait_newest_in_flight();
      __builtin_amdgcn_s_barrier();  // step 0 visible
      while (!l_done) {
        issue_next();
        wait_newest_in_flight();
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
      wait_newest_in_flight();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // step 0 visible
    int gs = 0;
    for (; gs + 2 < total_steps; ++gs) {
      // ring slot (gs+2)%3 held step gs-1: the consumers' reads of it completed before the previous barrier
      issue_next();             // step gs + 2
      wait_newest_in_flight();  // step gs + 1 landed
      __builtin_amdgcn_s_barrier();
    }
    for (; gs < total_steps; ++gs) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ================= consumer =================
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, g = lane >> 4;
  const int xfrag = (wm * WM + frow) * 128, wfrag = A_BYTES + (wn * WN + frow) * 128;
  const int sw = lane & 7;
  const int coff0 = (g ^ sw) << 4, coff1 = ((g + 4) ^ sw) << 4;

  if constexpr (Q) {
    if (wave == 0) {
      bool pk0 = q_pk, pk1 = q_pk;
      ps_q_mbox_write(mbox, 0, ps_q_resolve(a.queue, blockIdx.x & 7, lane, 0, ntiles, q_tk, pk0));
      ps_q_mbox_write(mbox, 1, ps_q_resolve(a.queue, blockIdx.x & 7, lane, 0, ntiles, q_tk1, pk1));
      q_pk = pk0 || pk1;
    }
    __builtin_amdgcn_s_barrier();
    first = ps_q_mbox_read(mbox, 0);
    if (first < 0) {
      if (wave == 0) ps_q_block_done(a.queue, lane, gridDim.x);
      return;
    }
  }
  __builtin_amdgcn_s_barrier();  // step 0 visible
  int cur = 0;
  [[maybe_unused]] int q_seq = 0;
  for (int tile = first; Q ? tile >= 0 : tile < ntiles; tile = Q ? ps_q_mbox_read(mbox, ++q_seq) : tile + G) {
    if constexpr (Q) {  // the ticket of this block's tile q_seq + 2: in flight until the tile's last K-step
      if (wave == 0) ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_pk, q_tk);
    }
    f32x4 acc[MI][WI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wf0[WI], xf0[MI], wf1[WI], xf1[MI];  // fragments of the first / second K-half of a step
    // One K-half: issue the NR = WI + MI fragment reads of (stage st, chunk offset coff) into (wfn, xfn) INTERLEAVED with the
    // MFMAs on (wfo, xfo), whose operands are already in registers: read, 2 MFMAs, read, 2 MFMAs, ... so the matrix pipe
    // starts right behind the barrier / the wait instead of idling through a burst of 11-12 LDS instructions (the order is
    // pinned with scheduling barriers; hipcc otherwise hoists all reads to the top).
    // Split types (Tr::split): a K-line's halves are the hi and the lo fragments of 32 channels and the K-step runs THREE MFMA groups on them,
    //   A: read hi -> (wf0, xf0)  |  MFMAs x_hi(previous step) . w_lo(previous step) = (xf0 being replaced, wf1)
    //   B: read lo -> (wf1, xf1)  |  MFMAs x_hi . w_hi = (xf0, wf0)
    //   C: (no reads)             |  MFMAs x_lo . w_hi = (xf1, wf0)
    // with the two register sets the plain loop already keeps.  In A the pixel fragments xf0 are operand AND read destination: with three MFMAs
    // per read slot, the MFMA that uses xf0[j] (slot <= (4 j + 3) / 3) is issued before the read that replaces it (slot WI + j) -- program order,
    // which the in-order wave keeps.
    auto half = [&](const unsigned char* st, int coff, u32x4 (&wfn)[WI], u32x4 (&xfn)[MI], const u32x4 (&wfo)[WI], const u32x4 (&xfo)[MI],
                    bool do_mma) {
      constexpr int NR = WI + MI, NM = MI * WI, PER = Tr::split ? 3 : PS_READ_PER;
      static_assert(!Tr::split || (NM - 1) / 3 < WI + (NM - 1) / WI, "split: every MFMA on the old pixel fragments precedes their replacement");
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r < WI) wfn[r] = *reinterpret_cast<const u32x4*>(st + wfrag + r * 2048 + coff);
        else xfn[r - WI] = *reinterpret_cast<const u32x4*>(st + xfrag + (r - WI) * 2048 + coff);
        __builtin_amdgcn_sched_barrier(0);
        if (do_mma) {
#pragma unroll
          for (int q = r * PER; q < (r + 1) * PER && q < NM; ++q) Tr::mma(wfo[q % WI], xfo[q / WI], acc[q / WI][q % WI]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (do_mma) {
#pragma unroll
        for (int q = NR * PER; q < NM; ++q) Tr::mma(wfo[q % WI], xfo[q / WI], acc[q / WI][q % WI]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    auto group_c = [&]() {  // split types: x_lo . w_hi, both already in registers
      if constexpr (Tr::split) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < WI; ++i) Tr::mma(wf0[i], xf1[mi], acc[mi][i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    {  // first K-step of the tile: nothing to overlap the first reads with
      const unsigned char* st = smem + cur * STAGE;
      half(st, coff0, wf0, xf0, wf1, xf1, false);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      half(st, coff1, wf1, xf1, wf0, xf0, true);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      group_c();
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
    }
    for (int s = 1; s < nsteps; ++s) {
      if constexpr (Q) {
        if (s == nsteps - 1 && wave == 0)
          ps_q_mbox_write(mbox, q_seq + 2, ps_q_resolve(a.queue, blockIdx.x & 7, lane, 0, ntiles, q_tk, q_pk));
      }
      const unsigned char* st = smem + cur * STAGE;
      // this step's first K-half is read behind the previous step's outstanding MFMAs (plain: its second half; split: x_hi . w_lo)
      if constexpr (Tr::split) half(st, coff0, wf0, xf0, wf1, xf0, true);
      else half(st, coff0, wf0, xf0, wf1, xf1, true);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      half(st, coff1, wf1, xf1, wf0, xf0, true);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all reads of this slot are done: the loaders may refill it
      group_c();
      __builtin_amdgcn_s_barrier();
      cur = (cur == 2) ? 0 : cur + 1;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < WI; ++i) Tr::mma(wf1[i], Tr::split ? xf0[mi] : xf1[mi], acc[mi][i]);  // (split: the last step's x_hi . w_lo)
    int tm, tn;
    ps_tile_of_block(tile, a.ntn, a.ntm, tm, tn, a.supertile);
    conv_epilogue<typename Tr::epi, MI, WI, SPLIT ? 2 : 0>(a, acc, tm * BM + wm * WM, tn * BN + wn * WN, lane);
  }
  if constexpr (Q) {
    if (wave == 0) ps_q_block_done(a.queue, lane, gridDim.x);  // (every draw of this block has returned: the last one was resolved in the last tile)
  }
}

// Head epilogue of conv_gemm256_kernel<.., HEAD> (inference): instead of storing the activated tile -- conv6, the net's second largest
// tensor, written only to be read back by the 4096 -> C head (models/revise_net.py:50) -- the wave reduces it on the spot:
//   v[m][ch] = T(max(acc * scale[ch] + shift[ch], 0))          (rounded to the storage type, exactly what conv6 would have held)
//   part[j][m][c] = sum over this wave's 64 channels of v[m][ch] * head_w[c][ch],   j = (cout tile) * 4 + (wave column)
// and head_reduce_kernel adds the parts in j order.  A lane owns 16 channels of one pixel per fragment row; the four lane groups of a
// pixel meet through two cross-lane adds.
template <typename T, int MI, int WI>
__device__ __forceinline__ void conv_head_epilogue(const IgemmArgs& a, f32x4 (&acc)[MI][WI], int mbase, int cbase, int lane, int part_idx) {
  static_assert(WI == 4, "a lane owns 16 channels");
  const int frow = lane & 15, g = lane >> 4;
  const int cb = cbase + 16 * g;
  const ps_epilogue& e = a.epi;
  float sc[16], sh[16];
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    const float4 q = e.scale ? *reinterpret_cast<const float4*>(e.scale + cb + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 r = e.shift ? *reinterpret_cast<const float4*>(e.shift + cb + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[i] = q.x; sc[i + 1] = q.y; sc[i + 2] = q.z; sc[i + 3] = q.w;
    sh[i] = r.x; sh[i + 1] = r.y; sh[i + 2] = r.z; sh[i + 3] = r.w;
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float v = fmaxf(acc[mi][i >> 2][i & 3] * sc[i] + sh[i], 0.f);
      if constexpr (std::is_same<T, __bf16>::value) acc[mi][i >> 2][i & 3] = __uint_as_float(static_cast<uint32_t>(ps_f32_to_bf16(v)) << 16);
      else acc[mi][i >> 2][i & 3] = static_cast<float>(static_cast<_Float16>(v));
    }
  const long long plane = (long long)a.epi_M * a.head_c;
  float* part = a.head_part + (long long)part_idx * plane;
  for (int c = 0; c < a.head_c; ++c) {
    float w[16];
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      const float4 q = *reinterpret_cast<const float4*>(a.head_w + (long long)c * a.Cd + cb + i);
      w[i] = q.x; w[i + 1] = q.y; w[i + 2] = q.z; w[i + 3] = q.w;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float p = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) p += acc[mi][i >> 2][i & 3] * w[i];
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      const int m = mbase + mi * 16 + frow;
      if (g == 0 && m < a.epi_M) part[(long long)m * a.head_c + c] = p;
    }
  }
}
// cam[i] = part[0][i] + part[1][i] + ... (i over M x C), in part order: one thread per element, fixed summation order
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, float* __restrict__ cam, long long n, int nparts) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float s = part[i];
    for (int j = 1; j < nparts; ++j) s += part[(long long)j * n + i];
    cam[i] = s;
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 tile kernel for the PLAIN GEMMs of the net: 1x1 stride-1 layers (forward and data gradient; the bottleneck units'
// K-concatenated shortcut + last conv included), 16-bit operands.
//   The ws2 kernel stages (256 + 128) x 128 B per K-step for 256 x 128 x 64 MACs and is bound by the L2 -> LDS fill rate (~65 GB/s per
//   CU, NOTES 7.1); a 256 x 256 tile stages 64 KiB for twice the MACs: 1.5x fewer staged bytes (and LDS-DMA pieces) per FLOP.  Its
//   64 K accumulators need all eight waves of the block as MFMA waves (128 accumulator registers each), so there are no loader waves:
//   the block's two wave GROUPS (waves 0-3 and 4-7, one wave of each per SIMD) alternate roles, offset by one barrier -- while a
//   group runs the 16 MFMAs of one output QUADRANT (64 pixels x 32 couts x one 64-deep K-tile), its SIMD partners issue their next
//   fragment reads and their share of the LDS-DMA, so the matrix pipe of every SIMD always has an MFMA wave (the structure of the
//   programming guide's 256^2 "8-phase" GEMM, restated for this data layout).
//   * waves: (grp = wave >> 2) picks the pixel half, (wc = wave & 3) the 64-cout quarter: per wave 128 pixels x 64 couts = 8 x 4
//     fragments -- the same per-wave shape, fragment layout and epilogue as the ws2 consumers.
//   * LDS: two K-tile buffers of [256 pixel rows | 256 weight rows] x 128 B, XOR-swizzled exactly like the other kernels' images.
//   * per K-tile and group four phases  L(q) | barrier | C(q) | barrier :  L = this quadrant's new fragment reads + ONE quarter tile
//     (64 rows, 8 KiB: two 1-KiB pieces per wave) of LDS-DMA + counted vmcnt + lgkmcnt(0); C = 16 MFMAs.  Quadrants
//     (A0,B0) (A0,B1) (A1,B1) (A1,B0): reads 8 / 4 / 8 / 4 (B0 of the NEXT K-tile is read in the last phase, into its second register set).
//   * staging schedule (intervals between barriers; G0's L(q_j) of K-tile t is interval 8t + 2j, G1's 8t + 2j + 1): group g stages
//     quarters g and g + 2 of A(t+1) in L(q0), L(q1) and of B(t+2) in L(q2), L(q3).  WAR: every L ends with lgkmcnt(0) BEFORE its
//     barrier, so a quarter may be re-staged from the interval after its last read: B(t) is last read in interval 8t + 3 (restaged
//     from 8t + 4), A rows of group h in interval 8t + 4 + h (restaged from 8(t+1) + ...).  RAW: a wave's pieces issued in one of its L
//     phases have landed by the end of its L phase after next (vmcnt(4) = the two newest phases' pieces may be in flight), which is
//     at least two intervals before any wave reads them.
// ------------------------------------------------------------------------------------------------
template <typename Tr, bool HEAD = false>
__global__ __launch_bounds__(512, 2) void conv_gemm256_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T [[maybe_unused]];
  static_assert(sizeof(T) == 2, "16-bit operands");
  constexpr int BM = 256, BN = 256, MI = 8, WI = 4;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, BUF = A_BYTES + B_BYTES;  // 64 KiB per K-tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wc = wave & 3;
  int tm, tn;
  ps_tile_of_block(ps_xcd_remap(blockIdx.x, gridDim.x), a.ntn, a.ntm, tm, tn, a.supertile);
  const int m0 = tm * BM, n0 = tn * BN;
  const int NT = a.klines;

  // ---- staging: this group's quarters {grp, grp + 2} of the pixel rows and of the weight rows; wave wc owns pieces 2 wc, 2 wc + 1
  const int srow = lane >> 3;
  const int chunk_off = ((lane & 7) ^ srow) << 4;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
  unsigned aoff[2][2], woff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int rowq = (grp + 2 * h) * 64 + (2 * wc + p) * 8 + srow;  // row of the 256-row tile
      const int m = m0 + rowq;
      aoff[h][p] = m < a.M ? (unsigned)(m * (int)a.pix_bytes) + chunk_off : PAD_ROW;
      const int within = rowq & 63, fi = within >> 4, rho = within & 15;  // weight rows are permuted inside each 64-row wave block
      const int cout = n0 + (rowq & ~63) + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
      woff[h][p] = (unsigned)(cout * (int)a.wrow_bytes) + chunk_off;
    }
  // pc: which of the wave's two pieces of the quarter (-1: both)
  auto stage_a = [&](int buf, int h, int kt, int pc = -1) {
    unsigned char* dst = smem + buf * BUF + ((grp + 2 * h) * 64 + 2 * wc * 8) * 128;
    if (pc != 1) BLDS16(rsA, dst, aoff[h][0], kt * 128);
    if (pc != 0) BLDS16(rsA, dst + 1024, aoff[h][1], kt * 128);
  };
  auto stage_b = [&](int buf, int h, int kt, int pc = -1) {
    unsigned char* dst = smem + buf * BUF + A_BYTES + ((grp + 2 * h) * 64 + 2 * wc * 8) * 128;
    if (pc != 1) BLDS16(rsB, dst, woff[h][0], kt * 128);
    if (pc != 0) BLDS16(rsB, dst + 1024, woff[h][1], kt * 128);
  };

  // ---- fragments
  const int frow = lane & 15, g = lane >> 4, sw = lane & 7;
  const int coff[2] = {(g ^ sw) << 4, ((g + 4) ^ sw) << 4};
  const int xbase = (grp * 128 + frow) * 128, wbase = A_BYTES + (wc * 64 + frow) * 128;
  f32x4 acc[MI][WI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 xf[4][2];      // A0 or A1: 4 pixel fragments x 2 K-halves
  u32x4 wf0[2][2][2];  // B0 (cout fragments 0, 1) of the current and of the next K-tile: [set][fragment][K-half]
  u32x4 wf1[2][2];     // B1 (cout fragments 2, 3)
  auto read_x = [&](const unsigned char* cb, int sub) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) xf[j][kh] = *reinterpret_cast<const u32x4*>(cb + xbase + (4 * sub + j) * 2048 + coff[kh]);
  };
  auto read_w = [&](const unsigned char* cb, int sub, u32x4 (&w)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) w[i][kh] = *reinterpret_cast<const u32x4*>(cb + wbase + (2 * sub + i) * 2048 + coff[kh]);
  };
#ifndef PS_GEMM256_PRIO
#define PS_GEMM256_PRIO 1  // A/B builds: 0 = no priority change, 1 = the MFMA phase at priority 1 (default), 2 = the LOAD phase at priority 1
#endif
#ifndef PS_GEMM256_SPLIT_DMA
#define PS_GEMM256_SPLIT_DMA 0  // A/B: 1 = a load phase issues ONE of the quarter's two LDS-DMA pieces, the compute phase the other (between its K-halves)
#endif
  // 16 MFMAs; the same accumulator recurs every 8th.  mid(): issued between the two K-halves (PS_GEMM256_SPLIT_DMA: the quarter's second piece)
  auto mma_quadrant = [&](int msub, int nsub, const u32x4 (&w)[2][2], auto mid) {
    if constexpr (PS_GEMM256_PRIO == 1) __builtin_amdgcn_s_setprio(1);
    if constexpr (PS_GEMM256_PRIO == 2) __builtin_amdgcn_s_setprio(0);
    if constexpr (Tr::split) {
      // split types: the K-tile's two 64-byte halves are the hi and the lo fragments of the same 32 logical channels (bf16x3_t) -- three MFMA
      // groups on the registers the plain loop holds anyway: x_hi w_hi, x_hi w_lo, x_lo w_hi (24 MFMAs per quadrant instead of 16)
#pragma unroll
      for (int grp3 = 0; grp3 < 3; ++grp3) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) Tr::mma(w[i][grp3 == 1 ? 1 : 0], xf[j][grp3 == 2 ? 1 : 0], acc[4 * msub + j][2 * nsub + i]);
        if (grp3 == 0) {
          __builtin_amdgcn_sched_barrier(0);
          mid();
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) Tr::mma(w[i][kh], xf[j][kh], acc[4 * msub + j][2 * nsub + i]);
      if (kh == 0) {
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }
    if constexpr (PS_GEMM256_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if constexpr (PS_GEMM256_PRIO == 2) __builtin_amdgcn_s_setprio(1);
  };
  // end of a load phase: the pieces of the two newest load phases may stay in flight; every LDS read retired BEFORE the barrier
  // End of a load phase.  The phase's LDS reads retire AFTER the barrier (lgkmcnt(0) in front of the first MFMA), so their latency
  // overlaps the barrier -- except where the NEXT interval already re-stages what was just read: G1's B1 reads of L(q1) (interval
  // 8t + 3) and G0's LDS-DMA of L(q2) (interval 8t + 4) into the same rows; there the reads are drained before the barrier.
  auto end_load = [&](bool staged, bool drain_reads) {
    // (split DMA: the pieces of this load phase, of the previous compute phase and of the previous load phase may stay in flight)
    if (staged) {
      if constexpr (PS_GEMM256_SPLIT_DMA) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
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
  constexpr int LP = PS_GEMM256_SPLIT_DMA ? 0 : -1;  // piece(s) issued by a load phase

  // ---- prologue: A(0), B(0), B(1)
  stage_a(0, 0, 0);
  stage_a(0, 1, 0);
  stage_b(0, 0, 0);
  stage_b(0, 1, 0);
  if (NT > 1) {
    stage_b(1, 0, 1);
    stage_b(1, 1, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_w(smem, 0, wf0[0]);
  if (grp == 1) __builtin_amdgcn_s_barrier();  // stagger: G1 runs one barrier behind G0

  auto ktile = [&](auto parity, int t) {
    constexpr int P = decltype(parity)::value;
    const unsigned char* cb = smem + P * BUF;
    const bool s1 = t + 1 < NT, s2 = t + 2 < NT;
    // L(q0): A0
    read_x(cb, 0);
    if (s1) stage_a(P ^ 1, 0, t + 1, LP);
    end_load(s1, false);
    mma_quadrant(0, 0, wf0[P], [&]() { if (PS_GEMM256_SPLIT_DMA && s1) stage_a(P ^ 1, 0, t + 1, 1); });
    end_compute();
    // L(q1): B1
    read_w(cb, 1, wf1);
    if (s1) stage_a(P ^ 1, 1, t + 1, LP);
    end_load(s1, grp == 1);
    mma_quadrant(0, 1, wf1, [&]() { if (PS_GEMM256_SPLIT_DMA && s1) stage_a(P ^ 1, 1, t + 1, 1); });
    end_compute();
    // L(q2): A1
    read_x(cb, 1);
    if (s2) stage_b(P, 0, t + 2, LP);
    end_load(s2, false);
    mma_quadrant(1, 1, wf1, [&]() { if (PS_GEMM256_SPLIT_DMA && s2) stage_b(P, 0, t + 2, 1); });
    end_compute();
    // L(q3): B0 of the NEXT K-tile (resident since two K-tiles ago)
    if (s1) read_w(smem + (P ^ 1) * BUF, 0, wf0[P ^ 1]);
    if (s2) stage_b(P, 1, t + 2, LP);
    end_load(s2, false);
    mma_quadrant(1, 0, wf0[P], [&]() { if (PS_GEMM256_SPLIT_DMA && s2) stage_b(P, 1, t + 2, 1); });
    end_compute();
  };
#ifndef PS_GEMM256_PHASES
#define PS_GEMM256_PHASES 8  // A/B: 4 = two load + two compute phases per K-tile (half tiles of 32 MFMAs) instead of four + four (quadrants of 16):
                             // bit-identical results, EQUAL speed within 0.5 % on every layer (profiles/r03_gemm256_vs_ws2.txt) -- the barriers are not what a K-tile costs
#endif
  // Four-phase form: G0's intervals for K-tile t are 4t (L: B1, A0 + its two quarters of A(t+1)), 4t + 1 (C: A0 x B0, A0 x B1),
  // 4t + 2 (L: A1, B0 of t + 1 + its two quarters of B(t+2)), 4t + 3 (C: A1 x B1, A1 x B0); G1 one interval later.  vmcnt(4) at the end
  // of a load phase = only that phase's four pieces in flight, i.e. a piece has landed by the end of its wave's NEXT load phase and is
  // visible one barrier later: A(t+1) quarters {0, 2} (issued 4t, visible 4t + 3, read from 4t + 4), {1, 3} (4t + 1 -> 4t + 4, read
  // from 4t + 6); B(t+2) quarters {0, 2} (4t + 2 -> 4t + 5), {1, 3} (4t + 3 -> 4t + 6, B0(t+2) read from 4t + 6).  WAR: A(t-1)'s quarters
  // were last read in 4t - 4 .. 4t - 1 (quarter 3 by G1 in 4t - 1, re-staged by G1 itself in 4t + 1); B(t) is last read by G1's B1 reads in
  // 4t + 1 and re-staged by G0 in 4t + 2: G1 retires those four reads (issued FIRST: lgkmcnt(8)) before its barrier.
  auto mma_half = [&](int msub, const u32x4 (&w0)[2][2], const u32x4 (&w1)[2][2]) {
    __builtin_amdgcn_s_setprio(1);
    constexpr int NG = Tr::split ? 3 : 2;  // (split: hi.hi, x_hi w_lo, x_lo w_hi -- see mma_quadrant)
#pragma unroll
    for (int kg = 0; kg < NG; ++kg) {
      const int wk = Tr::split ? (kg == 1 ? 1 : 0) : kg, xk = Tr::split ? (kg == 2 ? 1 : 0) : kg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 2; ++i) Tr::mma(w0[i][wk], xf[j][xk], acc[4 * msub + j][i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) Tr::mma(w1[i][wk], xf[j][xk], acc[4 * msub + j][2 + i]);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  auto end_load4 = [&](bool staged, bool drain_first4) {
    if (staged) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (drain_first4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  auto ktile4 = [&](auto parity, int t) {
    constexpr int P = decltype(parity)::value;
    const unsigned char* cb = smem + P * BUF;
    const bool s1 = t + 1 < NT, s2 = t + 2 < NT;
    // L(h0): B1 (first: G1 retires these before the barrier), A0
    read_w(cb, 1, wf1);
    __builtin_amdgcn_sched_barrier(0);
    read_x(cb, 0);
    if (s1) {
      stage_a(P ^ 1, 0, t + 1);
      stage_a(P ^ 1, 1, t + 1);
    }
    end_load4(s1, grp == 1);
    mma_half(0, wf0[P], wf1);
    end_compute();
    // L(h1): A1, B0 of the next K-tile
    read_x(cb, 1);
    if (s1) read_w(smem + (P ^ 1) * BUF, 0, wf0[P ^ 1]);
    if (s2) {
      stage_b(P, 0, t + 2);
      stage_b(P, 1, t + 2);
    }
    end_load4(s2, false);
    mma_half(1, wf0[P], wf1);
    end_compute();
  };
  int t = 0;
  if constexpr (PS_GEMM256_PHASES == 4) {
    for (; t + 1 < NT; t += 2) {
      ktile4(std::integral_constant<int, 0>{}, t);
      ktile4(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < NT) ktile4(std::integral_constant<int, 0>{}, t);
  } else {
    for (; t + 1 < NT; t += 2) {
      ktile(std::integral_constant<int, 0>{}, t);
      ktile(std::integral_constant<int, 1>{}, t + 1);
    }
    if (t < NT) ktile(std::integral_constant<int, 0>{}, t);
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  if constexpr (HEAD) conv_head_epilogue<T, MI, WI>(a, acc, m0 + grp * 128, n0 + wc * 64, lane, tn * 4 + wc);
  else conv_epilogue<typename Tr::epi, MI, WI, 0>(a, acc, m0 + grp * 128, n0 + wc * 64, lane);
}

// ------------------------------------------------------------------------------------------------
// Halo kernel: 3x3 STRIDE-1 convolutions on feature maps whose width is a multiple of 28 (28 / 56 / 112; dilation <= 4; forward and
// data gradient), persistent
// 224 x 128 tiles, 4 consumer + 4 loader waves as conv_igemm_ws2_kernel -- but the pixel operand is staged as a WINDOW with
// its halo instead of one gathered tile per tap.
//   The ws2 kernel is bound by L2 -> LDS fill bandwidth, not by the matrix pipe (r01: with the loads ablated the consumers
//   run 26-35 % faster; every CU pulls (224 + 128) x 128 B per K-step = ~64 GB/s, ~16.5 TB/s chip-wide, the measured ceiling
//   of LDS gathers from L2): each input row is fetched nine times, once per tap.  Here a tile is an 8-row x 28-column block
//   (global rows n*H + p; whole rows on the 28-wide maps); for one 64-channel K-line and one tap ROW ty the loaders stage the
//   8 x (28 + 2d) window  rows [R0 + (ty-1)d, +8) x cols [-d, 28 + d)  once (zero-filled outside the image), and the three
//   taps tx = 0..2 of that row are three K-steps whose pixel fragments are read from the same window at a column shift of
//   (tx-1)d: fill traffic per K-step drops from 45 KiB to 26-28 KiB.
//   * K order is (K-line, ty, tx) instead of (tap, K-line): sums are re-associated relative to the other kernels.
//   * for a given tap row, window row wr serves exactly one produced row (global row R0 + wr), so vertical padding -- and rows
//     that would come from the neighbouring image when a tile straddles two images (28 % 8 != 0) -- are zero-filled by the
//     LOADER per window row.
//   * the window is stored COLUMN-major (one 1-KiB DMA instruction = one window column of 8 rows) and a fragment is an
//     8-row x 2-column patch: the swizzle term row & 7 is then the window ROW, independent of the tap column, so a lane's
//     fragment address for tap tx is a kernel-constant centre address + a scalar -- no per-step address arithmetic.
//   * LDS: 2 windows x 36 KiB + 3 weight stages x 16 KiB; the window for (K-line, ty) + 1 is issued during the first K-step
//     of the current one (behind that step's weights, so the in-order vmcnt lets it stay in flight for two steps); weights
//     run two K-steps ahead.
// ------------------------------------------------------------------------------------------------
// Diagnostic build -DPS_HALO_STAMPS: the halo kernel's consumer waves time the three segments of a K-step with s_memtime (read right
// behind the existing lgkmcnt(0) waits / the barrier, where the LDS queue is empty) and add them up per wave: [first half: reads +
// MFMAs + wait], [second half], [barrier].  ps_debug_read_stamps copies the sums out.  (+ ~3 short SMEM round trips per K-step.)
#ifdef PS_HALO_STAMPS
__device__ unsigned long long g_halo_stamps[256 * 4 * 8];  // [block][consumer wave][seg0, seg1, seg2, steps, shader cycles, 100 MHz ticks, tail MFMAs, epilogue]
#define PS_STAMP(var)                                            \
  do {                                                           \
    var = __builtin_amdgcn_s_memtime();                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           \
  } while (0)
#endif

// A/B build -DPS_HALO_NO_LOADS (results WRONG): the halo kernel's loaders stage only the first window / weight steps of a block -- the
// consumers' rate with nothing arriving in LDS.  (left = items still to issue, counted down from the block's total)
#ifdef PS_HALO_NO_LOADS
#define PS_HALO_STAGE(left, total) ((left) >= (total) - 4)
#else
#define PS_HALO_STAGE(left, total) true
#endif

// WI = cout fragments per consumer wave: 4 = the 128-cout tile; 2 = a 64-cout HALF tile (8 KiB of weights per K-step, half the MFMAs):
// the tiles of a launch's partial last round are issued as half tiles by a second launch (see the dispatcher: a 3.5-round layer leaves
// half the chip idle for a whole round; as half tiles the same work occupies every CU for ~0.6 of a round).
// Q: tiles behind the block's first one come from the launch's ticket queue (a.queue; ps_internal.h) instead of the static schedule: consumer
// wave 0 draws tile s + 1 when tile s starts and publishes it in front of the barrier of K-step nst - 7 (nst = 9 klines >= 18 K-steps per tile: the
// dispatcher sends one-K-line problems to the static kernel); the loaders' cursors cross into tile s + 1 from K-step nst - 6 on.  As LATE as that
// allows on purpose: collecting the ticket is a vmcnt(0) in a wave whose memory queue still holds the previous tile's epilogue stores, and while
// the memory system is busy those take microseconds (collected at K-step 8 the queue cost 2-6 % on the 3x3 layers of a training step).
// SK: stream-K over the partial last round (IgemmArgs::sk_*).  A block's work is a list of ITEMS (tile, K-lines [kl0, kl1)): its sk_dp whole tiles,
// then the pieces of its K-line share -- the three cursors (windows, weights, consumers) walk the same list; a piece that is not a whole tile ends
// in the slab hand-off instead of the epilogue.
template <typename Tr, int TW, int NW = 3, int WI = 4, bool Q = false, bool SK = false>  // TW = 28 (maps of 224-pixel tiles: 28 / 56 / 112 wide) or 32 (256-pixel tiles: 32 / 64 / 128 wide); NW = weight ring depth
__global__ __launch_bounds__(512, 2) void conv_igemm_halo_kernel(const IgemmArgs a) {
  static_assert(!(Q && SK), "stream-K runs on the static schedule");
  typedef typename Tr::elem T [[maybe_unused]];
  static_assert(TW == 28 || TW == 32, "tile width");
  static_assert(WI == 4 || WI == 2, "cout fragments per wave");
  constexpr int BN = 32 * WI, MI = TW / 4, WN = 16 * WI, TR = 8;  // tile = 8 rows x TW columns = 224 | 256 pixels, BN = 128 | 64 couts
  constexpr int WPW = BN / 32;  // weight DMA instructions per loader wave and K-step (4 | 2)
  constexpr int WJ = (TW + 8 + 3) / 4, WIN_BYTES = WJ * 4 * 1024, B_BYTES = BN * 128;  // window of <= TW + 8 columns: 9 | 10 DMAs per loader wave
  static_assert(WJ == 9 || WJ == 10, "window DMAs per loader wave");
  static_assert(NW >= 3 && NW <= 5 && 2 * WIN_BYTES + NW * B_BYTES <= 160 * 1024, "LDS budget");
  constexpr int W_OFF = 2 * WIN_BYTES;
  constexpr int MB_OFF = W_OFF + NW * B_BYTES;  // Q: the mailbox (16 bytes behind the weight ring)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int first, G, ntiles;  // this block's tiles: first, first + G, ... < ntiles  (Q: first, then whatever the queue hands out)
  [[maybe_unused]] unsigned q_tk = 0;  // Q, consumer wave 0: the ticket in flight
  [[maybe_unused]] bool q_peek = false; // Q (wave-uniform): whether the next draw also looks at the other classes' counters
  if constexpr (Q) {
    // EVERY tile comes from the queue, the first one too: a block that becomes resident late (its CU was held by another kernel) finds the
    // queue drained and leaves, instead of adding a whole tile to the launch's critical path.  The first ticket is drawn here, at the
    // very top, and collected behind the waves' tile-independent set-up (one extra block barrier).
    G = 0;
    first = -1;
    ntiles = a.ntm * a.ntn;
    q_peek = ps_q_count(ntiles, blockIdx.x & 7) <= 64;
#ifndef PS_Q_STATIC_TICKETS
    if (wave == 0) ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_peek, q_tk);
#endif
  } else {
    ps_block_items(blockIdx.x, gridDim.x, a.ntm * a.ntn, a.nb, a.tpb, first, G, ntiles);
  }
  // SK: this block's item list (wave-uniform scalars).  sblk = its stream-K index: blocks that share an XCD get consecutive indices, i.e.
  // neighbouring K-line shares -- the parts of one tile then mostly meet in one L2 (speed only)
  [[maybe_unused]] int sk_g0 = 0, sk_g1 = 0, sk_nitems = 0, sk_mylines = 0;
  [[maybe_unused]] const int sblk = SK ? first : 0;  // (ps_block_items: first = ps_xcd_remap(blockIdx.x, gridDim.x) under tpb = 0, grid = nb)
  if constexpr (SK) {
    sk_g0 = (int)((long long)a.sk_lines * sblk / a.nb);
    sk_g1 = (int)((long long)a.sk_lines * (sblk + 1) / a.nb);
    sk_nitems = a.sk_dp + (sk_g1 > sk_g0 ? (sk_g1 - 1) / a.klines - sk_g0 / a.klines + 1 : 0);
    sk_mylines = a.sk_dp * a.klines + (sk_g1 - sk_g0);
    if (sk_nitems == 0) return;  // (fewer K-lines than blocks)
  }
  [[maybe_unused]] auto sk_item = [&](int i, int& tile, int& kl0, int& kl1) {  // item i of this block
    if (i < a.sk_dp) {
      tile = first + i * G;
      kl0 = 0;
      kl1 = a.klines;
      return;
    }
    const int j = i - a.sk_dp;
    const int line = j == 0 ? sk_g0 : (sk_g0 / a.klines + j) * a.klines;
    const int tl = line / a.klines;
    tile = a.sk_tile0 + tl;
    kl0 = line - tl * a.klines;
    const int rest = sk_g1 - line;
    kl1 = kl0 + rest < a.klines ? kl0 + rest : a.klines;
  };
  const int nwin = 3 * a.klines;             // windows (K-line, ty) per tile, three K-steps each
  const int my_tiles = Q ? 1 : (ntiles - first + G - 1) / G;
  // Q: the loaders' three countdowns (windows, weight steps, consumed steps) count what is KNOWN to exist -- the first tile, plus one tile
  // every time the window cursor learns from the mailbox that there is a next one -- so every test below reads the same in both modes (and
  // hipcc proves the same things about them: the steady-state path must stay free of per-issue end tests)
  const int total_steps = SK ? sk_mylines * 9 : Q ? nwin * 3 : my_tiles * nwin * 3;
  [[maybe_unused]] const unsigned mbox = ps_q_mbox_addr(smem + MB_OFF);
  const int H = a.Hs, W = a.Ws, NH = a.M / W;  // stride 1: produced grid == source grid; NH = global rows n*H + p
  const int ncb = W / TW;                    // column blocks per row (W is a multiple of TW)
  const int dabs = a.dstep < 0 ? -a.dstep : a.dstep;
  const int WP = TW + 2 * dabs;              // window columns (<= 36)

  if (wave >= 4) {
    // ================= loader =================
    PS_LOADER_SETPRIO();
    const int lw = wave - 4;
    const int srow = lane >> 3;              // window row this lane stages (of every window column)
    const int chunk_off = ((lane & 7) ^ srow) << 4;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, (int)a.wgt_bytes, 0x00020000);
    // DMA instruction t = 4j + lw stages window COLUMN t (8 rows x 128 B = the instruction's 1 KiB): source column X0 + t - d
    int tcol[WJ];
#pragma unroll
    for (int j = 0; j < WJ; ++j) {
      const int t = j * 4 + lw;
      tcol[j] = t < WP ? t - dabs : -(1 << 20);
    }
    const int lane_off = srow * W * (int)a.pix_bytes + chunk_off;
    if constexpr (Q) {  // the block's first tile (published by consumer wave 0 in front of this barrier)
      __builtin_amdgcn_s_barrier();
      first = ps_q_mbox_read(mbox, 0);
      if (first < 0) return;
    }
    // --- cursor of the next WINDOW to stage: (tile, K-line, ty).  Window row wr of tap row ty serves exactly ONE produced row,
    // global row R0 + wr: it holds source row p + (ty-1)*dstep of the SAME image, or zeros (vertical padding; produced rows
    // past the tensor's end; rows that would come from the neighbouring image when a tile straddles two images).
    int a_tile = first, a_kl = 0, a_ty = 0, a_buf = 0, a_left = SK ? sk_mylines * 3 : Q ? nwin : my_tiles * nwin;
    [[maybe_unused]] int a_item = 0, a_kl_end = a.klines, b_item = 0, b_kl_end = a.klines;  // SK: item cursors of the window / weight streams
    // Q: the cursors cross a tile boundary in the order windows (K-step nst - 6), weights (nst - NW), and the windows do not cross the next one
    // before the weights have crossed this one: ONE mailbox read per tile, handed down in a register
    [[maybe_unused]] int a_seq = 0, q_next = -1;
    int c_left = total_steps;  // consumed steps left (Q: that are known of)
    int b_tile = first, b_kl = 0, b_tap = 0, b_slot = 0, b_left = total_steps;  // cursor of the next WEIGHT tile to stage (see below)
    if constexpr (SK) {
      sk_item(0, a_tile, a_kl, a_kl_end);
      b_tile = a_tile, b_kl = a_kl, b_kl_end = a_kl_end;
    }
    int a_R0 = 0, a_X0 = 0, prow = 0;  // prow: image row p of the produced row this lane's window row serves (per tile)
    auto window_tile_setup = [&](int tile) {
      int tm, tn;
      ps_tile_of_block(tile, a.ntn, a.ntm, tm, tn, a.supertile);
      tm += a.tm0;
      const int rb = tm / ncb;
      a_R0 = rb * TR;
      a_X0 = (tm - rb * ncb) * TW;
      const int gr = a_R0 + srow;
      prow = gr < NH ? gr - (gr / H) * H : -(1 << 20);
    };
    window_tile_setup(a_tile);
    auto issue_window = [&]() -> bool {
      if (a_left == 0) return false;
      const int shift = (a_ty - 1) * a.dstep;
      const int base = ((a_R0 + shift) * W + a_X0) * (int)a.pix_bytes + lane_off;
      const bool row_ok = (unsigned)(prow + shift) < (unsigned)H;
      unsigned char* dst = smem + a_buf * WIN_BYTES;
      const int ko = a_kl * 128;
#pragma unroll
      for (int j = 0; j < WJ; ++j) {
        const bool ok = row_ok && (unsigned)(a_X0 + tcol[j]) < (unsigned)W;  // left / right of the image: zero padding
        if (PS_HALO_STAGE(a_left, my_tiles * nwin)) BLDS16(rsA, dst + (j * 4 + lw) * 1024, ok ? (unsigned)(base + tcol[j] * (int)a.pix_bytes) : PAD_ROW, ko);
      }
      --a_left;
      a_buf ^= 1;
      if (++a_ty == 3) {
        a_ty = 0;
        if constexpr (SK) {
          if (++a_kl == a_kl_end && a_left) {
            sk_item(++a_item, a_tile, a_kl, a_kl_end);
            window_tile_setup(a_tile);
          }
        } else
        if (++a_kl == a.klines) {
          a_kl = 0;
          if constexpr (Q) {
            a_tile = q_next = ps_q_mbox_read(mbox, ++a_seq);
            if (a_tile >= 0) {
              a_left = nwin;
              b_left += nwin * 3;  // (declared below)
              c_left += nwin * 3;
            }
          } else {
            a_tile += G;
          }
          if (a_left) window_tile_setup(a_tile);
        }
      }
      return true;
    };
    // --- cursor of the next WEIGHT tile to stage: (tile, K-line, tap)
    unsigned woff[WPW];
    auto weights_setup = [&](int tile) {
      int tm, tn;
      ps_tile_of_block(tile, a.ntn, a.ntm, tm, tn, a.supertile);
      tm += a.tm0;
      const int n0 = tn * BN;
#pragma unroll
      for (int j = 0; j < WPW; ++j) {
        const int rb = (j * 4 + lw) * 8 + srow;
        const int wg = rb / WN, within = rb % WN, fi = within >> 4, rho = within & 15;
        const int cout = n0 + wg * WN + 4 * WI * (rho >> 2) + 4 * fi + (rho & 3);
        woff[j] = (unsigned)(cout * a.wrow_bytes) + chunk_off;
      }
    };
    weights_setup(b_tile);
    const int tap_bytes = a.klines * 128;  // Cs * esize
    auto issue_weights = [&]() -> bool {
      if (b_left == 0) return false;
      unsigned char* dst = smem + W_OFF + b_slot * B_BYTES;
      const int wk = b_tap * tap_bytes + b_kl * 128;
#pragma unroll
      for (int j = 0; j < WPW; ++j)
        if (PS_HALO_STAGE(b_left, total_steps)) BLDS16(rsB, dst + (j * 4 + lw) * 1024, woff[j], wk);
      --b_left;
      b_slot = (b_slot == NW - 1) ? 0 : b_slot + 1;
      if (++b_tap == 9) {
        b_tap = 0;
        if constexpr (SK) {
          if (++b_kl == b_kl_end && b_left) {
            sk_item(++b_item, b_tile, b_kl, b_kl_end);
            weights_setup(b_tile);
          }
        } else
        if (++b_kl == a.klines) {
          b_kl = 0;
          if constexpr (Q) {
            b_tile = q_next;
          } else {
            b_tile += G;
          }
          if (b_left) weights_setup(b_tile);
        }
      }
      return true;
    };
    // s_waitcnt vmcnt(n): all but the n newest DMAs of this wave have landed (the count needs a literal)
    auto wait_allow = [&](int n) {
      switch (n) {
#define PS_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        PS_VMCNT_CASE(2) PS_VMCNT_CASE(4) PS_VMCNT_CASE(6) PS_VMCNT_CASE(8) PS_VMCNT_CASE(12) PS_VMCNT_CASE(16)
        PS_VMCNT_CASE(9) PS_VMCNT_CASE(11) PS_VMCNT_CASE(13) PS_VMCNT_CASE(15) PS_VMCNT_CASE(17) PS_VMCNT_CASE(21) PS_VMCNT_CASE(25)
        PS_VMCNT_CASE(10) PS_VMCNT_CASE(14) PS_VMCNT_CASE(18) PS_VMCNT_CASE(22) PS_VMCNT_CASE(26)
#undef PS_VMCNT_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };
    // The weights run NW - 1 K-steps ahead, the next window is issued during the first step of the current one.  The pipeline is bound
    // by (bytes in flight) / (fill latency) -- a K-step took the same ~0.7 us with 7 or 8 pixel fragments per wave (r02: 224- vs
    // 256-pixel tiles), i.e. it waits for its operands, and with a 3-deep weight ring only 32 KiB of weights + one window were in flight
    // per CU.  vmcnt is ONE in-order queue: to consume the weights of step gs + 1 everything issued after them may stay in flight
    // -- the younger weight steps and, until its own third step, the window issued behind them.
    int w_issued = 0;        // weight K-steps issued so far
    int after_win = 0;       // weight K-steps issued after the window that is still pending
    bool win_pending = false;
    issue_window();
    for (int i = 0; i < NW - 1; ++i) w_issued += issue_weights() ? 1 : 0;
    wait_allow(WPW * (w_issued > 0 ? w_issued - 1 : 0));  // window 0 (oldest) and the weights of step 0 have landed
    __builtin_amdgcn_s_barrier();
    int r = 0;
    // STEADY STATE, three K-steps (one window) at a time while every issue below is known to succeed: the waits are then compile-time
    // literals -- queue behind the weights of step gs + 1: r = 0, 1: the younger weight steps and the whole next window; r = 2: the
    // window is older than them, only the weight steps issued after IT may stay in flight.  The loader waves' issue stream is part of
    // the K-step's critical path (NOTES 7.19 / 7.20): the general step below re-derives `allow` from four counters and reaches its
    // `s_waitcnt` through a 5-level compare tree (a switch over 18 literals) -- ~40 scalar instructions and half a dozen branches more
    // per K-step than this form.
    constexpr int ALLOW01 = WPW * (NW - 2) + WJ, ALLOW2 = WPW * ((NW - 2) < 2 ? (NW - 2) : 2);
    for (int gs = 0; Q ? c_left > 0 : gs < total_steps;) {
      if (r == 0 && b_left >= 3 && a_left >= 1) {
        issue_weights();
        issue_window();
        wait_allow(ALLOW01);
        __builtin_amdgcn_s_barrier();
        issue_weights();
        wait_allow(ALLOW01);
        __builtin_amdgcn_s_barrier();
        issue_weights();
        wait_allow(ALLOW2);
        __builtin_amdgcn_s_barrier();
        w_issued += 3;
        after_win = 2;
        win_pending = false;
        gs += 3;
        if constexpr (Q) c_left -= 3;
        continue;
      }
      if (issue_weights()) { ++w_issued; ++after_win; }   // weights of step gs + NW - 1
      if (r == 0) {
        win_pending = issue_window();                     // the NEXT window, behind this step's weights
        after_win = 0;
      }
      const int younger = w_issued - (gs + 2) > 0 ? w_issued - (gs + 2) : 0;  // weight steps issued after those of step gs + 1
      int allow;
      if (r == 2) {  // the next window must have landed as well: only the weight steps issued after it may stay in flight
        allow = WPW * (win_pending ? (younger < after_win ? younger : after_win) : younger);
        win_pending = false;
      } else {
        allow = WPW * younger + (win_pending ? WJ : 0);
      }
      wait_allow(allow);
      __builtin_amdgcn_s_barrier();
      r = (r == 2) ? 0 : r + 1;
      ++gs;
      if constexpr (Q) --c_left;
    }
    return;
  }

  // ================= consumer =================
  // waves 2x2: wm = column half of the tile (TW / 2 columns = MI fragments of 8 rows x 2 columns), wn = cout half
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 15, g = lane >> 4;
  const int wfrag = W_OFF + (wn * WN + frow) * 128;
  const int sw = lane & 7;
  const int coff0 = (g ^ sw) << 4, coff1 = ((g + 4) ^ sw) << 4;
  // The window is stored COLUMN-major (LDS row = 8 * window column + window row): a fragment's 16 lanes read 8 rows of 2
  // adjacent columns -- conflict-free with the usual chunk ^ (row & 7) swizzle, whose row & 7 is the WINDOW ROW and therefore
  // does not change with the tap column: the address for tap tx is the centre address + (tx-1)*dstep KiB, a scalar.
  const int tr = frow & 7;
  int xa[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int c = wm * (TW / 2) + 2 * mi + (frow >> 3) + dabs;  // window column of the centre tap
    xa[mi] = c * 1024 + tr * 128 + ((g ^ tr) << 4);
  }
  if constexpr (Q) {
    if (wave == 0) {
#ifdef PS_Q_STATIC_TICKETS
      ps_q_mbox_write(mbox, 0, (int)blockIdx.x < ntiles ? ps_xcd_remap(blockIdx.x, gridDim.x) : -1);
#else
      ps_q_mbox_write(mbox, 0, ps_q_resolve(a.queue, blockIdx.x & 7, lane, 0, ntiles, q_tk, q_peek));
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
  // Staggered start (a.stagger): the blocks of a persistent launch run their tiles in lock step, so all CUs reach the epilogue within the
  // same few microseconds and its loads / stores (172 KB per tile with a residual and two outputs) queue up at the memory fabric while every
  // matrix pipe idles (profiles/r05t_halo_epilogue_stamps.txt).  Phase-shifting the blocks of each XCD spreads that burst over the tile period.
  // (Measured, debug library only: no gain at 1-3 x 2048 cycles in 2 / 4 / 8 phases on any layer -- the epilogue's rate is the CU's own store
  // path, not the fabric's: profiles/r05t_halo_stagger_ab.txt.)
#ifdef PS_DEBUG_HOOKS
  if (a.stagger > 0) {
    const int ph = ((int)blockIdx.x >> 3) % a.stagger_phases;
    for (int i = 0; i < ph * a.stagger; ++i) __builtin_amdgcn_s_sleep(32);  // 32 x 64 cycles
  }
#endif
  __builtin_amdgcn_s_barrier();  // window 0 / weights of step 0 visible

  int cur = 0, wbuf = 0;
#ifdef PS_HALO_STAMPS
  unsigned long long st_seg0 = 0, st_seg1 = 0, st_seg2 = 0, st_steps = 0, st_tail = 0, st_epi = 0, st_prev;
  PS_STAMP(st_prev);
  const unsigned long long st_c0 = st_prev, st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  [[maybe_unused]] int q_seq = 0;
  [[maybe_unused]] int sk_i = 0, kl0 = 0, kl1 = a.klines;  // SK: the consumers' item cursor
  int tile = first;
  if constexpr (SK) sk_item(0, tile, kl0, kl1);
  for (; SK ? sk_i < sk_nitems : (Q ? tile >= 0 : tile < ntiles);) {
    const int nst = SK ? 9 * (kl1 - kl0) : 3 * nwin;  // K-steps of this item
    if constexpr (Q) {
#ifndef PS_Q_STATIC_TICKETS  // (diagnostic build: the queue's code paths fed with the static schedule -- no atomics; what the restructuring alone costs)
      if (wave == 0) ps_q_draw_begin(a.queue, blockIdx.x & 7, lane, q_peek, q_tk);  // the ticket of this block's tile q_seq + 1: in flight until K-step nst - 7
#endif
    }
    f32x4 acc[MI][WI];
#ifndef PS_HALO_MFMA32_TIMING
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    u32x4 wf0[WI], xf0[MI], wf1[WI], xf1[MI];
#ifdef PS_HALO_MFMA32_TIMING
    // A/B TIMING build (results WRONG): the same reads, loaders, barriers and accumulator registers, but every PAIR of 16x16x32 MFMAs is
    // replaced by ONE v_mfma_f32_32x32x16 (same FLOPs, half the MFMA instructions, 32-cycle issue gaps that hide the ds_read_b128
    // issue): what the consumer loop would gain from the 32x32 shape before anyone rewrites the fragment layout and the epilogue
    // (tools/mfma_shape_probe.hip: +25 % for a lone MFMA wave per SIMD; profiles/r03_mfma_shape_probe.txt).
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 acc16[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc16[mi][e] = 0.f;
#endif
    // One K-half: the NR = WI + MI fragment reads interleaved with the MFMAs whose operands are already in registers
    // (see conv_igemm_ws2_kernel); a pixel fragment address = centre address + scalar (window buffer, tap column) [^ K-half].
    auto half = [&](const unsigned char* wst, int wcoff, int soff, int flip, u32x4 (&wfn)[WI], u32x4 (&xfn)[MI], const u32x4 (&wfo)[WI],
                    const u32x4 (&xfo)[MI], bool do_mma) {
#ifdef PS_HALO_MFMA32_TIMING
      constexpr int NR = WI + MI, NM = MI * WI / 2, PER = 1;
      auto mma32 = [&](int q) {
        acc16[q % MI] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfo[(q / MI) % WI]), __builtin_bit_cast(bf16x8, xfo[q % MI]),
                                                                acc16[q % MI], 0, 0, 0);
      };
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r < WI) wfn[r] = *reinterpret_cast<const u32x4*>(wst + r * 2048 + wcoff);
        else xfn[r - WI] = *reinterpret_cast<const u32x4*>(smem + ((xa[r - WI] ^ flip) + soff));
        __builtin_amdgcn_sched_barrier(0);
        if (do_mma) {
          if (r < NM) mma32(r);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (do_mma) {
#pragma unroll
        for (int q = NR; q < NM; ++q) mma32(q);
      }
      __builtin_amdgcn_sched_barrier(0);
      (void)PER;
      return;
#else
      constexpr int NR = WI + MI, NM = MI * WI, PER = Tr::split ? 3 : PS_READ_PER;  // (split types: three MFMA groups per K-line, see conv_igemm_ws2_kernel)
      static_assert(!Tr::split || (NM - 1) / 3 < WI + (NM - 1) / WI, "split: every MFMA on the old pixel fragments precedes their replacement");
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#ifndef PS_HALO_NO_READS  // (diagnostic build, results WRONG: MFMAs on stale registers -- the bare MFMA stream between the barriers)
        if (r < WI) wfn[r] = *reinterpret_cast<const u32x4*>(wst + r * 2048 + wcoff);
        else xfn[r - WI] = *reinterpret_cast<const u32x4*>(smem + ((xa[r - WI] ^ flip) + soff));
#else
        if (r < WI) asm volatile("" : "+v"(wfn[r]));
        else asm volatile("" : "+v"(xfn[r - WI]));
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (do_mma) {
#pragma unroll
          for (int q = r * PER; q < (r + 1) * PER && q < NM; ++q) Tr::mma(wfo[q % WI], xfo[q / WI], acc[q / WI][q % WI]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (do_mma) {
#pragma unroll
        for (int q = NR * PER; q < NM; ++q) Tr::mma(wfo[q % WI], xfo[q / WI], acc[q / WI][q % WI]);
      }
      __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto kstep = [&](int tx, bool first_of_tile) {
      const unsigned char* wst = smem + cur * B_BYTES + wfrag;
      const int soff = wbuf * WIN_BYTES + (tx - 1) * a.dstep * 1024;
#ifndef PS_HALO_NO_CONSUMERS  // (A/B build, results WRONG: the consumers only keep the barriers -- what the loaders alone sustain)
      if constexpr (Tr::split) half(wst, coff0, soff, 0, wf0, xf0, wf1, xf0, !first_of_tile);  // x_hi(previous step) . w_lo(previous step) while this step's hi arrives
      else half(wst, coff0, soff, 0, wf0, xf0, wf1, xf1, !first_of_tile);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef PS_HALO_STAMPS
      unsigned long long t_a, t_b, t_c;
      PS_STAMP(t_a);
      st_seg0 += t_a - st_prev;
#endif
      __builtin_amdgcn_sched_barrier(0);
      half(wst, coff1, soff, 64, wf1, xf1, wf0, xf0, true);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (Tr::split) {  // x_lo . w_hi, both in registers
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < WI; ++i) Tr::mma(wf0[i], xf1[mi], acc[mi][i]);
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef PS_HALO_STAMPS
      PS_STAMP(t_b);
      st_seg1 += t_b - t_a;
#endif
#endif
      __builtin_amdgcn_s_barrier();
#ifdef PS_HALO_STAMPS
      PS_STAMP(t_c);
      st_seg2 += t_c - t_b;
      st_prev = t_c;
      ++st_steps;
#endif
      cur = (cur == NW - 1) ? 0 : cur + 1;
    };
    // first K-step of the tile peeled (no previous MFMAs to overlap its reads with), then tx = 1, 2, 0, 1, 2, ...
    kstep(0, true);
    int tx = 1;
    for (int s = 1; s < nst; ++s) {
      if constexpr (Q) {
#ifdef PS_Q_STATIC_TICKETS
        if (s == 3 * nwin - 7 && wave == 0) ps_q_mbox_write(mbox, q_seq + 1, tile + (int)gridDim.x < ntiles ? tile + (int)gridDim.x : -1);
#else
        if (s == 3 * nwin - 7 && wave == 0)
          ps_q_mbox_write(mbox, q_seq + 1, ps_q_resolve(a.queue, blockIdx.x & 7, lane, G, ntiles, q_tk, q_peek));
#endif
      }
      kstep(tx, false);
      if (++tx == 3) { tx = 0; wbuf ^= 1; }
    }
    // (after the last step tx wrapped to 0 and wbuf moved on to the next tile's first window)
#ifndef PS_HALO_MFMA32_TIMING
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < WI; ++i) Tr::mma(wf1[i], Tr::split ? xf0[mi] : xf1[mi], acc[mi][i]);  // (split: the last step's x_hi . w_lo)
#ifdef PS_HALO_STAMPS
    {
      asm volatile("s_nop 15\n s_nop 15" ::: "memory");  // (the MFMA queue drains before the stamp: the last results' 16 passes)
      unsigned long long t_m;
      PS_STAMP(t_m);
      st_tail += t_m - st_prev;
      st_prev = t_m;
    }
#endif
#else
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[mi][i][r] = acc16[mi][(4 * i + r) & 15] + __uint_as_float(wf1[i][r] & 0xffu) + __uint_as_float(xf1[mi][r] & 0xffu);
#endif
    int tm, tn;
    ps_tile_of_block(tile, a.ntn, a.ntm, tm, tn, a.supertile);
      tm += a.tm0;
    const int rb = tm / ncb;
    bool finish = true;  // whether this wave runs the tile's epilogue
    if constexpr (SK) {
      if (kl1 - kl0 < a.klines) {
        // A PART of a tile: hand the wave's accumulators over.  Slab stores are write-through (sc1: they leave this XCD's L2, no release
        // fence needed), drained by this wave's vmcnt(0); then ONE agent-scope add on the (tile, wave) arrival counter.  The wave whose
        // add returns nparts - 1 came last: every other part is complete in memory, it reads them with sc1 loads (L2-served, never this
        // CU's L1: MI355X_MICROARCH.md, inter-workgroup visibility) IN PART ORDER and finishes the tile.
        const int tl = tile - a.sk_tile0, x0 = tl * a.klines, x1 = x0 + a.klines - 1;
        const int bfirst = (int)((((long long)x0 + 1) * a.nb - 1) / a.sk_lines), blast = (int)((((long long)x1 + 1) * a.nb - 1) / a.sk_lines);
        const int nparts = blast - bfirst + 1, part = sblk - bfirst;
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)a.sk_slabs, 0, (int)a.sk_slab_bytes, 0x00020000);
        constexpr int SLAB = MI * WI * 1024;  // bytes per (part, wave)
        const int tbase = tl * a.sk_maxparts * 4 * SLAB + wave * SLAB + lane * 16;
        const int mine = tbase + part * 4 * SLAB;
        // The owner of a tile's HEAD part (kl0 == 0) reaches it as the LAST item of its share, the owners of the later parts reach theirs first
        // (at most a whole share earlier): it usually finds every other part complete.  One look at the counter then saves its own slab --
        // a third of the exchange's bytes, which is what the exchange costs (all CUs hand over within the same few microseconds).
        bool skip_store = false;
        if (kl0 == 0) {
          unsigned seen = 0;
          if (lane == 0) seen = __hip_atomic_load(a.sk_cnt + tl * 4 + wave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          skip_store = (int)__builtin_amdgcn_readfirstlane((int)seen) == nparts - 1;  // every other part has drained its stores and counted itself
        }
        unsigned old = (unsigned)(nparts - 1);
        if (!skip_store) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int i = 0; i < WI; ++i)
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[mi][i]), rsS, mine + (mi * WI + i) * 1024, 0, 16);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          old = 0;
          if (lane == 0) old = __hip_atomic_fetch_add(a.sk_cnt + tl * 4 + wave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          old = (unsigned)__builtin_amdgcn_readfirstlane((int)old);
        }
        finish = (int)old == nparts - 1;
        if (finish) {
          if (lane == 0) __hip_atomic_store(a.sk_cnt + tl * 4 + wave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-armed for the stream's next launch
          // the sum in part order 0, 1, ...: with two parts a + b is the same either way round (own part stays in registers); with more the
          // own part is re-read from its slab in its place, so that the result does not depend on who came last
          // the sum in part order 0, 1, ...: the head part (part 0: this wave's own whenever it skipped its store) comes first, so the own
          // accumulators are the running sum's start; otherwise the own part sits at `part` and is re-read from its slab in its place -- the
          // result never depends on who came last.  (Two parts: a + b either way round.)
          if (nparts == 2 || part == 0) {
            for (int pp = 0; pp < nparts; ++pp) {
              if (pp == part) continue;
              const int src = tbase + pp * 4 * SLAB;
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                u32x4 t[WI];
#pragma unroll
                for (int i = 0; i < WI; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(rsS, src + (mi * WI + i) * 1024, 0, 16);
#pragma unroll
                for (int i = 0; i < WI; ++i) acc[mi][i] += __builtin_bit_cast(f32x4, t[i]);
              }
            }
          } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int i = 0; i < WI; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int pp = 0; pp < nparts; ++pp) {
              const int src = tbase + pp * 4 * SLAB;
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) {
                u32x4 t[WI];
#pragma unroll
                for (int i = 0; i < WI; ++i) t[i] = __builtin_amdgcn_raw_buffer_load_b128(rsS, src + (mi * WI + i) * 1024, 0, 16);
#pragma unroll
                for (int i = 0; i < WI; ++i) acc[mi][i] += __builtin_bit_cast(f32x4, t[i]);
              }
            }
          }
        }
      }
    }
    // (A row-wise version of this epilogue -- every tile-shaped operand transposed through 2 KiB of LDS scratch per wave so that each
    // store / load instruction covers eight whole 128-byte rows, 3 x faster per CU in tools/epilogue_shape_probe.hip -- was built, bit-identical,
    // and measured 25 % SLOWER in this kernel (consumer stamps: 6.1 k instead of 4.8 k cycles store-only, 17-18 k instead of 13-14 k with a
    // residual and two outputs): tools/experiments/r05_halo_rows_epilogue.diff, profiles/r05u_halo_rows_epilogue_stamps.txt.)
    if (finish) conv_epilogue<typename Tr::epi, MI, WI, 1>(a, acc, rb * TR * W + (tm - rb * ncb) * TW + wm * (TW / 2), tn * BN + wn * WN, lane);
#ifdef PS_HALO_STAMPS
    {
      unsigned long long t_e;
      PS_STAMP(t_e);  // the epilogue is not part of segment 0 of the next tile's first step
      st_epi += t_e - st_prev;
      st_prev = t_e;
    }
#endif
    if constexpr (SK) {
      if (++sk_i < sk_nitems) sk_item(sk_i, tile, kl0, kl1);
    } else if constexpr (Q) {
      tile = ps_q_mbox_read(mbox, ++q_seq);
    } else {
      tile += G;
    }
  }
  if constexpr (Q) {
#ifndef PS_Q_STATIC_TICKETS
    if (wave == 0) ps_q_block_done(a.queue, lane, gridDim.x);  // (every draw of this block has returned: the last one was resolved in the last tile)
#endif
  }
#ifdef PS_HALO_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
    unsigned long long* o = g_halo_stamps + (blockIdx.x * 4 + wave) * 8;
    o[0] = st_seg0; o[1] = st_seg1; o[2] = st_seg2; o[3] = st_steps;
    o[4] = __builtin_amdgcn_s_memtime() - st_c0; o[5] = __builtin_amdgcn_s_memrealtime() - st_r0;
    o[6] = st_tail; o[7] = st_epi;
  }
#endif
}

#ifdef PS_DEBUG_HOOKS
// ------------------------------------------------------------------------------------------------
// EXPERIMENT, debug library only (r04; built, bit-identical to the three launches it replaces, measured NO faster and therefore not used
// by the model: profiles/r04_front_fusion.txt).  The front of the net in one launch: conv1a (3x3, 3 -> 64, stride 1, pad 1, NCHW f32 image)
// + the first ResBlock's BN + ReLU (the activation `a`), consumed in place by that block's two stride-2 convolutions -- shortcut =
// conv_branch1(a) (1x1 s2, raw) and a2 = relu(bn(conv_branch2a(a))) (3x3 s2, pad 1) -- models/resnet38d.py:123,161-162 + ResBlock.forward
// :28-41 of b2.  `a` ([N, H, W, 64]: 411 MB at bs = 64, the net's largest tensor) is never written to HBM; it exists one row at a time in LDS.
// Applicable where `a` is not needed again: inference, and training of the models that freeze b2 (revise_net.py:27, the segmentation model).
//   block = ONE output row (n, oy): Wo = 16 NF output pixels x 128 + 128 produced channels; 4 waves = 4 cout groups of 32, each over the whole row;
//   for r = 0..2 (rows 2 oy - 1 + r of `a`): phase 1 -- conv1a of the row's 2 Wo + 1 window pixels (columns -1 .. 2 Wo - 1) on the MFMA
//   (K = 27 padded to 32: im2col operands gathered from the image rows staged in LDS as 16-bit values), BN + ReLU, zero outside the image,
//   written to LDS as 64-channel rows in two column-parity planes (a stride-2 tap then reads CONSECUTIVE rows: conflict-free with the usual
//   chunk ^ (row & 7) swizzle); phase 2 -- the three taps (r, 0..2) of the 3x3 stride-2 conv (+ the 1x1 on the centre tap, same pixel
//   fragments), weights straight from global memory (L2-resident, 160 KB) into MFMA operands, one tap ahead.  Plain block barriers between
//   the phases; two 4-wave blocks per CU (36 KiB of LDS each).
//   Arithmetic as conv1a_lowp_kernel + the 16-bit conv kernels: image and conv1a weights rounded to the storage type, f32 accumulation,
//   `a` rounded to the storage type before the stride-2 convs read it (what the unfused path stores).
//   Measured (bs = 64, 224 x 224, bf16): 432 us against 148 + 74 + 208 us for conv1a, the 1x1 and the 3x3 stride-2 launches: every phase of a
//   block is a chain of exposed latencies (image rows and weight fragments from L2, the im2col gather, seven barriers per output row) with only
//   two blocks per CU to overlap them; a persistent variant with the constants in LDS and batched weight / image prefetches spilled 180
//   registers (679 us).  What it needs is the other kernels' structure: loader waves that stage image rows and weight taps through LDS ahead of
//   the MFMA waves, and `a` rows kept across output rows (two of three are recomputed here).
struct FrontArgs {
  const float* image;   // [N][3][H][W]
  const float* w1a;     // [64][3][3][3]
  const float* sc0;     // BN affine of `a` (64)
  const float* sh0;
  const unsigned char* w_b1;  // [128][64] 16-bit
  const unsigned char* w_2a;  // [128][9][64] 16-bit
  int N, H, W;
  IgemmArgs eb1, e2a;   // epilogue descriptors of the two outputs (conv_epilogue reads epi, Wo, Ho, M, epi_M, Cd, m_off)
};

template <typename Tr, int NF>
__global__ __launch_bounds__(256, 2) void conv_front_s2_kernel(const FrontArgs f) {
  typedef typename Tr::elem T;
  constexpr int WO = 16 * NF, WC = 2 * WO + 1, PLANE = WO + 1;   // window columns c = 0 .. 2 WO <-> image columns c - 1
  constexpr int WIN_BYTES = 2 * PLANE * 128;                       // one `a` row: [column parity][index] rows of 64 channels
  constexpr int IW = 2 * WO + 4;                                   // staged image columns jc = 0 .. 2 WO + 2 <-> image columns jc - 2
  constexpr int IMG_OFF = WIN_BYTES;                               // ... followed by the staged image rows: 3 x 5 x IW 16-bit values
  constexpr int PF = (WC + 15) / 16;                               // window pixel fragments per row
  constexpr int MI = NF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, g = lane >> 4;
  const int Ho = f.H / 2;
  const int img_n = blockIdx.x / Ho, oy = blockIdx.x - img_n * Ho;
  auto to_bits = [](float v) -> unsigned {
    if constexpr (std::is_same<T, __bf16>::value) return ps_f32_to_bf16(v);
    else return ps_f32_to_f16(v);
  };
  // ---- the five image rows 2 oy - 2 .. 2 oy + 2 (zero outside the image), three channels, as 16-bit values
  for (int i = tid; i < 3 * 5 * IW; i += 256) {
    const int ch = i / (5 * IW), rem = i - ch * 5 * IW, j = rem / IW, jc = rem - j * IW;
    const int iy = 2 * oy - 2 + j, ix = jc - 2;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)f.H && (unsigned)ix < (unsigned)f.W) v = f.image[(((long long)img_n * 3 + ch) * f.H + iy) * f.W + ix];
    reinterpret_cast<unsigned short*>(smem + IMG_OFF)[i] = (unsigned short)to_bits(v);
  }
  // ---- conv1a weights as MFMA A operands: fragment i = couts 16 i .. 16 i + 15, lane (row n16, K group g) holds K = 8 g .. 8 g + 7 of
  // k = c * 9 + ky * 3 + kx (27 real, zero above); and the lane's 8 constant byte offsets into the staged image
  u32x4 w1f[4];
  int koff[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * g + j, kc = k < 27 ? k : 0;
    const int c = kc / 9, ky = (kc - 9 * c) / 3, kx = kc - 9 * c - 3 * ky;
    koff[j] = ((c * 5 + ky) * IW + kx) * 2;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * g + j;
      h[j] = k < 27 ? to_bits(f.w1a[(16 * i + n16) * 27 + k]) : 0u;
    }
    w1f[i] = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
  }
  // BN affine of `a` for this lane's couts 16 i + 4 g + e
  float sc0[4][4], sh0[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a4 = *reinterpret_cast<const float4*>(f.sc0 + 16 * i + 4 * g), b4 = *reinterpret_cast<const float4*>(f.sh0 + 16 * i + 4 * g);
    sc0[i][0] = a4.x; sc0[i][1] = a4.y; sc0[i][2] = a4.z; sc0[i][3] = a4.w;
    sh0[i][0] = b4.x; sh0[i][1] = b4.y; sh0[i][2] = b4.z; sh0[i][3] = b4.w;
  }
  // ---- phase 2 role: cout group `wave` (32 couts: two fragments in the epilogue's channel order) over the row's NF pixel fragments
  const int n0 = 32 * wave;
  f32x4 acc2[MI][2], acc1[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < 2; ++i) acc2[mi][i] = acc1[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // weight fragment (i, kh) of tap t: row rho = n16 is cout n0 + 8 (rho >> 2) + 4 i + (rho & 3), K elements kh * 32 + 8 g .. + 7
  const int wrow = n0 + 8 * (n16 >> 2) + (n16 & 3);
  auto load_w2 = [&](int t, u32x4 (&wf)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kh = 0; kh < 2; ++kh)
        wf[i][kh] = *reinterpret_cast<const u32x4*>(f.w_2a + (((long long)(wrow + 4 * i) * 9 + t) * 64 + kh * 32 + 8 * g) * 2);
  };
  u32x4 wb1[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) wb1[i][kh] = *reinterpret_cast<const u32x4*>(f.w_b1 + ((long long)(wrow + 4 * i) * 64 + kh * 32 + 8 * g) * 2);
  u32x4 wcur[2][2], wnxt[2][2];
  load_w2(0, wcur);
  __syncthreads();  // image rows staged
  for (int r = 0; r < 3; ++r) {
    // ---- phase 1: `a` row 2 oy - 1 + r into the window (wave w takes fragments w, w + 4, ...)
    const bool row_ok = (unsigned)(2 * oy - 1 + r) < (unsigned)f.H;
    for (int pf = wave; pf < PF; pf += 4) {
      const int c = 16 * pf + n16;                      // window column of this lane's pixel (image column c - 1)
      const int cc = c < WC ? c : WC - 1;
      const unsigned char* base = smem + IMG_OFF + (r * IW + cc) * 2;
      unsigned h[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) h[j] = *reinterpret_cast<const unsigned short*>(base + koff[j]);
      if (g == 3) { h[3] = h[4] = h[5] = h[6] = h[7] = 0u; }  // K = 27 .. 31
      const u32x4 xf = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
      const bool ok = row_ok && c >= 1 && c < WC && c - 1 < f.W;
      const int R = (c & 1) * PLANE + (c >> 1);         // LDS row of the pixel: [parity][index]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 d = f32x4{0.f, 0.f, 0.f, 0.f};
        Tr::mma(w1f[i], xf, d);                         // d[e] = cout 16 i + 4 g + e of pixel n16
        unsigned q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = ok ? to_bits(fmaxf(d[e] * sc0[i][e] + sh0[i][e], 0.f)) : 0u;
        if (c < WC) {
          const int chunk = 2 * i + (g >> 1);           // 16-byte chunk of couts 16 i + 4 g .. + 3, its 8-byte half g & 1
          *reinterpret_cast<uint2*>(smem + R * 128 + ((chunk ^ (R & 7)) << 4) + (g & 1) * 8) = uint2{q[0] | (q[1] << 16), q[2] | (q[3] << 16)};
        }
      }
    }
    __syncthreads();
    // ---- phase 2: taps (r, 0..2); the 1x1 conv rides on the centre tap
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int t = r * 3 + kx;
      if (t < 8) load_w2(t + 1, wnxt);
      const int Rb = (kx & 1) * PLANE + (kx >> 1) + n16;  // window row of output pixel n16 for this tap
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        u32x4 xf[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const int R = Rb + 16 * mi;
          xf[mi] = *reinterpret_cast<const u32x4*>(smem + R * 128 + (((kh * 4 + g) ^ (R & 7)) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            Tr::mma(wcur[i][kh], xf[mi], acc2[mi][i]);
            if (r == 1 && kx == 1) Tr::mma(wb1[i][kh], xf[mi], acc1[mi][i]);
          }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) wcur[i][kh] = wnxt[i][kh];
    }
    __syncthreads();  // the window may be overwritten
  }
  // ---- epilogues: a2 = relu(bn(acc2)) and the raw shortcut, the row's pixels (n, oy, 0 .. Wo - 1)
  const int mbase = (img_n * Ho + oy) * WO;
  conv_epilogue<T, MI, 2, 0>(f.e2a, acc2, mbase, n0, lane);
  conv_epilogue<T, MI, 2, 0>(f.eb1, acc1, mbase, n0, lane);
}
#endif  // PS_DEBUG_HOOKS

#ifdef PS_DEBUG_HOOKS
constexpr bool kDebugBuild = true;
#else
constexpr bool kDebugBuild = false;
#endif
PS_TUNABLE g_use_glds = 2;  // staging mode: 0 registers, 1 global_load_lds, 2 buffer_load ... lds
PS_TUNABLE g_use_pp = 0;       // experimental ping-pong kernel (correct, slower: r01 measurements)
PS_TUNABLE g_use_ws = 1;       // wave-specialised (loader/consumer) kernel for big problems
PS_TUNABLE g_use_ws2 = 1;
PS_TUNABLE g_use_halo = 1;     // window + halo staging for 3x3 stride-1 layers (width a multiple of 28): 0 off, 1 for big 16-bit problems, 2 forced      // large-tile wave-specialised kernel: 0 off, 1 by cost model, 256 / 224 force that pixel tile
#ifndef PS_HALO_SK
#define PS_HALO_SK 2  // (A/B builds: -DPS_HALO_SK=0|1)
#endif
PS_TUNABLE g_halo_sk = PS_HALO_SK;  // halo kernel: stream-K finish of a partial last round (needs ps_epilogue.sk_ws): 0 off, 1 on unless gpu_shared, 2 on
PS_TUNABLE g_halo_stagger = 0, g_halo_stagger_phases = 4;  // halo kernel: staggered start of a launch's blocks (IgemmArgs::stagger), units of 2048 cycles
PS_TUNABLE g_halo_tail = 1;   // halo kernel: a partial last round (<= half the CUs) as a second launch of 64-cout half tiles: 0 off, 1 on
PS_TUNABLE g_halo_ring = 3;   // weight ring depth of the halo kernel (3 | 4 | 5 stages of 16 KiB; 256-pixel tiles: <= 4)
PS_TUNABLE g_gemm256 = 1;      // 256 x 256 tile kernel for the plain GEMMs (1x1 stride-1, 16-bit): 0 off, 1 by shape, 2 whenever legal
PS_TUNABLE g_gemm256_min_klines = 32, g_gemm256_min_cd = 1024;  // the by-shape rule's thresholds: K / 64 and produced channels
PS_TUNABLE g_gemm256_min_tiles = 0;  // ... and its minimum launch size in tiles: 0 = 1.5 rounds of the device's CUs (384 on the 256-CU MI355X), > 0 forces a count
PS_TUNABLE g_gemm256_tail = 1; // gemm256: a partial last round of at most half the CUs goes to a second launch on the gathered-tile kernels: 0 off, 1 on
PS_TUNABLE g_use_3stage = 0;  // experimental 256x128 three-stage kernel: correct but slower than two 128x128 blocks per CU (r01 measurements)
PS_TUNABLE g_ablate = 0;
PS_TUNABLE g_supertile = 4;  // measured best of {0,4,8,16} on the wide 28x28 layers (r01)
PS_TUNABLE g_force_bm = 0;  // testing: 112 or 128 forces the pixel-tile height of the 128-cout kernel
PS_TUNABLE g_force_bn = 0;  // testing: 64 or 128 forces the 2-stage tile width
PS_TUNABLE g_s2split = 1;  // stride-2 3x3 data gradient as four parity-class launches: 0 off, 1 for big 16-bit problems, 2 whenever legal

template <typename Tr, int BM, int BN, int WMW, int WNW>
int launch_igemm(const IgemmArgs& a0, hipStream_t stream) {
  IgemmArgs a = a0;
  a.ntn = a.Cd / BN;
  a.ntm = (a.M + BM - 1) / BM;
  const int grid = a.ntm * a.ntn;
  const size_t lds = 2 * (BM * 128 + BN * 128);
  const dim3 block(64 * WMW * WNW);
  // (the alternative staging modes, like every kernel variant below that only a `ps_debug_set_*` switch can select, are instantiated in
  // the debug library only: the product library carries no code it cannot reach)
#ifdef PS_DEBUG_HOOKS
  if (g_use_glds == 1) {
    hipLaunchKernelGGL((conv_igemm_kernel<Tr, BM, BN, WMW, WNW, 1>), dim3(grid), block, lds, stream, a);
  } else if (g_use_glds == 0) {
    hipLaunchKernelGGL((conv_igemm_kernel<Tr, BM, BN, WMW, WNW, 0>), dim3(grid), block, lds, stream, a);
  } else
#endif
  {
    hipLaunchKernelGGL((conv_igemm_kernel<Tr, BM, BN, WMW, WNW, 2>), dim3(grid), block, lds, stream, a);
  }
  PS_CHECK_LAUNCH("conv_igemm");
  return PS_OK;
}

int check_geom(const ps_conv_geom* g) {
  PS_REQUIRE(g != nullptr, "conv: null geometry");
  PS_REQUIRE(ps_conv_dtype_ok(g->dtype), "conv: dtype %d unsupported", g->dtype);
  PS_REQUIRE(g->ksize == 1 || g->ksize == 3, "conv: ksize %d unsupported (1 or 3)", g->ksize);
  PS_REQUIRE(g->stride == 1 || g->stride == 2, "conv: stride %d unsupported (1 or 2)", g->stride);
  PS_REQUIRE(g->dilation >= 1 && g->dilation <= 64, "conv: dilation %d unsupported", g->dilation);
  PS_REQUIRE(g->n > 0 && g->h > 0 && g->w > 0, "conv: empty input %dx%dx%d", g->n, g->h, g->w);
  PS_REQUIRE(g->tiles_per_block >= 0 && g->tiles_per_block <= 4096, "conv: tiles_per_block %d out of range", g->tiles_per_block);
  PS_REQUIRE(g->gpu_shared == 0 || g->gpu_shared == 1, "conv: gpu_shared %d (0 or 1)", g->gpu_shared);
  PS_REQUIRE(g->cus_reserved >= 0 && g->cus_reserved <= 4096, "conv: cus_reserved %d out of range", g->cus_reserved);
  PS_REQUIRE(g->tile_queue == 0 || g->tile_queue == 1, "conv: tile_queue %d (0 or 1)", g->tile_queue);
  const int es = ps_esize(g->dtype);
  PS_REQUIRE((g->cin * es) % 128 == 0 && (g->cout * es) % 128 == 0,
             "conv: cin=%d cout=%d must be multiples of %d channels", g->cin, g->cout, 128 / es);
  PS_REQUIRE(g->cin % 64 == 0 && g->cout % 64 == 0, "conv: cin=%d cout=%d must be multiples of 64", g->cin, g->cout);
  PS_REQUIRE(g->ldc_x >= ps_planes(g->dtype) * g->cin && g->ldc_y >= ps_planes(g->dtype) * g->cout, "conv: channel strides smaller than channel counts");
  PS_REQUIRE((g->ldc_x * es) % 16 == 0 && (g->ldc_y * es) % 16 == 0, "conv: channel strides must be 16-byte multiples");
  const long long ho = (g->h - 1) / g->stride + 1, wo = (g->w - 1) / g->stride + 1;
  PS_REQUIRE((long long)g->n * g->h * g->w < (1LL << 31) && (long long)g->n * ho * wo < (1LL << 31), "conv: too many pixels");
  return PS_OK;
}

int check_epilogue(const ps_epilogue* e, int dtype, const char* who, int cd) {
  PS_REQUIRE(e != nullptr, "%s: null epilogue", who);
  PS_REQUIRE(e->mode >= PS_EPI_NONE && e->mode <= PS_EPI_RELUBWD, "%s: bad epilogue mode %d", who, e->mode);
  PS_REQUIRE(e->out_raw || e->mode != PS_EPI_NONE, "%s: epilogue produces no output", who);
  PS_REQUIRE(e->mode == PS_EPI_NONE || e->out, "%s: epilogue mode %d needs out", who, e->mode);
  PS_REQUIRE(e->mode != PS_EPI_RELUBWD || e->mask_src, "%s: RELUBWD epilogue needs mask_src", who);
  const int es = ps_esize(dtype);
  const struct { const void* p; int ldc; const char* nm; } t[] = {
      {e->add0, e->ldc_add0, "add0"}, {e->out_raw, e->ldc_raw, "out_raw"}, {e->mask_src, e->ldc_mask, "mask_src"},
      {e->add1, e->ldc_add1, "add1"}, {e->out, e->ldc_out, "out"}};
  for (const auto& x : t) {
    if (!x.p) continue;
    PS_REQUIRE(ps_aligned16(x.p) && (x.ldc * es) % 16 == 0 && x.ldc > 0, "%s: epilogue tensor %s misaligned (ptr %p ldc %d)", who, x.nm, x.p, x.ldc);
    PS_REQUIRE(x.ldc >= ps_planes(dtype) * cd, "%s: epilogue tensor %s has channel stride %d < %d produced channels x %d planes", who, x.nm, x.ldc, cd, ps_planes(dtype));
  }
  if (e->out_hi) {  // the plain 16-bit copy of `out`'s hi halves (split types)
    PS_REQUIRE(ps_planes(dtype) == 2 && e->out, "%s: out_hi goes with the split types' `out`", who);
    PS_REQUIRE(ps_aligned16(e->out_hi) && e->ldc_hi >= cd && (e->ldc_hi * 2) % 16 == 0, "%s: out_hi misaligned or narrower than %d channels (ldc %d)", who, cd, e->ldc_hi);
  }
  return PS_OK;
}

int set_extents(IgemmArgs& a, long long src_bytes, long long wgt_bytes, int es) {
  // 32-bit buffer offsets; bit 31 marks padding rows
  PS_REQUIRE(src_bytes < (1LL << 31) && wgt_bytes < (1LL << 31), "conv: tensor larger than 2 GiB (%lld / %lld bytes)", src_bytes, wgt_bytes);
  a.src_bytes = (unsigned)src_bytes;
  a.wgt_bytes = (unsigned)wgt_bytes;
  {  // the epilogue addresses its tensors through buffer descriptors of M rows with 32-bit byte offsets
    const ps_epilogue& e = a.epi;
    long long ld = 0;
    if (e.add0 && e.ldc_add0 > ld) ld = e.ldc_add0;
    if (e.out_raw && e.ldc_raw > ld) ld = e.ldc_raw;
    if (e.mask_src && e.ldc_mask > ld) ld = e.ldc_mask;
    if (e.add1 && e.ldc_add1 > ld) ld = e.ldc_add1;
    if (e.out && e.ldc_out > ld) ld = e.ldc_out;
    PS_REQUIRE((long long)a.M * ld * es < (1LL << 31), "conv: epilogue tensor larger than 2 GiB (%lld rows x %lld channels)", (long long)a.M, ld);
  }
  a.ablate = g_ablate;
  a.epi_M = g_ablate == 3 ? 0 : a.M;
  a.supertile = g_supertile;
  return PS_OK;
}

// Which wave-specialised kernel serves a (produced pixels M, produced channels Cd) problem; shared by the dispatcher and
// ps_conv_variant().  Returns PS_CONV_WS2_256 / _224 / PS_CONV_WS_128 / _112, or 0 if neither applies.
//   * large tile (one block per CU): the busiest CU's pixel rows decide between 224- and 256-pixel tiles; 256 wins ties
//     (fewer passes over the weights) -- measured r01: 1x1 2048->4096 prefers 256 at equal quantisation;
//   * it beats the two-blocks-per-CU kernel whenever there are >= 2 cout tiles (r01: +5..+12 % on the 28x28 layers, equal on
//     256-channel 56x56); single-cout-tile layers (128 channels @112x112, 18 K-steps) keep the small tile, whose second
//     resident block hides the epilogue.
// Column-block width of the halo kernel's 8-row tiles for a feature map of width w: 28 (224-pixel tiles; the 28 / 56 / 112 / 224-wide
// maps of 224 x 224 inputs) or 32 (256-pixel tiles; the 32 / 64 / 128 / 256-wide maps of 256 x 256 inputs, stages 2 and 4 of the
// reference: infer_pseudo_masks.py:50, infer_revise_masks.py:46); 0 if neither divides w.
// CUs the persistent kernels of a launch may count on (ps_conv_geom.cus_reserved; at least a quarter of the device)
static int usable_cus(int reserved) {
  const int n = ps_num_cus();
  return reserved <= 0 ? n : (n - reserved > n / 4 ? n - reserved : n / 4);
}

// Stream-K plan of a halo launch: T tiles of `klines` K-lines on nb CUs.  full = whole rounds that stay data-parallel, R = tiles of the partial
// round, cut into nb equal K-line shares.  Off (on = false) when there is no partial round, when the round is nearly full (nothing to win against
// the slab traffic), or when a tile would be spread over more than 8 blocks (tiny R: the half-tile tail launch serves those).
struct SkPlan {
  bool on;
  int full, R, lines, maxparts;
  long long ws_bytes;
};
constexpr long long kSkCounterBytes = 4096;
static SkPlan halo_sk_plan(long long T, int nb, int klines, int slab_bytes_per_part) {
  SkPlan p{false, 0, 0, 0, 0, 0};
  if (nb <= 0 || klines <= 0) return p;
  p.full = (int)(T / nb);
  p.R = (int)(T % nb);
  p.lines = p.R * klines;
  // lines < 4 nb: a share of fewer than four K-lines (36 K-steps) does not pay for the slab exchange -- measured per layer, profiles/r05c_convbench_stream_k.txt:
  // shares of 2 lines lose 2-6 %, of 4-6 lines win 0-1.5 %, of 8-12 lines win 3-7 %.  (And every block must own at least one line: the part numbering.)
  // full == 0 (fewer tiles than CUs: the whole launch would be parts): measured to LOSE 6-15 % against one tile per CU on T of the CUs at T / nb >= 0.66
  // (profiles/r05o_convbench_sk_small_batches.txt: every tile is cut, twice the slabs per CU) -- the static schedule stays
  if (p.full == 0 || p.R == 0 || p.R * 100LL > nb * 92LL || p.R > 1000 || p.lines < 4LL * nb) return p;
  for (int t = 0; t < p.R; ++t) {  // (the kernel's own formula for the blocks a tile is spread over)
    const long long x0 = (long long)t * klines, x1 = x0 + klines - 1;
    const int parts = (int)(((x1 + 1) * nb - 1) / p.lines) - (int)(((x0 + 1) * nb - 1) / p.lines) + 1;
    p.maxparts = parts > p.maxparts ? parts : p.maxparts;
  }
  if (p.maxparts > 8) return p;
  p.ws_bytes = kSkCounterBytes + (long long)p.R * p.maxparts * slab_bytes_per_part;
  if (p.ws_bytes >= (1LL << 31)) return p;
  p.on = true;
  return p;
}

// dynamic LDS of a halo launch: two windows + the weight ring + the ticket mailbox
static constexpr unsigned halo_lds_bytes(int tw, int nw, int wi = 4) {
  return 2u * (tw == 28 ? 9 : 10) * 4096u + (unsigned)nw * (wi == 4 ? 16384u : 8192u) + 16u;
}
static int halo_tile_width(int w) { return (w <= 0 || w > 256) ? 0 : (w % 28 == 0 ? 28 : (w % 32 == 0 ? 32 : 0)); }

static int pick_ws_variant(long long M, int Cd, int esize, bool halo_ok) {
  if (g_use_glds != 2 || Cd % 128 != 0) return 0;
  const long long n128 = Cd / 128;
  // 3x3 stride-1 layers whose 224-pixel tiles are whole feature-map rows: window + halo staging (less LDS fill traffic)
  // (from HALF a round of tiles on: one tile per CU on half the chip already beats the small-tile kernels' 0.22-0.24 of peak, and with the stream-K
  // workspace every CU gets a share of the K-lines -- run.sh's own batch size is 16, i.e. 224 tiles on the 512-channel layers: round 5)
  if (halo_ok && g_use_halo && (g_use_halo == 2 || (g_use_ws2 == 1 && esize == 2 && ((M + 223) / 224) * n128 >= ps_num_cus() / 2))) return PS_CONV_HALO;
  if (g_use_ws2 && (g_use_ws2 > 1 || (esize == 2 && n128 >= 2 && ((M + 255) / 256) * n128 >= 256))) {
    const long long t256 = (M + 255) / 256, t224 = (M + 223) / 224;
    const long long c256 = ((t256 * n128 + 255) / 256) * 256, c224 = ((t224 * n128 + 255) / 256) * 224;
    if (g_use_ws2 == 256) return PS_CONV_WS2_256;
    if (g_use_ws2 == 224) return PS_CONV_WS2_224;
    return c224 * 103 < c256 * 100 ? PS_CONV_WS2_224 : PS_CONV_WS2_256;
  }
  if (g_use_ws && (((M + 127) / 128) * n128 >= 512 || g_use_ws == 2)) {
    // pixel-tile height by the busiest CU's load (two blocks resident per CU)
    const long long t128 = (M + 127) / 128, t112 = (M + 111) / 112;
    const long long c128 = ((t128 * n128 + 255) / 256) * 128, c112 = ((t112 * n128 + 255) / 256) * 112;
    return (g_force_bm == 112 || (g_force_bm == 0 && c112 * 100 < c128 * 90)) ? PS_CONV_WS_112 : PS_CONV_WS_128;
  }
  return 0;
}

// The 256 x 256 tile kernel serves plain GEMMs: one tap, no gather arithmetic, 16-bit operands, whole 256-cout tiles, K >= 256.
// By shape (g_gemm256 == 1): K >= 2048, >= 1024 produced channels and >= 1.5 rounds of the device's CUs in tiles (384 on MI355X; first 512: at bs = 32 the
// 1024-channel layers have 392 tiles, and taking them is +1.0-1.2 % of a stage-3 step, profiles/r03_train_ab_gemm256_rule.txt).  Measured r03 on sustained single-kernel loops
// (profiles/r03_gemm256_vs_ws2.txt): +13..27 % on the >= 2048-channel layers, +-5 % on the K = 1024 / 1024-channel ones; on the WHOLE step
// (profiles/r03_train_ab_gemm256_rule.txt) the 1024-channel layers with K >= 2048 -- the data gradients 1024 <- 2048 / 2560 and the forward
// 2048 -> 1024 -- are worth another +0.8-1.0 % of a training step, +0.3 % of an inference pass; K = 1024 layers stay on ws2 (no change).
static bool use_gemm256(long long M, int Cd, int esize, int taps, int mul, int div_shift, int klines) {
  if (!g_gemm256 || g_use_glds != 2 || esize != 2 || taps != 1 || mul != 1 || div_shift != 0 || Cd % 256 != 0 || klines < 4) return false;
  if (g_gemm256 == 2) return true;
  const long long min_tiles = g_gemm256_min_tiles > 0 ? g_gemm256_min_tiles : 3LL * ps_num_cus() / 2;  // (a partitioned device has fewer CUs per round)
  return ((M + 255) / 256) * (Cd / 256) >= min_tiles && klines >= g_gemm256_min_klines && Cd >= g_gemm256_min_cd;
}

template <typename Tr>
int dispatch_bn(const IgemmArgs& a, hipStream_t s, bool allow_gemm256 = true) {
  if constexpr (sizeof(typename Tr::elem) == 2) {
    // (one tile per block, dispatched by the hardware: this kernel re-balances around CUs held by a communication kernel by itself, so
    // the tiles_per_block launch option of the persistent kernels does not apply to it)
    // (split types, round 5: the same kernel with a third MFMA group per quadrant; the rule counts LOGICAL K: a split K-line holds 32 channels)
    if (allow_gemm256 && use_gemm256(a.M, a.Cd, 2, a.taps, a.mul, a.div_shift, Tr::split ? a.klines / 2 : a.klines)) {
      IgemmArgs b = a;
      b.ntn = a.Cd / 256;
      b.ntm = (a.M + 255) / 256;
      // The partial last round: T = ntm x ntn tiles on nb CUs leave R = T mod nb tiles that would keep R CUs busy for a whole tile time
      // (1568 tiles of a 2048-channel layer at bs = 64: 6.125 rounds).  When R is at most half a round and a whole number of pixel
      // tiles, those pixel rows go to a second launch on the gathered-tile kernels (256 x 128 or smaller tiles: <= 2 R <= nb blocks of
      // about 0.6 tile times), with every tensor pointer advanced to the tail's first row.  Same MFMA chain per output element.
      const int nb = ps_num_cus();
      const long long T = (long long)b.ntm * b.ntn, R = T % nb;
      int tail_ptiles = 0;
      if (g_gemm256_tail && !a.shared && T > nb && R > 0 && 2 * R <= nb && R % b.ntn == 0 && a.epi_M == a.M) tail_ptiles = (int)(R / b.ntn);
      b.ntm -= tail_ptiles;
      hipLaunchKernelGGL((conv_gemm256_kernel<Tr>), dim3((unsigned)(b.ntm * b.ntn)), dim3(512), 2 * (256 * 128 + 256 * 128), s, b);
      PS_CHECK_LAUNCH("conv_gemm256");
      if (tail_ptiles > 0) {
        IgemmArgs c = a;
        const int es = (int)sizeof(typename Tr::elem);
        const long long m_off = (long long)b.ntm * 256;
        c.m_off = (int)m_off;
        c.src = a.src + m_off * a.pix_bytes;
        c.src_bytes = (unsigned)(a.src_bytes - m_off * a.pix_bytes);
        c.M = a.M - (int)m_off;
        c.epi_M = c.M;
        auto adv = [&](const void* ptr, int ldc) { return ptr ? static_cast<const void*>(static_cast<const unsigned char*>(ptr) + m_off * ldc * es) : nullptr; };
        c.epi.add0 = adv(a.epi.add0, a.epi.ldc_add0);
        c.epi.out_raw = const_cast<void*>(adv(a.epi.out_raw, a.epi.ldc_raw));
        c.epi.mask_src = adv(a.epi.mask_src, a.epi.ldc_mask);
        c.epi.add1 = adv(a.epi.add1, a.epi.ldc_add1);
        c.epi.out = const_cast<void*>(adv(a.epi.out, a.epi.ldc_out));
        c.epi.out_hi = const_cast<void*>(adv(a.epi.out_hi, a.epi.ldc_hi));  // (split types: the plain 16-bit copy of `out`'s hi halves has rows of its own)
        return dispatch_bn<Tr>(c, s, false);
      }
      return PS_OK;
    }
  }
#ifdef PS_DEBUG_HOOKS  // experimental kernels (measured slower, kept for the variant sweeps of the parity suite)
  // big problems: 256 x 128 tile, 3-stage LDS-DMA ring (1 block of 8 waves per CU)
  if (g_use_3stage && g_use_glds == 2 && a.Cd % 128 == 0 && (long long)((a.M + 255) / 256) * (a.Cd / 128) >= 256) {
    IgemmArgs b = a;
    b.ntn = a.Cd / 128;
    b.ntm = (a.M + 255) / 256;
    const int grid = b.ntm * b.ntn;
    hipLaunchKernelGGL((conv_igemm3_kernel<Tr>), dim3(grid), dim3(512), 3 * (256 * 128 + 128 * 128), s, b);
    PS_CHECK_LAUNCH("conv_igemm3");
    return PS_OK;
  }
  // big problems: ping-pong kernel, one 8-wave block per CU; pick the group height by the busiest CU's load
  if (g_use_pp && g_use_glds == 2 && a.Cd % 128 == 0) {
    const long long n128 = a.Cd / 128;
    const long long b256 = ((a.M + 255) / 256) * n128, b224 = ((a.M + 223) / 224) * n128;
    if (b256 >= 512 || g_use_pp == 2) {
      const long long c256 = ((b256 + 255) / 256) * 256, c224 = ((b224 + 255) / 256) * 224;
      IgemmArgs b = a;
      b.ntn = (int)n128;
      b.ntm = (int)(((g_force_bm == 0 && c224 < c256) || g_force_bm == 112) ? (a.M + 223) / 224 : (a.M + 255) / 256);
      if ((g_force_bm == 0 && c224 < c256) || g_force_bm == 112) {
        hipLaunchKernelGGL((conv_igemm_pp_kernel<Tr, 112>), dim3((unsigned)b224), dim3(512), 3 * (224 * 128 + 128 * 128) + 1024, s, b);
      } else {
        hipLaunchKernelGGL((conv_igemm_pp_kernel<Tr, 128>), dim3((unsigned)b256), dim3(512), 3 * (256 * 128 + 128 * 128) + 1024, s, b);
      }
      PS_CHECK_LAUNCH("conv_igemm_pp");
      return PS_OK;
    }
  }
#endif
  const int adil = a.dstep < 0 ? -a.dstep : a.dstep;
  const bool halo_ok = a.taps == 9 && a.mul == 1 && a.div_shift == 0 && a.Hs == a.Ho && a.Ws == a.Wo && halo_tile_width(a.Ws) != 0 && adil <= 4;
  if (const int v = pick_ws_variant(a.M, a.Cd, (int)sizeof(typename Tr::elem), halo_ok)) {
    IgemmArgs b = a;
    b.ntn = a.Cd / 128;
    // (f32 problems reach the halo / large-tile kernels only through the debug switches: pick_ws_variant)
    constexpr bool kLargeTiles = kDebugBuild || sizeof(typename Tr::elem) == 2;
    if constexpr (kLargeTiles) {
    if (v == PS_CONV_HALO) {
      const int tw = halo_tile_width(a.Ws);
      b.ntm = (a.M / a.Ws + 7) / 8 * (a.Ws / tw);  // blocks of 8 global rows x column blocks of tw
      b.nb = usable_cus(a.reserved);
      b.tpb = a.tpb;
      b.stagger = g_halo_stagger;
      b.stagger_phases = g_halo_stagger_phases > 0 ? g_halo_stagger_phases : 1;
      // The partial last round.  With T = ntm x ntn tiles on nb CUs, R = T mod nb <= nb / 2 tiles would keep R CUs busy for a whole round
      // (512-channel layers at bs=64: 896 tiles = 3.5 rounds).  Those R tiles -- whole pixel tiles, R % ntn == 0 -- go to a second launch
      // as 2 R half tiles of 64 couts (half the MFMAs, 8 instead of 16 KiB of weights per K-step: ~0.6 of a round on every CU).
      // Not while another stream of the process shares the GPU (gpu_shared: the two-stream backward): its blocks take the idle CUs, and the
      // split then COSTS 2 % of a training step (profiles/r03_tail_split_two_streams.txt).
      // Stream-K finish of the partial round (ps_epilogue.sk_ws given): one launch, every CU busy to the end.
      if constexpr (sizeof(typename Tr::elem) == 2) {
        // (224-pixel tiles only: the 256-pixel instance -- 8 pixel fragments per wave, 128 accumulator registers -- did not fit its slab hand-off into 256
        // registers (174 spilled) and measured -10 % on the 512-channel layers, +-2 % on the others: profiles/r05v_convbench_tw32_stream_k.txt)
        if (g_halo_sk && tw == 28 && (g_halo_sk == 2 || !a.shared) && a.tpb == 0 && !a.use_queue && a.epi.sk_ws) {
          const int slab = 7 * 4 * 1024 * 4;  // MI x WI fragments x 1 KiB x 4 consumer waves
          const SkPlan p = halo_sk_plan((long long)b.ntm * b.ntn, b.nb, a.klines, slab);
          if (p.on && a.epi.sk_ws_bytes >= p.ws_bytes && (reinterpret_cast<uintptr_t>(a.epi.sk_ws) & 255u) == 0) {
            b.sk_dp = p.full;
            b.sk_tile0 = p.full * b.nb;
            b.sk_lines = p.lines;
            b.sk_maxparts = p.maxparts;
            b.sk_cnt = static_cast<unsigned*>(a.epi.sk_ws);
            b.sk_slabs = reinterpret_cast<float*>(static_cast<unsigned char*>(a.epi.sk_ws) + kSkCounterBytes);
            b.sk_slab_bytes = (unsigned)(p.ws_bytes - kSkCounterBytes);
            const dim3 sgrid((unsigned)b.nb);
            hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 3, 4, false, true>), sgrid, dim3(512), halo_lds_bytes(28, 3), s, b);
            PS_CHECK_LAUNCH("conv_igemm_halo<stream-K>");
            return PS_OK;
          }
        }
      }
      int tail_ptiles = 0;
      if constexpr (sizeof(typename Tr::elem) == 2) {
        const long long T = (long long)b.ntm * b.ntn, R = T % b.nb;
        if (g_halo_tail && a.tpb == 0 && !a.shared && T > b.nb && R > 0 && 2 * R <= b.nb && R % b.ntn == 0 && a.Cd % 64 == 0) tail_ptiles = (int)(R / b.ntn);
      }
      b.ntm -= tail_ptiles;
      // tile_queue: only where there is something to hand out (more tiles than blocks) and a tile is long enough to draw one ahead
      if constexpr (sizeof(typename Tr::elem) == 2) {
        if (a.use_queue && (long long)b.ntm * b.ntn > b.nb && b.klines >= 2) {
          b.queue = ps_queue_slot(s);
          PS_REQUIRE(b.queue != nullptr, "conv: no ticket counters (hipMalloc failed)");
          const dim3 qgrid((unsigned)b.nb);
          if (tw == 28) hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 3, 4, true>), qgrid, dim3(512), halo_lds_bytes(28, 3), s, b);
          else hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 32, 3, 4, true>), qgrid, dim3(512), halo_lds_bytes(32, 3), s, b);
          PS_CHECK_LAUNCH("conv_igemm_halo<queue>");
          goto halo_tail;
        }
      }
      {
      const dim3 hgrid(ps_persistent_grid((long long)b.ntm * b.ntn, b.nb, b.tpb));
      if (tw == 28) {
#ifdef PS_DEBUG_HOOKS
        if (g_halo_ring == 5) hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 5>), hgrid, dim3(512), halo_lds_bytes(28, 5), s, b);
        else if (g_halo_ring == 4) hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 4>), hgrid, dim3(512), halo_lds_bytes(28, 4), s, b);
        else
#endif
        hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 3>), hgrid, dim3(512), halo_lds_bytes(28, 3), s, b);
      } else {
#ifdef PS_DEBUG_HOOKS
        if (g_halo_ring >= 4) hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 32, 4>), hgrid, dim3(512), halo_lds_bytes(32, 4), s, b);
        else
#endif
        hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 32, 3>), hgrid, dim3(512), halo_lds_bytes(32, 3), s, b);
      }
      PS_CHECK_LAUNCH("conv_igemm_halo");
      }
    halo_tail:
      if constexpr (sizeof(typename Tr::elem) == 2) {
        if (tail_ptiles > 0) {
          IgemmArgs c = b;
          c.tm0 = b.ntm;
          c.ntm = tail_ptiles;
          c.ntn = a.Cd / 64;
          const dim3 tgrid(ps_persistent_grid((long long)c.ntm * c.ntn, c.nb, 0));
          if (tw == 28) hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 28, 3, 2>), tgrid, dim3(512), halo_lds_bytes(28, 3, 2), s, c);
          else hipLaunchKernelGGL((conv_igemm_halo_kernel<Tr, 32, 3, 2>), tgrid, dim3(512), halo_lds_bytes(32, 3, 2), s, c);
          PS_CHECK_LAUNCH("conv_igemm_halo<tail>");
        }
      }
      return PS_OK;
    }
    }
    const int bm = v == PS_CONV_WS2_256 ? 256 : v == PS_CONV_WS2_224 ? 224 : v == PS_CONV_WS_128 ? 128 : 112;
    b.ntm = (a.M + bm - 1) / bm;
    const dim3 grid((unsigned)(b.ntm * b.ntn));
    b.nb = usable_cus(a.reserved);
    b.tpb = a.tpb;
    const dim3 pgrid(ps_persistent_grid((long long)b.ntm * b.ntn, b.nb, b.tpb));  // persistent: one block per CU (per batch)
    if constexpr (kLargeTiles) {
      // tile_queue: where there is something to hand out and a tile has at least THREE K-steps (the loaders stage two K-steps ahead: with
      // two-step tiles they cross into tile s + 2 one barrier before the consumers publish it at the end of tile s)
      if constexpr (sizeof(typename Tr::elem) == 2) {
        if (a.use_queue && (v == PS_CONV_WS2_256 || v == PS_CONV_WS2_224) && (long long)b.ntm * b.ntn > b.nb && a.taps * a.klines >= 3) {
          b.queue = ps_queue_slot(s);
          PS_REQUIRE(b.queue != nullptr, "conv: no ticket counters (hipMalloc failed)");
          const dim3 qgrid((unsigned)b.nb);
          if (v == PS_CONV_WS2_256) hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 256, false, true>), qgrid, dim3(512), 3 * (256 * 128 + 128 * 128) + 16, s, b);
          else hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 224, false, true>), qgrid, dim3(512), 3 * (224 * 128 + 128 * 128) + 16, s, b);
          PS_CHECK_LAUNCH("conv_igemm_ws2<queue>");
          return PS_OK;
        }
      }
      if (v == PS_CONV_WS2_256) {
        hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 256>), pgrid, dim3(512), 3 * (256 * 128 + 128 * 128), s, b);
        PS_CHECK_LAUNCH("conv_igemm_ws2");
        return PS_OK;
      }
      if (v == PS_CONV_WS2_224) {
        hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 224>), pgrid, dim3(512), 3 * (224 * 128 + 128 * 128), s, b);
        PS_CHECK_LAUNCH("conv_igemm_ws2");
        return PS_OK;
      }
    }
    if (v == PS_CONV_WS_112) hipLaunchKernelGGL((conv_igemm_ws_kernel<Tr, 112>), grid, dim3(512), 2 * (112 * 128 + 128 * 128) + 1024, s, b);
    else hipLaunchKernelGGL((conv_igemm_ws_kernel<Tr, 128>), grid, dim3(512), 2 * (128 * 128 + 128 * 128) + 1024, s, b);
    PS_CHECK_LAUNCH("conv_igemm_ws");
    return PS_OK;
  }
  // Tile choice.  Two blocks are resident per CU: the launch takes as long as the busiest CU needs for its
  // ceil(blocks / 256) blocks, so compare pixel-tile heights by (blocks on the busiest CU) x (tile height).
  const long long t128 = (a.M + 127) / 128, t112 = (a.M + 111) / 112;
  if (a.Cd % 128 == 0 && g_force_bn != 64) {
    const long long n128 = a.Cd / 128;
    const long long c128 = ((t128 * n128 + 255) / 256) * 128, c112 = ((t112 * n128 + 255) / 256) * 112;
    const bool big = t128 * n128 >= 512 || g_force_bn == 128;
    if (big && (g_force_bm == 112 || (g_force_bm == 0 && c112 * 100 < c128 * 90))) return launch_igemm<Tr, 112, 128, 1, 4>(a, s);
#ifdef PS_DEBUG_HOOKS
    if (big && g_force_bm == 1288) return launch_igemm<Tr, 128, 128, 2, 4>(a, s);  // experiment: 8 waves of 64x32
    if (big && g_force_bm == 1289) return launch_igemm<Tr, 128, 128, 4, 2>(a, s);  // experiment: 8 waves of 32x64
#endif
    if (big) return launch_igemm<Tr, 128, 128, 2, 2>(a, s);
  }
  return launch_igemm<Tr, 128, 64, 4, 1>(a, s);
}

}  // namespace

// Ticket counters of the queue-mode launches: per (device, stream) a ring of counter blocks (nine counters, one 128-byte line each), zeroed once (on that stream); every launch takes the next
// block of its stream's ring.  A launch leaves its block zeroed (ps_q_block_done) and launches of one stream run in order, so a block is
// never shared by two launches in flight.
namespace {
struct QueueRing { unsigned* base; unsigned next; };
std::mutex g_ring_mu;
std::map<std::pair<int, hipStream_t>, QueueRing> g_rings;
}  // namespace

unsigned* ps_queue_slot(hipStream_t stream) {
  constexpr int kSlots = 256, kDwords = PS_Q_SLOT_DWORDS;
  // keyed by the STREAM's device (not the calling thread's current one: a stream of device 1 used while device 0 is current keeps its own ring)
  int dev = 0;
  hipDevice_t sdev;
  if (stream != nullptr && hipStreamGetDevice(stream, &sdev) == hipSuccess) dev = (int)sdev;
  else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_ring_mu);
  auto it = g_rings.find({dev, stream});
  if (it == g_rings.end()) {
    // first queue-mode launch on this stream: one 288 KiB allocation (hipMalloc synchronises the device once, and fails under stream capture:
    // call ps_queue_prepare(stream) beforehand where that matters)
    unsigned* d = nullptr;
    if (hipMalloc(&d, kSlots * kDwords * sizeof(unsigned)) != hipSuccess) return nullptr;
    // zeroed ON THE LAUNCH STREAM: a null-stream hipMemset is not ordered with kernels on a non-blocking stream (torch's side streams), and
    // recycled device memory is not zero -- the first launches of a new stream would draw from garbage counters
    if (hipMemsetAsync(d, 0, kSlots * kDwords * sizeof(unsigned), stream) != hipSuccess) { (void)hipFree(d); return nullptr; }
    it = g_rings.emplace(std::make_pair(dev, stream), QueueRing{d, 0u}).first;
  }
  QueueRing& r = it->second;
  unsigned* p = r.base + (size_t)(r.next % kSlots) * kDwords;
  ++r.next;
  return p;
}

// Allocate (and zero) the ticket-counter ring of `stream` now, outside any launch path.
extern "C" int ps_queue_prepare(void* stream) {
  PS_REQUIRE(ps_queue_slot(static_cast<hipStream_t>(stream)) != nullptr, "queue_prepare: hipMalloc / hipMemsetAsync failed");
  return PS_OK;
}

// Free every ticket-counter ring of this process (a destroyed stream's ring would otherwise be inherited by a new stream that re-uses its handle,
// and a queue-mode kernel that was aborted leaves its counters non-zero): the caller guarantees that no queue-mode launch is in flight.
extern "C" int ps_queue_release(void) {
  std::lock_guard<std::mutex> lock(g_ring_mu);
  for (auto& kv : g_rings) (void)hipFree(kv.second.base);
  g_rings.clear();
  return PS_OK;
}


#ifdef PS_DEBUG_HOOKS
extern "C" void ps_debug_set_glds(int on) { g_use_glds = on; }  // 0 | 1 | 2
extern "C" void ps_debug_set_3stage(int on) { g_use_3stage = on; }
extern "C" void ps_debug_set_bn(int bn) { g_force_bn = bn; }
extern "C" void ps_debug_set_bm(int bm) { g_force_bm = bm; }
extern "C" void ps_debug_set_ablate(int v) { g_ablate = v; }
extern "C" void ps_debug_set_pp(int v) { g_use_pp = v; }
extern "C" void ps_debug_set_ws(int v) { g_use_ws = v; }
extern "C" void ps_debug_set_ws2(int v) { g_use_ws2 = v; }
extern "C" void ps_debug_set_halo(int v) { g_use_halo = v; }
extern "C" void ps_debug_set_gemm256(int v) { g_gemm256 = v; }
extern "C" void ps_debug_set_gemm256_tail(int v) { g_gemm256_tail = v; }
extern "C" void ps_debug_set_gemm256_rule(int code) { g_gemm256_min_klines = code / 10000; g_gemm256_min_cd = code % 10000; }
extern "C" void ps_debug_set_gemm256_min_tiles(int v) { g_gemm256_min_tiles = v; }
#ifdef PS_HALO_STAMPS
extern "C" int ps_debug_read_stamps(unsigned long long* host_out) {  // 256 x 4 x 8 values; synchronises the device
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_halo_stamps), sizeof(g_halo_stamps)) == hipSuccess ? 0 : -2;
}
#endif
extern "C" void ps_debug_set_halo_ring(int v) { g_halo_ring = v; }
extern "C" void ps_debug_set_halo_tail(int v) { g_halo_tail = v; }
extern "C" void ps_debug_set_halo_sk(int v) { g_halo_sk = v; }
extern "C" void ps_debug_set_halo_stagger(int v) { g_halo_stagger = v & 0xff; g_halo_stagger_phases = (v >> 8) ? (v >> 8) : 4; }
extern "C" void ps_debug_set_supertile(int v) { g_supertile = v; }
// every tunable of this translation unit back to its default (one list, next to the definitions: the tools call ps_debug_reset())
void ps_debug_reset_igemm(void) {
  g_use_glds = 2; g_use_pp = 0; g_use_ws = 1; g_use_ws2 = 1; g_use_halo = 1; g_halo_sk = PS_HALO_SK; g_halo_stagger = 0; g_halo_stagger_phases = 4; g_halo_tail = 1; g_halo_ring = 3; g_gemm256 = 1;
  g_gemm256_min_klines = 32; g_gemm256_min_cd = 1024; g_gemm256_min_tiles = 0; g_gemm256_tail = 1; g_use_3stage = 0; g_ablate = 0;
  g_supertile = 4; g_force_bm = 0; g_force_bn = 0; g_s2split = 1;
}
#endif

extern "C" int ps_conv_supported(const ps_conv_geom* g) { return check_geom(g) == PS_OK ? 1 : 0; }

extern "C" int ps_conv_variant(const ps_conv_geom* g, int32_t dgrad) {
  if (check_geom(g) != PS_OK) return -1;
  const long long ho = (g->h - 1) / g->stride + 1, wo = (g->w - 1) / g->stride + 1;
  const long long M = dgrad ? (long long)g->n * g->h * g->w : (long long)g->n * ho * wo;
  if (g_use_3stage + g_use_pp != 0) return PS_CONV_OTHER;
  if (use_gemm256(M, dgrad ? g->cin : g->cout, ps_esize(g->dtype), g->ksize * g->ksize, dgrad ? 1 : g->stride, dgrad && g->stride == 2 ? 1 : 0,
                  (dgrad ? g->cout : g->cin) * ps_esize(g->dtype) / 128))  // (split types: the rule counts logical K -- as the dispatcher)
    return PS_CONV_GEMM256;
  // both directions of a stride-1 3x3 layer gather on the input grid h x w
  const bool halo_ok = g->ksize == 3 && g->stride == 1 && halo_tile_width(g->w) != 0 && g->dilation <= 4;
  const int v = pick_ws_variant(M, dgrad ? g->cin : g->cout, ps_esize(g->dtype), halo_ok);
  return v ? v : PS_CONV_4WAVE;
}

extern "C" int64_t ps_conv_sk_workspace_bytes(const ps_conv_geom* g, int32_t dgrad) {
  if (!g || check_geom(g) != PS_OK || !g_halo_sk || ps_esize(g->dtype) != 2 || g->tiles_per_block != 0 || g->tile_queue != 0) return 0;
  if (g_halo_sk != 2 && g->gpu_shared) return 0;
  if (ps_conv_variant(g, dgrad) != PS_CONV_HALO) return 0;
  const int tw = halo_tile_width(g->w);
  if (tw != 28) return 0;  // (224-pixel tiles only: see the dispatcher)
  const long long M = (long long)g->n * g->h * g->w;  // stride 1: both directions produce on the input grid
  const long long ntm = (M / g->w + 7) / 8 * (g->w / tw), ntn = (dgrad ? g->cin : g->cout) / 128;
  const int klines = ps_planes(g->dtype) * (dgrad ? g->cout : g->cin) * 2 / 128;
  const SkPlan p = halo_sk_plan(ntm * ntn, usable_cus(g->cus_reserved), klines, 7 * 4 * 1024 * 4);
  return p.on ? p.ws_bytes : 0;
}

// ---- 1x1 conv + BN + ReLU + narrow head in one launch (inference) ------------------------------------------------------------------
static bool head_geom_ok(const ps_conv_geom* g, int classes) {
  if (check_geom(g) != PS_OK || classes < 1 || classes > 8) return false;
  const int es = ps_esize(g->dtype);
  return es == 2 && g->ksize == 1 && g->stride == 1 && g->cout % 256 == 0 && (g->cin * es) / 128 >= 4;
}

extern "C" int64_t ps_conv1x1_head_workspace_floats(const ps_conv_geom* g, int32_t classes) {
  if (!g || !head_geom_ok(g, classes)) return 0;
  return (int64_t)(g->cout / 64) * g->n * g->h * g->w * classes;
}

extern "C" int ps_conv1x1_head_fwd(const ps_conv_geom* g, const void* x, const void* w_fwd, const float* scale, const float* shift, const float* w_head,
                                   int32_t classes, float* workspace, int64_t workspace_floats, float* cam, void* stream) {
  PS_REQUIRE(g && x && w_fwd && w_head && workspace && cam, "conv1x1_head_fwd: null argument");
  PS_REQUIRE(head_geom_ok(g, classes), "conv1x1_head_fwd: geometry not served (16-bit 1x1 stride-1, cout %% 256 == 0, cin >= 256, 1..8 classes): %s",
             ps_last_error());
  PS_REQUIRE(ps_aligned16(x) && ps_aligned16(w_fwd) && ps_aligned16(w_head) && ps_aligned16(workspace), "conv1x1_head_fwd: misaligned pointer");
  const int64_t need = ps_conv1x1_head_workspace_floats(g, classes);
  PS_REQUIRE(workspace_floats >= need, "conv1x1_head_fwd: workspace of %lld floats, need %lld", (long long)workspace_floats, (long long)need);
  const int es = 2;
  IgemmArgs a{};
  a.src = static_cast<const unsigned char*>(x);
  a.wgt = static_cast<const unsigned char*>(w_fwd);
  a.Hs = a.Ho = g->h; a.Ws = a.Wo = g->w;
  a.M = g->n * g->h * g->w;
  a.mul = 1; a.dstep = 1; a.div_shift = 0;
  a.taps = 1; a.ctr = 0;
  a.klines = g->cin * es / 128;
  a.pix_bytes = (long long)g->ldc_x * es;
  a.wrow_bytes = (long long)g->cin * es;
  a.Cd = g->cout;
  a.epi = ps_epilogue{};
  a.epi.mode = PS_EPI_BNRELU;
  a.epi.scale = scale;
  a.epi.shift = shift;
  a.head_w = w_head;
  a.head_part = workspace;
  a.head_c = classes;
  const long long src_bytes = (long long)a.M * a.pix_bytes, wgt_bytes = (long long)g->cout * a.wrow_bytes;
  PS_REQUIRE(src_bytes < (1LL << 31) && wgt_bytes < (1LL << 31), "conv1x1_head_fwd: tensor larger than 2 GiB");
  a.src_bytes = (unsigned)src_bytes;
  a.wgt_bytes = (unsigned)wgt_bytes;
  a.epi_M = a.M;
  a.supertile = g_supertile;
  a.ntn = g->cout / 256;
  a.ntm = (a.M + 255) / 256;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(a.ntm * a.ntn));
  if (g->dtype == PS_BF16) hipLaunchKernelGGL((conv_gemm256_kernel<TraitsBF16, true>), grid, dim3(512), 2 * (256 * 128 + 256 * 128), s, a);
  else hipLaunchKernelGGL((conv_gemm256_kernel<TraitsF16, true>), grid, dim3(512), 2 * (256 * 128 + 256 * 128), s, a);
  PS_CHECK_LAUNCH("conv_gemm256<head>");
  const long long nel = (long long)a.M * classes;
  hipLaunchKernelGGL(head_reduce_kernel, dim3((unsigned)std::min<long long>((nel + 255) / 256, 4096)), dim3(256), 0, s, workspace, cam, nel, g->cout / 64);
  PS_CHECK_LAUNCH("head_reduce");
  return PS_OK;
}

extern "C" int ps_conv2d_fwd(const ps_conv_geom* g, const void* x, const void* w_fwd, const ps_epilogue* epi, void* stream) {
  if (int rc = check_geom(g)) return rc;
  if (int rc = check_epilogue(epi, g->dtype, "conv2d_fwd", g->cout)) return rc;
  PS_REQUIRE(x && w_fwd && ps_aligned16(x) && ps_aligned16(w_fwd), "conv2d_fwd: null or misaligned x/w");
  const int es = ps_esize(g->dtype);
  IgemmArgs a{};
  a.src = static_cast<const unsigned char*>(x);
  a.wgt = static_cast<const unsigned char*>(w_fwd);
  a.Hs = g->h; a.Ws = g->w;
  a.Ho = (g->h - 1) / g->stride + 1; a.Wo = (g->w - 1) / g->stride + 1;
  a.M = g->n * a.Ho * a.Wo;
  a.mul = g->stride; a.dstep = g->dilation; a.div_shift = 0;
  a.taps = g->ksize * g->ksize; a.ctr = g->ksize / 2;
  const int kc = ps_planes(g->dtype) * g->cin;  // stored channels per tap (split formats: 32-channel blocks [hi | lo], each one 128-byte K-line)
  a.klines = kc * es / 128;
  a.pix_bytes = (long long)g->ldc_x * es;
  a.wrow_bytes = (long long)a.taps * kc * es;
  a.Cd = g->cout;
  a.epi = *epi;
  a.tpb = g->tiles_per_block;
  a.shared = g->gpu_shared;
  a.reserved = g->cus_reserved;
  a.use_queue = g->tile_queue;
  a.queue = nullptr;
  if (int rc = set_extents(a, (long long)g->n * g->h * g->w * a.pix_bytes, (long long)g->cout * a.wrow_bytes, es)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g->dtype == PS_BF16X3) return dispatch_bn<TraitsBF16X3>(a, s);
  if (g->dtype == PS_F16X3) return dispatch_bn<TraitsF16X3>(a, s);
  if (g->dtype == PS_BF16) return dispatch_bn<TraitsBF16>(a, s);
  if (g->dtype == PS_F16) return dispatch_bn<TraitsF16>(a, s);
  return dispatch_bn<TraitsF32>(a, s);
}

#ifdef PS_DEBUG_HOOKS
extern "C" void ps_debug_set_s2split(int v) { g_s2split = v; }
#endif

// Stride-2 3x3 data gradient: an output pixel's parity (p & 1, q & 1) decides which taps meet dy at integer positions -- 1, 2, 2 or 4
// of the 9.  Gathering all 9 per pixel (and multiplying the zero rows) does 4x the work; one launch per parity class stages and
// multiplies only its live taps and writes its pixels of the full gradient through the epilogue's sub-grid row map.
template <typename Tr>
static int dgrad_s2_split(const IgemmArgs& a0, hipStream_t s) {
  for (int cy = 0; cy < 2; ++cy)
    for (int cx = 0; cx < 2; ++cx) {
      IgemmArgs b = a0;
      b.Hf = a0.Ho; b.Wf = a0.Wo;
      b.Ho = (a0.Ho - cy + 1) / 2; b.Wo = (a0.Wo - cx + 1) / 2;
      const int nimg = a0.M / (a0.Ho * a0.Wo);
      b.M = nimg * b.Ho * b.Wo;
      if (b.M == 0) continue;
      b.oy = cy; b.ox = cx;
      b.tap_mask = 0;
      for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx)
          if (((cy + (ty - 1) * a0.dstep) & 1) == 0 && ((cx + (tx - 1) * a0.dstep) & 1) == 0) b.tap_mask |= 1 << (ty * 3 + tx);
      b.ntn = a0.Cd / 128;
      const long long t256 = (b.M + 255) / 256, t224 = (b.M + 223) / 224;
      const long long c256 = ((t256 * b.ntn + 255) / 256) * 256, c224 = ((t224 * b.ntn + 255) / 256) * 224;
      const bool use224 = c224 * 103 < c256 * 100;
      b.ntm = (int)(use224 ? t224 : t256);
      b.nb = usable_cus(a0.reserved);
      b.tpb = a0.tpb;
      const dim3 pgrid(ps_persistent_grid((long long)b.ntm * b.ntn, b.nb, b.tpb));
      if (use224) hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 224, true>), pgrid, dim3(512), 3 * (224 * 128 + 128 * 128), s, b);
      else hipLaunchKernelGGL((conv_igemm_ws2_kernel<Tr, 256, true>), pgrid, dim3(512), 3 * (256 * 128 + 128 * 128), s, b);
      PS_CHECK_LAUNCH("conv_igemm_ws2<split>");
    }
  return PS_OK;
}

static bool dgrad_s2_split_ok(const ps_conv_geom* g, const ps_epilogue* epi) {
  if (!g_s2split || g->stride != 2 || g->ksize != 3 || g->dilation != 1 || ps_esize(g->dtype) != 2 || epi->drop) return false;
  if (g_use_glds != 2 || !g_use_ws2 || g->cin % 128 != 0) return false;
  return g_s2split == 2 || ((long long)g->n * g->h * g->w / 4 / 224) * (g->cin / 128) >= 256;  // >= one round per class
}

extern "C" int ps_conv2d_dgrad(const ps_conv_geom* g, const void* dy, const void* w_dgrad, const ps_epilogue* epi, void* stream) {
  if (int rc = check_geom(g)) return rc;
  if (int rc = check_epilogue(epi, g->dtype, "conv2d_dgrad", g->cin)) return rc;
  PS_REQUIRE(dy && w_dgrad && ps_aligned16(dy) && ps_aligned16(w_dgrad), "conv2d_dgrad: null or misaligned dy/w");
  const int es = ps_esize(g->dtype);
  IgemmArgs a{};
  a.src = static_cast<const unsigned char*>(dy);
  a.wgt = static_cast<const unsigned char*>(w_dgrad);
  a.Hs = (g->h - 1) / g->stride + 1; a.Ws = (g->w - 1) / g->stride + 1;  // dy's grid
  a.Ho = g->h; a.Wo = g->w;                                              // produces dx on x's grid
  a.M = g->n * g->h * g->w;
  a.mul = 1; a.dstep = -g->dilation; a.div_shift = g->stride == 2 ? 1 : 0;
  a.taps = g->ksize * g->ksize; a.ctr = g->ksize / 2;
  const int kc = ps_planes(g->dtype) * g->cout;
  a.klines = kc * es / 128;
  a.pix_bytes = (long long)g->ldc_y * es;
  a.wrow_bytes = (long long)a.taps * kc * es;
  a.Cd = g->cin;
  a.epi = *epi;
  a.tpb = g->tiles_per_block;
  a.shared = g->gpu_shared;
  a.reserved = g->cus_reserved;
  a.use_queue = g->tile_queue;
  a.queue = nullptr;
  if (int rc = set_extents(a, (long long)g->n * a.Hs * a.Ws * a.pix_bytes, (long long)g->cin * a.wrow_bytes, es)) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dgrad_s2_split_ok(g, epi))
    return g->dtype == PS_BF16X3 ? dgrad_s2_split<TraitsBF16X3>(a, s) : g->dtype == PS_F16X3 ? dgrad_s2_split<TraitsF16X3>(a, s)
           : g->dtype == PS_BF16 ? dgrad_s2_split<TraitsBF16>(a, s) : dgrad_s2_split<TraitsF16>(a, s);
  if (g->dtype == PS_BF16X3) return dispatch_bn<TraitsBF16X3>(a, s);
  if (g->dtype == PS_F16X3) return dispatch_bn<TraitsF16X3>(a, s);
  if (g->dtype == PS_BF16) return dispatch_bn<TraitsBF16>(a, s);
  if (g->dtype == PS_F16) return dispatch_bn<TraitsF16>(a, s);
  return dispatch_bn<TraitsF32>(a, s);
}

// ---- (debug library) conv1a + BN + ReLU + the first ResBlock's two stride-2 convolutions in one launch (conv_front_s2_kernel)
#ifdef PS_DEBUG_HOOKS
extern "C" int ps_debug_conv_front_s2_supported(int32_t dtype, int32_t n, int32_t h, int32_t w) {
  return (dtype == PS_BF16 || dtype == PS_F16) && n > 0 && h > 0 && h % 2 == 0 && (w == 224 || w == 256) && (long long)n * (h / 2) * (w / 2) < (1LL << 24) ? 1 : 0;
}

extern "C" int ps_debug_conv_front_s2(int32_t dtype, int32_t n, int32_t h, int32_t w, const float* image, const float* w1a, const float* scale0,
                                const float* shift0, const void* w_b1, const void* w_2a, void* out_b1, int32_t ldc_b1, const float* scale1,
                                const float* shift1, void* out_2a, int32_t ldc_2a, void* stream) {
  PS_REQUIRE(ps_debug_conv_front_s2_supported(dtype, n, h, w), "conv_front_s2: dtype %d, %d x %d x %d unsupported (bf16 / fp16; width 224 or 256; even height)", dtype, n, h, w);
  PS_REQUIRE(image && w1a && scale0 && shift0 && w_b1 && w_2a && out_b1 && out_2a && scale1 && shift1, "conv_front_s2: null pointer");
  PS_REQUIRE(ldc_b1 >= 128 && ldc_2a >= 128 && ldc_b1 % 8 == 0 && ldc_2a % 8 == 0, "conv_front_s2: channel strides");
  PS_REQUIRE(ps_aligned16(w_b1) && ps_aligned16(w_2a) && ps_aligned16(out_b1) && ps_aligned16(out_2a) && ps_aligned16(scale0) && ps_aligned16(shift0), "conv_front_s2: misaligned");
  FrontArgs f{};
  f.image = image; f.w1a = w1a; f.sc0 = scale0; f.sh0 = shift0;
  f.w_b1 = static_cast<const unsigned char*>(w_b1);
  f.w_2a = static_cast<const unsigned char*>(w_2a);
  f.N = n; f.H = h; f.W = w;
  const int ho = h / 2, wo = w / 2;
  for (IgemmArgs* e : {&f.eb1, &f.e2a}) {
    e->Ho = ho; e->Wo = wo; e->M = e->epi_M = n * ho * wo; e->Cd = 128; e->m_off = 0;
  }
  f.eb1.epi.mode = PS_EPI_NONE; f.eb1.epi.out_raw = out_b1; f.eb1.epi.ldc_raw = ldc_b1;
  f.e2a.epi.mode = PS_EPI_BNRELU; f.e2a.epi.scale = scale1; f.e2a.epi.shift = shift1; f.e2a.epi.out = out_2a; f.e2a.epi.ldc_out = ldc_2a;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(n * ho));
  const int nf = wo / 16;
  const size_t lds = (size_t)2 * (wo + 1) * 128 + (size_t)3 * 5 * (2 * wo + 4) * 2;
  if (dtype == PS_BF16) {
    if (nf == 7) hipLaunchKernelGGL((conv_front_s2_kernel<TraitsBF16, 7>), grid, dim3(256), lds, s, f);
    else hipLaunchKernelGGL((conv_front_s2_kernel<TraitsBF16, 8>), grid, dim3(256), lds, s, f);
  } else {
    if (nf == 7) hipLaunchKernelGGL((conv_front_s2_kernel<TraitsF16, 7>), grid, dim3(256), lds, s, f);
    else hipLaunchKernelGGL((conv_front_s2_kernel<TraitsF16, 8>), grid, dim3(256), lds, s, f);
  }
  PS_CHECK_LAUNCH("conv_front_s2");
  return PS_OK;
}
#endif  // PS_DEBUG_HOOKS
