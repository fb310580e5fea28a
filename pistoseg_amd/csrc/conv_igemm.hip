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

namespace {

__device__ __attribute__((aligned(256))) unsigned char g_zero_line[256];  // source of padding rows

struct IgemmArgs {
  const unsigned char* src;   // gathered activation (forward: x, dgrad: dy)
  const unsigned char* wgt;   // [Cd][taps][Cs] rows of contiguous K
  int Hs, Ws, Ho, Wo, M;      // source dims, produced grid, produced pixels
  int mul, dstep, div_shift;  // gather arithmetic (see file header)
  int taps, klines;           // 1|9, Cs*esize/128
  int ctr;                    // centre tap coordinate (0 for 1x1, 1 for 3x3)
  long long pix_bytes;        // source channel stride in bytes
  long long wrow_bytes;       // taps*Cs*esize
  int ntn;                    // number of cout tiles
  int Cd;                     // produced channels
  ps_epilogue epi;
};

struct TraitsBF16 {
  typedef __bf16 elem;
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& x, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
  }
};
struct TraitsF32 {
  typedef float elem;
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

template <typename Tr, int BN, bool GLDS>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const IgemmArgs a) {
  typedef typename Tr::elem T;
  constexpr int BM = 128;
  constexpr int WM = (BN == 128) ? 64 : 32;  // pixels per wave
  constexpr int MI = WM / 16;                // pixel fragments per wave
  constexpr int BROWS = BN / 32;             // weight rows staged per lane
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bid = ps_xcd_remap(blockIdx.x, gridDim.x);
  const int tn = bid % a.ntn, tm = bid / a.ntn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (BN == 128) ? (wave >> 1) : wave;
  const int wn = (BN == 128) ? (wave & 1) : 0;

  // ---------------- staging state ----------------
  // pixel rows: LDS row R = wave*32 + j*8 + (lane>>3); the lane moves source chunk (lane&7)^(R&7) to position lane&7
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
      py[j] = -(1 << 20);  // never valid
      px[j] = 0;
      nb[j] = 0;
    }
  }
  // weight rows: LDS row Rb = wave*(BN/4) + j*8 + (lane>>3) holds cout n0 + 64*(Rb>>6) + perm(Rb&63), where
  // perm(16*i + rho) = 16*(rho>>2) + 4*i + (rho&3)  (so that accumulator register r of fragment i, lane
  // group g is cout 16*g + 4*i + r: 16 contiguous couts per lane)
  const unsigned char* wptr[BROWS];
#pragma unroll
  for (int j = 0; j < BROWS; ++j) {
    const int rb = wave * (BN / 4) + j * 8 + srow;
    const int within = rb & 63, fi = within >> 4, rho = within & 15;
    const int cout = n0 + (rb & ~63) + 16 * (rho >> 2) + 4 * fi + (rho & 3);
    wptr[j] = a.wgt + (long long)cout * a.wrow_bytes + chunk_off;
  }
  const unsigned char* aptr[4];
  int tap = 0, kl = 0;
  long long wk = 0;  // running K byte offset in a weight row

  auto tap_pointers = [&](int t) {
    const int ty = (a.taps == 1) ? a.ctr : t / 3, tx = (a.taps == 1) ? a.ctr : t - (t / 3) * 3;
    const int dy = (ty - a.ctr) * a.dstep, dx = (tx - a.ctr) * a.dstep;
    const int dmask = (1 << a.div_shift) - 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int yn = py[j] + dy, xn = px[j] + dx;
      const int y = yn >> a.div_shift, x = xn >> a.div_shift;
      const bool ok = (yn >= 0) && (xn >= 0) && (((yn | xn) & dmask) == 0) && (y < a.Hs) && (x < a.Ws);
      aptr[j] = ok ? a.src + (long long)(nb[j] + y * a.Ws + x) * a.pix_bytes + chunk_off : nullptr;
    }
  };
  tap_pointers(0);

  u32x4 ra[4], rb_[BROWS];  // register staging (GLDS == false)
  auto stage_issue = [&](int buf) {
    unsigned char* sa = smem + buf * STAGE + (wave * 32) * 128 + lane * 16;
    unsigned char* sb = smem + buf * STAGE + A_BYTES + (wave * (BN / 4)) * 128 + lane * 16;
    const long long ko = (long long)kl * 128;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned char* g = aptr[j] ? aptr[j] + ko : g_zero_line + (lane & 7) * 16;
      if constexpr (GLDS) GLDS16(g, sa + j * 1024 - lane * 16);
      else ra[j] = *reinterpret_cast<const u32x4*>(g);
    }
#pragma unroll
    for (int j = 0; j < BROWS; ++j) {
      const unsigned char* g = wptr[j] + wk;
      if constexpr (GLDS) GLDS16(g, sb + j * 1024 - lane * 16);
      else rb_[j] = *reinterpret_cast<const u32x4*>(g);
    }
    wk += 128;
    if (++kl == a.klines) {
      kl = 0;
      if (++tap < a.taps) tap_pointers(tap);
    }
  };
  auto stage_commit = [&](int buf) {  // register staging only: registers -> LDS
    if constexpr (!GLDS) {
      unsigned char* sa = smem + buf * STAGE + (wave * 32) * 128 + lane * 16;
      unsigned char* sb = smem + buf * STAGE + A_BYTES + (wave * (BN / 4)) * 128 + lane * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4*>(sa + j * 1024) = ra[j];
#pragma unroll
      for (int j = 0; j < BROWS; ++j) *reinterpret_cast<u32x4*>(sb + j * 1024) = rb_[j];
    }
  };

  // ---------------- accumulate ----------------
  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[mi][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, g = lane >> 4;
  const int nsteps = a.taps * a.klines;
  // fragment byte offsets inside a stage (row&7 == lane&7 for both operands)
  const int xoff = (wm * WM + frow) * 128, woff = A_BYTES + (wn * 64 + frow) * 128;
  const int sw = lane & 7;

  stage_issue(0);
  stage_commit(0);
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const int cur = s & 1;
    if (s + 1 < nsteps) stage_issue(cur ^ 1);
    const unsigned char* st = smem + cur * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((g + 4 * kk) ^ sw) << 4;
      u32x4 wf[4], xf[MI];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *reinterpret_cast<const u32x4*>(st + woff + i * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const u32x4*>(st + xoff + mi * 2048 + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int i = 0; i < 4; ++i) Tr::mma(wf[i], xf[mi], acc[mi][i]);
    }
    if (s + 1 < nsteps) stage_commit(cur ^ 1);
    __syncthreads();
  }

  // ---------------- epilogue ----------------
  const ps_epilogue& e = a.epi;
  const int cb = n0 + wn * 64 + 16 * g;  // this lane's 16 contiguous produced channels
  float sc[16], sh[16];
  if (e.mode != PS_EPI_NONE) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sc[i] = e.scale ? e.scale[cb + i] : 1.f;
      sh[i] = (e.shift && e.mode == PS_EPI_BNRELU) ? e.shift[cb + i] : 0.f;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * WM + mi * 16 + frow;
    if (m >= a.M) continue;
    float v[16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[4 * i + r] = acc[mi][i][r];
    if (e.add0) {
      float t[16];
      load16<T>(reinterpret_cast<const T*>(e.add0) + (long long)m * e.ldc_add0 + cb, t);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] += t[i];
    }
    if (e.out_raw) store16<T>(reinterpret_cast<T*>(e.out_raw) + (long long)m * e.ldc_raw + cb, v);
    if (e.mode == PS_EPI_NONE) continue;
    float dm[16];
    if (e.drop) {
      const int n = m / (a.Ho * a.Wo);
      const float* d = e.drop + (long long)n * a.Cd + cb;
#pragma unroll
      for (int i = 0; i < 16; ++i) dm[i] = d[i];
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) dm[i] = 1.f;
    }
    if (e.mode == PS_EPI_BNRELU) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = fmaxf(v[i] * sc[i] + sh[i], 0.f) * dm[i];
    } else {  // PS_EPI_RELUBWD
      float ms[16];
      load16<T>(reinterpret_cast<const T*>(e.mask_src) + (long long)m * e.ldc_mask + cb, ms);
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = ms[i] > 0.f ? v[i] * sc[i] * dm[i] : 0.f;
      if (e.add1) {
        float t[16];
        load16<T>(reinterpret_cast<const T*>(e.add1) + (long long)m * e.ldc_add1 + cb, t);
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] += t[i];
      }
    }
    store16<T>(reinterpret_cast<T*>(e.out) + (long long)m * e.ldc_out + cb, v);
  }
}

static int g_use_glds = 1;

template <typename Tr, int BN>
int launch_igemm(const IgemmArgs& a, int ntm, hipStream_t stream) {
  const int grid = ntm * a.ntn;
  const size_t lds = 2 * (128 * 128 + BN * 128);
  if (g_use_glds) {
    hipLaunchKernelGGL((conv_igemm_kernel<Tr, BN, true>), dim3(grid), dim3(256), lds, stream, a);
  } else {
    hipLaunchKernelGGL((conv_igemm_kernel<Tr, BN, false>), dim3(grid), dim3(256), lds, stream, a);
  }
  PS_CHECK_LAUNCH("conv_igemm");
  return PS_OK;
}

int check_geom(const ps_conv_geom* g) {
  PS_REQUIRE(g != nullptr, "conv: null geometry");
  PS_REQUIRE(g->dtype == PS_F32 || g->dtype == PS_BF16, "conv: dtype %d unsupported", g->dtype);
  PS_REQUIRE(g->ksize == 1 || g->ksize == 3, "conv: ksize %d unsupported (1 or 3)", g->ksize);
  PS_REQUIRE(g->stride == 1 || g->stride == 2, "conv: stride %d unsupported (1 or 2)", g->stride);
  PS_REQUIRE(g->dilation >= 1 && g->dilation <= 64, "conv: dilation %d unsupported", g->dilation);
  PS_REQUIRE(g->n > 0 && g->h > 0 && g->w > 0, "conv: empty input %dx%dx%d", g->n, g->h, g->w);
  const int es = ps_esize(g->dtype);
  PS_REQUIRE((g->cin * es) % 128 == 0 && (g->cout * es) % 128 == 0,
             "conv: cin=%d cout=%d must be multiples of %d channels", g->cin, g->cout, 128 / es);
  PS_REQUIRE(g->cin % 64 == 0 && g->cout % 64 == 0, "conv: cin=%d cout=%d must be multiples of 64", g->cin, g->cout);
  PS_REQUIRE(g->ldc_x >= g->cin && g->ldc_y >= g->cout, "conv: channel strides smaller than channel counts");
  PS_REQUIRE((g->ldc_x * es) % 16 == 0 && (g->ldc_y * es) % 16 == 0, "conv: channel strides must be 16-byte multiples");
  const long long ho = (g->h - 1) / g->stride + 1, wo = (g->w - 1) / g->stride + 1;
  PS_REQUIRE((long long)g->n * g->h * g->w < (1LL << 31) && (long long)g->n * ho * wo < (1LL << 31), "conv: too many pixels");
  return PS_OK;
}

int check_epilogue(const ps_epilogue* e, int dtype, const char* who) {
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
  }
  return PS_OK;
}

template <typename Tr>
int dispatch_bn(const IgemmArgs& a, hipStream_t s) {
  const int ntm = (a.M + 127) / 128;
  // BN = 128 whenever it divides Cd and there are enough tiles to fill 256 CUs twice over; else 64.
  if (a.Cd % 128 == 0 && (long long)ntm * (a.Cd / 128) >= 512) {
    IgemmArgs b = a;
    b.ntn = a.Cd / 128;
    return launch_igemm<Tr, 128>(b, ntm, s);
  }
  IgemmArgs b = a;
  b.ntn = a.Cd / 64;
  return launch_igemm<Tr, 64>(b, ntm, s);
}

}  // namespace

extern "C" void ps_debug_set_glds(int on) { g_use_glds = on; }

extern "C" int ps_conv_supported(const ps_conv_geom* g) { return check_geom(g) == PS_OK ? 1 : 0; }

extern "C" int ps_conv2d_fwd(const ps_conv_geom* g, const void* x, const void* w_fwd, const ps_epilogue* epi, void* stream) {
  if (int rc = check_geom(g)) return rc;
  if (int rc = check_epilogue(epi, g->dtype, "conv2d_fwd")) return rc;
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
  a.klines = g->cin * es / 128;
  a.pix_bytes = (long long)g->ldc_x * es;
  a.wrow_bytes = (long long)a.taps * g->cin * es;
  a.Cd = g->cout;
  a.epi = *epi;
  return g->dtype == PS_BF16 ? dispatch_bn<TraitsBF16>(a, static_cast<hipStream_t>(stream))
                             : dispatch_bn<TraitsF32>(a, static_cast<hipStream_t>(stream));
}

extern "C" int ps_conv2d_dgrad(const ps_conv_geom* g, const void* dy, const void* w_dgrad, const ps_epilogue* epi, void* stream) {
  if (int rc = check_geom(g)) return rc;
  if (int rc = check_epilogue(epi, g->dtype, "conv2d_dgrad")) return rc;
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
  a.klines = g->cout * es / 128;
  a.pix_bytes = (long long)g->ldc_y * es;
  a.wrow_bytes = (long long)a.taps * g->cout * es;
  a.Cd = g->cin;
  a.epi = *epi;
  return g->dtype == PS_BF16 ? dispatch_bn<TraitsBF16>(a, static_cast<hipStream_t>(stream))
                             : dispatch_bn<TraitsF32>(a, static_cast<hipStream_t>(stream));
}
