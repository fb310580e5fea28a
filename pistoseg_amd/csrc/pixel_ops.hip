// Per-pixel, HBM-bound kernels: bilinear resize (fwd + gather-form bwd), softmax cross-entropy (fwd+bwd fused),
// CAM/logit -> mask reductions and the confusion matrix.  One lane per pixel, channel loops in registers,
// wavefront (64-lane) shuffles for the reductions.
#include <math.h>

#include <algorithm>

#include "ps_internal.h"

namespace {

struct T4 {
  unsigned char* ptr;
  int dtype, n, c, h, w;
  long long sn, sc, sh, sw;
};
static T4 to_t4(const ps_tensor4* t) {
  T4 r;
  r.ptr = static_cast<unsigned char*>(t->ptr);
  r.dtype = t->dtype; r.n = t->n; r.c = t->c; r.h = t->h; r.w = t->w;
  r.sn = t->sn; r.sc = t->sc; r.sh = t->sh; r.sw = t->sw;
  return r;
}
__device__ __forceinline__ float t4_load(const T4& t, long long off) {
  return ps_ld_dt(t.ptr, t.dtype, off);
}
__device__ __forceinline__ void t4_store(const T4& t, long long off, float v) {
  ps_st_dt(t.ptr, t.dtype, off, v);
}

// torch's area_pixel_compute_scale / _source_index (aten/src/ATen/native/UpSample.h), f32 arithmetic.
__host__ __device__ inline float interp_scale(int in, int out, bool align) {
  if (align) return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  return (float)in / (float)out;
}
__device__ __forceinline__ void interp_index(float scale, int dst, int in, bool align, int& i0, int& i1, float& l0, float& l1) {
  float src;
  if (align) src = scale * (float)dst;
  else {
    src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
  }
  i0 = (int)src;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

// CT > 0 / SDT: channel count and source dtype at compile time -- a pixel's 4 x C source values are then loaded back to back (with the
// run-time loop and the dtype switch inside every load, each load sat in its own basic block and was consumed before the next was
// issued: 27 us for the 28 -> 224 upsample of a 64-tile batch); CT == 0: any channel count, the loop as written.
template <int CT, int SDT>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T4 src, const T4 dst, int align, float sy, float sx) {
  const long long total = (long long)dst.n * dst.h * dst.w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % dst.w);
    const long long t = i / dst.w;
    const int oy = (int)(t % dst.h), n = (int)(t / dst.h);
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    interp_index(sy, oy, src.h, align, y0, y1, ly0, ly1);
    interp_index(sx, ox, src.w, align, x0, x1, lx0, lx1);
    const long long b = (long long)n * src.sn;
    const long long o = (long long)n * dst.sn + (long long)oy * dst.sh + (long long)ox * dst.sw;
    const long long o00 = b + y0 * src.sh + x0 * src.sw, o01 = b + y0 * src.sh + x1 * src.sw;
    const long long o10 = b + y1 * src.sh + x0 * src.sw, o11 = b + y1 * src.sh + x1 * src.sw;
    if constexpr (CT > 0) {
      float v[CT][4];
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const long long bc = (long long)c * src.sc;
        v[c][0] = ps_ld_dt(src.ptr, SDT, o00 + bc); v[c][1] = ps_ld_dt(src.ptr, SDT, o01 + bc);
        v[c][2] = ps_ld_dt(src.ptr, SDT, o10 + bc); v[c][3] = ps_ld_dt(src.ptr, SDT, o11 + bc);
      }
#pragma unroll
      for (int c = 0; c < CT; ++c)
        t4_store(dst, o + (long long)c * dst.sc, ly0 * (lx0 * v[c][0] + lx1 * v[c][1]) + ly1 * (lx0 * v[c][2] + lx1 * v[c][3]));
    } else {
      for (int c = 0; c < dst.c; ++c) {
        const long long bc = (long long)c * src.sc;
        const float v00 = ps_ld_dt(src.ptr, SDT, o00 + bc), v01 = ps_ld_dt(src.ptr, SDT, o01 + bc);
        const float v10 = ps_ld_dt(src.ptr, SDT, o10 + bc), v11 = ps_ld_dt(src.ptr, SDT, o11 + bc);
        t4_store(dst, o + (long long)c * dst.sc, ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11));
      }
    }
  }
}

// Gather-form backward: one wave per source pixel (n, iy, ix); lanes sweep the window of destination
// pixels whose interpolation footprint touches it, then a wave reduction per channel.  Deterministic.
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T4 dd, const T4 ds, int align, float sy, float sx, int wy, int wx) {
  const int lane = threadIdx.x & 63;
  const long long wave_id = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long total = (long long)ds.n * ds.h * ds.w;
  if (wave_id >= total) return;
  const int ix = (int)(wave_id % ds.w);
  const long long t = wave_id / ds.w;
  const int iy = (int)(t % ds.h), n = (int)(t / ds.h);
  // candidate destination window: every oy with i0 in {iy-1, iy}
  int oy_lo, ox_lo;
  {
    const float inv_y = sy > 0.f ? 1.f / sy : 0.f, inv_x = sx > 0.f ? 1.f / sx : 0.f;
    // smallest dst whose source coordinate reaches iy-1 (align: src = s*dst; else src = s*(dst+.5)-.5)
    const float hb = align ? 0.f : 0.5f;
    oy_lo = sy > 0.f ? max(0, (int)floorf(((float)iy - 1.f + hb) * inv_y - hb) - 1) : 0;
    ox_lo = sx > 0.f ? max(0, (int)floorf(((float)ix - 1.f + hb) * inv_x - hb) - 1) : 0;
  }
  constexpr int MAXC = 8;
  for (int c0 = 0; c0 < ds.c; c0 += MAXC) {
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    for (int k = lane; k < wy * wx; k += 64) {
      const int oy = oy_lo + k / wx, ox = ox_lo + k % wx;
      if (oy >= dd.h || ox >= dd.w) continue;
      int y0, y1, x0, x1;
      float ly0, ly1, lx0, lx1;
      interp_index(sy, oy, ds.h, align, y0, y1, ly0, ly1);
      interp_index(sx, ox, ds.w, align, x0, x1, lx0, lx1);
      const float wyv = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
      const float wxv = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
      const float wgt = wyv * wxv;
      if (wgt == 0.f) continue;
      const long long o = (long long)n * dd.sn + (long long)oy * dd.sh + (long long)ox * dd.sw;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c0 + c < ds.c) acc[c] = fmaf(wgt, t4_load(dd, o + (long long)(c0 + c) * dd.sc), acc[c]);
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      if (c0 + c < ds.c) {
        const float s = ps_wave_sum(acc[c]);
        if (lane == 0) t4_store(ds, (long long)n * ds.sn + (long long)(c0 + c) * ds.sc + (long long)iy * ds.sh + (long long)ix * ds.sw, s);
      }
    }
  }
}

// Separable form of the same sum for destination rows of up to 1024 pixels: one block per SOURCE ROW (n, iy).  Pass 1 -- the threads own
// destination columns and fold the destination rows whose footprint touches iy into per-column sums (row weights are block-uniform;
// NCHW gradients are read along x, coalesced, each row by the two source rows it feeds); pass 2 -- one thread per (ix, c) folds its
// window of column sums from LDS.  Deterministic; reads the gradient ~2x instead of the ~7x (three quarters of them weight-zero
// candidates) of the wave-per-source-pixel form.
// The row loop is branch-free and unrolled -- rows past the end are clamped and weigh 0 like the window's slack rows, the channel count
// of a chunk and the gradient's dtype are template arguments -- so that the loads of several rows are in flight together: with a
// `continue` per row and a run-time `if (c < channels)` around every load each load sat in its own basic block and the block paid
// one memory latency per LOAD (63 of them: 47 us; measured by ablation, pass 2 is 9 us).
constexpr int BWD_MAXC = 8, BWD_SLOTS = 4;
template <int DDT, int CC>
__global__ __launch_bounds__(256) void bilinear_bwd_rows_kernel(const T4 dd, const T4 ds, int align, float sy, float sx, int wy, int wx) {
  extern __shared__ float cols[];  // [channel of the chunk][dd.w]
  const int tid = threadIdx.x;
  const int iy = (int)(blockIdx.x % (unsigned)ds.h), n = (int)(blockIdx.x / (unsigned)ds.h);
  const float hb = align ? 0.f : 0.5f;
  const float inv_y = sy > 0.f ? 1.f / sy : 0.f, inv_x = sx > 0.f ? 1.f / sx : 0.f;
  const int oy_lo = sy > 0.f ? max(0, (int)floorf(((float)iy - 1.f + hb) * inv_y - hb) - 1) : 0;
  for (int c0 = 0; c0 < ds.c; c0 += CC) {  // (a last partial chunk re-reads the last channel and does not store the copies)
    for (int ox0 = 0; ox0 < dd.w; ox0 += 256) {
      const int ox = ox0 + tid;
      const long long ocol = (long long)n * dd.sn + (long long)min(ox, dd.w - 1) * dd.sw;
      long long och[CC];
#pragma unroll
      for (int c = 0; c < CC; ++c) och[c] = ocol + (long long)min(c0 + c, ds.c - 1) * dd.sc;
      float acc[CC];
#pragma unroll
      for (int c = 0; c < CC; ++c) acc[c] = 0.f;
#pragma unroll 4
      for (int r = 0; r < wy; ++r) {
        const int oyr = oy_lo + r, oy = min(oyr, dd.h - 1);
        int y0, y1;
        float ly0, ly1;
        interp_index(sy, oy, ds.h, align, y0, y1, ly0, ly1);
        const float wyv = oyr < dd.h ? (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f) : 0.f;
        const long long orow = (long long)oy * dd.sh;
#pragma unroll
        for (int c = 0; c < CC; ++c) acc[c] = fmaf(wyv, ps_ld_dt(dd.ptr, DDT, och[c] + orow), acc[c]);
      }
      if (ox < dd.w) {
#pragma unroll
        for (int c = 0; c < CC; ++c) cols[c * dd.w + ox] = acc[c];
      }
    }
    __syncthreads();
    const int cc = min(CC, ds.c - c0);
    for (int j = tid; j < ds.w * cc; j += 256) {
      const int ix = j / cc, c = j - ix * cc;
      const int ox_lo = sx > 0.f ? max(0, (int)floorf(((float)ix - 1.f + hb) * inv_x - hb) - 1) : 0;
      float sum = 0.f;
      for (int k = 0; k < wx; ++k) {
        const int ox = ox_lo + k;
        if (ox >= dd.w) break;
        int x0, x1;
        float lx0, lx1;
        interp_index(sx, ox, ds.w, align, x0, x1, lx0, lx1);
        const float wxv = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
        sum = fmaf(wxv, cols[c * dd.w + ox], sum);
      }
      t4_store(ds, (long long)n * ds.sn + (long long)(c0 + c) * ds.sc + (long long)iy * ds.sh + (long long)ix * ds.sw, sum);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// softmax cross-entropy, mean over ALL pixels
// ------------------------------------------------------------------------------------------------
constexpr int CE_BLOCKS = 1024;
// channel loop: fully unrolled for a compile-time count CT > 0, a plain loop over the run-time count otherwise
template <int CT, class F>
__device__ __forceinline__ void for_ch(int c, F&& f) {
  if constexpr (CT > 0) {
#pragma unroll
    for (int k = 0; k < CT; ++k) f(k);
  } else {
    for (int k = 0; k < c; ++k) f(k);
  }
}

// CT > 0: the channel count is a compile-time constant -- a pixel's logits are loaded ONCE, back to back, into registers (with a run-time
// count the three channel loops re-read them and every load is consumed before the next is issued: one memory latency per load);
// CT == 0: any channel count, the loops as written.  `fast`: pixel -> (image, offset) by magic-number division (indices < 2^31).
template <int CT>
__global__ __launch_bounds__(256) void ce_kernel(const float* __restrict__ z, const long long* __restrict__ tgt, float* __restrict__ dz,
                                                 float* __restrict__ partials, float gscale, int n, int c_rt, long long hw, int ignore,
                                                 int fast, const FastDiv div_hw) {
  __shared__ float red[4];
  const int c = CT ? CT : c_rt;
  const long long total = (long long)n * hw;
  float local = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = fast ? (long long)fdiv((uint32_t)i, div_hw) : i / hw, pix = i - img * hw;
    const float* zp = z + img * c * hw + pix;
    const long long t = tgt[i];
    float zv[CT ? CT : 1];
    if constexpr (CT > 0) {
#pragma unroll
      for (int k = 0; k < CT; ++k) zv[k] = zp[k * hw];
    }
    auto zk = [&](int k) { return CT ? zv[CT ? k : 0] : zp[k * hw]; };
    float mx = -INFINITY;
    for_ch<CT>(c, [&](int k) { mx = fmaxf(mx, zk(k)); });
    float se = 0.f;
    for_ch<CT>(c, [&](int k) { se += expf(zk(k) - mx); });
    const float lse = mx + logf(se);
    const bool live = (t != ignore) && t >= 0 && t < c;
    if (live) {
      float zt = 0.f;
      if constexpr (CT > 0) {
#pragma unroll
        for (int k = 0; k < CT; ++k) zt = (k == (int)t) ? zv[k] : zt;
      } else {
        zt = zp[t * hw];
      }
      local += lse - zt;
    }
    if (dz) {
      float* dp = dz + img * c * hw + pix;
      for_ch<CT>(c, [&](int k) {
        float g = 0.f;
        if (live) g = (expf(zk(k) - lse) - (k == t ? 1.f : 0.f)) * gscale;
        dp[k * hw] = g;
      });
    }
  }
  local = ps_wave_sum(local);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void ce_finish_kernel(const float* __restrict__ partials, int nparts, float inv_total, float* __restrict__ out) {
  __shared__ float red[4];
  float v = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) v += partials[i];
  v = ps_wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_total;
}


// ------------------------------------------------------------------------------------------------
// multiclass Dice loss (smp.losses.DiceLoss(mode='multiclass', from_logits=True, smooth=0, eps=1e-7,
// dims=(0,2)) as used by models/mosaic_module.py:65-68,108 -- third-party arithmetic, PARITY UNPINNED).
// sums[0..C) = sum p*t, sums[C..2C) = sum (p + t), sums[2C..3C) = sum t   over kept pixels
// ------------------------------------------------------------------------------------------------
constexpr int DICE_BLOCKS = 512, DICE_MAXC = 8;
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ z, const long long* __restrict__ tgt, float* __restrict__ partials,
                                                        int n, int c, long long hw, int ignore) {
  __shared__ float red[4][3 * DICE_MAXC];
  float acc[3 * DICE_MAXC];
#pragma unroll
  for (int i = 0; i < 3 * DICE_MAXC; ++i) acc[i] = 0.f;
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    const long long t = tgt[i];
    if (t == ignore) continue;
    const float* zp = z + img * c * hw + pix;
    float mx = -INFINITY;
    for (int k = 0; k < c; ++k) mx = fmaxf(mx, zp[k * hw]);
    float se = 0.f;
    for (int k = 0; k < c; ++k) se += expf(zp[k * hw] - mx);
    const float lse = mx + logf(se);
#pragma unroll
    for (int k = 0; k < DICE_MAXC; ++k) {
      if (k < c) {
        const float p = expf(zp[k * hw] - lse), tt = (k == t) ? 1.f : 0.f;
        acc[k] += p * tt;
        acc[DICE_MAXC + k] += p + tt;
        acc[2 * DICE_MAXC + k] += tt;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 3 * DICE_MAXC; ++i) {
    const float v = ps_wave_sum(acc[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3 * DICE_MAXC)
    partials[blockIdx.x * 3 * DICE_MAXC + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(64) void dice_finish_kernel(const float* __restrict__ partials, int nparts, int c, float* __restrict__ sums,
                                                         float* __restrict__ loss) {
  __shared__ float s[3 * DICE_MAXC];
  if (threadIdx.x < 3 * DICE_MAXC) {
    float v = 0.f;
    for (int b = 0; b < nparts; ++b) v += partials[b * 3 * DICE_MAXC + threadIdx.x];
    s[threadIdx.x] = v;
    sums[threadIdx.x] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float l = 0.f;
    for (int k = 0; k < c; ++k) {
      const float score = 2.f * s[k] / fmaxf(s[DICE_MAXC + k], 1e-7f);
      l += (s[2 * DICE_MAXC + k] > 0.f) ? (1.f - score) : 0.f;
    }
    loss[0] = l / (float)c;
  }
}
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ z, const long long* __restrict__ tgt, const float* __restrict__ sums,
                                                       float* __restrict__ dz, float gscale, int n, int c, long long hw, int ignore) {
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = i / hw, pix = i - img * hw;
    const long long t = tgt[i];
    const float* zp = z + img * c * hw + pix;
    float* dp = dz + img * c * hw + pix;
    if (t == ignore) {
      for (int k = 0; k < c; ++k) dp[k * hw] = 0.f;
      continue;
    }
    float mx = -INFINITY;
    for (int k = 0; k < c; ++k) mx = fmaxf(mx, zp[k * hw]);
    float se = 0.f;
    for (int k = 0; k < c; ++k) se += expf(zp[k * hw] - mx);
    const float lse = mx + logf(se);
    float p[DICE_MAXC], g[DICE_MAXC];
    float dot = 0.f;
    for (int k = 0; k < c; ++k) {
      p[k] = expf(zp[k * hw] - lse);
      const float I = sums[k], S = sums[DICE_MAXC + k], T = sums[2 * DICE_MAXC + k];
      float gk = 0.f;
      if (T > 0.f && S > 1e-7f) gk = (-2.f * ((k == t) ? 1.f : 0.f) / S + 2.f * I / (S * S)) / (float)c;
      g[k] = gk;
      dot += p[k] * gk;
    }
    for (int k = 0; k < c; ++k) dp[k * hw] = gscale * p[k] * (g[k] - dot);
  }
}

// ------------------------------------------------------------------------------------------------
// CAM / logit -> mask
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool better(float v, float best) { return v > best || (v != v && best == best); }

// (CT / fast: as ce_kernel -- compile-time channel count = the pixel's values loaded once, back to back)
template <int CT>
__global__ __launch_bounds__(256) void argmax_mask_kernel(const float* __restrict__ x, const float* __restrict__ label,
                                                          const uint8_t* __restrict__ tissue, uint8_t* __restrict__ mask,
                                                          float* __restrict__ entropy, int mode, int softmax_first, int first_ch, int n,
                                                          int c_rt, long long hw, int fast, const FastDiv div_hw) {
  const int c = CT ? CT : c_rt;
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long img = fast ? (long long)fdiv((uint32_t)i, div_hw) : i / hw, pix = i - img * hw;
    const float* xp = x + img * c * hw + pix;
    const float* lb = label ? label + img * c : nullptr;
    float xv[CT ? CT : 1];
    if constexpr (CT > 0) {
#pragma unroll
      for (int k = 0; k < CT; ++k) xv[k] = xp[k * hw];
    }
    auto xk = [&](int k) { return CT ? xv[CT ? k : 0] : xp[k * hw]; };
    int best_i = 0;
    float ent = 0.f;
    if (mode == PS_MASK_FILL) {
      float lsum = 0.f;
      for_ch<CT>(c, [&](int k) { lsum += lb[k]; });
      if (lsum == 1.f) {  // single tissue type: constant mask (patch_label.index(1)), zero entropy
        int first1 = 0;
        for_ch<CT>(c, [&](int j) {
          const int k = c - 1 - j;
          if (lb[k] == 1.f) first1 = k;
        });
        best_i = first1;
      } else {
        float mx = -INFINITY;
        for_ch<CT>(c, [&](int k) { mx = fmaxf(mx, lb[k] == 0.f ? -1e10f : xk(k)); });
        float se = 0.f;
        for_ch<CT>(c, [&](int k) { se += expf((lb[k] == 0.f ? -1e10f : xk(k)) - mx); });
        float best = -INFINITY;
        for_ch<CT>(c, [&](int k) {
          const float p = expf((lb[k] == 0.f ? -1e10f : xk(k)) - mx) / se;
          ent -= p * logf(p + 1e-10f);
          if (k == 0 || better(p, best)) { best = p; best_i = k; }
        });
      }
      if (tissue && tissue[i] == 0) best_i = c;
    } else {
      float mx = -INFINITY, se = 1.f;
      if (softmax_first) {
        for_ch<CT>(c, [&](int k) {
          if (k >= first_ch) mx = fmaxf(mx, xk(k));
        });
        se = 0.f;
        for_ch<CT>(c, [&](int k) {
          if (k >= first_ch) se += expf(xk(k) - mx);
        });
      }
      float best = 0.f;
      for_ch<CT>(c, [&](int k) {
        if (k < first_ch) return;
        float v = xk(k);
        if (softmax_first) v = expf(v - mx) / se;
        if (mode == PS_MASK_MUL) v *= lb[k];
        if (k == first_ch || better(v, best)) { best = v; best_i = k - first_ch; }
      });
    }
    mask[i] = (uint8_t)best_i;
    if (entropy) entropy[i] = ent;
  }
}

__global__ __launch_bounds__(256) void confusion_kernel(const uint8_t* __restrict__ pred, const long long* __restrict__ gt,
                                                        unsigned long long* __restrict__ cm, long long npix, int nc) {
  __shared__ unsigned int hist[256];
  if (threadIdx.x < 256) hist[threadIdx.x] = 0;
  __syncthreads();
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long long)gridDim.x * 256) {
    // loss.py:63 casts the mask with .byte() before the range test
    const int g = (int)(uint8_t)gt[i], p = pred[i];
    if (g < nc && p < nc) atomicAdd(&hist[g * nc + p], 1u);
  }
  __syncthreads();
  if (threadIdx.x < nc * nc && hist[threadIdx.x]) atomicAdd(&cm[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

// loss.py:28-53 on the device: per-tissue IoU, their mean, the frequency-weighted IoU -- f64, the sums taken in numpy's order (row / column
// sums of exact integers; the mean and the weighted sum left to right, as numpy's pairwise sum does for fewer than 8 terms... and, for up to
// 16 classes, in its 8-accumulator block order -- see iou_sum), so that the values equal the host computation bit for bit.
__device__ inline double iou_sum(const double* v, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += v[i];
    return n ? s : 0.0;
  }
  double r[8];  // numpy pairwise_sum, 8 <= n <= 128: eight strided partial sums, combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the remainder
  for (int j = 0; j < 8; ++j) r[j] = v[j];
  int i = 8;
  for (; i + 8 <= n; i += 8)
    for (int j = 0; j < 8; ++j) r[j] += v[i + j];
  double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) s += v[i];
  return s;
}

__global__ void iou_from_confusion_kernel(const long long* __restrict__ cm, int nc, double* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double rows[16], cols[16], iou[16], w[16];
  double total_rows[16];
  for (int i = 0; i < nc; ++i) {
    double r[16], c[16];
    for (int j = 0; j < nc; ++j) {
      r[j] = (double)cm[i * nc + j];
      c[j] = (double)cm[j * nc + i];
    }
    rows[i] = iou_sum(r, nc);
    cols[i] = iou_sum(c, nc);
    total_rows[i] = rows[i];
  }
  const double total = iou_sum(total_rows, nc);  // (integers below 2^53: any order gives the same value)
  int nw = 0;
  for (int i = 0; i < nc; ++i) {
    const double d = (double)cm[i * nc + i];
    double v = d / (rows[i] + cols[i] - d);  // 0/0 -> NaN
    const double freq = rows[i] / total;     // NaN for an empty matrix: `freq > 0` is then false everywhere and the sum is empty (0.0)
    if (freq > 0) w[nw++] = freq * v;
    iou[i] = (v != v) ? 0.0 : v;             // Tissue_Intersection_over_Union: iou[np.isnan(iou)] = 0
    out[2 + i] = iou[i];
  }
  out[0] = iou_sum(iou, nc) / (double)nc;  // np.mean = pairwise sum / n
  out[1] = iou_sum(w, nw);
}

static inline int grid_for(long long items, int per_block, int cap = 2048) {
  long long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

int check_t4(const ps_tensor4* t, const char* who) {
  PS_REQUIRE(t && t->ptr, "%s: null tensor", who);
  PS_REQUIRE(ps_dtype_ok(t->dtype), "%s: dtype %d unsupported", who, t->dtype);
  PS_REQUIRE(t->n > 0 && t->c > 0 && t->h > 0 && t->w > 0, "%s: empty tensor", who);
  return PS_OK;
}

}  // namespace

extern "C" int ps_bilinear_fwd(const ps_tensor4* src, const ps_tensor4* dst, int32_t align, void* stream) {
  if (int rc = check_t4(src, "bilinear_fwd")) return rc;
  if (int rc = check_t4(dst, "bilinear_fwd")) return rc;
  PS_REQUIRE(src->n == dst->n && src->c == dst->c, "bilinear_fwd: batch/channel mismatch");
  const float sy = interp_scale(src->h, dst->h, align != 0), sx = interp_scale(src->w, dst->w, align != 0);
  const long long total = (long long)dst->n * dst->h * dst->w;
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream), to_t4(src), to_t4(dst), align, sy, sx);
  };
#define PS_BIL_FWD_C(DT)                                    \
  switch (src->c) {                                        \
    case 1: launch(bilinear_fwd_kernel<1, DT>); break;     \
    case 2: launch(bilinear_fwd_kernel<2, DT>); break;     \
    case 3: launch(bilinear_fwd_kernel<3, DT>); break;     \
    case 4: launch(bilinear_fwd_kernel<4, DT>); break;     \
    case 5: launch(bilinear_fwd_kernel<5, DT>); break;     \
    default: launch(bilinear_fwd_kernel<0, DT>); break;    \
  }
  if (src->dtype == PS_F32) { PS_BIL_FWD_C(PS_F32) }
  else if (src->dtype == PS_BF16) { PS_BIL_FWD_C(PS_BF16) }
  else { PS_BIL_FWD_C(PS_F16) }
#undef PS_BIL_FWD_C
  PS_CHECK_LAUNCH("bilinear_fwd");
  return PS_OK;
}

extern "C" int ps_bilinear_bwd(const ps_tensor4* ddst, const ps_tensor4* dsrc, int32_t align, void* stream) {
  if (int rc = check_t4(ddst, "bilinear_bwd")) return rc;
  if (int rc = check_t4(dsrc, "bilinear_bwd")) return rc;
  PS_REQUIRE(ddst->n == dsrc->n && ddst->c == dsrc->c, "bilinear_bwd: batch/channel mismatch");
  const float sy = interp_scale(dsrc->h, ddst->h, align != 0), sx = interp_scale(dsrc->w, ddst->w, align != 0);
  // window of destination pixels per source pixel: ~2/scale (+ slack for the floor/ceil and rounding)
  const int wy = sy > 0.f ? (int)ceilf(2.f / sy) + 4 : ddst->h;
  const int wx = sx > 0.f ? (int)ceilf(2.f / sx) + 4 : ddst->w;
  const long long rows = (long long)dsrc->n * dsrc->h;
  if (ddst->w <= 256 * BWD_SLOTS && rows < (1LL << 31)) {
    const size_t lds = (size_t)std::min<int>(BWD_MAXC, dsrc->c) * ddst->w * sizeof(float);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const T4 a = to_t4(ddst), b = to_t4(dsrc);
#define PS_BIL_ROWS(DT, CC) hipLaunchKernelGGL((bilinear_bwd_rows_kernel<DT, CC>), dim3((unsigned)rows), dim3(256), lds, st, a, b, align, sy, sx, wy, wx)
#define PS_BIL_ROWS_C(DT)                                                                                   \
  switch (std::min<int>(BWD_MAXC, dsrc->c)) {                                                              \
    case 1: PS_BIL_ROWS(DT, 1); break;                                                                     \
    case 2: PS_BIL_ROWS(DT, 2); break;                                                                     \
    case 3: PS_BIL_ROWS(DT, 3); break;                                                                     \
    case 4: PS_BIL_ROWS(DT, 4); break;                                                                     \
    case 5: PS_BIL_ROWS(DT, 5); break;                                                                     \
    case 6: PS_BIL_ROWS(DT, 6); break;                                                                     \
    case 7: PS_BIL_ROWS(DT, 7); break;                                                                     \
    default: PS_BIL_ROWS(DT, 8); break;                                                                    \
  }
    if (ddst->dtype == PS_F32) { PS_BIL_ROWS_C(PS_F32) }
    else if (ddst->dtype == PS_BF16) { PS_BIL_ROWS_C(PS_BF16) }
    else { PS_BIL_ROWS_C(PS_F16) }
#undef PS_BIL_ROWS_C
#undef PS_BIL_ROWS
    PS_CHECK_LAUNCH("bilinear_bwd_rows");
    return PS_OK;
  }
  const long long waves = (long long)dsrc->n * dsrc->h * dsrc->w;
  PS_REQUIRE((waves + 3) / 4 < (1LL << 31), "bilinear_bwd: too many source pixels");
  hipLaunchKernelGGL(bilinear_bwd_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), to_t4(ddst),
                     to_t4(dsrc), align, sy, sx, wy, wx);
  PS_CHECK_LAUNCH("bilinear_bwd");
  return PS_OK;
}

extern "C" int64_t ps_ce_workspace_floats(void) { return CE_BLOCKS; }

extern "C" int ps_softmax_ce(const float* logits, const int64_t* target, float* loss_out, float* dlogits, float grad_scale, int32_t n,
                             int32_t c, int32_t h, int32_t w, int32_t ignore_index, float* partials, void* stream) {
  PS_REQUIRE(logits && target && loss_out && partials, "softmax_ce: null argument");
  PS_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "softmax_ce: empty input");
  const long long hw = (long long)h * w, total = (long long)n * hw;
  const int grid = grid_for(total, 256, CE_BLOCKS);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int fast = total < (1LL << 31) ? 1 : 0;
  const FastDiv div_hw = make_fastdiv((uint32_t)std::min<long long>(hw, 0x7fffffff));
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, s, logits, (const long long*)target, dlogits, partials, grad_scale / (float)total, n, c, hw,
                       ignore_index < 0 ? -1 : ignore_index, fast, div_hw);
  };
  switch (c) {
    case 2: launch(ce_kernel<2>); break;
    case 3: launch(ce_kernel<3>); break;
    case 4: launch(ce_kernel<4>); break;
    case 5: launch(ce_kernel<5>); break;
    default: launch(ce_kernel<0>); break;
  }
  PS_CHECK_LAUNCH("softmax_ce");
  hipLaunchKernelGGL(ce_finish_kernel, dim3(1), dim3(256), 0, s, partials, grid, 1.f / (float)total, loss_out);
  PS_CHECK_LAUNCH("softmax_ce_finish");
  return PS_OK;
}

extern "C" int64_t ps_dice_workspace_floats(void) { return DICE_BLOCKS * 3 * DICE_MAXC + 3 * DICE_MAXC; }

extern "C" int ps_dice_loss(const float* logits, const int64_t* target, float* loss_out, float* dlogits, float grad_scale, int32_t n,
                            int32_t c, int32_t h, int32_t w, int32_t ignore_index, float* workspace, void* stream) {
  PS_REQUIRE(logits && target && loss_out && workspace, "dice_loss: null argument");
  PS_REQUIRE(n > 0 && c > 0 && c <= DICE_MAXC && h > 0 && w > 0, "dice_loss: bad shape (C <= %d)", DICE_MAXC);
  const long long hw = (long long)h * w, total = (long long)n * hw;
  const int grid = grid_for(total, 256 * 4, DICE_BLOCKS);
  float* sums = workspace + (long long)DICE_BLOCKS * 3 * DICE_MAXC;
  const int ign = ignore_index < 0 ? -1 : ignore_index;
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(dice_sums_kernel, dim3(grid), dim3(256), 0, s, logits, (const long long*)target, workspace, n, c, hw, ign);
  PS_CHECK_LAUNCH("dice_sums");
  hipLaunchKernelGGL(dice_finish_kernel, dim3(1), dim3(64), 0, s, workspace, grid, c, sums, loss_out);
  PS_CHECK_LAUNCH("dice_finish");
  if (dlogits) {
    hipLaunchKernelGGL(dice_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, logits, (const long long*)target, sums, dlogits, grad_scale, n, c,
                       hw, ign);
    PS_CHECK_LAUNCH("dice_bwd");
  }
  return PS_OK;
}

extern "C" int ps_argmax_mask(const float* x, const float* label, const uint8_t* tissue, uint8_t* mask_out, float* entropy_out,
                              int32_t mode, int32_t softmax_first, int32_t first_ch, int32_t n, int32_t c, int32_t h, int32_t w,
                              void* stream) {
  PS_REQUIRE(x && mask_out, "argmax_mask: null argument");
  PS_REQUIRE(mode >= PS_MASK_PLAIN && mode <= PS_MASK_FILL, "argmax_mask: bad mode %d", mode);
  PS_REQUIRE(mode == PS_MASK_PLAIN || label, "argmax_mask: mode %d needs label", mode);
  PS_REQUIRE(n > 0 && c > 0 && c < 255 && h > 0 && w > 0 && first_ch >= 0 && first_ch < c, "argmax_mask: bad shape");
  const long long hw = (long long)h * w;
  const int fast = (long long)n * hw < (1LL << 31) ? 1 : 0;
  const FastDiv div_hw = make_fastdiv((uint32_t)std::min<long long>(hw, 0x7fffffff));  // only used when `fast` (then hw < 2^31), as in ps_softmax_ce
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(grid_for((long long)n * hw, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, label, tissue, mask_out,
                       entropy_out, mode, softmax_first, mode == PS_MASK_FILL ? 0 : first_ch, n, c, hw, fast, div_hw);
  };
  switch (c) {
    case 2: launch(argmax_mask_kernel<2>); break;
    case 3: launch(argmax_mask_kernel<3>); break;
    case 4: launch(argmax_mask_kernel<4>); break;
    case 5: launch(argmax_mask_kernel<5>); break;
    default: launch(argmax_mask_kernel<0>); break;
  }
  PS_CHECK_LAUNCH("argmax_mask");
  return PS_OK;
}

extern "C" int ps_iou_from_confusion(const int64_t* cm, int32_t num_class, double* out, void* stream) {
  PS_REQUIRE(cm && out, "iou_from_confusion: null argument");
  PS_REQUIRE(num_class >= 1 && num_class <= 16, "iou_from_confusion: num_class %d unsupported (1..16)", num_class);
  hipLaunchKernelGGL(iou_from_confusion_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (const long long*)cm, num_class, out);
  PS_CHECK_LAUNCH("iou_from_confusion");
  return PS_OK;
}

extern "C" int ps_confusion_accum(const uint8_t* pred, const int64_t* gt, int64_t* cm, int64_t npix, int32_t num_class, void* stream) {
  PS_REQUIRE(pred && gt && cm && npix >= 0, "confusion_accum: null argument");
  PS_REQUIRE(num_class >= 1 && num_class <= 16, "confusion_accum: num_class %d unsupported (1..16)", num_class);
  if (npix == 0) return PS_OK;
  hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(npix, 256 * 16, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), pred,
                     (const long long*)gt, (unsigned long long*)cm, (long long)npix, num_class);
  PS_CHECK_LAUNCH("confusion_accum");
  return PS_OK;
}
