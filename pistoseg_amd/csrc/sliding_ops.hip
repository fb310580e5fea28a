// Sliding-window evaluation on the device:
//   * per-tile softmax -> scatter-add into per-image f64 probability canvases + coverage counts
//     (models/segmentation_module.py:141-161, segmentation_test.py:141-183, OEEM/classification/prepare_seg_inputs.py:121-131);
//   * canvas / count -> bilinear resize (align_corners = False, f64 like the reference's numpy / torch-double arithmetic)
//     -> accumulate into the full-resolution canvas (segmentation_module.py:166-178, prepare_seg_inputs.py:133-138);
//   * canvas -> argmax mask (loss.py:55-57 with probs = True; NaN = maximum, as torch.argmax);
//   * the d4 test-time-augmentation view transforms and their inverse + merge (ttach d4_transform: horizontal flip x rot90,
//     infer_pseudo_masks.py:96, mosaic_module.py:76).
// All HBM-bound, one thread per pixel, coalesced along the fastest dimension of the side that is written.
#include <algorithm>

#include "ps_internal.h"

namespace {

constexpr int SW_MAXC = 8;

// ------------------------------------------------------------------------------------------------
// tile scores [N, C, H, W] f32 -> (softmax over C) -> canvas[y0 + y, x0 + x, :] += p ; count[y0 + y, x0 + x] += 1 for
// y < vh, x < vw.  Overlapping tiles of one launch meet in f64 atomics (summation order of overlaps is not fixed: the sums
// agree with the reference's sequential f64 adds to ~1 ulp of f64).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_scatter_kernel(const float* __restrict__ scores, const ps_tile_dst* __restrict__ tiles,
                                                              int c, int h, int w, int apply_softmax) {
  const int j = blockIdx.y;
  const ps_tile_dst t = tiles[j];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= h * w) return;
  const int y = pix / w, x = pix - y * w;
  if (y >= t.vh || x >= t.vw) return;
  const float* src = scores + (long long)j * c * h * w + pix;
  float v[SW_MAXC];
#pragma unroll
  for (int k = 0; k < SW_MAXC; ++k) v[k] = k < c ? src[(long long)k * h * w] : 0.f;
  if (apply_softmax) {
    float m = v[0];
#pragma unroll
    for (int k = 1; k < SW_MAXC; ++k)
      if (k < c) m = fmaxf(m, v[k]);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < SW_MAXC; ++k)
      if (k < c) {
        v[k] = expf(v[k] - m);
        s += v[k];
      }
#pragma unroll
    for (int k = 0; k < SW_MAXC; ++k)
      if (k < c) v[k] = v[k] / s;
  }
  const long long cy = t.y0 + y, cx = t.x0 + x;
  const long long cpix = cy * t.canvas_w + cx;
#pragma unroll
  for (int k = 0; k < SW_MAXC; ++k)
    if (k < c) {
      double* dst = t.channels_last ? t.canvas + cpix * c + k : t.canvas + (long long)k * t.canvas_h * t.canvas_w + cpix;
      atomicAdd(dst, (double)v[k]);
    }
  atomicAdd(t.count + cpix, 1.0);
}

// value of canvas channel k at (y, x): sum / count, with the reference's treatment of uncovered pixels
__device__ __forceinline__ double canvas_value(const double* __restrict__ src, const double* __restrict__ cnt, double div, long long p,
                                               long long plane, int c, int k, int channels_last, int zero_uncovered) {
  const double s = channels_last ? src[p * c + k] : src[(long long)k * plane + p];
  double d = cnt ? cnt[p] : div;
  if (zero_uncovered && d < 1.0) d = 1.0;  // sum_counter[sum_counter < 1] = 1 (prepare_seg_inputs.py:131)
  return s / d;                            // else 0 / 0 = NaN as in numpy (segmentation_module.py:166)
}

// torch's area_pixel_compute_source_index + guard_index_and_lambda in double (aten/src/ATen/native/UpSample.h)
__device__ __forceinline__ void src_index(double scale, int dst, int in_size, int& i0, int& i1, double& l0, double& l1) {
  double r = scale * (dst + 0.5) - 0.5;
  if (r < 0.0) r = 0.0;
  i0 = min((int)r, in_size - 1);
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = fmin(fmax(r - i0, 0.0), 1.0);
  l0 = 1.0 - l1;
}

__global__ __launch_bounds__(256) void canvas_resize_accum_kernel(const double* __restrict__ src, const double* __restrict__ cnt, double div,
                                                                  int hs, int ws, double* __restrict__ dst, double* __restrict__ dcnt,
                                                                  int hd, int wd, int c, int channels_last, int zero_uncovered,
                                                                  int accumulate) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= (long long)hd * wd) return;
  const int y = (int)(p / wd), x = (int)(p - (long long)y * wd);
  int y0, y1, x0, x1;
  double ly0, ly1, lx0, lx1;
  src_index((double)hs / hd, y, hs, y0, y1, ly0, ly1);
  src_index((double)ws / wd, x, ws, x0, x1, lx0, lx1);
  const long long plane_s = (long long)hs * ws, plane_d = (long long)hd * wd;
  const long long p00 = (long long)y0 * ws + x0, p01 = (long long)y0 * ws + x1, p10 = (long long)y1 * ws + x0, p11 = (long long)y1 * ws + x1;
#pragma unroll
  for (int k = 0; k < SW_MAXC; ++k) {
    if (k >= c) break;
    const double v00 = canvas_value(src, cnt, div, p00, plane_s, c, k, channels_last, zero_uncovered);
    const double v01 = canvas_value(src, cnt, div, p01, plane_s, c, k, channels_last, zero_uncovered);
    const double v10 = canvas_value(src, cnt, div, p10, plane_s, c, k, channels_last, zero_uncovered);
    const double v11 = canvas_value(src, cnt, div, p11, plane_s, c, k, channels_last, zero_uncovered);
    // (v00*wx0 + v01*wx1)*wy0 + (v10*wx0 + v11*wx1)*wy1: torch's Interpolate<2>::eval association, no fused multiply-adds
    const double top = __dadd_rn(__dmul_rn(v00, lx0), __dmul_rn(v01, lx1));
    const double bot = __dadd_rn(__dmul_rn(v10, lx0), __dmul_rn(v11, lx1));
    // same-size "resize": torch copies the input (upsample_bilinear2d's shortcut), so a NaN neighbour with weight 0 does not leak
    const double val = (hs == hd && ws == wd) ? v00 : __dadd_rn(__dmul_rn(top, ly0), __dmul_rn(bot, ly1));
    double* d = channels_last ? dst + p * c + k : dst + (long long)k * plane_d + p;
    *d = accumulate ? *d + val : val;
  }
  if (dcnt) dcnt[p] = accumulate ? dcnt[p] + 1.0 : 1.0;
}

// argmax over channels of canvas / count (first maximum; NaN counts as the maximum, so an uncovered pixel -> 0);
// optional overwrite where gt == bg_value (segmentation_test.py:201 `mask_pred[mask == 3] = 3`)
__global__ __launch_bounds__(256) void canvas_argmax_kernel(const double* __restrict__ src, const double* __restrict__ cnt, long long npix, int c,
                                                            int channels_last, const uint8_t* __restrict__ gt, int bg_value,
                                                            uint8_t* __restrict__ pred) {
  const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  int best = 0;
  double bv = canvas_value(src, cnt, 1.0, p, npix, c, 0, channels_last, 0);
#pragma unroll
  for (int k = 1; k < SW_MAXC; ++k) {
    if (k >= c) break;
    const double v = canvas_value(src, cnt, 1.0, p, npix, c, k, channels_last, 0);
    if (!(bv != bv) && (v > bv || v != v)) {  // keep the first NaN; otherwise strictly greater wins (first maximum)
      bv = v;
      best = k;
    }
  }
  if (gt && bg_value >= 0 && gt[p] == bg_value) best = bg_value;
  pred[p] = (uint8_t)best;
}

// ------------------------------------------------------------------------------------------------
// d4 views.  A view is (hflip, k): x -> flip(x, W) if hflip -> rot90(x, k, dims (H, W)) (ttach HorizontalFlip then Rotate90);
// its inverse on the model output is rot90(., 4 - k) then flip.  Square tiles.
// forward:  dst[n, c, :, :] = view(src[n, c])
// inverse:  dst[n, c, :, :] (+)= unview(src[n, c])
// torch.rot90(x, 1, (2, 3)) = x.flip(3).transpose(2, 3):  out[i][j] = in[j][S - 1 - i].
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void rot_src(int k, int s, int i, int j, int& si, int& sj) {  // out[i][j] = in[si][sj] for rot90(in, k)
  switch (k & 3) {
    case 0: si = i; sj = j; break;
    case 1: si = j; sj = s - 1 - i; break;
    case 2: si = s - 1 - i; sj = s - 1 - j; break;
    default: si = s - 1 - j; sj = i; break;
  }
}

// One 32 x 32 LDS tile per block (grid: tiles x planes): both the reads and the writes run along rows, whatever the view.  (One
// thread per output element reading its source directly reads a COLUMN per wave for the odd rotations -- 64 lines per load
// instruction, 43 us for a 64-tile batch's 38.5 MB -- and needs a 64-bit division per element.)
__device__ __forceinline__ void d4_src(int s, int hflip, int k, int inverse, int i, int j, int& si, int& sj) {
  if (!inverse) {
    rot_src(k, s, i, j, si, sj);
    if (hflip) sj = s - 1 - sj;
  } else {
    const int fj = hflip ? s - 1 - j : j;
    rot_src(4 - k, s, i, fj, si, sj);
  }
}
__global__ __launch_bounds__(256) void d4_view_tiled_kernel(const float* __restrict__ src, float* __restrict__ dst, long long planes, int s, int hflip,
                                                            int k, int inverse, int accumulate, int tiles_per_side) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int i0 = (int)(blockIdx.x / (unsigned)tiles_per_side) * 32, j0 = (int)(blockIdx.x % (unsigned)tiles_per_side) * 32;
  for (long long pl = blockIdx.y; pl < planes; pl += gridDim.y) {
    const float* sp = src + pl * s * s;
    float* dp = dst + pl * s * s;
    if (!(k & 1)) {  // no transposition: an output row is a (possibly reversed) source row -- straight through registers
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 8 * r, j = j0 + tx;
        if (i < s && j < s) {
          int si, sj;
          d4_src(s, hflip, k, inverse, i, j, si, sj);
          const float v = sp[(long long)si * s + sj];
          float* d = dp + (long long)i * s + j;
          *d = accumulate ? *d + v : v;
        }
      }
      continue;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // element (i0 + tx, j0 + jj) of the output: consecutive tx walk along a source row
      const int jj = ty + 8 * r, i = i0 + tx, j = j0 + jj;
      if (i < s && j < s) {
        int si, sj;
        d4_src(s, hflip, k, inverse, i, j, si, sj);
        tile[jj][tx] = sp[(long long)si * s + sj];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = ty + 8 * r, i = i0 + ii, j = j0 + tx;
      if (i < s && j < s) {
        const float v = tile[tx][ii];
        float* d = dp + (long long)i * s + j;
        *d = accumulate ? *d + v : v;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long long n, float divisor) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = x[i] / divisor;
}

static inline unsigned blocks_for(long long items) { return (unsigned)((items + 255) / 256); }

}  // namespace

extern "C" int ps_softmax_scatter_accum(const float* scores, int32_t n, int32_t c, int32_t h, int32_t w, const ps_tile_dst* tiles_dev,
                                        int32_t apply_softmax, void* stream) {
  PS_REQUIRE(scores && tiles_dev, "softmax_scatter_accum: null argument");
  PS_REQUIRE(n > 0 && h > 0 && w > 0 && c >= 1 && c <= SW_MAXC, "softmax_scatter_accum: bad shape n=%d c=%d h=%d w=%d (c <= %d)", n, c, h, w, SW_MAXC);
  PS_REQUIRE(n <= 65535, "softmax_scatter_accum: at most 65535 tiles per launch");
  hipLaunchKernelGGL(softmax_scatter_kernel, dim3(blocks_for((long long)h * w), n), dim3(256), 0, static_cast<hipStream_t>(stream), scores,
                     tiles_dev, c, h, w, apply_softmax);
  PS_CHECK_LAUNCH("softmax_scatter_accum");
  return PS_OK;
}

extern "C" int ps_canvas_resize_accum(const double* src, const double* src_count, double src_div, int32_t hs, int32_t ws, double* dst,
                                      double* dst_count, int32_t hd, int32_t wd, int32_t c, int32_t channels_last, int32_t zero_uncovered,
                                      int32_t accumulate, void* stream) {
  PS_REQUIRE(src && dst, "canvas_resize_accum: null argument");
  PS_REQUIRE(hs > 0 && ws > 0 && hd > 0 && wd > 0 && c >= 1 && c <= SW_MAXC, "canvas_resize_accum: bad shape");
  hipLaunchKernelGGL(canvas_resize_accum_kernel, dim3(blocks_for((long long)hd * wd)), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     src_count, src_div, hs, ws, dst, dst_count, hd, wd, c, channels_last, zero_uncovered, accumulate);
  PS_CHECK_LAUNCH("canvas_resize_accum");
  return PS_OK;
}

extern "C" int ps_canvas_argmax(const double* canvas, const double* count, int32_t h, int32_t w, int32_t c, int32_t channels_last,
                                const uint8_t* gt, int32_t bg_value, uint8_t* pred, void* stream) {
  PS_REQUIRE(canvas && pred, "canvas_argmax: null argument");
  PS_REQUIRE(h > 0 && w > 0 && c >= 1 && c <= SW_MAXC, "canvas_argmax: bad shape");
  const long long npix = (long long)h * w;
  hipLaunchKernelGGL(canvas_argmax_kernel, dim3(blocks_for(npix)), dim3(256), 0, static_cast<hipStream_t>(stream), canvas, count, npix, c,
                     channels_last, gt, bg_value, pred);
  PS_CHECK_LAUNCH("canvas_argmax");
  return PS_OK;
}

extern "C" int ps_d4_view(const float* src, float* dst, int64_t planes, int32_t side, int32_t hflip, int32_t k, int32_t inverse,
                          int32_t accumulate, void* stream) {
  PS_REQUIRE(src && dst && src != dst, "d4_view: null or aliased argument");
  PS_REQUIRE(planes > 0 && side > 0 && k >= 0 && k < 4, "d4_view: bad argument");
  // every view goes through the LDS-tiled kernel (16 us per 64-tile batch for the transposing views against 43 us read column-wise, and
  // 22 us for the direct kernel's row-to-row views with their per-element 64-bit division)
  const int tps = (side + 31) / 32;
  hipLaunchKernelGGL(d4_view_tiled_kernel, dim3((unsigned)(tps * tps), (unsigned)std::min<long long>(planes, 65535)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, (long long)planes, side, hflip, k, inverse, accumulate, tps);
  PS_CHECK_LAUNCH("d4_view");
  return PS_OK;
}

extern "C" int ps_scale_inplace(float* x, int64_t n, float divisor, void* stream) {
  PS_REQUIRE(x && n >= 0 && divisor != 0.f, "scale_inplace: bad argument");
  if (n == 0) return PS_OK;
  hipLaunchKernelGGL(scale_kernel, dim3(blocks_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), x, (long long)n, divisor);
  PS_CHECK_LAUNCH("scale_inplace");
  return PS_OK;
}
