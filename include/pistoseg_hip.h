/*
 * pistoseg_hip.h -- C-ABI of libpistoseg_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the
 * PistoSeg segmentation hot path.
 *
 * The reference (Vison307/PistoSeg) has no FFI of its own: every arithmetic step on its hot path is a
 * stock torch op call site.  Each entry point below replaces one of those call sites;
 * the citation after "replaces:" is the reference file:line (relative to the reference root).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  All pointers are DEVICE pointers unless a
 *    parameter says "host".  No ownership transfer, no allocation, no synchronisation inside a call:
 *    every function enqueues work on `stream` (a hipStream_t passed as void*) and returns.
 *  - return value: 0 on success, negative ps_status on error; ps_last_error() gives a message
 *    (thread-local).
 *  - dtype: PS_F32 = exact-f32 MFMA path (parity), PS_BF16 / PS_F16 = 16-bit storage / f32 accumulate (throughput;
 *    PS_F16 training needs loss scaling: see ps_softmax_ce grad_scale and ps_adamw_step grad_inv_scale).
 *  - activation layout: channels-last  [N, H, W, C]  with an explicit channel stride `ldc` (elements per
 *    pixel), so a tensor may be a channel slice of a wider buffer.  Per-channel vectors (BN scale/shift)
 *    and all loss/optimizer state are f32.  Network inputs/outputs at the API edge are the reference's
 *    NCHW f32 (conv1a reads NCHW, the upsample writes NCHW).
 *  - conv weight layouts:  W_fwd[cout][kh][kw][cin]  (== an OIHW tensor in torch.channels_last memory
 *    format) and  W_dgrad[cin][kh][kw][cout]  (made by ps_weight_transpose).
 */
#ifndef PISTOSEG_HIP_H
#define PISTOSEG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_VERSION 226 /* major*10000 + minor*100 + patch */

typedef enum ps_status {
  PS_OK = 0,
  PS_ERR_ARG = -1,     /* bad shape / unsupported geometry / misaligned pointer */
  PS_ERR_LAUNCH = -2,  /* hipLaunchKernel or other HIP runtime failure */
  PS_ERR_NOGPU = -3    /* no gfx950 device visible */
} ps_status;

typedef enum ps_dtype {
  PS_F32 = 0, PS_BF16 = 1, PS_F16 = 2,
  /* Split bf16 ("bf16x3") -- the convolutions only.  A value is hi + lo with hi = bf16(v), lo = bf16(v - hi) (16 significant bits).  A tensor of C
   * logical channels (C % 32 == 0) stores 2 C bf16 channels per pixel in blocks of 32 logical channels, [hi(32) | lo(32)] = one 128-byte K-line;
   * channel strides count stored elements (>= 2 C).  Weights W_fwd[cout][tap][2 cin], W_dgrad[cin][tap][2 cout] use the same layout along K
   * (ps_convert_rows).  The forward / data-gradient kernels stage such tensors like plain bf16 ones and multiply every staged K-line three times
   * (x_hi w_hi + x_hi w_lo + x_lo w_hi, f32 accumulation): relative error ~2^-16 per product instead of bf16's 2^-8, at 3 MFMAs per product and the
   * staging of 2.  Epilogue tensors (add0, out_raw, mask_src, add1, out) are split tensors.  The weight gradient: see ps_conv_geom.wgrad_terms.
   * The path that meets the reference's fp32 results to 1e-4 (models/resnet38d.py:156-188 computes in fp32) at a fraction of the exact-f32 MFMA's cost. */
  PS_BF16X3 = 3,
  /* The same layout and products with fp16 halves: hi = fp16(v) (11 significant bits), lo = fp16(v - hi) -- 22 bits while lo stays a normal number
   * (|v| >= 2^-3), an absolute error of at most 2^-25 below that (the 16-bit MFMA keeps fp16 subnormals: tools/f16_denorm_probe.hip), against
   * the bf16 split's 2^-16 relative.  Range is fp16's: |v| <= 65504, and gradients need the loss scale of the PS_F16 path. */
  PS_F16X3 = 4
} ps_dtype;

int ps_version(void);
const char* ps_last_error(void);
/* Number of visible HIP devices (0 if none); never initialises a context beyond hipGetDeviceCount. */
int ps_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on MFMA).  replaces: F.conv2d behind every nn.Conv2d of
 * models/resnet38d.py:16-24,60-71,123-146 and models/revise_net.py:13-19 (k in {1,3}, stride in {1,2},
 * dilation in {1,2,4}, padding == dilation for 3x3 and 0 for 1x1, bias-free), plus the eval-mode
 * BatchNorm+ReLU (+Dropout2d) that follows it (resnet38d.py:28-29,39-40,75-90,186) and the residual
 * add (resnet38d.py:43,93), fused into the epilogue.
 * ---------------------------------------------------------------------------------------------- */
typedef struct ps_conv_geom {
  int32_t dtype;          /* ps_dtype of activations and weights */
  int32_t n, h, w;        /* the conv's INPUT activation [n, h, w, cin] (forward sense) */
  int32_t cin, cout;      /* cin*esize and cout*esize must be multiples of 128 bytes; see ps_conv_supported */
  int32_t ksize;          /* 1 or 3 */
  int32_t stride;         /* 1 or 2 */
  int32_t dilation;       /* >= 1 */
  int32_t ldc_x;          /* channel stride of the forward input activation (elements) */
  int32_t ldc_y;          /* channel stride of the forward output activation / its gradient */
  int32_t tiles_per_block; /* launch option of the persistent kernels: 0 = one block per CU lives for the whole launch (best when the
                            * GPU runs nothing else); n > 0 = blocks are dispatched in batches and take n work items each, so the
                            * hardware re-balances when another kernel (an RCCL all-reduce overlapping the backward) holds some CUs.
                            * Results are bit-identical for every value. */
  int32_t gpu_shared;     /* launch option: 1 = another stream of this process keeps the GPU busy beside this launch (the weight gradients
                            * of the two-stream backward): the partial last round of a launch is then NOT re-issued as a second launch of
                            * smaller tiles -- the co-running kernel's blocks fill those CUs, and the split costs 2 % of a training step
                            * (profiles/r03_tail_split_two_streams.txt).  0 = the launch has the GPU to itself.  Bit-identical results. */
  int32_t cus_reserved;   /* launch option of the persistent kernels: size the grid (and the static tile schedule) for `#CUs - cus_reserved` compute
                            * units, leaving the rest to a co-running kernel that HOLDS its CUs for as long as it lives -- an RCCL collective beside
                            * the backward.  A persistent block that cannot become resident serialises its whole static share behind the others
                            * (+33 % on a training step beside a 16-CU kernel); with the reservation every block starts at once and the launch costs
                            * the ideal #CUs / (#CUs - reserved).  0 = all CUs.  Bit-identical results for every value. */
  int32_t wgrad_terms;    /* ps_conv2d_wgrad[_det] on the split types only: 0 / 1 = dW from the hi halves (x_hi dy_hi: one launch of the 16-bit kernel);
                            * 3 = + x_hi dy_lo + x_lo dy_hi (three launches).  The lo terms are 2^-8 (bf16) / 2^-11 (fp16) of a product and add up
                            * incoherently over the pixels: 1.8e-4 of a flip-free step's gradient in fp16x3, nothing measurable at training sizes. */
  int32_t tile_queue;     /* launch option of the persistent kernels: 1 = a block's first work item is static, every further one is drawn from
                            * per-XCD ticket counters one item ahead (ps_internal.h: ps_q_*), so blocks that start late or share their CU with
                            * another kernel take fewer items instead of delaying the launch.  Replaces tiles_per_block / cus_reserved where the
                            * co-running kernel's footprint is not known in advance.  0 = the static schedule.  Bit-identical results. */
} ps_conv_geom;

/* Epilogue applied to the f32 accumulator `acc` of every produced element (pixel m, channel c):
 *   v = acc + (add0 ? add0[m,c] : 0)
 *   if (out_raw) out_raw[m,c] = v
 *   mode PS_EPI_BNRELU : out[m,c] = max(v*scale[c] + shift[c], 0) * (drop ? drop[n(m),c] : 1)
 *   mode PS_EPI_RELUBWD: out[m,c] = (mask_src[m,c] > 0 ? v*scale[c]*(drop ? drop[n(m),c] : 1) : 0)
 *                                   + (add1 ? add1[m,c] : 0)
 * scale/shift may be NULL (1 / 0).  All activation-typed pointers use the geom's dtype. */
enum { PS_EPI_NONE = 0, PS_EPI_BNRELU = 1, PS_EPI_RELUBWD = 2 };
typedef struct ps_epilogue {
  const void* add0;     int32_t ldc_add0;  int32_t _pad0;
  void*       out_raw;  int32_t ldc_raw;   int32_t mode;
  const float* scale;
  const float* shift;
  const float* drop;                        /* [n, C] f32 dropout multipliers (>= 0: 0 or 1/(1-p)), or NULL.  Where all rows of a wave lie in
                                             * one image they are folded into scale/shift: max(v*(scale*drop) + shift*drop, 0) */
  const void* mask_src; int32_t ldc_mask;  int32_t _pad1;
  const void* add1;     int32_t ldc_add1;  int32_t _pad2;
  void*       out;      int32_t ldc_out;   int32_t _pad3;
  void*       out_hi;   int32_t ldc_hi;    int32_t _pad4;  /* split types only (NULL otherwise): `out`'s hi halves ALSO as a plain 16-bit tensor
                                             * [M, >= C] (bf16 for PS_BF16X3, fp16 for PS_F16X3): contiguous rows for the weight gradient, which
                                             * contracts over pixels (ps_conv2d_wgrad on the plain type with x = such a copy: the x_hi dy_hi term
                                             * at the plain kernel's speed instead of a gather of every other 64 bytes) */
  void*       sk_ws;    int64_t sk_ws_bytes;               /* optional scratch for the STREAM-K finish of a launch's partial last round (3x3 stride-1 layers
                                             * on the halo kernel): with T tiles on #CUs compute units the last T mod #CUs tiles are cut along K into equal
                                             * shares, one per CU; tiles that span CUs are added up through f32 slabs here, in a fixed order (bit-identical from
                                             * run to run, same MFMA chain per element up to the split points).  >= ps_conv_sk_workspace_bytes(g, dgrad)
                                             * bytes, 256-byte aligned, ZEROED ONCE by the caller when allocated (its first 4 KiB are arrival counters that
                                             * every launch leaves zeroed) and used by ONE stream at a time.  The split points depend on the CU count the launch is sized for, and
                                             * the other schedules (tile_queue, tiles_per_block) do not split: results with sk_ws agree with theirs up to f32 re-association; WITHOUT it
                                             * every launch option is bit-identical.  NULL / too small: the launch keeps the static
                                             * schedule (half-tile tail launch or an idle partial round). */
} ps_epilogue;

/* 1 if the MFMA implicit-GEMM path handles this geometry, else 0 (message in ps_last_error). */
int ps_conv_supported(const ps_conv_geom* g);
/* Which implicit-GEMM instantiation ps_conv2d_fwd (dgrad = 0) / ps_conv2d_dgrad (dgrad = 1) will launch for this geometry
 * (profiling labels; -1 if the geometry is unsupported). */
enum {
  PS_CONV_4WAVE = 1,    /* conv_igemm_kernel: 4 waves stage and compute (small / narrow problems) */
  PS_CONV_WS_128 = 2,   /* conv_igemm_ws_kernel<128>: 4 loader + 4 consumer waves, 128x128 tile, two blocks per CU */
  PS_CONV_WS_112 = 3,   /* conv_igemm_ws_kernel<112> */
  PS_CONV_WS2_256 = 4,  /* conv_igemm_ws2_kernel<256>: 256x128 tile, one block per CU, 3-stage ring */
  PS_CONV_WS2_224 = 5,  /* conv_igemm_ws2_kernel<224> */
  PS_CONV_OTHER = 6,    /* an experimental kernel forced through the testing hooks */
  PS_CONV_GEMM256 = 8,  /* conv_gemm256_kernel: plain GEMMs (1x1 stride-1, 16-bit), 256x256 tile, all eight waves compute, two wave groups alternate load / MFMA phases */
  PS_CONV_HALO = 7      /* conv_igemm_halo_kernel: 3x3 stride-1, width % 28 == 0 (224x128 tile of 8 rows x 28 columns) or % 32 == 0 (256x128, 8 x 32), pixel window + halo staged once per tap row */
};
int ps_conv_variant(const ps_conv_geom* g, int32_t dgrad);
/* Bytes of ps_epilogue.sk_ws with which ps_conv2d_fwd (dgrad = 0) / ps_conv2d_dgrad (dgrad = 1) finishes this geometry's partial last round by
 * stream-K; 0 when the launch would not use it (no partial round, another kernel family, feature maps that are not a multiple of 28 wide, batch / queue launch options, ...). */
int64_t ps_conv_sk_workspace_bytes(const ps_conv_geom* g, int32_t dgrad);
/* 1 if ps_conv2d_wgrad will launch conv_wgrad_ws2_kernel (256x128 tile, persistent, wave-specialised), 0 for conv_wgrad_kernel, -1 unsupported
 * (2: conv_wgrad256_kernel, an experiment that only the debug library can select). */
int ps_conv_wgrad_variant(const ps_conv_geom* g);

/* Ticket counters of the `tile_queue` launch option: a 288 KiB ring per (device of the stream, stream), allocated and zeroed at the first queue-mode
 * launch of a stream (one hipMalloc: synchronises the device once, not allowed under stream capture) -- or ahead of time by ps_queue_prepare.
 * ps_queue_release frees all rings (no queue-mode launch may be in flight): after destroying streams, or after an aborted launch. */
int ps_queue_prepare(void* stream);
int ps_queue_release(void);

/* y = conv(x, W_fwd) with epilogue.  x: [n,h,w,cin]; produces [n,ho,wo,cout], ho = (h-1)/stride+1. */
int ps_conv2d_fwd(const ps_conv_geom* g, const void* x, const void* w_fwd, const ps_epilogue* epi, void* stream);

/* dx = conv_transpose(dy, W) with epilogue.  dy: [n,ho,wo,cout] (ldc_y); produces [n,h,w,cin].
 * replaces: autograd of F.conv2d w.r.t. its input (Lightning backward / l.backward(),
 * revise_pseudo_labels.py:300). */
int ps_conv2d_dgrad(const ps_conv_geom* g, const void* dy, const void* w_dgrad, const ps_epilogue* epi, void* stream);

/* dW_fwd[cout][kh][kw][cin] (f32) += sum over pixels dy[m,cout] * x[m@tap,cin].  ACCUMULATES with f32
 * atomics (split over pixel ranges): zero dw first.  replaces: autograd of F.conv2d w.r.t. its weight. */
int ps_conv2d_wgrad(const ps_conv_geom* g, const void* x, const void* dy, float* dw, void* stream);
/* The same result WITHOUT atomics -- bit-identical from run to run: every pixel range (split-K part) of a weight tile stores its f32
 * partial sums into its own slice of a caller workspace, and a second kernel adds the slices to dw in range order (one thread per
 * element: a fixed summation order).  dw is accumulated into, as above.  workspace: >= ps_conv2d_wgrad_det_workspace_bytes(g) bytes,
 * 16-byte aligned, contents need not be initialised and are clobbered; 0 bytes (NULL allowed) when the problem is not split -- every
 * element of dw then receives exactly one addition and the atomic kernel is already deterministic.
 * replaces: autograd of F.conv2d w.r.t. its weight under the reference's determinism switches --
 * torch.use_deterministic_algorithms(True) (revise_pseudo_labels.py:140-146), pl.Trainer(deterministic=True)
 * (segmentation_train.py:153-160). */
int64_t ps_conv2d_wgrad_det_workspace_bytes(const ps_conv_geom* g);
int ps_conv2d_wgrad_det(const ps_conv_geom* g, const void* x, const void* dy, float* dw, void* workspace, int64_t workspace_bytes,
                        void* stream);

/* dst[cin][taps][cout] = src[cout][taps][cin]; same 16-bit dtype on both sides, f32 -> f32, or f32 -> bf16/f16 (cast). */
int ps_weight_transpose(int32_t src_dtype, int32_t dst_dtype, const void* src, void* dst, int32_t cout, int32_t taps,
                        int32_t cin, void* stream);
/* The same for many tensors in ONE launch (the per-step refresh of the data-gradient layouts after an optimizer step).
 * dst_ld >= cout is the row stride of dst in elements: dst[(ci * taps + tap) * dst_ld + co] = src[(co * taps + tap) * cin + ci];
 * two items with the same dst_ld whose dst pointers are offset by the first's cout build the K-concatenated weights of two 1x1
 * convolutions side by side.  All items share the (src, dst) dtype pair. */
#define PS_WT_MAX_ITEMS 48 /* items per launch; longer lists are cut into several launches */
typedef struct ps_wt_item {
  const void* src;
  void* dst;
  int32_t cout, taps, cin, dst_ld;
} ps_wt_item;
int ps_weight_transpose_batched(int32_t src_dtype, int32_t dst_dtype, int32_t n_items, const ps_wt_item* items, void* stream);
/* Row-wise conversion between f32 and a storage format (exactly one side is PS_F32): rows of c logical channels (c % 8 == 0), pitches
 * ld_src / ld_dst in ELEMENTS of the respective side (channel slices of wider buffers are fine).  f32 -> PS_BF16 / PS_F16 (RNE cast),
 * f32 -> PS_BF16X3 / PS_F16X3 (c % 32 == 0; 2 c stored 16-bit channels per row in blocks [hi(32) | lo(32)], hi = round16(v), lo = round16(v - hi);
 * activations and weights alike, `pattern` is ignored and kept for ABI stability) and back (16-bit -> f32 exactly; split -> hi + lo).
 * replaces: nothing in the reference (it computes in fp32 throughout, models/resnet38d.py:156-188); this is the edge between the split /
 * 16-bit conv stack and the f32 head, loss and RFM kernels (models/revise_net.py:50-75), and the weight layout maker of the split path. */
int ps_convert_rows(const void* src, int32_t src_fmt, int64_t ld_src, void* dst, int32_t dst_fmt, int64_t ld_dst, int64_t rows, int32_t c,
                    int32_t pattern, void* stream);
/* Strided row copy: dst[r][0..row_bytes) = src[r][0..row_bytes), r < rows (pitches in bytes; everything a multiple of 16).  Used to lay
 * two weight matrices side by side along K, so that a bottleneck unit's shortcut conv and its last 1x1 conv (and their data gradients)
 * run as ONE GEMM over concatenated channels (models/resnet38d.py:76-97: `branch1 + branch2`). */
int ps_copy_rows(const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes, int64_t rows, int64_t row_bytes, void* stream);
/* dst = (bf16)src, n elements.  (per-step refresh of the bf16 forward weights from the f32 master arena) */
int ps_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
/* dst = (dst_dtype)src for dst_dtype in {PS_BF16, PS_F16}. */
int ps_cast_f32_lowp(const float* src, void* dst, int32_t dst_dtype, int64_t n, void* stream);

/* conv1a: 3x3, 3 -> 64, stride 1, pad 1, reads the reference's NCHW f32 image, writes channels-last
 * activation out[m,c] = max(conv*scale[c]+shift[c], 0) (b2.bn_branch2a fused) and/or the raw conv.
 * replaces: models/resnet38d.py:123,161 (+ :28-29 of the first ResBlock).  w: f32 [64][3][3][3] (OIHW).
 * PS_F32 output: exact-f32 MFMA; PS_BF16 / PS_F16 output: image and weights are rounded to the storage type (f32 accumulate), like every
 * other convolution of the 16-bit paths. */
int ps_conv1a_fwd(int32_t out_dtype, const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                  void* out_act, void* out_raw, int32_t n, int32_t h, int32_t w, void* stream);

/* fc8 head (4096 -> C, C <= 8, 1x1, no bias) on dropout7(conv6).  replaces: models/revise_net.py:50.
 *   cam[m,c] = sum_k x[m,k] * (drop ? drop[n(m),k] : 1) * w[c,k]      cam: f32 [M, C] (pixel-major)   */
int ps_fc8_fwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, float* cam, int32_t m_total,
               int32_t pix_per_image, int32_t k, int32_t c, void* stream);
/* INFERENCE fusion of a wide 1x1 convolution, the eval-mode BN + ReLU behind it and the narrow head on its output:
 *   cam[m, c] = sum_ch T(max(conv(x, W)[m, ch] * scale[ch] + shift[ch], 0)) * w_head[c * cout + ch]        (T = the storage type's rounding)
 * without materialising the activated tensor.  replaces: b7's last 1x1 conv (the K-concatenated shortcut + branch2b2 of models/resnet38d.py:
 * 76-97) + `relu(bn7(.))` (:186) + `fc8` (models/revise_net.py:50) when dropout is off -- conv6 (411 MB at bs = 64) is otherwise written only to
 * be read back and reduced to C channels.  16-bit dtypes, 1x1 stride-1, cout % 256 == 0, cin >= 256, 1 <= classes <= 8; every (64-channel
 * slice, pixel, class) partial sum goes to `workspace` (ps_conv1x1_head_workspace_floats(g, classes) floats, 0 = geometry not served: use
 * ps_conv2d_fwd + ps_fc8_fwd) and a second kernel adds the slices in order (deterministic).  cam: f32 [M, classes] pixel-major. */
int64_t ps_conv1x1_head_workspace_floats(const ps_conv_geom* g, int32_t classes);
int ps_conv1x1_head_fwd(const ps_conv_geom* g, const void* x, const void* w_fwd, const float* scale, const float* shift, const float* w_head,
                        int32_t classes, float* workspace, int64_t workspace_floats, float* cam, void* stream);
/* General narrow 1x1 head: cam[m, c] (=|+=) sum_k x[m, k] * drop[n, k] * w[c * ldw + k] + bias[c]  (bias / drop may be NULL).
 * Lets a head over concatenated features run as one call per feature map without materialising the concat:
 * replaces `fc_cam(torch.cat([conv4, conv5, conv6], dim=1))` (OEEM/classification/network/wide_resnet.py:166-186). */
int ps_fc_head_fwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, int32_t ldw, const float* bias, const float* drop,
                   float* cam, int32_t accumulate, int32_t m_total, int32_t ppi, int32_t k, int32_t c, void* stream);
/* backward of ps_fc8_fwd fused with the ReLU(bn7) mask:  dx[m,k] = (x[m,k] > 0) * scale7[k] * drop * sum_c dcam[m,c] w[c,k]
 * (dx typed like x, written with stride ldc_dx), and dw[c,k] += sum_m dcam[m,c] * x[m,k]*drop (f32 atomics). */
int ps_fc8_bwd(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, const float* scale7,
               const float* dcam, void* dx, int32_t ldc_dx, float* dw, int32_t m_total, int32_t pix_per_image, int32_t k,
               int32_t c, void* stream);
/* The same with a caller-provided workspace of ps_fc8_bwd_workspace_floats() floats (16-byte aligned; contents need not be
 * initialised and are clobbered): the per-block partial sums of dw are stored there and added to dw by a second kernel instead
 * of going to dw with float atomics (9.6 M atomics on 12 K addresses at the 64 x 28 x 28 x 4096 training shape).
 * workspace == NULL behaves like ps_fc8_bwd. */
int64_t ps_fc8_bwd_workspace_floats(int32_t m_total, int32_t pix_per_image, int32_t k, int32_t c);
int ps_fc8_bwd_ws(int32_t dtype, const void* x, int32_t ldc_x, const float* w, const float* drop, const float* scale7,
                  const float* dcam, void* dx, int32_t ldc_dx, float* dw, int32_t m_total, int32_t pix_per_image, int32_t k,
                  int32_t c, float* workspace, int64_t workspace_floats, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-pixel kernels (HBM-bound)
 * ---------------------------------------------------------------------------------------------- */
/* Bilinear resize, F.interpolate(mode='bilinear') semantics.  replaces: models/revise_net.py:64,78-86,93,
 * revise_pseudo_labels.py:273-274 (align_corners=1) and infer_pseudo_masks.py:89-90 (align_corners=0).
 * src / dst are strided 4-d views (f32 or bf16 elements): both NCHW and channels-last views are
 * expressible; arithmetic is f32 with torch's index rules (area_pixel_compute_source_index). */
typedef struct ps_tensor4 {
  void* ptr;
  int32_t dtype;          /* ps_dtype of the elements */
  int32_t n, c, h, w;
  int32_t _pad;
  int64_t sn, sc, sh, sw; /* element strides */
} ps_tensor4;
int ps_bilinear_fwd(const ps_tensor4* src, const ps_tensor4* dst, int32_t align_corners, void* stream);
/* dsrc = d(bilinear)/d(src)^T * ddst, gather form (deterministic, no atomics); dsrc is overwritten. */
int ps_bilinear_bwd(const ps_tensor4* ddst, const ps_tensor4* dsrc, int32_t align_corners, void* stream);

/* Per-pixel softmax cross-entropy over NCHW f32 logits, int64 targets.
 * replaces: models/segmentation_module.py:63-66,101-102 -- CrossEntropyLoss(reduction='none', ignore_index)
 * followed by torch.mean over ALL n*h*w pixels (ignored pixels add 0 but count in the denominator).
 *   loss_out[0] = (1/(n*h*w)) * sum_pix [t != ignore] (logsumexp(z) - z_t)
 *   dlogits     = grad_scale/(n*h*w) * (softmax(z) - onehot(t)) for t != ignore, else 0   (if dlogits != NULL)
 * ignore_index < 0 disables ignoring.  partials: f32 workspace of >= ps_ce_workspace_floats() floats. */
int64_t ps_ce_workspace_floats(void);
int ps_softmax_ce(const float* logits, const int64_t* target, float* loss_out, float* dlogits, float grad_scale,
                  int32_t n, int32_t c, int32_t h, int32_t w, int32_t ignore_index, float* partials, void* stream);

/* Multiclass Dice loss over NCHW f32 logits (softmax inside), smp.losses.DiceLoss(mode='multiclass'[, ignore_index])
 * as called by models/mosaic_module.py:65-68,108.  THIRD-PARTY arithmetic (segmentation-models-pytorch 0.3.0,
 * not vendored): restated from its public definition, parity unpinned.  Per class over batch+space of the kept
 * pixels: 1 - 2*sum(p*t)/max(sum(p+t), 1e-7), zeroed for classes absent from the target, mean over classes.
 * workspace: >= ps_dice_workspace_floats() floats.  dlogits (optional) = grad_scale * d loss / d logits. */
int64_t ps_dice_workspace_floats(void);
int ps_dice_loss(const float* logits, const int64_t* target, float* loss_out, float* dlogits, float grad_scale, int32_t n,
                 int32_t c, int32_t h, int32_t w, int32_t ignore_index, float* workspace, void* stream);

/* CAM/logit -> mask reduction.  replaces: infer_revise_masks.py:137-143 (mode PS_MASK_MUL: argmax over
 * channels first_ch.. of x*label), infer_pseudo_masks.py:76-85 (mode PS_MASK_FILL: channels with label 0
 * are filled with -1e10, softmax, entropy = -sum p*log(p+1e-10), argmax of p; single-label tiles get the
 * constant mask and zero entropy; tissue==0 pixels get index C), loss.py:57-60 (mode PS_MASK_PLAIN:
 * argmax of softmax(x) / of x).  First maximum wins ties, NaN is treated as maximal (torch.argmax).
 * x: NCHW f32 [n,c,h,w]; label: f32 [n,c] or NULL; tissue: u8 [n,h,w] or NULL; mask_out: u8 [n,h,w];
 * entropy_out: f32 [n,h,w] or NULL. */
enum { PS_MASK_PLAIN = 0, PS_MASK_MUL = 1, PS_MASK_FILL = 2 };
int ps_argmax_mask(const float* x, const float* label, const uint8_t* tissue, uint8_t* mask_out, float* entropy_out,
                   int32_t mode, int32_t softmax_first, int32_t first_ch, int32_t n, int32_t c, int32_t h, int32_t w,
                   void* stream);

/* Confusion matrix accumulate, loss.py:16-26 as called (rows = ground truth, cols = prediction; ground
 * truth >= num_class dropped).  cm: int64 [num_class*num_class], accumulated with atomics. */
int ps_confusion_accum(const uint8_t* pred, const int64_t* gt, int64_t* cm, int64_t npix, int32_t num_class, void* stream);

/* loss.py:28-53 evaluated on the device, so that `mIoUMask.forward` (called once per training step, models/segmentation_module.py:108-109)
 * needs no device->host copy: out f64[2 + num_class] = { Mean_Intersection_over_Union, Frequency_Weighted_Intersection_over_Union,
 * Tissue_Intersection_over_Union[0..num_class) } of the int64 [num_class*num_class] matrix, numpy's f64 evaluation order (bit-identical to the
 * host computation on the copied matrix). */
int ps_iou_from_confusion(const int64_t* cm, int32_t num_class, double* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RFM head (stage 3/4, models/revise_net.py) and the feature-consistency losses
 * (revise_pseudo_labels.py:115-138,253-282).  f32 arithmetic.
 * ---------------------------------------------------------------------------------------------- */
/* Batched strided GEMM  C[b][m][n] = alpha * sum_k A[b][m*sam + k*sak] * B[b][k*sbk + n*sbn]  (element strides;
 * operand dtypes f32 or bf16).  replaces: torch.matmul(q.transpose(1,2), k) (revise_net.py:72) and its autograd. */
int ps_bgemm(int32_t a_dtype, int32_t b_dtype, int32_t c_dtype, const void* A, const void* B, void* C, int32_t batch, int32_t M,
             int32_t N, int32_t K, int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn, int64_t scb, int64_t scm,
             int64_t scn, float alpha, void* stream);
/* In-place softmax of `rows` contiguous rows of length `len`.  replaces: F.softmax(A, dim=1) (revise_net.py:73) -- the
 * affinity is kept TRANSPOSED (P[n][j][i] = A[n][i][j]) so the reference's dim=1 softmax runs along rows. */
int ps_softmax_rows(float* x, int64_t rows, int32_t len, void* stream);
/* R[n,j,cc] = sum_i P[n][j][i] * V[n,i,cc]  (cc <= 24).  replaces: torch.matmul(cam, A) in Net.RFM (revise_net.py:93-94)
 * for the three normalised maps at once (V = [cam_n | pmask_n | pcam_n], pixel-major). */
int ps_rfm_apply(const float* P, const float* V, float* R, int32_t n, int32_t np, int32_t cc, void* stream);
/* Backward of ps_rfm_apply + ps_softmax_rows w.r.t. the pre-softmax scores, in place over P:
 * dS[n][j][i] = P[n][j][i] * (sum_c dR[n,j,c]*V[n,i,c] - sum_c dR[n,j,c]*R[n,j,c]).  (V is no-grad: revise_net.py:32.) */
int ps_affinity_softmax_bwd(float* P_inout, const float* dR, const float* V, const float* R, int32_t n, int32_t np, int32_t cc,
                            void* stream);
/* CAM normalisation into an f32 destination with element strides (dn, dc, dp) over (sample, channel, pixel).
 * mode 0: get_norm_cam_d (revise_net.py:29-41); mode 1: max_norm(src)*label with channel 0 rebuilt as 1 - max fg
 * (revise_pseudo_labels.py:132-138,268-272; label [n,c] f32 or NULL).  2 <= C <= 8. */
int ps_norm_cam(const ps_tensor4* src, float* dst, int64_t dn, int64_t dc, int64_t dp, const float* label, int32_t mode, void* stream);

int64_t ps_loss_workspace_floats(void);
/* loss_rfm (revise_pseudo_labels.py:263-265): loss_out[0] (+)= mean over [n,1..c-1,h,w] of |a*label - b*label|; if da/db
 * are given they are ACCUMULATED with grad_scale * d loss.  a, b: NCHW f32; label: [n,c] f32. */
int ps_l1_masked(const float* a, const float* b, const float* label, float* da, float* db, float* loss_out, int32_t accumulate,
                 float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, float* partials, void* stream);
/* out[n,c,p] = | max_onehot(ref)[n,c,p] - rv[n,c,p]*label[n,c] |  (revise_pseudo_labels.py:125-130,275-276). */
int ps_ecr_tensor(const float* ref, const float* rv, const float* label, float* out, int32_t n, int32_t c, int32_t h, int32_t w,
                  void* stream);
/* d rv (+)= -sign(max_onehot(ref) - rv*label) * label * grad_scale on the elements selected by ps_topk_select
 * (thr/take from it; tie_counter: zeroed int32[n] scratch). */
int ps_ecr_bwd(const float* ref, const float* rv, const float* label, const float* t, const float* thr, const int32_t* take,
               int32_t* tie_counter, float* drv, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* The same with a DETERMINISTIC choice among the elements exactly equal to the threshold: the first take[n] of them in a fixed order
 * (pixel segments, 256-pixel chunks, channel, lane) instead of first come, first served -- torch.use_deterministic_algorithms(True) of
 * revise_pseudo_labels.py:140-146.  seg_counts: int32 scratch of >= ps_tie_workspace_ints(n, h, w) entries (contents need not be
 * initialised). */
int64_t ps_tie_workspace_ints(int32_t n, int32_t h, int32_t w);
int ps_ecr_bwd_det(const float* ref, const float* rv, const float* label, const float* t, const float* thr, const int32_t* take,
                   int32_t* seg_counts, int64_t seg_counts_ints, float* drv, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w,
                   void* stream);
/* Per-row top-k by radix select (k largest if largest != 0 else k smallest).  replaces: torch.topk(...)[0] followed by
 * mean/sum (revise_pseudo_labels.py:120-122,277-278).  thr[row] = k-th value, take[row] = number of elements equal to thr
 * that belong to the selection, sums[row] = sum of the selected values (of relu(values) if relu != 0). */
int ps_topk_select(const float* x, int32_t rows, int64_t row_len, int32_t k, int32_t largest, int32_t relu, float* thr, int32_t* take,
                   float* sums, void* stream);
/* The same with every row spread over many blocks (one launch per radix pass + a summation and a finishing launch): the single-block
 * version walks each row five times on one CU.  workspace: ps_topk_select_workspace_bytes(rows) bytes, 16-byte aligned, contents
 * clobbered; NULL falls back to ps_topk_select.  Results are identical (thr, take exactly; sums up to the order of a float sum). */
int64_t ps_topk_select_workspace_bytes(int32_t rows);
int ps_topk_select_ws(const float* x, int32_t rows, int64_t row_len, int32_t k, int32_t largest, int32_t relu, float* thr, int32_t* take,
                      float* sums, void* workspace, int64_t workspace_bytes, void* stream);
/* out[0] (+)= scale * sum_i x[i]   (fixed-order reduction of per-row sums). */
int ps_sum_scaled(const float* x, int32_t n, float scale, float* out, int32_t accumulate, void* stream);
/* out[nc] = mean over hw of x[nc, hw]   (F.adaptive_avg_pool2d(cam, 1), revise_pseudo_labels.py:253). */
int ps_gap(const float* x, float* out, int32_t nc, int64_t hw, void* stream);
/* F.multilabel_soft_margin_loss(gap[:,1:], label[:,1:]) (revise_pseudo_labels.py:255): loss_out[0] (+)= loss,
 * dgap[n,c] = grad_scale * d loss / d gap (channel 0 gets 0). */
int ps_softmargin(const float* gap, const float* label, float* dgap, float* loss_out, int32_t accumulate, float grad_scale, int32_t n,
                  int32_t c, void* stream);
/* dx[nc, hw] += dgap[nc] / hw */
int ps_gap_bwd(const float* dgap, float* dx, int32_t nc, int64_t hw, void* stream);
/* m[n,p] = max over channels 1.. of x[n,c,p]*label[n,c], arg[n,p] = that channel (first max)
 * (adaptive_min_pooling_loss((cam_rv*label)[:,1:]), revise_pseudo_labels.py:115-119,254). */
int ps_chmax(const float* x, const float* label, float* m, uint8_t* arg, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* dx[n,arg,p] += grad_scale*label[n,arg] for the k smallest m (per ps_topk_select) that are > 0. */
int ps_minpool_bwd(const float* m, const uint8_t* arg, const float* label, const float* thr, const int32_t* take, int32_t* tie_counter,
                   float* dx, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);
/* Deterministic tie choice, as ps_ecr_bwd_det. */
int ps_minpool_bwd_det(const float* m, const uint8_t* arg, const float* label, const float* thr, const int32_t* take, int32_t* seg_counts,
                       int64_t seg_counts_ints, float* dx, float grad_scale, int32_t n, int32_t c, int32_t h, int32_t w, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimisers over a flat f32 arena.  replaces: torch.optim.AdamW (models/segmentation_module.py:86-90)
 * and utils.PolyOptimizer / torch.optim.SGD (utils.py:166-187).
 * ---------------------------------------------------------------------------------------------- */
/* Decoupled-weight-decay Adam, torch.optim.AdamW semantics (eps outside the bias-corrected sqrt):
 *   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * Optionally refreshes a bf16 shadow copy of p (p_bf16 may be NULL). */
int ps_adamw_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* Same, with the gradient multiplied by grad_inv_scale first (undoing fp16 loss scaling) and a 16-bit shadow of dtype
 * shadow_dtype in {PS_BF16, PS_F16} (p_shadow may be NULL). */
int ps_adamw_step_scaled(float* p, const float* g, float* m, float* v, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step, float grad_inv_scale, void* stream);
/* The same update with the overflow check of dynamic loss scaling ON THE DEVICE (no host round trip per step): state[0] = optimiser steps applied so
 * far, state[1] = non-finite elements of this step's gradient (written by ps_nonfinite_count into &state[1]; zero it first).  When state[1] != 0 the
 * launch leaves p / m / v / shadow untouched; otherwise it applies AdamW with bias corrections from t = state[0] + 1 and then advances state[0].
 * replaces: torch.cuda.amp.GradScaler.step's `if not found_inf: optimizer.step()` around AdamW.step (Lightning's fp16 precision plugin,
 * models/segmentation_module.py:86-90). */
int ps_adamw_step_guarded(float* p, const float* g, float* m, float* v, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int32_t* state, float grad_inv_scale, void* stream);
/* torch.optim.SGD with momentum (dampening 0, no nesterov) and L2 weight decay:
 *   g' = g + wd*p;  buf = first ? g' : mom*buf + g';  p -= lr*buf */
int ps_sgd_step(float* p, const float* g, float* buf, void* p_bf16, int64_t n, float lr, float momentum,
                float weight_decay, int32_t first_step, void* stream);
/* Same, with the gradient multiplied by grad_inv_scale first and a 16-bit shadow of dtype shadow_dtype in {PS_BF16, PS_F16}. */
int ps_sgd_step_scaled(float* p, const float* g, float* buf, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr,
                       float momentum, float weight_decay, int32_t first_step, float grad_inv_scale, void* stream);
/* The same update with the overflow check of dynamic loss scaling on the device, as ps_adamw_step_guarded: state[0] = steps applied so far,
 * state[1] = non-finite elements of this step's gradient.  When state[1] != 0 nothing is written; otherwise `first_step` is state[0] == 0 and, with
 * poly_max_step > 0, lr is multiplied by utils.PolyOptimizer's schedule (1 - t / poly_max_step) ** poly_power at t = state[0] (utils.py:176-182; held
 * at t = poly_max_step - 1 beyond it).  A step over several parameter groups is several calls on one stream, `advance` != 0 on the last only
 * (state[0] += 1 when the step was applied). */
int ps_sgd_step_guarded(float* p, const float* g, float* buf, void* p_shadow, int32_t shadow_dtype, int64_t n, float lr, float momentum,
                        float weight_decay, int32_t* state, int32_t advance, int32_t poly_max_step, float poly_power, float grad_inv_scale,
                        void* stream);

/* ---- sliding-window evaluation ------------------------------------------------------------------ */
/* Where one tile of a batch lands: its valid (un-padded) region [0, vh) x [0, vw) is added at (y0, x0) of a per-image f64 canvas.
 * canvas: [canvas_h, canvas_w, C] if channels_last (models/segmentation_module.py:156: np.zeros((h_, w_, 3))) else
 * [C, canvas_h, canvas_w] (OEEM/classification/prepare_seg_inputs.py:121: np.zeros((num_of_class, w_, h_))); count: [canvas_h, canvas_w].
 * An array of these lives in DEVICE memory. */
typedef struct ps_tile_dst {
  double* canvas;
  double* count;
  int32_t canvas_h, canvas_w;
  int32_t y0, x0, vh, vw;
  int32_t channels_last, _pad;
} ps_tile_dst;

/* For every tile j < n: p = apply_softmax ? softmax_C(scores[j]) : scores[j] (f32, [n, C, h, w]);
 * canvas_j[y0 + y, x0 + x, :] += p[:, y, x] and count_j[y0 + y, x0 + x] += 1 for y < vh, x < vw (f64 atomics: tiles of one launch may overlap).
 * replaces: the per-sample loop `probs = torch.softmax(output_, dim=0).cpu().numpy(); pred_big_mask_dict_ms[key][...] += probs;
 * cnt_big_mask_dict_ms[key][...] += 1` (models/segmentation_module.py:141-161, segmentation_test.py:141-183) and
 * `sum_cam[:, y:y+side, x:x+side] += crop; sum_counter[...] += 1` (prepare_seg_inputs.py:124-130).  The caller checks bounds. */
int ps_softmax_scatter_accum(const float* scores, int32_t n, int32_t c, int32_t h, int32_t w, const ps_tile_dst* tiles_dev,
                             int32_t apply_softmax, void* stream);

/* dst (=|+=) bilinear_{align_corners=False}(src / d) in f64, d = src_count (per pixel) or the scalar src_div when src_count is NULL;
 * zero_uncovered: counts < 1 become 1 (prepare_seg_inputs.py:131), otherwise 0/0 = NaN as in numpy; dst_count (optional) (=|+=) 1.
 * replaces: `mask /= cnt; mask = F.interpolate(torch.from_numpy(mask...), (h, w), mode='bilinear'); pred_big_mask_dict[idx] += mask;
 * cnt_big_mask_dict[idx] += 1` (segmentation_module.py:166-178) and the three F.interpolate calls of prepare_seg_inputs.py:133-138. */
int ps_canvas_resize_accum(const double* src, const double* src_count, double src_div, int32_t hs, int32_t ws, double* dst,
                           double* dst_count, int32_t hd, int32_t wd, int32_t c, int32_t channels_last, int32_t zero_uncovered,
                           int32_t accumulate, void* stream);

/* pred[y, x] = argmax_C(canvas / count) (first maximum; NaN is the maximum, as torch.argmax), then pred = bg_value where
 * gt == bg_value (gt may be NULL / bg_value < 0: no overwrite).
 * replaces: `mask_pred /= cnt; big_mask_iou(torch.from_numpy(mask_pred...), ..., probs=True)` (segmentation_module.py:181-185,
 * loss.py:55-57) and `mask_pred[mask == 3] = 3` (segmentation_test.py:201). */
int ps_canvas_argmax(const double* canvas, const double* count, int32_t h, int32_t w, int32_t c, int32_t channels_last,
                     const uint8_t* gt, int32_t bg_value, uint8_t* pred, void* stream);

/* One view of the d4 group on `planes` square side x side f32 planes: forward dst = rot90(hflip ? flip_W(src) : src, k);
 * inverse dst (=|+=) (hflip ? flip_W : id)(rot90(src, 4 - k)).  replaces: ttach d4_transform()'s augment_image / deaugment_mask
 * (tta.SegmentationTTAWrapper, infer_pseudo_masks.py:96, mosaic_module.py:76; third-party ttach==0.0.3: parity unpinned). */
int ps_d4_view(const float* src, float* dst, int64_t planes, int32_t side, int32_t hflip, int32_t k, int32_t inverse,
               int32_t accumulate, void* stream);
/* x /= divisor (the 'mean' merge of the eight views). */
int ps_scale_inplace(float* x, int64_t n, float divisor, void* stream);

/* Dropout2d multipliers of one training step in ONE launch.  out: flat f32 buffer of plan->nseg <= 8 segments (one per Dropout2d
 * module: [n, channels] each, segment k = elements [end[k-1], end[k])); out[e] = (u >= p[k]) / (1 - p[k]) with u uniform on a 24-bit
 * grid from Philox4x32-10 (key = seed, counter = (e / 4, offset)): the same (seed, offset) always gives the same masks, a new offset
 * per step a new draw.  replaces: the Bernoulli draw inside nn.Dropout2d (models/resnet38d.py:63,67,85,90; models/revise_net.py:11,50);
 * the multiplication is fused into the conv / fc8 kernels (ps_epilogue.drop).  torch's own RNG stream cannot be reproduced: parity
 * tests inject the masks on both sides. */
typedef struct ps_dropout_plan { int32_t nseg; int32_t _pad; int64_t end[8]; float p[8]; } ps_dropout_plan;
int ps_dropout2d_masks(float* out, const ps_dropout_plan* plan, uint64_t seed, uint64_t offset, void* stream);

/* *count += number of inf/nan elements of g[0..n) (caller zeroes count).  Used by fp16 dynamic loss scaling: an overflowed
 * activation gradient reaches the f32 gradient arena as inf/nan, and the step is then skipped. */
int ps_nonfinite_count(const float* g, int64_t n, int32_t* count, void* stream);

/* The library keeps NO process-global mutable state: per-device properties (CU count) are cached, errors are thread-local, every
 * tuning decision is a pure function of the geometry passed in.  Testing / ablation hooks live in pistoseg_hip_debug.h and exist only
 * in libpistoseg_hip_debug.so (built with -DPS_DEBUG_HOOKS). */

#ifdef __cplusplus
}
#endif
#endif /* PISTOSEG_HIP_H */
