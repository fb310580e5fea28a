/*
 * pistoseg_hip_debug.h -- testing and ablation hooks of libpistoseg_hip_debug.so (the same sources built with -DPS_DEBUG_HOOKS).
 *
 * NOT part of the product ABI: libpistoseg_hip.so exports none of these symbols and its kernel-selection tunables are compile-time
 * constants.  The debug library exports the whole product ABI (pistoseg_hip.h) plus the process-global switches below, which let the
 * parity suite force every staging / tiling variant through the same entry points (all variants compute identical results) and let
 * tools/conv_bench.py, tools/k_sweep.py and tools/hog_probe.py run timing ablations.  Never link the product against this header.
 */
#ifndef PISTOSEG_HIP_DEBUG_H
#define PISTOSEG_HIP_DEBUG_H

#include "pistoseg_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Testing hook: `blocks` workgroups of 256 threads each occupy a CU slot (and `lds_bytes` of its LDS: >= 40 KiB keeps the persistent
 * conv blocks off that CU) for `usec` microseconds -- a stand-in for a communication kernel sharing the GPU (tools/hog_probe.py). */
int ps_debug_hog(int32_t blocks, int32_t usec, int32_t lds_bytes, void* stream);

/* Testing hook: how conv operands are staged into LDS: 2 (default) LDS-DMA through buffer descriptors
 * (buffer_load ... lds; padding rows are out-of-range lanes, which the DMA zero-fills), 1 LDS-DMA with flat
 * addresses (global_load_lds; padding rows read a zero page), 0 through registers.  All variants compute identical
 * results; the tests run the parity suite over all of them. */
void ps_debug_set_glds(int on);
/* Testing hook: 1 lets large problems use the experimental 256x128 three-stage kernel; 0 (default) uses the 128-pixel kernels. */
void ps_debug_set_3stage(int on);
/* Testing hook: 64 / 128 force the cout-tile width of the 2-stage conv kernel, 0 (default) picks by problem size. */
void ps_debug_set_bn(int bn);
/* Testing hook: 112 / 128 force the pixel-tile height of the 128-cout conv kernel, 0 (default) picks the better-balanced. */
void ps_debug_set_bm(int bm);
/* Timing experiments only (results become WRONG): 1 = the conv kernels stage their first K-steps and then stop loading (consumer-only
 * rate); 3 = the epilogue touches no memory (per-tile store cost). */
void ps_debug_set_ablate(int v);
/* Testing hook: 1 = big problems use the experimental 8-wave ping-pong conv kernel, 0 (default) = never, 2 = always (cout % 128 == 0). */
void ps_debug_set_pp(int v);
/* Testing hook: 1 (default) = big problems use the wave-specialised (4 loader + 4 consumer waves) conv kernel, 0 = never, 2 = always. */
void ps_debug_set_ws(int v);
/* Testing hook: large-tile (256|224 x 128, one block per CU) wave-specialised kernel: 0 off, 1 (default) chosen by the cost model, 256 / 224 forced. */
void ps_debug_set_ws2(int v);
/* Testing hook: window + halo kernel for 3x3 stride-1 layers (width a multiple of 28): 0 off, 1 (default) for big 16-bit problems, 2 forced. */
void ps_debug_set_halo(int v);
/* Testing hook: conv_gemm256_kernel (256x256 tile, plain 1x1 stride-1 GEMMs, 16-bit): 0 off, 1 (default) by shape, 2 whenever legal. */
void ps_debug_set_gemm256(int v);
/* Testing hook: gemm256's partial last round as a second launch on the gathered-tile kernels: 0 off, 1 (default) on. */
void ps_debug_set_gemm256_tail(int v);
void ps_debug_set_gemm256_rule(int code); /* the by-shape rule's thresholds: code = min K-lines (K / 64) * 10000 + min produced channels; default 32 * 10000 + 1024 */
void ps_debug_set_gemm256_min_tiles(int v); /* the by-shape rule's minimum number of 256 x 256 tiles in a launch (default 384 = 1.5 rounds of the 256 CUs) */
/* Tuning hook: weight ring depth of the halo kernel: 3, 4 or 5 stages of 16 KiB (256-pixel tiles: at most 4). */
void ps_debug_set_halo_ring(int v);
/* Testing hook: halo kernel, partial last round as a second launch of 64-cout half tiles: 0 off, 1 (default) on. */
void ps_debug_set_halo_tail(int v);
void ps_debug_set_halo_stagger(int v); /* staggered start of the halo kernel's blocks: low byte = step in units of 2048 shader cycles (0 = off, default), bits 8.. = phases per XCD (0 -> 4) */
void ps_debug_set_halo_sk(int v);   /* stream-K finish of the halo kernel's partial last round: 0 off, 1 on unless gpu_shared, 2 on (default) -- 224-pixel tiles only */
void ps_debug_set_s2split(int v);  /* stride-2 3x3 data gradient as four parity-class launches: 0 off, 1 big 16-bit problems (default), 2 whenever legal */
/* Testing hook: 1 (default) = 128x128 weight-gradient tiles use the wave-specialised variant, 0 = the 4-wave kernel. */
void ps_debug_set_wgrad_ws(int v);
/* Testing hook: large-tile persistent weight-gradient kernel (256x128 tile, one block per CU): 0 off, 1 (default) for big 16-bit problems, 2 forced. */
void ps_debug_set_wgrad_ws2(int v);
/* Experiment hook: conv_wgrad256_kernel (256x256 tile; measured slower than the ws2 kernel, r03): 0 (default) off, 1 by shape, 2 whenever legal
 * (stride 1, cout % 256 == 0, cin % 256 == 0, 16-bit). */
void ps_debug_set_wgrad256(int v);
/* Timing experiments only (results WRONG): 1 = the large-tile weight-gradient kernel skips its atomics, 2 = plain stores instead,
 * 3 = it stages its first three K-steps only (consumer-only rate). */
void ps_debug_set_wgrad_ablate(int v);
/* Tuning hook: per-item overhead (in 64-pixel K-steps) the large-tile weight-gradient kernel's split-K cost model assumes. */
void ps_debug_set_wgrad_ovh(int v);
/* Testing hook: cout tiles per super-column of the conv block raster (default 4; 0 = plain row-major). */
void ps_debug_set_supertile(int v);
/* Testing hook: weight-gradient block order: 0 pixel range slowest, 1 pixel range fastest, -1 (default) chosen by shape. */
/* every tunable above and below back to its library default */
void ps_debug_reset(void);
void ps_debug_set_wgrad_raster(int v);
/* 3x3 stride-1 weight gradients: 1 (default) = padding validity from the precomputed lane-mask table, 0 = per-row tracking in the loaders */
void ps_debug_set_wgrad_vtab(int v);

/* EXPERIMENT (r04; correct, not faster than the three launches it replaces: profiles/r04_front_fusion.txt).  conv1a + the first ResBlock's entry in ONE launch, for the uses in which conv1a's activation is not needed again (inference; training
 * of the models that freeze b2, models/revise_net.py:27):
 *   a      = max(conv1a(image) * scale0 + shift0, 0)            (3x3, 3 -> 64, stride 1, pad 1; never written to memory)
 *   out_b1 = conv_branch1(a)                                    (1x1, stride 2, 64 -> 128, raw)            w_b1: [128][64]
 *   out_2a = max(conv_branch2a(a) * scale1 + shift1, 0)         (3x3, stride 2, pad 1, 64 -> 128)          w_2a: [128][3][3][64]
 * replaces: models/resnet38d.py:123,161-162 and ResBlock.forward :28-41 of b2 (x_bn_relu, branch1, the first conv of branch2 + its BN + ReLU).
 * dtype PS_BF16 / PS_F16 (weights and outputs; image and conv1a weights f32, rounded to dtype like ps_conv1a_fwd does); image NCHW f32
 * [n][3][h][w] with w = 224 or 256 and h even; outputs channels-last [n, h/2, w/2, >= 128] with the given channel strides.  Same arithmetic
 * as ps_conv1a_fwd followed by the two ps_conv2d_fwd launches, up to the summation order inside conv1a. */
int ps_debug_conv_front_s2_supported(int32_t dtype, int32_t n, int32_t h, int32_t w);
int ps_debug_conv_front_s2(int32_t dtype, int32_t n, int32_t h, int32_t w, const float* image, const float* w1a, const float* scale0,
                     const float* shift0, const void* w_b1, const void* w_2a, void* out_b1, int32_t ldc_b1, const float* scale1,
                     const float* shift1, void* out_2a, int32_t ldc_2a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PISTOSEG_HIP_DEBUG_H */
